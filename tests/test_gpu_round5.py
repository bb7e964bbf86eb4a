"""-m gpu tests added in round 5.

Part 1 -- the WHOLE of BASELINE.json configs[4] on the one GPU a box has: the 100M x 1024 Float16 corpus (204.8 GB) as EIGHT
12.5M-row handles in one `mvfgpu_shardset` (the 8-way split north_star names; the lists travel by device copies instead of
RCCL because the shards share a device) and as ONE 100M-row handle -- the two must return the same rows and the same score
bits for all 1024 queries (merge(top-k per shard) == top-k(global)) -- and two of the queries against the
oracle's score of ALL 100M rows.  The shards select on their stored Float16 rows (MVF_I8_SHADOW=0): eight int8 shadows do not
fit beside 204.8 GB of rows, and the single handle has no room for one either.
"""
import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import assert_exact, assert_float_topk, oracle_scores_all_rows

pytestmark = pytest.mark.gpu
SEED = 0x4D564631


def test_cfg5_100m_x_1024_f16_l2_eight_shards_vs_one_handle_vs_the_oracle_over_all_rows(oracle, monkeypatch):
    n_shards, shard_rows, dim, nq, k = 8, 12_500_000, 1024, 1024, 100
    n = n_shards * shard_rows
    monkeypatch.setenv("MVF_I8_SHADOW", "0")  # read once per handle, at creation
    q = oracle.synth_queries(SEED + 1, nq, dim, 1)
    shards = [G.GpuCorpus.synthetic(shard_rows, dim, 1, SEED, row0=i * shard_rows) for i in range(n_shards)]
    try:
        with G.ShardSet(shards) as ss:
            info = ss.info()
            assert info.n_shards == n_shards and info.rows == n
            merged = ss.search(q, k, G.L2)
    finally:
        for s in shards:
            s.close()
    with G.GpuCorpus.synthetic(n, dim, 1, SEED) as c:
        assert c.info().rows == n
        one = c.search(q, k, G.L2)
        assert not c.info().shadows & 5
    # the same handle with the default environment (round 5): no room for the int8 shadow of all 100M rows, so a PREFIX of
    # them is shadowed and the search runs as two row ranges whose lists are merged -- same rows, same score bits
    monkeypatch.delenv("MVF_I8_SHADOW")
    with G.GpuCorpus.synthetic(n, dim, 1, SEED) as c:
        part = c.search(q, k, G.L2)
        assert c.info().shadows & 4, "expected an int8 shadow of a prefix of the rows beside 204.8 GB of Float16 rows"
    assert (part.indices == one.indices).all() and (part.scores.view(np.uint32) == one.scores.view(np.uint32)).all()
    assert (merged.indices == one.indices).all(), "merge(top-k per shard) != top-k(one handle)"
    assert (merged.scores.view(np.uint32) == one.scores.view(np.uint32)).all()
    assert merged.indices.max() < n and (np.sort(merged.indices, axis=1)[:, 1:] != np.sort(merged.indices, axis=1)[:, :-1]).all()
    hit_shards = np.unique((merged.indices // shard_rows).astype(np.int64))
    assert len(hit_shards) == n_shards, "1024 x 100 results that miss a whole shard of a uniform corpus"
    sel = [5, 1000]
    all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, 1, 0, q[sel], chunk=500_000)
    for j, qi in enumerate(sel):
        assert_float_topk(0, one.scores[qi], one.indices[qi], all_sc[j], None, q[qi], k)


# ---------------------------------------------------------------------------------------------------------------------
# Part 2 -- mvfgpu_search_fetch with k far beyond the corpus: the reference takes any k and returns min(k, n)
# items (examples/similarity_search.rs:143, :159-168); the payload staging is min(k, rows) rows per query, not k.
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("with_ids", [False, True])
def test_search_fetch_with_k_far_beyond_the_corpus(oracle, with_ids):
    n, dim, k, nq = 300, 64, 200_000, 3
    rows = oracle.synth_rows(SEED + 3, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 4, nq, dim, 0)
    dead = np.zeros(n, bool)
    dead[::7] = True
    with G.GpuCorpus.from_array(rows) as c:
        before = c.info().device_bytes
        if with_ids:
            c.set_vector_ids(np.arange(n, dtype=np.uint64) * 3 + 11)
        c.set_tombstones(np.packbits(dead, bitorder="little"))
        res, vec = c.search_fetch(q, k, G.L2)
        grown = c.info().device_bytes - before
    live = np.nonzero(~dead)[0]
    sc, idx, _ = oracle.search(rows[live], 0, 0, q, len(live))
    want = live[idx.astype(np.int64)]
    assert vec.shape == (nq, k, dim) and res.indices.shape == (nq, k)
    for qi in range(nq):
        got = res.indices[qi][:len(live)].astype(np.int64)
        pos = (got - 11) // 3 if with_ids else got
        assert (pos == want[qi]).all()
        assert np.allclose(res.scores[qi][:len(live)], sc[qi], rtol=1e-5)
        assert (res.indices[qi][len(live):] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        assert (vec[qi][:len(live)] == rows[pos]).all()
        assert not vec[qi][len(live):].any()  # padding results: rows left as the caller allocated them (zeros here)
    # 3 queries x 200k results x 256 B = 154 MB of zero rows if the staging were sized by k; min(k, rows) rows are 230 KB
    assert grown < 64 << 20, f"{grown} bytes of scratch for a 300-row corpus"


# ---------------------------------------------------------------------------------------------------------------------
# Part 3 -- an int8 selection shadow of a PREFIX of the rows (round 5): a corpus whose shadow does not fit beside it as a
# whole is searched as two row ranges (int8 selection over the prefix, the f16 / f32 kernels over the rest) whose lists are
# merged.  MVF_I8_SHADOW_ROWS forces the split on corpora of test size: the answers must be those of the unsplit search, bit
# for bit, and the oracle's.
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dt,metric,nq,k", [(1, 0, 300, 10), (1, 2, 70, 100), (0, 1, 257, 25), (0, 2, 1024, 100)])
def test_int8_shadow_of_a_prefix_of_the_rows_two_ranges_merged(oracle, monkeypatch, dt, metric, nq, k):
    n, dim = 400_000, 96
    rows = oracle.synth_rows(SEED + 11, 0, n, dim, dt)
    q = oracle.synth_queries(SEED + 12, nq, dim, dt)
    dead = np.zeros(n, bool)
    dead[5::11] = True
    ids = (np.arange(n, dtype=np.uint64) * 7 + 3) if metric == 2 else None

    def run(env):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        with G.GpuCorpus.from_array(rows) as c:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
            if ids is not None:
                c.set_vector_ids(ids)
            res = c.search(q, k, metric)
            sh = c.info().shadows
        for key in env:
            monkeypatch.delenv(key)
        return res, sh

    whole, sh_whole = run({})
    split, sh_split = run({"MVF_I8_SHADOW_ROWS": "262144"})
    assert sh_whole & 1 and not sh_whole & 4, "the unsplit handle selects on an int8 shadow of all rows"
    assert sh_split & 4 and not sh_split & 1, "MVF_I8_SHADOW_ROWS shadows a prefix only"
    assert (split.indices == whole.indices).all()
    assert (split.scores.view(np.uint32) == whole.scores.view(np.uint32)).all()
    live = np.nonzero(~dead)[0]
    sel = [0, nq // 2, nq - 1]
    rows32 = rows[live].astype(np.float32)
    for qi in sel:
        all_sc = oracle.scores(rows[live], dt, metric, q[qi])[0]
        got = split.indices[qi].astype(np.int64)
        pos = (got - 3) // 7 if ids is not None else got
        assert not dead[pos].any()
        local = np.searchsorted(live, pos)
        assert_float_topk(metric, split.scores[qi], local.astype(np.uint64), all_sc, rows32, q[qi].astype(np.float32), k)


def test_prefix_shadow_with_overflowing_candidate_regions_is_repaired_once_behind_the_merge(oracle, monkeypatch):
    """Tiny candidate regions overflow in both ranges: the flagged queries are redone exactly over the WHOLE corpus, after
    the merge (a repair inside a range would return rows of the other range twice)."""
    n, dim, nq, k = 300_000, 64, 300, 20
    rows = oracle.synth_rows(SEED + 13, 0, n, dim, 1)
    q = oracle.synth_queries(SEED + 14, nq, dim, 1)
    monkeypatch.setenv("MVF_I8_SHADOW_ROWS", "196608")
    monkeypatch.setenv("MVF_K2_REGION_RECORDS", "4096")
    with G.GpuCorpus.from_array(rows) as c:
        res = c.search(q, k, G.COSINE)
        assert c.info().shadows & 4
        repaired = c.last_timing().repaired_queries if hasattr(c.last_timing(), "repaired_queries") else None
    sc, idx, _ = oracle.search(rows, 1, 2, q, k)
    rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(res.indices, idx)])
    assert rec >= 0.999, rec
    assert np.allclose(res.scores, sc, rtol=0, atol=1e-5)
    for r in res.indices:
        assert len(set(r.tolist())) == k, "a row returned twice"


# ---------------------------------------------------------------------------------------------------------------------
# Part 4 -- the hand-written select + sorts behind k > 1024 (csrc/sort_topk.hip): the long-list forms that the older tests
# do not reach -- a select whose survivors exceed 131072 entries (8-bit digit passes with the scan launch) and the sort of a
# whole shard of that length -- on rows with few distinct scores (ties by position everywhere), bit-exact against the oracle.
# ---------------------------------------------------------------------------------------------------------------------

def test_select_and_sort_of_long_lists_with_ties(oracle):
    rng = np.random.default_rng(5)
    n, dim = 300_000, 16
    rows = rng.integers(-2, 3, (n, dim)).astype(np.int8)
    q = rng.integers(-2, 3, (5, dim)).astype(np.int8)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_profiling(True)
        for k in (140_000, 200_000, n):      # select (k < n / 2) / full sort / full ranking
            got = c.search(q, k, G.INNER_PRODUCT)
            assert c.last_timing().scan_kernel == 8
            assert_exact(got, *oracle.search(rows, 2, 1, q, k))
        one = c.search(q[0], 131_073, G.L2)  # one query per pass, survivors one past the 11-bit form's limit
        assert_exact(one, *oracle.search(rows, 2, 0, q[:1], 131_073))


# ---------------------------------------------------------------------------------------------------------------------
# Part 5 -- the schedule switches of the round's second half change WHEN thresholds tighten and how the phases are cut,
# never what is returned: the same handle under every value of each switch (mvfgpu_corpus_reload_tuning re-reads them)
# returns the same rows and the same score bits, and they are the oracle's.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,metric,nq,k", [(0, 2, 16, 100), (0, 0, 100, 10), (1, 2, 128, 100), (0, 2, 400, 100), (1, 0, 1030, 50)])
def test_phase_schedule_switches_do_not_change_the_results(oracle, monkeypatch, dt, metric, nq, k):
    n, dim = 1_200_000, 128
    q = oracle.synth_queries(SEED + 1, nq, dim, dt)
    with G.GpuCorpus.synthetic(n, dim, dt, SEED) as c:
        base = c.search(q, k, metric)
        sel = [0, nq // 2, nq - 1]
        all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, dt, metric, q[sel], chunk=400_000)
        for j, qi in enumerate(sel):
            assert_float_topk(metric, base.scores[qi], base.indices[qi], all_sc[j], None, q[qi], k)
        for var, values in (("MVF_K2_GROWTH_SMALL", ("4", "8")), ("MVF_K2_GROWTH", ("3", "6")), ("MVF_QS_REFINE_PHASES", ("0", "1", "3")),
                            ("MVF_K2_DIRECT64", ("0",))):
            for v in values:
                monkeypatch.setenv(var, v)
                c.reload_tuning()
                r = c.search(q, k, metric)
                assert (r.indices == base.indices).all(), f"{var}={v}: other rows"
                assert (r.scores.view(np.uint32) == base.scores.view(np.uint32)).all(), f"{var}={v}: other score bits"
            monkeypatch.delenv(var)
        c.reload_tuning()


# ---------------------------------------------------------------------------------------------------------------------
# Part 6 -- rows of more than 256 16-byte vectors (dim > 1024 Float32, > 2048 Float16): the final re-scoring takes the block
# kernel there (rescore_score_kernel), which since round 5 writes its exact keys into the list's upper half like the wave kernel.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,dim,metric,nq,k", [(0, 1536, 2, 70, 100), (0, 2048, 0, 300, 10), (1, 2560, 1, 130, 50)])
def test_batched_search_on_rows_longer_than_256_vectors(oracle, dt, dim, metric, nq, k):
    n = 60_000
    q = oracle.synth_queries(SEED + 1, nq, dim, dt)
    with G.GpuCorpus.synthetic(n, dim, dt, SEED) as c:
        got = c.search(q, k, metric)
        c.set_scan_path(1)  # the streaming kernel: exact, one pass per four queries
        ref = c.search(q, k, metric)
    assert (got.indices == ref.indices).all(), "batched route != streaming kernel"
    sel = [0, nq - 1]
    all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, dt, metric, q[sel], chunk=20_000)
    rows = oracle.synth_rows(SEED, 0, n, dim, dt).astype(np.float32) if metric == 1 else None
    for j, qi in enumerate(sel):
        assert_float_topk(metric, got.scores[qi], got.indices[qi], all_sc[j], rows, q[qi], k)
