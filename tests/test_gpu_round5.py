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

from _util import assert_float_topk, oracle_scores_all_rows

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
