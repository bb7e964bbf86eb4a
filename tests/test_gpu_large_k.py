"""-m gpu tests of a search for ANY k (round 4, second session).

The reference takes `k: usize` without a limit (examples/similarity_search.rs:143; its heap holds k + 1 entries whatever k
is, :166-168, and what is left is sorted, :172-173).  Beyond MVFGPU_K_PER_PASS = 1024 the library has two exact
formulations: passes of the streaming kernel behind a floor (k <= MVFGPU_K_BY_PASSES = 16384) and the WHOLE-SHARD SORT
(the streaming kernel writes every row's order key, a device-wide sort ranks them; any k).  It picks the cheaper one;
MVF_LARGE_K = 1 | 2 forces passes | the sort.  Here: both against the oracle and against each other, on every dump site of
the kernel (lane-group widths 1 / 4 / 8 / 16 / 64, one and four queries per pass), every data type and metric, with ties,
deletions, vector ids, k at, beyond and far beyond the rows.  Everything through the C ABI; the oracle is the checker."""
import ctypes as C

import numpy as np
import pytest

from metrovector_amd import _lib, errors as E, gpu as G

from _util import assert_exact, assert_float_topk

pytestmark = pytest.mark.gpu
SEED = 0x4D564631
PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def forced(c, monkeypatch, mode):
    """mode: 1 = passes, 2 = the whole-shard sort, 0 = the library's choice."""
    if mode:
        monkeypatch.setenv("MVF_LARGE_K", str(mode))
    else:
        monkeypatch.delenv("MVF_LARGE_K", raising=False)
    c.reload_tuning()


def same(a, b):
    return (a.indices == b.indices).all() and (a.raw == b.raw).all() and \
        (a.scores.view(np.uint32) == b.scores.view(np.uint32)).all()


# dim per (dtype): row sizes that take the lane groups 1, 4, 8, 16 and 64 of the streaming kernel
SHAPES = [(2, 16), (2, 64), (3, 128), (1, 96), (0, 96), (0, 768), (1, 100)]


@pytest.mark.parametrize("dtype,dim", SHAPES)
@pytest.mark.parametrize("metric", [G.L2, G.INNER_PRODUCT, G.COSINE])
def test_sort_and_passes_agree_and_match_the_oracle(oracle, monkeypatch, dtype, dim, metric):
    n, k = (60_000 if dim < 768 else 20_000), 3000
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, 6, dim, dtype)
    with G.GpuCorpus.from_array(rows, index_base=5_000_000_000) as c:
        c.set_profiling(True)
        forced(c, monkeypatch, 2)
        s6 = c.search(q, k, metric)               # four queries + two
        assert c.last_timing().scan_kernel == 8
        s1 = c.search(q[5], k, metric)            # one query per pass
        forced(c, monkeypatch, 1)
        p6 = c.search(q, k, metric)
        assert c.last_timing().scan_kernel == 1
    assert same(s6, p6), "whole-shard sort and passes differ"
    if dtype in (2, 3) or dim != 768:             # (3-KiB float rows: one query sums on 64 lanes, four on 16 -- other last bits)
        assert same(G.SearchResult(s6.scores[5:], s6.indices[5:], s6.raw[5:]), s1)
    osc, oidx, oraw = oracle.search(rows, dtype, metric, q, k, index_base=5_000_000_000)
    if dtype in (2, 3):
        assert_exact(s6, osc, oidx, oraw)
    else:
        rows32 = rows.astype(np.float32)
        for qi in (0, 3, 5):
            sc = oracle.scores(rows, dtype, metric, q[qi])[0]
            assert_float_topk(metric, s6.scores[qi], s6.indices[qi], sc, rows32, q[qi], k, index_base=5_000_000_000)
        assert_float_topk(metric, s1.scores[0], s1.indices[0], sc, rows32, q[5], k, index_base=5_000_000_000)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
def test_k_far_beyond_the_pass_formulation(oracle, dtype):
    """k = 50 000, k = n and k = n + 1000 on 120k rows: only the sort serves these (the library's own choice)."""
    n, dim = 120_000, 48
    metric = [G.COSINE, G.L2, G.INNER_PRODUCT, G.L2][dtype]
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, 2, dim, dtype)
    all_sc = [oracle.scores(rows, dtype, metric, q[qi])[0] for qi in range(2)]
    rows32 = rows.astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        for k in (50_000, n, n + 1000):
            got = c.search(q, k, metric)
            if dtype in (2, 3):
                assert_exact(got, *oracle.search(rows, dtype, metric, q, k))
            else:
                for qi in range(2):
                    assert_float_topk(metric, got.scores[qi], got.indices[qi], all_sc[qi], rows32, q[qi], k)
            if k > n:
                assert (got.indices[:, n:] == PAD).all()
                assert sorted(got.indices[0, :n].tolist()) == list(range(n))    # a full ranking: every row exactly once


def test_ties_deletions_ids_and_exhaustion_on_the_sort_path(oracle, monkeypatch):
    """Few distinct scores (ties everywhere: the order inside a tie group is the row position), 40 % of the rows deleted,
    vector ids, k beyond the live rows -- the sort path against the oracle on the live rows and against the passes."""
    rng = np.random.default_rng(11)
    n, dim = 30_000, 32
    rows = rng.integers(-3, 4, (n, dim)).astype(np.int8)
    q = rng.integers(-3, 4, (6, dim)).astype(np.int8)
    dead = np.zeros(n, bool)
    dead[rng.choice(n, 12_000, replace=False)] = True
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(10**12)
    live = np.nonzero(~dead)[0]
    with G.GpuCorpus.from_array(rows) as c:
        forced(c, monkeypatch, 2)
        for k in (1025, 5000, 16384, 16385, 29_999):
            got = c.search(q, k, G.INNER_PRODUCT)
            assert_exact(got, *oracle.search(rows, 2, 1, q, k))
        c.set_tombstones(np.packbits(dead, bitorder="little"))
        c.set_vector_ids(ids)
        for k, metric in ((16384, G.L2), (20_000, G.COSINE), (18_000, G.INNER_PRODUCT)):
            got = c.search(q, k, metric)
            osc, oidx, oraw = oracle.search(rows[live], 2, metric, q, k)
            kk = min(k, len(live))
            assert (got.indices[:, :kk] == ids[live[oidx[:, :kk].astype(np.int64)]]).all()
            assert (got.raw[:, :kk] == oraw[:, :kk]).all()
            assert (got.scores[:, :kk].view(np.uint32) == osc[:, :kk].view(np.uint32)).all()
            assert (got.indices[:, kk:] == PAD).all()                # deleted rows are never returned: padding behind the live ones
            if k <= 16384:
                forced(c, monkeypatch, 1)
                assert same(got, c.search(q, k, metric))
                forced(c, monkeypatch, 2)


def test_nan_scores_rank_behind_every_number_and_in_front_of_the_padding(oracle, monkeypatch):
    rng = np.random.default_rng(5)
    n, dim, k = 5000, 24, 6000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    bad = rng.choice(n, 40, replace=False)
    rows[bad, 3] = np.nan
    q = rng.standard_normal(dim).astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        forced(c, monkeypatch, 2)
        got = c.search(q, k, G.L2)
    assert np.isfinite(got.scores[0, :n - 40]).all()
    assert np.isnan(got.scores[0, n - 40:n]).all() and sorted(got.indices[0, n - 40:n].tolist()) == sorted(bad.tolist())
    assert (got.indices[0, n - 40:n] == np.sort(bad).astype(np.uint64)).all()      # ties (all NaN) by row position
    assert (got.indices[0, n:] == PAD).all() and (got.scores[0, n:] == np.inf).all()
    # ... and with deletions in between (dead rows and NaN rows must not tie: the sort orders by key only)
    dead = np.zeros(n, bool)
    dead[rng.choice(n, 1500, replace=False)] = True
    dead[bad[:10]] = True
    live_bad = np.sort(np.setdiff1d(bad, np.nonzero(dead)[0]))
    nlive = int((~dead).sum())
    with G.GpuCorpus.from_array(rows) as c:
        c.set_tombstones(np.packbits(dead, bitorder="little"))
        forced(c, monkeypatch, 2)
        got = c.search(q, k, G.L2)
        forced(c, monkeypatch, 1)
        by_passes = c.search(q, k, G.L2)
    assert same(got, by_passes)
    nb = len(live_bad)
    assert np.isfinite(got.scores[0, :nlive - nb]).all() and not dead[got.indices[0, :nlive].astype(np.int64)].any()
    assert (got.indices[0, nlive - nb:nlive] == live_bad.astype(np.uint64)).all() and np.isnan(got.scores[0, nlive - nb:nlive]).all()
    assert (got.indices[0, nlive:] == PAD).all()


def test_a_search_after_the_sort_path_is_unchanged(oracle, monkeypatch):
    """The dump launches use the floor instantiation of the kernel and the handle's candidate buffers: ordinary searches
    before and after return the same bits."""
    n, dim = 80_000, 128
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, 4, dim, 0)
    with G.GpuCorpus.from_array(rows) as c:
        before = c.search(q, 100, G.COSINE)
        forced(c, monkeypatch, 2)
        big = c.search(q, 2000, G.COSINE)
        after = c.search(q, 100, G.COSINE)
        forced(c, monkeypatch, 1)
        big_p = c.search(q, 2000, G.COSINE)
    assert same(before, after) and same(big, big_p)
    assert (big.indices[:, :100] == before.indices).all()


def test_k_beyond_the_abi_limit_is_refused_before_any_device_call():
    rows = np.zeros((8, 16), np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        with pytest.raises(E.InvalidArgument):
            c.search_device(1, 0, 16, 1, 2**31 + 1, G.L2, 1, 1)    # MVFGPU_MAX_K = 2^31; the pointers are never touched
        rc = _lib.gpu().mvfgpu_search_device(c._h, 0, C.c_void_p(1), 0, 16, 1, 0, C.c_void_p(1), C.c_void_p(1), None, None)
        assert rc != 0


# ---------------------------------------------------------------------------------------------------------------------
# Cross-shard merges beyond one block's LDS (nlists * k > 8192): a device-wide sort per query.  Same order as the LDS merge
# and the host merge: (score order, list order, rank in the list).
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype,metric", [(0, G.L2), (2, G.INNER_PRODUCT), (3, G.COSINE)])
@pytest.mark.parametrize("packed", [False, True])
def test_device_merge_of_long_lists(oracle, dtype, metric, packed):
    import torch
    rng = np.random.default_rng(3)
    n, dim, nq, nl = 60_000, 16, 3, 5
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype) if dtype == 0 else rng.integers(-2, 3, (n, dim)).astype([None, None, np.int8, np.uint8][dtype])
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype) if dtype == 0 else rng.integers(0, 3, (nq, dim)).astype(rows.dtype)
    cuts = [0, 9000, 9000, 31_000, 52_500, n]                   # ragged shards, one of them empty
    for k in (1700, 9000, 20_000):                              # 5 x 1700 = 8500 > 8192: the first size the sort serves
        parts = [oracle.search(rows[a:b], dtype, metric, q, k, index_base=int(a)) for a, b in zip(cuts[:-1], cuts[1:])]
        S, I, R = (np.stack([p[j] for p in parts]) for j in range(3))
        want_s, want_i, want_r = oracle.search(rows, dtype, metric, q, k)
        oS = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        oI = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        oR = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        if packed:   # one list = { u64 indices | f32 scores | i32 raw } (MVFGPU_PACKED_LIST_BYTES): what the all-gather delivers
            buf = np.concatenate([np.concatenate([I[l].view(np.uint8).ravel(), S[l].view(np.uint8).ravel(), R[l].view(np.uint8).ravel()])
                                  for l in range(nl)])
            dP = torch.from_numpy(buf).cuda()
            _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_packed_device(dP.data_ptr(), nl, nq, k, metric, dtype, oS.data_ptr(),
                                                                      oI.data_ptr(), oR.data_ptr(), 0, None))
        else:
            dS, dI, dR = torch.from_numpy(S).cuda(), torch.from_numpy(I.view(np.int64)).cuda(), torch.from_numpy(R).cuda()
            _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_device(dS.data_ptr(), dI.data_ptr(), dR.data_ptr(), nl, nq, k, metric, dtype,
                                                               oS.data_ptr(), oI.data_ptr(), oR.data_ptr(), 0, None))
        torch.cuda.synchronize()
        got = G.SearchResult(oS.cpu().numpy(), oI.cpu().numpy().view(np.uint64), oR.cpu().numpy())
        hm = G.merge_topk_host(S, I, R, metric, dtype)
        assert same(got, hm), "device merge (sort) differs from the host merge"
        assert_exact(got, want_s, want_i, want_r)               # merge(top-k per shard) == top-k(global), ties by position


def test_shardset_with_k_in_the_thousands(oracle, monkeypatch):
    """Three shards behind mvfgpu_shardset_search, k = 1024 (the LDS merge's last size), 5000 and 20 000 (per-shard searches
    by the whole-shard sort, merge by the device-wide sort): the oracle's answer over all rows, bit for bit (Int8)."""
    rng = np.random.default_rng(8)
    n, dim, nq = 45_000, 48, 5
    rows = rng.integers(-5, 6, (n, dim)).astype(np.int8)
    q = rng.integers(-5, 6, (nq, dim)).astype(np.int8)
    cuts = [0, 20_000, 21_000, n]
    shards = [G.GpuCorpus.from_array(rows[a:b], index_base=int(a)) for a, b in zip(cuts[:-1], cuts[1:])]
    try:
        with G.ShardSet(shards) as ss:
            for k in (1024, 5000, 20_000):
                got = ss.search(q, k, G.L2)
                assert_exact(got, *oracle.search(rows, 2, 0, q, k))
    finally:
        for c in shards:
            c.close()


def test_buffers_beyond_a_gib_come_from_the_stream_ordered_allocator(oracle):
    """20M x 16 Int8 rows, two queries (a four-query dump pass): 1.28 GB of composites -- not kept with the handle but taken
    from hipMallocAsync for the search and given back behind it.  Bit-exact against the oracle over all 20M rows; a second
    search (the pool hands the block out again) returns the same bits; the handle's scratch stays small."""
    from _util import oracle_topk_all_rows
    n, dim, k = 20_000_000, 16, 3000
    q = oracle.synth_queries(SEED + 1, 2, dim, 2)
    with G.GpuCorpus.synthetic(n, dim, 2, SEED) as c:
        before = c.info().device_bytes
        a = c.search(q, k, G.INNER_PRODUCT)
        b = c.search(q, k, G.INNER_PRODUCT)
        assert c.info().device_bytes - before < (64 << 20)
    assert same(a, b)
    assert_exact(a, *oracle_topk_all_rows(oracle, SEED, 0, n, dim, 2, 1, q, k))


def test_sharded_searcher_with_k_in_the_thousands_over_rccl(oracle, tmp_path):
    """One process per GPU (metrovector_amd/sharded.py) at k = 3000 and k = 12 000: the local search by the whole-shard sort, the
    packed list through a 1-rank nccl (= RCCL) all-gather, the merge in LDS (3000 entries) and by the device-wide sort (12 000 >
    8192).  Same bits as the plain search; Int8 rows bit-exact against the oracle."""
    import os
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from metrovector_amd import gpu as G
from metrovector_amd.sharded import ShardedSearcher
from oracle import mvf_oracle as O
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, dim, dtype, metric = 30011, 48, 2, 1
rows = O.synth_rows(777, 0, n, dim, dtype)
q = O.synth_queries(778, 3, dim, dtype)
c = G.GpuCorpus.from_array(rows, index_base=5, device=0)
tq = torch.from_numpy(q.copy()).cuda()
for k in (3000, 12000):
    plain = [t.clone() for t in ShardedSearcher(c).search(tq, k, metric)]
    got = ShardedSearcher(c, always_exchange=True).search(tq, k, metric)
    torch.cuda.synchronize()
    for a, b in zip(plain, got):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b), k
    osc, oidx, oraw = O.search(rows, dtype, metric, q, k, index_base=5)
    assert (got[1].cpu().numpy().view(np.uint64) == oidx).all() and (got[2].cpu().numpy() == oraw).all(), k
c.close()
dist.destroy_process_group()
print("rccl ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29743", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl ok" in out.stdout


def test_whole_shard_sort_of_a_hundred_million_rows(oracle, monkeypatch):
    """n = 100M (16-byte Int8 rows): the sort's scratch for a four-query pass is 6.4 GB from the stream-ordered allocator, the
    radix sort runs on 10^8 entries per query.  Sort and passes must agree at k = 16384; k = 50 000 continues the same list."""
    n, dim = 100_000_000, 16
    q = oracle.synth_queries(SEED + 1, 2, dim, 2)
    with G.GpuCorpus.synthetic(n, dim, 2, SEED) as c:
        forced(c, monkeypatch, 2)
        a = c.search(q, 16384, G.INNER_PRODUCT)
        big = c.search(q[0], 50_000, G.INNER_PRODUCT)
        forced(c, monkeypatch, 1)
        b = c.search(q, 16384, G.INNER_PRODUCT)
    assert same(a, b)
    assert (big.indices[0, :16384] == a.indices[0]).all() and (big.raw[0, :16384] == a.raw[0]).all()
    assert (np.diff(big.raw[0].astype(np.int64)) <= 0).all() and len(set(big.indices[0].tolist())) == 50_000
