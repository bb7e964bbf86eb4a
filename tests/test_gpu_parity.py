"""Parity of the HIP path (libmvf_gpu.so through its C ABI) against the CPU
oracle — the -m gpu tier.  Bit-exact for Int8/UInt8; within 1e-5 for f32/f16
(tolerances in tests/_util.py)."""
import os

import numpy as np
import pytest

from metrovector_amd import errors as E
from metrovector_amd import gpu as G
from tests._util import PAD, assert_exact, assert_float_topk, recall_at_k

pytestmark = pytest.mark.gpu

SEED = 0x4D564631  # "MVF1" 


def _check(oracle, dtype, metric, rows, q, k, index_base=0):
    with G.GpuCorpus.from_array(rows, index_base=index_base) as c:
        res = c.search(q, k, metric)
    q2 = q if q.ndim == 2 else q[None]
    if dtype in (2, 3):
        osc, oidx, oraw = oracle.search(rows, dtype, metric, q2, k, index_base=index_base)
        assert_exact(res, osc, oidx, oraw)
    else:
        rows32 = rows.astype(np.float32)
        for i in range(q2.shape[0]):
            sc, _, _ = oracle.scores(rows, dtype, metric, q2[i])
            assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q2[i], k, index_base)
        assert (res.raw == 0).all()
    return res


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(1000, 128, 1, 10), (5000, 768, 3, 100), (777, 4, 1, 5), (3000, 13, 5, 7),
                                   (100, 100, 2, 128), (2049, 48, 9, 33), (1, 16, 1, 1), (513, 1024, 2, 64)])
def test_all_dtypes_metrics_shapes(oracle, dtype, metric, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    _check(oracle, dtype, metric, rows, q, k)


def test_reference_example_known_answers(golden):
    """examples/similarity_search.rs dataset + queries: intended (nearest) answers."""
    g = golden["similarity_search_60x4"]
    X = np.array(g["rows_bits"], np.uint32).view(np.float32)
    with G.GpuCorpus.from_array(X) as c:
        for case in g["cases"]:
            res = c.search(np.array(case["query"], np.float32), case["k"], G.L2)
            assert res.indices[0].tolist() == case["intended_nearest"]["indices"]
            want = np.array(case["intended_nearest"]["score_bits"], np.uint32).view(np.float32)
            np.testing.assert_allclose(res.scores[0], want, rtol=1e-5, atol=1e-7)


def test_find_top_k_similar_on_mvf_file(golden, golden_dir):
    """The drop-in, end to end: open the .mvf, map_vector_range -> HBM -> search."""
    from metrovector_amd.reader import MvfReader
    from metrovector_amd.search import find_top_k_similar
    r = MvfReader.open(os.path.join(golden_dir, "clusters_60x4_f32.mvf"))
    space = r.vector_space(r.vector_space_names()[0])
    for case in golden["similarity_search_60x4"]["cases"]:
        top = find_top_k_similar(space, case["query"], case["k"])
        assert [t.index for t in top] == case["intended_nearest"]["indices"]
        want = np.array(case["intended_nearest"]["score_bits"], np.uint32).view(np.float32)
        np.testing.assert_allclose([t.score for t in top], want, rtol=1e-5, atol=1e-7)
        assert (top[0].vector == space.get_vector(top[0].index).as_f32()).all()
    with pytest.raises(E.DimensionMismatch, match="expected 4, got 3"):
        find_top_k_similar(space, [1, 2, 3], 5)


def test_find_top_k_similar_batch_agrees_with_the_single_query_drop_in(golden, golden_dir):
    from metrovector_amd import MvfReader, find_top_k_similar, find_top_k_similar_batch, upload_space
    r = MvfReader.open(os.path.join(golden_dir, "clusters_60x4_f32.mvf"))
    space = r.vector_space(r.vector_space_names()[0])
    cases = golden["similarity_search_60x4"]["cases"]
    qs = [c["query"] for c in cases] * 20            # 60+ queries: the batched (MFMA) path even on a tiny space
    with upload_space(space) as corpus:
        batch = find_top_k_similar_batch(space, qs, 5, corpus=corpus, with_vectors=True)
        for q, hits in zip(qs, batch):
            single = find_top_k_similar(space, q, 5, corpus=corpus)
            assert [h.index for h in hits] == [s.index for s in single]
            np.testing.assert_allclose([h.score for h in hits], [s.score for s in single], rtol=1e-5, atol=1e-7)
            assert all((h.vector == s.vector).all() for h, s in zip(hits, single))
    with pytest.raises(E.BuildError):
        find_top_k_similar_batch(space, [1, 2, 3, 4], 5)      # 1-D: not a batch


def test_multi_space_file_all_dtypes(oracle, golden_dir):
    from metrovector_amd.reader import MvfReader
    from metrovector_amd.search import find_top_k_similar, upload_space
    r = MvfReader.open(os.path.join(golden_dir, "multi_space.mvf"))
    src = np.load(os.path.join(golden_dir, "multi_space_src.npz"))
    rng = np.random.default_rng(5)
    # f32 cosine (metric taken from the file), unaligned block offsets downstream
    q = rng.standard_normal(24).astype(np.float32)
    top = find_top_k_similar(r.vector_space("f32_cos"), q, 6)
    sc, idx, _ = oracle.search(src["A"], 0, 2, q, 6)
    assert [t.index for t in top] == idx[0].tolist()
    top = find_top_k_similar(r.vector_space("f16_l2"), q, 6)
    sc, idx, _ = oracle.search(src["A"].astype(np.float16), 1, 0, q, 6)
    assert [t.index for t in top] == idx[0].tolist()
    qi = rng.integers(-128, 128, 20, dtype=np.int8)
    top = find_top_k_similar(r.vector_space("i8_dot"), qi, 50)
    sc, idx, raw = oracle.search(src["I8"], 2, 1, qi, 50)
    assert [t.index for t in top] == idx[0].tolist() and [t.score for t in top] == sc[0].tolist()
    qu = rng.integers(0, 256, 7, dtype=np.uint8)
    sp = r.vector_space("u8_l2")
    with upload_space(sp) as c:
        assert (c.read_rows(0, 33) == src["U8"]).all()
        res = c.search(qu, 40, G.L2)
    sc, idx, raw = oracle.search(src["U8"], 3, 0, qu, 40)
    assert_exact(res, sc, idx, raw)


def test_integer_ties_break_by_index(oracle):
    # few distinct values -> massive score ties; order must be (score, index)
    rng = np.random.default_rng(1)
    rows = rng.integers(0, 2, (4000, 16), dtype=np.uint8)
    q = np.ones(16, np.uint8)
    _check(oracle, 3, 1, rows, q, 300)
    rows8 = rng.integers(-1, 2, (3000, 32)).astype(np.int8)
    _check(oracle, 2, 0, rows8, np.zeros(32, np.int8), 1024)


def test_float_ties_duplicates_and_all_equal(oracle):
    rows = np.tile(np.array([[0.25, -0.5, 1.0, 2.0]], np.float32), (1500, 1))
    q = np.array([1, 1, 1, 1], np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        for metric in (0, 1, 2):
            res = c.search(q, 20, metric)
            assert res.indices[0].tolist() == list(range(20))  # all scores equal -> ascending index
            assert len(set(res.scores[0].tolist())) == 1


def test_nan_rows_sort_last(oracle):
    rows = oracle.synth_rows(3, 0, 600, 8, 0)
    rows[5, 2] = np.nan
    rows[77, 0] = np.nan
    q = oracle.synth_queries(4, 1, 8, 0)[0]
    with G.GpuCorpus.from_array(rows) as c:
        res = c.search(q, 600, G.L2)
    assert sorted(res.indices[0, -2:].tolist()) == [5, 77]
    assert np.isnan(res.scores[0, -2:]).all() and not np.isnan(res.scores[0, :-2]).any()


def test_zero_norm_cosine_is_zero():
    rows = np.zeros((10, 8), np.float32)
    rows[3] = 1.0
    with G.GpuCorpus.from_array(rows) as c:
        res = c.search(np.ones(8, np.float32), 10, G.COSINE)
    assert res.indices[0, 0] == 3 and abs(res.scores[0, 0] - 1.0) < 1e-6
    assert (res.scores[0, 1:] == 0.0).all() and res.indices[0, 1:].tolist() == [0, 1, 2, 4, 5, 6, 7, 8, 9]


def test_empty_and_tiny_corpora():
    with G.GpuCorpus.from_array(np.zeros((0, 8), np.float32)) as c:
        res = c.search(np.ones(8, np.float32), 3, G.L2)
        assert (res.indices == PAD).all() and (res.scores == np.inf).all()
        res = c.search(np.ones(8, np.float32), 3, G.INNER_PRODUCT)
        assert (res.scores == -np.inf).all()
    with G.GpuCorpus.from_array(np.array([[1, 2, 3]], np.int8)) as c:
        res = c.search(np.array([1, 1, 1], np.int8), 2, G.INNER_PRODUCT)
        assert res.indices[0].tolist() == [0, int(PAD)] and res.raw[0, 0] == 6


def test_strided_and_misaligned_host_rows(oracle):
    # rows embedded in a wider host array at an odd byte offset (like an mmap'd block after an odd-sized one)
    base = oracle.synth_rows(8, 0, 300, 40, 0)
    buf = np.zeros(300 * 200 + 7, np.uint8)
    view = np.lib.stride_tricks.as_strided(buf[3:].view(np.uint8), (300, 160), (200, 1))
    view[:] = base.view(np.uint8).reshape(300, 160)
    ptr = buf.ctypes.data + 3
    q = oracle.synth_queries(9, 2, 40, 0)
    c = G.GpuCorpus.from_pointer(ptr, 300, 40, 0, 200, index_base=1000)
    try:
        assert (c.read_rows(0, 300).view(np.uint32) == base.view(np.uint32)).all()
        res = c.search(q, 10, G.L2)
    finally:
        c.close()
    ws, wi, _ = oracle.search(base, 0, 0, q, 10, index_base=1000)
    assert recall_at_k(res.indices, wi) == 1.0 and res.indices.min() >= 1000


def test_error_codes_mirror_reference():
    rows = np.zeros((8, 16), np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        with pytest.raises(E.DimensionMismatch, match="expected 16, got 8"):
            c.search(np.zeros(8, np.float32), 2)
        with pytest.raises(E.BuildError):
            c.search(np.zeros(16, np.int8), 2)
        with pytest.raises(E.InvalidArgument):
            c.search(np.zeros(16, np.float32), 0)
        with pytest.raises(E.InvalidArgument):
            c.search_device(1, 0, 16, 1, 2**31 + 1, 0, 1, 1)   # MVFGPU_MAX_K = 2^31: refused before any pointer is touched
        with pytest.raises(E.InvalidArgument):
            c.search(np.zeros(16, np.float32), 2, metric=255)
        with pytest.raises(E.IndexOutOfBounds):
            c.read_rows(4, 5)
    with G.GpuCorpus.from_array(np.zeros((8, 16), np.int8)) as c:
        with pytest.raises(E.BuildError):
            c.search(np.zeros(16, np.float32), 2)


def test_synthetic_generator_matches_oracle(oracle):
    for dtype in (0, 1, 2, 3):
        with G.GpuCorpus.synthetic(1000, 52, dtype, SEED, row0=12345) as c:
            got = c.read_rows(0, 1000)
            want = oracle.synth_rows(SEED, 12345, 1000, 52, dtype)
            assert (got.view(np.uint8) == want.view(np.uint8)).all()
            assert c.info().index_base == 12345


def test_device_pointer_api_and_shard_merge_device(oracle):
    """mvfgpu_search_device on torch-owned buffers + mvfgpu_merge_topk_device:
    merge(top-k per shard) == top-k(global)  (SURVEY.md §8e)."""
    import ctypes as C
    import torch
    from metrovector_amd import _lib
    n, dim, nq, k = 6000, 96, 5, 50
    for dtype, metric in ((0, 2), (1, 0), (2, 1)):
        rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
        q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
        qd = G.query_dtype_code(dtype)
        tq = torch.from_numpy(q.copy()).cuda()
        cuts = [0, 1500, 1501, 6000]
        S = torch.empty((3, nq, k), dtype=torch.float32, device="cuda")
        I = torch.empty((3, nq, k), dtype=torch.int64, device="cuda")
        R = torch.empty((3, nq, k), dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        shards = []
        for j, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            c = G.GpuCorpus.from_array(rows[a:b], index_base=a)
            shards.append(c)
            c.search_device(tq.data_ptr(), qd, dim, nq, k, metric, S[j].data_ptr(), I[j].data_ptr(), R[j].data_ptr(), stream)
        OS = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        OI = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        OR = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_device(S.data_ptr(), I.data_ptr(), R.data_ptr(), 3, nq, k, metric, dtype,
                                                           OS.data_ptr(), OI.data_ptr(), OR.data_ptr(), 0, C.c_void_p(stream)))
        torch.cuda.synchronize()
        for c in shards:
            c.close()
        gi = OI.cpu().numpy().view(np.uint64)
        ws, wi, wr = oracle.search(rows, dtype, metric, q, k)
        if dtype == 2:
            assert (gi == wi).all() and (OR.cpu().numpy() == wr).all()
        else:
            assert recall_at_k(gi, wi) >= 0.99
            np.testing.assert_allclose(OS.cpu().numpy(), ws, rtol=1e-4, atol=1e-5)
        # host merge of the same per-shard lists agrees with the device merge
        hm = G.merge_topk_host(S.cpu().numpy(), I.cpu().numpy().view(np.uint64), R.cpu().numpy(), metric, dtype)
        assert (hm.indices == gi).all()
        # the same lists in the packed all-gather layout {u64 indices | f32 scores | i32 raw} per shard
        nk = nq * k
        packed = torch.empty((3, 2 * nk), dtype=torch.int64, device="cuda")
        for j in range(3):
            packed[j, :nk] = I[j].reshape(-1)
            tail = packed[j, nk:].view(torch.int32)
            tail[:nk] = S[j].reshape(-1).view(torch.int32)
            tail[nk:] = R[j].reshape(-1)
        PS, PI, PR = torch.empty_like(OS), torch.empty_like(OI), torch.empty_like(OR)
        _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_packed_device(packed.data_ptr(), 3, nq, k, metric, dtype, PS.data_ptr(),
                                                                  PI.data_ptr(), PR.data_ptr(), 0, C.c_void_p(stream)))
        torch.cuda.synchronize()
        assert torch.equal(PI, OI) and torch.equal(PS.view(torch.int32), OS.view(torch.int32)) and torch.equal(PR, OR)


def test_concurrent_searches_on_one_handle(oracle):
    import threading
    rows = oracle.synth_rows(SEED, 0, 20000, 64, 0)
    qs = oracle.synth_queries(SEED + 1, 8, 64, 0)
    want = oracle.search(rows, 0, 0, qs, 10)[1]
    out = [None] * 8
    with G.GpuCorpus.from_array(rows) as c:
        def work(i):
            out[i] = c.search(qs[i], 10, G.L2).indices[0]
        ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    assert recall_at_k(np.stack(out), want) == 1.0


# ---------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle cannot
# score 7.7e9 elements in seconds; rows are regenerated on demand instead).
# ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def corpus_10m():
    c = G.GpuCorpus.synthetic(10_000_000, 768, 0, SEED)
    yield c
    c.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_full_size_cfg2_properties(oracle, corpus_10m, metric):
    n, dim, k = 10_000_000, 768, 100
    c = corpus_10m
    q = oracle.synth_queries(SEED + 1, 1, dim, 0)[0]
    res = c.search(q, k, metric)
    idx = res.indices[0].astype(np.int64)
    assert len(set(idx.tolist())) == k and idx.min() >= 0 and idx.max() < n
    # (1) every returned row re-scored by the oracle from regenerated bytes
    rows = np.stack([oracle.synth_rows(SEED, int(i), 1, dim, 0)[0] for i in idx])
    sc, _, _ = oracle.scores(rows, 0, metric, q)
    tol = 1e-5 * np.abs(sc) if metric == 0 else 1e-5
    assert (np.abs(sc - res.scores[0]) <= tol).all()
    sign = 1 if metric == 0 else -1
    assert (np.diff(sign * res.scores[0]) >= 0).all()
    # (2) no row of a 200k-row oracle-scored sample beats the k-th result unless it was returned
    kth = sign * float(res.scores[0, -1])
    r0 = 3_333_333
    sample = oracle.synth_rows(SEED, r0, 200_000, dim, 0)
    ssc, _, _ = oracle.scores(sample, 0, metric, q)
    better = np.nonzero(sign * ssc.astype(np.float64) < kth - 1e-5 * abs(kth))[0] + r0
    assert set(better.tolist()) <= set(idx.tolist())
    # (3) planted neighbour: a query equal to a corpus row must find it first with distance 0 / cosine 1
    planted = 8_765_432
    pq = oracle.synth_rows(SEED, planted, 1, dim, 0)[0]
    pres = c.search(pq, 5, metric)
    assert pres.indices[0, 0] == planted
    assert abs(pres.scores[0, 0] - (0.0 if metric == 0 else 1.0)) < 1e-5


def test_full_size_shard_merge_invariance(oracle, corpus_10m):
    """top-k(whole) == merge(top-k(shards)) at full size, using device shards of the same synthetic corpus."""
    dim, k = 768, 100
    q = oracle.synth_queries(SEED + 1, 2, dim, 0)
    whole = corpus_10m.search(q, k, G.COSINE)
    cuts = [0, 2_500_000, 6_000_001, 10_000_000]
    S, I = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        with G.GpuCorpus.synthetic(b - a, dim, 0, SEED, row0=a) as c:
            r = c.search(q, k, G.COSINE)
            S.append(r.scores)
            I.append(r.indices)
    merged = G.merge_topk_host(np.stack(S), np.stack(I), None, G.COSINE, 0)
    assert (merged.indices == whole.indices).all()
    assert (merged.scores.view(np.uint32) == whole.scores.view(np.uint32)).all()


# ---------------------------------------------------------------------------
# K2: MFMA batched path (f32, cosine / dot)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("metric", [1, 2])
@pytest.mark.parametrize("shape", [(20000, 768, 200, 100), (5000, 100, 33, 10), (300, 64, 128, 500), (9000, 36, 40, 1),
                                   (4097, 128, 32, 64)])
def test_batched_mfma_path(oracle, metric, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    with G.GpuCorpus.from_array(rows, index_base=77) as c:
        c.set_scan_path(2)
        res = c.search(q, k, metric)
        c.set_scan_path(1)
        ref = c.search(q, k, metric)   # streaming path, same queries
    rows32 = rows.astype(np.float32)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k, 77)
    assert recall_at_k(res.indices, ref.indices) >= 0.999


# ---------------------------------------------------------------------------
# K2 on Float32 corpora through the scaled-f16 SHADOW (scan path 3): the f16 MFMA kernel only selects, the kept
# rows are re-scored from the f32 rows -- results must be those of the exact paths
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(20000, 768, 200, 100), (5000, 100, 33, 10), (300, 64, 128, 500), (9000, 36, 40, 1),
                                   (4097, 128, 32, 64)])
def test_batched_f32_shadow_path(oracle, metric, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    q[0] *= 1e6
    q[1] *= 1e-9
    with G.GpuCorpus.from_array(rows, index_base=77) as c:
        c.set_scan_path(3)
        c.set_profiling(True)
        res = c.search(q, k, metric)
        assert c.last_timing().scan_kernel == 4       # the shadow really ran
        c.set_profiling(False)
        c.set_scan_path(2)
        exact = c.search(q, k, metric)                # exact f32 MFMA kernel
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k, 77)
    assert recall_at_k(res.indices, exact.indices) >= 0.999


def test_batched_f32_shadow_row_dynamic_range(oracle):
    """Rows spanning 60 orders of magnitude, elements far below their row's largest, zero rows, and rows holding
    Inf / NaN: every row is scaled by its own power of two before it is rounded to f16, and whatever the shadow
    loses the re-scoring restores."""
    n, dim, nq, k = 12000, 96, 36, 25
    rng = np.random.default_rng(21)
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    rows *= (10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)[:, None]
    rows[::11, ::3] *= 1e-7                       # tiny elements next to large ones
    rows[5] = 0.0
    rows[17, 3] = np.inf
    rows[23, 0] = np.nan
    rows[29] = 1.0e35                             # sum x^2 overflows, the dot products do not
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    for metric in (2, 1, 0):
        with G.GpuCorpus.from_array(rows) as c:
            c.set_scan_path(3)
            res = c.search(q, k, metric)
        for i in range(nq):
            sc, _, _ = oracle.scores(rows, 0, metric, q[i])
            assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("cluster", [300, 3000])
def test_batched_f32_shadow_dense_cluster_at_the_kth_rank(oracle, metric, cluster):
    """As the Float16 case below: score gaps inside the cluster are far smaller than the shadow's error bound
    (2^-10 |q||x|), so the cluster either rides in the margin and is re-scored, or overflows the candidate
    budget and the query is repaired by the streaming kernel."""
    n, dim, nq, k = 20000, 192, 40, 50
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    rng = np.random.default_rng(12)
    base = rng.standard_normal(dim).astype(np.float32)
    where = rng.choice(n, cluster, replace=False)
    rows[where] = base * 0.5 + rng.standard_normal((cluster, dim)).astype(np.float32) * 1.5e-3
    q = (base[None, :] + rng.standard_normal((nq, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(3)
        res = c.search(q, k, metric)
    members = set(where.tolist())
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k)
        assert set(res.indices[i].tolist()) <= members


@pytest.mark.parametrize("dtype,metric", [(1, 2), (1, 0), (2, 1), (3, 0), (0, 2)])
def test_batched_register_staged_reference_kernel_agrees(oracle, dtype, metric, monkeypatch):
    """MVF_K2_DMA=0 selects the register-staged f16/int8 MFMA kernel (the A/B reference of the default LDS-DMA
    ring): both must return the same top-k -- bit-identical on integer spaces, identical after the exact
    re-scoring on float spaces (dtype 0 runs the f16 kernel on the shadow of the f32 rows)."""
    n, dim, nq, k = 30000, 200, 70, 40
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(3)
        monkeypatch.setenv("MVF_K2_DMA", "0")
        c.reload_tuning()   # the switches are read once per handle, at creation
        ref = c.search(q, k, metric)
        monkeypatch.delenv("MVF_K2_DMA")
        c.reload_tuning()
        res = c.search(q, k, metric)
    assert (res.indices == ref.indices).all()
    assert (res.scores.view(np.uint32) == ref.scores.view(np.uint32)).all()
    assert (res.raw == ref.raw).all()


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
def test_batched_odd_shapes_sweep(oracle, dtype):
    """The MFMA kernels pad everything (rows to 16 B, k to 64-B k-tiles, queries to the 256-query tile, rows to
    256-row tiles): sweep dimensions that are not multiples of any of those -- down to one element -- with row
    counts and batch sizes on both sides of the tile edges.  Float32 goes through the f16 shadow (path 3)."""
    rng = np.random.default_rng(1234 + dtype)
    dims = [1, 2, 7, 8, 15, 17, 31, 33, 63, 65, 100, 127, 129, 200, 257]
    for dim in dims:
        n = int(rng.choice([255, 256, 257, 1000, 4097, 5003]))
        nq = int(rng.choice([32, 33, 255, 256, 257]))
        k = int(rng.choice([1, 7, 64, 100]))
        metric = int(rng.integers(0, 3))
        rows = oracle.synth_rows(SEED + dim, 0, n, dim, dtype)
        q = oracle.synth_queries(SEED + 1 + dim, nq, dim, dtype)
        with G.GpuCorpus.from_array(rows) as c:
            c.set_scan_path(3)
            res = c.search(q, k, metric)
        if dtype in (2, 3):
            osc, oidx, oraw = oracle.search(rows, dtype, metric, q, k)
            assert_exact(res, osc, oidx, oraw)
        else:
            rows32 = rows.astype(np.float32)
            for i in range(0, nq, 7):
                sc, _, _ = oracle.scores(rows, dtype, metric, q[i])
                assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k)


@pytest.mark.parametrize("dtype,metric", [(2, 1), (1, 2), (0, 0)])
def test_batched_results_are_deterministic(oracle, dtype, metric):
    """The LDS-DMA kernel reads a stage only after a counted wait and a barrier; a read that raced its DMA would
    pass a single comparison whenever the DMA happened to win.  Repeated searches must be bit-identical
    (scripts/soak_k2.py does the same at full size, hundreds of times)."""
    n, dim, nq, k = 1_500_000, 256, 300, 50
    with G.GpuCorpus.synthetic(n, dim, dtype, SEED) as c:
        q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
        c.set_scan_path(3)
        first = c.search(q, k, metric)
        for _ in range(12):
            r = c.search(q, k, metric)
            assert (r.indices == first.indices).all()
            assert (r.scores.view(np.uint32) == first.scores.view(np.uint32)).all()
            assert (r.raw == first.raw).all()


@pytest.mark.parametrize("dtype,metric", [(0, 2), (1, 0), (2, 1), (3, 2)])
def test_batched_max_k(oracle, dtype, metric):
    """k = MVFGPU_MAX_K = 1024: the per-query candidate budget (4096 slots per phase, 2048 carried) is at its
    tightest and the phases grow only 2x at a time."""
    n, dim, nq, k = 150_000, 64, 40, 1024
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(3)
        res = c.search(q, k, metric)
    if dtype in (2, 3):
        osc, oidx, oraw = oracle.search(rows, dtype, metric, q, k)
        assert_exact(res, osc, oidx, oraw)
    else:
        rows32 = rows.astype(np.float32)
        for i in range(0, nq, 5):
            sc, _, _ = oracle.scores(rows, dtype, metric, q[i])
            assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k)


def test_device_bytes_accounts_for_the_f16_shadow(oracle):
    """mvfgpu_corpus_get_info.device_bytes is what a caller budgets HBM with: the shadow a batched search builds for a
    Float32 corpus (+50 %: dim*2 bytes per row rounded up to 16, plus 4 bytes of scale per row) must show up in it."""
    n, dim = 40_000, 100
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, 64, dim, 0)
    with G.GpuCorpus.from_array(rows) as c:
        before = c.info().device_bytes
        assert before >= n * 400
        c.set_scan_path(2)
        c.search(q, 10, G.COSINE)
        exact_only = c.info().device_bytes
        c.set_scan_path(3)
        c.search(q, 10, G.COSINE)
        with_shadow = c.info().device_bytes
        c.set_scan_path(5)
        c.search(q, 10, G.COSINE)
        with_shadow8 = c.info().device_bytes
    assert with_shadow - exact_only >= n * (208 + 4)      # pitch16 = 208 B for 100 halves
    assert with_shadow8 - with_shadow >= n * (112 + 4)    # the int8 shadow: 112 B for 100 bytes
    assert exact_only > before                            # norms + K2 scratch


# ---------------------------------------------------------------------------
# Scan path 4: one or two queries on a Float32 corpus stream its f16 shadow (K1, dt1x unit), margin + exact re-score
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(60000, 768, 1, 100), (20000, 100, 2, 10), (300, 64, 1, 500), (9000, 36, 2, 1),
                                   (5003, 7, 1, 64)])
def test_stream_shadow_path(oracle, metric, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    with G.GpuCorpus.from_array(rows, index_base=77) as c:
        c.set_scan_path(4)
        c.set_profiling(True)
        res = c.search(q, k, metric)
        assert c.last_timing().scan_kernel == 5       # K1 really streamed the shadow
        c.set_profiling(False)
        c.set_scan_path(1)
        exact = c.search(q, k, metric)                # K1 on the stored f32 rows
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k, 77)
    assert recall_at_k(res.indices, exact.indices) >= 0.999


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("cluster", [40, 3000])
def test_stream_shadow_dense_cluster_at_the_kth_rank(oracle, metric, cluster):
    """Score gaps inside the cluster are far below the shadow's error bound.  40 near-identical rows around the
    k-th rank fit inside the k' = 2k candidates and are re-scored; with 3000 of them every one of the k' candidates
    is inside the margin, rows beyond the cut may be too, and the query is redone by the exact kernel."""
    n, dim, k = 20000, 192, 50
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    rng = np.random.default_rng(12)
    base = rng.standard_normal(dim).astype(np.float32)
    where = rng.choice(n, cluster, replace=False)
    rows[where] = base * 0.5 + rng.standard_normal((cluster, dim)).astype(np.float32) * 1.5e-3
    q = (base[None, :] + rng.standard_normal((2, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(4)
        res = c.search(q, k, metric)
    for i in range(2):
        sc, _, _ = oracle.scores(rows, 0, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k)


def test_stream_shadow_row_dynamic_range(oracle):
    n, dim, k = 12000, 96, 25
    rng = np.random.default_rng(21)
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    rows *= (10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)[:, None]
    rows[::11, ::3] *= 1e-7
    rows[5] = 0.0
    rows[17, 3] = np.inf
    rows[23, 0] = np.nan
    q = oracle.synth_queries(SEED + 1, 2, dim, 0)
    for metric in (2, 1, 0):
        with G.GpuCorpus.from_array(rows) as c:
            c.set_scan_path(4)
            res = c.search(q, k, metric)
        for i in range(2):
            sc, _, _ = oracle.scores(rows, 0, metric, q[i])
            assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows, q[i], k)


@pytest.mark.parametrize("dtype,metric,nq,k", [(0, 2, 20000, 10), (1, 0, 6000, 100), (2, 1, 9000, 7), (0, 1, 2000, 1024)])
def test_batched_very_large_batches(oracle, dtype, metric, nq, k):
    """Thousands of queries in one call (tens of query tiles, hundreds of MB of per-query candidate buffers): a sample
    of the queries is checked against the oracle."""
    n, dim = 120_000, 64
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    with G.GpuCorpus.from_array(rows) as c:
        res = c.search(q, k, metric)          # automatic path: MFMA (shadowed for Float32)
    sample = list(range(0, nq, max(1, nq // 25))) + [nq - 1]
    if dtype in (2, 3):
        osc, oidx, oraw = oracle.search(rows, dtype, metric, q[sample], k)
        assert (res.indices[sample] == oidx).all() and (res.raw[sample] == oraw).all()
    else:
        rows32 = rows.astype(np.float32)
        for i in sample:
            sc, _, _ = oracle.scores(rows, dtype, metric, q[i])
            assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k)


def test_batched_path_single_query_forced(oracle):
    rows = oracle.synth_rows(SEED, 0, 3000, 96, 0)
    q = oracle.synth_queries(SEED + 1, 1, 96, 0)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, 20, G.COSINE)
    sc, _, _ = oracle.scores(rows, 0, 2, q[0])
    assert_float_topk(2, res.scores[0], res.indices[0], sc, rows, q[0], 20)


def test_batched_overflow_is_repaired_exactly(oracle):
    """Adversarial order: every row beats all previous ones, so a phase's
    survivors overflow the per-query candidate buffer; the flagged queries must
    be redone exactly by the streaming path."""
    n, dim, nq, k = 60000, 32, 32, 10
    rng = np.random.default_rng(9)
    base = rng.standard_normal(dim).astype(np.float32)
    scale = (np.arange(1, n + 1, dtype=np.float32) / n)[:, None]
    rows = (base[None, :] * scale).astype(np.float32)             # dot with +base grows with the row index
    q = np.tile(base, (nq, 1)) * rng.uniform(0.5, 2.0, (nq, 1)).astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, k, G.INNER_PRODUCT)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, 1, q[i])
        assert_float_topk(1, res.scores[i], res.indices[i], sc, rows, q[i], k)
    assert (res.indices[:, 0] == n - 1).all()


def test_sharded_searcher_two_ranks_one_gpu(oracle, tmp_path):
    """Rehearsal of the N>1 GPU path: two ranks (gloo, sharing cuda:0) run
    ShardedSearcher = search_device -> all-gather -> merge_topk_device and must
    both hold the global answer.  (The production backend is nccl/RCCL.)"""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from metrovector_amd import gpu as G
from metrovector_amd.sharded import ShardedSearcher, shard_range
from oracle import mvf_oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
for dtype, metric in ((0, 2), (2, 1)):
    n, dim, nq, k = 30001, 64, 6, 40
    lo, hi = shard_range(n, world, rank)
    c = G.GpuCorpus.synthetic(hi - lo, dim, dtype, 555, row0=lo, device=0)
    q = O.synth_queries(556, nq, dim, dtype)
    s, i, r = ShardedSearcher(c).search(torch.from_numpy(q.copy()).cuda(), k, metric)
    torch.cuda.synchronize()
    rows = O.synth_rows(555, 0, n, dim, dtype)
    ws, wi, wr = O.search(rows, dtype, metric, q, k)
    gi = i.cpu().numpy().view(np.uint64)
    if dtype == 2:
        assert (gi == wi).all() and (r.cpu().numpy() == wr).all()
    else:
        hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, wi))
        assert hits >= 0.99 * wi.size, hits
        assert np.allclose(s.cpu().numpy(), ws, atol=1e-5)
    c.close()
dist.destroy_process_group()
print("rank", rank, "ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2


def test_sharded_searcher_exchange_on_rccl_single_rank(oracle, tmp_path):
    """The exchange step on the PRODUCTION backend: a 1-rank nccl (= RCCL) group with always_exchange=True runs
    search_device -> all_gather_into_tensor (RCCL, packed list) -> merge_topk_packed_device on cuda:0.
    (More than one rank per GPU is refused by RCCL; the 2-rank rehearsal above stages through gloo.)"""
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
for dtype, metric, nq in ((0, 2, 1), (2, 1, 6), (1, 0, 40)):
    n, dim, k = 20011, 64, 33
    c = G.GpuCorpus.synthetic(n, dim, dtype, 777, device=0)
    q = O.synth_queries(778, nq, dim, dtype)
    tq = torch.from_numpy(q.copy()).cuda()
    plain = [t.clone() for t in ShardedSearcher(c).search(tq, k, metric)]
    ex = ShardedSearcher(c, always_exchange=True)
    for _ in range(3):   # repeated searches reuse the packed buffers
        got = ex.search(tq, k, metric)
    torch.cuda.synchronize()
    for a, b in zip(plain, got):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
    c.close()
dist.destroy_process_group()
print("rccl ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl ok" in out.stdout


# ---------------------------------------------------------------------------
# The remaining BASELINE.json configs as parity cases
# ---------------------------------------------------------------------------

def test_cfg1_reference_cpu_case_10k_x_128_l2_top10(oracle):
    """configs[0]: examples/similarity_search on CPU, 10k x 128 f32 Euclidean, single query, top-10.
    GPU vs the FAITHFUL restatement of the reference loop (intended = nearest semantics)."""
    rows = oracle.synth_rows(SEED, 0, 10_000, 128, 0)
    q = oracle.synth_queries(SEED + 1, 1, 128, 0)[0]
    idx, sc = oracle.find_top_k_similar_faithful(rows, 10_000, 128, 0, q, 10, farthest=False)
    with G.GpuCorpus.from_array(rows) as c:
        res = c.search(q, 10, G.L2)
    assert res.indices[0].tolist() == idx.tolist()
    np.testing.assert_allclose(res.scores[0], sc, rtol=1e-5)


def test_cfg4_50m_x_768_int8_dot_256_queries_bit_exact(oracle):
    """configs[3]: 50M x 768 Int8 dot, 256 batched queries — bit-exact vs the CPU on every returned row,
    plus an exhaustively scored 400k-row window."""
    n, dim, nq, k = 50_000_000, 768, 256, 100
    q = oracle.synth_queries(SEED + 1, nq, dim, 2)
    with G.GpuCorpus.synthetic(n, dim, 2, SEED) as c:
        res = c.search(q, k, G.INNER_PRODUCT)
    assert res.indices.max() < n
    r64 = None
    for qi in (0, 100, 255):
        idx = res.indices[qi].astype(np.int64)
        assert len(set(idx.tolist())) == k
        rows = np.stack([oracle.synth_rows(SEED, int(i), 1, dim, 2)[0] for i in idx])
        _, _, raw = oracle.scores(rows, 2, 1, q[qi])
        assert (raw == res.raw[qi]).all()                       # exact i32 dot of every returned row
        assert (res.scores[qi] == raw.astype(np.float32)).all()
        order = np.lexsort((idx, -raw.astype(np.int64)))
        assert (order == np.arange(k)).all()                    # sorted by (score desc, index asc)
    r0 = 31_000_000
    win = oracle.synth_rows(SEED, r0, 400_000, dim, 2)
    for qi in (7, 200):
        _, _, raw = oracle.scores(win, 2, 1, q[qi])
        kth = int(res.raw[qi, -1])
        better = np.nonzero(raw > kth)[0] + r0
        assert set(better.tolist()) <= set(res.indices[qi].tolist())


def test_cfg5_shard_12p5m_x_1024_f16_l2_batched(oracle):
    """configs[4], one GPU's shard: 12.5M x 1024 Float16 L2, batched queries, rows [25M, 37.5M) of the 100M corpus."""
    n, dim, nq, k, row0 = 12_500_000, 1024, 64, 100, 25_000_000
    q = oracle.synth_queries(SEED + 1, nq, dim, 1)
    with G.GpuCorpus.synthetic(n, dim, 1, SEED, row0=row0) as c:
        res = c.search(q, k, G.L2)
    assert res.indices.min() >= row0 and res.indices.max() < row0 + n
    for qi in (0, 63):
        idx = res.indices[qi].astype(np.int64)
        rows = np.stack([oracle.synth_rows(SEED, int(i), 1, dim, 1)[0] for i in idx])
        sc, _, _ = oracle.scores(rows, 1, 0, q[qi])
        np.testing.assert_allclose(res.scores[qi], sc, rtol=1e-5)
        assert (np.diff(res.scores[qi]) >= 0).all()
        win = oracle.synth_rows(SEED, row0 + 5_000_000, 200_000, dim, 1)
        wsc, _, _ = oracle.scores(win, 1, 0, q[qi])
        kth = float(res.scores[qi, -1])
        better = np.nonzero(wsc.astype(np.float64) < kth * (1 - 1e-5))[0] + row0 + 5_000_000
        assert set(better.tolist()) <= set(idx.tolist())


# ---------------------------------------------------------------------------
# K2 for the narrow types (scan_mfma16.hip): Int8 on i32 MFMA (bit-exact),
# Float16 on f16 MFMA with hi/lo-split queries
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [2, 3])
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(30000, 768, 256, 100), (5000, 100, 33, 10), (300, 64, 300, 500), (9001, 20, 40, 1),
                                   (4000, 33, 64, 16)])
def test_batched_mfma_int8_bit_exact(oracle, dtype, metric, shape):
    """Int8 on the i32 MFMA, UInt8 on the same MFMA shifted by 128 with exact corrections: bit-exact vs the CPU."""
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    with G.GpuCorpus.from_array(rows, index_base=5) as c:
        c.set_scan_path(2)
        res = c.search(q, k, metric)
    osc, oidx, oraw = oracle.search(rows, dtype, metric, q, k, index_base=5)
    assert_exact(res, osc, oidx, oraw)


def test_batched_mfma_uint8_extremes(oracle):
    """All-0 / all-255 rows and queries stress the shift corrections (sums at their extremes)."""
    rng = np.random.default_rng(11)
    rows = rng.integers(0, 256, (6000, 96), dtype=np.uint8)
    rows[0] = 0
    rows[1] = 255
    rows[2, ::2] = 255
    q = rng.integers(0, 256, (40, 96), dtype=np.uint8)
    q[0] = 0
    q[1] = 255
    for metric in (0, 1, 2):
        with G.GpuCorpus.from_array(rows) as c:
            c.set_scan_path(2)
            res = c.search(q, 50, metric)
        osc, oidx, oraw = oracle.search(rows, 3, metric, q, 50)
        assert_exact(res, osc, oidx, oraw)


def test_batched_mfma_int8_ties(oracle):
    rng = np.random.default_rng(2)
    rows = rng.integers(-1, 2, (20000, 48)).astype(np.int8)
    q = rng.integers(-2, 3, (64, 48)).astype(np.int8)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, 200, G.INNER_PRODUCT)
    osc, oidx, oraw = oracle.search(rows, 2, 1, q, 200)
    assert_exact(res, osc, oidx, oraw)


@pytest.mark.parametrize("metric", [1, 2])
@pytest.mark.parametrize("shape", [(20000, 1024, 128, 100), (5000, 100, 33, 10), (300, 72, 130, 400)])
def test_batched_mfma_f16(oracle, metric, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, 1)
    q = oracle.synth_queries(SEED + 1, nq, dim, 1)
    q[0] *= 1000.0      # the per-query power-of-two scaling must make magnitude irrelevant
    q[1] *= 1e-6
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, k, metric)
    rows32 = rows.astype(np.float32)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 1, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("cluster", [300, 3000])
def test_batched_f16_dense_cluster_at_the_kth_rank(oracle, metric, cluster):
    """Float16 rows are SELECTED with one f16 query plane (score error up to 2^-11 |q||x|) and the kept rows
    re-scored exactly.  A cluster of near-identical rows around the k-th rank has score gaps far below that
    error: 300 rows ride inside the margin and are re-scored; 3000 exceed the candidate budget and the query is
    repaired by the streaming kernel.  Either way the answer must be the exact one."""
    n, dim, nq, k = 20000, 192, 40, 50
    rows = oracle.synth_rows(SEED, 0, n, dim, 1).astype(np.float32)
    rng = np.random.default_rng(11)
    base = rng.standard_normal(dim).astype(np.float32)
    where = rng.choice(n, cluster, replace=False)
    rows[where] = base * 0.5 + rng.standard_normal((cluster, dim)).astype(np.float32) * 1.5e-3
    rows = rows.astype(np.float16)
    q = (base[None, :] + rng.standard_normal((nq, dim)).astype(np.float32) * 1e-2).astype(np.float32)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, k, metric)
    rows32 = rows.astype(np.float32)
    members = set(where.tolist())
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 1, metric, q[i])
        assert_float_topk(metric, res.scores[i], res.indices[i], sc, rows32, q[i], k)
        assert set(res.indices[i].tolist()) <= members


# ---------------------------------------------------------------------------
# K2 batched L2 on float spaces: GEMM-form selection with an error margin +
# exact (q-x)^2 re-scoring of the kept candidates
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("shape", [(20000, 768, 130, 100), (5000, 100, 33, 10), (300, 64, 128, 500), (9000, 36, 40, 1)])
def test_batched_l2_float(oracle, dtype, shape):
    n, dim, nq, k = shape
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
    with G.GpuCorpus.from_array(rows, index_base=3) as c:
        c.set_scan_path(2)
        res = c.search(q, k, G.L2)
    rows32 = rows.astype(np.float32)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, dtype, 0, q[i])
        assert_float_topk(0, res.scores[i], res.indices[i], sc, rows32, q[i], k, 3)


@pytest.mark.parametrize("dtype", [0, 1])
def test_batched_l2_near_duplicates_are_exact(oracle, dtype):
    """The GEMM form qq + xx - 2 dot cancels catastrophically for near neighbours (SURVEY.md §7 hard part 2):
    queries that ARE corpus rows, or a hair away from them, must still get distances within 1e-5 of the
    oracle (0 for the exact duplicate) — that is what the exact re-scoring is for."""
    n, dim, nq, k = 30000, 256, 64, 10
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    rng = np.random.default_rng(4)
    pick = rng.choice(n, nq, replace=False)
    q = rows[pick].astype(np.float32)
    q[1::2] += (rng.standard_normal((nq // 2, dim)) * 1e-4).astype(np.float32)   # tiny perturbations
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, k, G.L2)
    assert (res.indices[:, 0] == pick.astype(np.uint64)).all()
    assert (res.scores[0::2, 0] == 0.0).all()
    rows32 = rows.astype(np.float32)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, dtype, 0, q[i])
        assert_float_topk(0, res.scores[i], res.indices[i], sc, rows32, q[i], k)


def test_batched_l2_heterogeneous_norms(oracle):
    """Rows of very different magnitude inflate the error margin (it scales with the largest row norm): more
    candidates are carried or the query is repaired by K1 — the answer must stay exact either way."""
    n, dim, nq, k = 20000, 64, 40, 20
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    rows[::7] *= 300.0
    rows[5] *= 1e4
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_scan_path(2)
        res = c.search(q, k, G.L2)
    for i in range(nq):
        sc, _, _ = oracle.scores(rows, 0, 0, q[i])
        assert_float_topk(0, res.scores[i], res.indices[i], sc, rows, q[i], k)


def test_gather_rows_payload(oracle):
    """mvfgpu_corpus_gather_rows: the ScoredVector.vector payload served from HBM."""
    for dtype, dim in ((0, 13), (1, 40), (2, 7), (3, 100)):
        rows = oracle.synth_rows(SEED, 0, 5000, dim, dtype)
        with G.GpuCorpus.from_array(rows, index_base=10_000) as c:
            idx = np.array([10_000, 14_999, 12_345, 10_001, 12_345], np.uint64)
            got = c.gather_rows(idx)
            assert (got.view(np.uint8) == rows[(idx - 10_000).astype(np.int64)].view(np.uint8)).all()
            pad = c.gather_rows(np.array([0xFFFFFFFFFFFFFFFF, 10_002], np.uint64))
            assert (pad[0].view(np.uint8) == 0).all() and (pad[1].view(np.uint8) == rows[2].view(np.uint8)).all()
            with pytest.raises(E.IndexOutOfBounds):
                c.gather_rows(np.array([15_000], np.uint64))
            with pytest.raises(E.IndexOutOfBounds):
                c.gather_rows(np.array([9_999], np.uint64))
            assert c.gather_rows(np.array([], np.uint64)).shape == (0, dim)
