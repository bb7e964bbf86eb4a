"""-m gpu tests added in round 4.

Part 1 — the full-size BASELINE configs anchored to the ORACLE'S top-k over EVERY row of the corpus (not to sampled
windows): configs[3] (50M x 768 Int8 dot, 256 queries) bit-exact on indices, raw sums and score bits; a configs[4] shard
(12.5M x 1024 Float16 L2, 1024 queries) and configs[2] (10M x 768 Float32 cosine, 1024 queries; 16 of them, on the three
selection paths) with the tolerance-aware criterion of tests/_util.py (1e-5; boundary ties at the k-th rank may go either
way, everything else must be the oracle's list).  Everything goes through the C ABI (libmvf_gpu.so); the oracle is only
the checker."""
import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import assert_exact, assert_float_topk, oracle_scores_all_rows, oracle_topk_all_rows

pytestmark = pytest.mark.gpu
SEED = 0x4D564631


def test_cfg4_50m_x_768_int8_dot_256_queries_vs_the_oracle_over_all_rows(oracle):
    """BASELINE.json configs[3].  north_star: "bit-exact for Int8/UInt8 dot".  Four queries against the oracle's top-100
    over all 50M rows (indices, exact i32 sums, f32 score bits); eight further queries against the streaming kernel K1
    (another kernel family: v_dot4 lanes + butterfly instead of MFMA tiles + phased candidate lists)."""
    n, dim, nq, k = 50_000_000, 768, 256, 100
    q = oracle.synth_queries(SEED + 1, nq, dim, 2)
    sel = [0, 85, 170, 255]
    k1 = [1, 31, 64, 99, 128, 191, 222, 254]
    with G.GpuCorpus.synthetic(n, dim, 2, SEED) as c:
        res = c.search(q, k, G.INNER_PRODUCT)
        c.set_scan_path(1)
        ref = [c.search(q[qi], k, G.INNER_PRODUCT) for qi in k1]
        c.set_scan_path(0)
    for qi, r in zip(k1, ref):
        assert (r.indices[0] == res.indices[qi]).all() and (r.raw[0] == res.raw[qi]).all()
        assert (r.scores[0].view(np.uint32) == res.scores[qi].view(np.uint32)).all()
    osc, oidx, oraw = oracle_topk_all_rows(oracle, SEED, 0, n, dim, 2, 1, q[sel], k)
    assert_exact(G.SearchResult(res.scores[sel], res.indices[sel], res.raw[sel]), osc, oidx, oraw)


def test_cfg5_shard_12p5m_x_1024_f16_l2_1024_queries_vs_the_oracle_over_all_rows(oracle):
    """BASELINE.json configs[4], one GPU's shard (rows [25M, 37.5M) of the 100M-row corpus) at its batch size: four of the
    1024 queries against the oracle's score of every one of the shard's 12.5M rows."""
    n, dim, nq, k, row0 = 12_500_000, 1024, 1024, 100, 25_000_000
    q = oracle.synth_queries(SEED + 1, nq, dim, 1)
    sel = [0, 341, 700, 1023]
    with G.GpuCorpus.synthetic(n, dim, 1, SEED, row0=row0) as c:
        res = c.search(q, k, G.L2)             # default: int8-shadow selection + exact re-scoring
        c.set_scan_path(3)
        f16 = c.search(q, k, G.L2)             # the f16 MFMA kernel on the stored rows
        c.set_scan_path(0)
    all_sc = oracle_scores_all_rows(oracle, SEED, row0, n, dim, 1, 0, q[sel])
    for j, qi in enumerate(sel):
        assert_float_topk(0, res.scores[qi], res.indices[qi], all_sc[j], None, q[qi], k, index_base=row0)
        assert_float_topk(0, f16.scores[qi], f16.indices[qi], all_sc[j], None, q[qi], k, index_base=row0)


def test_cfg3_10m_x_768_f32_cosine_1024_queries_vs_the_oracle_over_all_rows(oracle):
    """BASELINE.json configs[2]: 16 of the 1024 queries against the oracle's score of every one of the 10M rows, on the
    default path (int8-shadow selection), scan path 3 (f16-shadow selection) and scan path 2 (exact f32 MFMA)."""
    n, dim, nq, k = 10_000_000, 768, 1024, 100
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    sel = [0, 63, 64, 200, 255, 256, 300, 411, 511, 512, 640, 767, 768, 900, 1000, 1023]
    got = {}
    with G.GpuCorpus.synthetic(n, dim, 0, SEED) as c:
        for path in (0, 3, 2):
            c.set_scan_path(path)
            got[path] = c.search(q, k, G.COSINE)
        c.set_scan_path(0)
        one = c.search(q[sel[5]], k, G.COSINE)  # the headline's kernel (K1 on the stored rows) on one of them
    all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, 0, 2, q[sel])
    for j, qi in enumerate(sel):
        for path, res in got.items():
            assert_float_topk(2, res.scores[qi], res.indices[qi], all_sc[j], None, q[qi], k)
    assert_float_topk(2, one.scores[0], one.indices[0], all_sc[5], None, q[sel[5]], k)


# ---------------------------------------------------------------------------------------------------------------------
# Part 2 -- the drop-in on a REAL multi-GB file:
# MvfReader::open is O(footer) for any size (src/reader.rs:45-79), map_vector_range(0, total) hands the whole space over
# (src/vectors/vector_space.rs:155-188).  A two-space file whose first block is > 4 GiB, so that the second block lies at
# an offset beyond 2^32: written by the C++ builder (streamed save), opened, checksum-validated beside the upload,
# searched single and batched.  The rows are the library generator's, so the answers must equal a
# GpuCorpus.synthetic search bit for bit.
# ---------------------------------------------------------------------------------------------------------------------

def _scratch_dir_with(free_bytes, tmp_path):
    """A directory with room for the file: $MVF_TEST_BIGFILE_DIR, pytest's tmp_path, else /dev/shm."""
    import os
    import shutil
    for d in (os.environ.get("MVF_TEST_BIGFILE_DIR"), str(tmp_path), "/dev/shm"):
        if d and os.path.isdir(d) and shutil.disk_usage(d).free >= free_bytes:
            return d
    pytest.skip(f"no scratch directory with {free_bytes / 2**30:.1f} GiB free for the multi-GB .mvf")


def test_multi_gb_mvf_file_open_upload_search(oracle, tmp_path):
    import os
    from metrovector_amd.builder import MvfBuilder
    from metrovector_amd.reader import DataType, DistanceMetric, MvfReader, VectorType
    from metrovector_amd.search import find_top_k_similar, find_top_k_similar_batch, upload_space
    n, dim, k = 1_500_000, 768, 100                      # 4 608 000 000 B of Float32 rows > 2^32
    n8, dim8 = 20_000, 64
    path = os.path.join(_scratch_dir_with(6 << 30, tmp_path), f"mvf_big_{os.getpid()}.mvf")
    try:
        b = MvfBuilder()
        b.add_vector_space("big", dim, VectorType.Dense, DistanceMetric.Cosine, DataType.Float32)
        b.add_vector_space("small_i8", dim8, VectorType.Dense, DistanceMetric.InnerProduct, DataType.Int8)
        b.reserve_vectors("big", n)
        buf = np.empty((250_000, dim), np.float32)
        for r0 in range(0, n, 250_000):
            b.add_vectors_raw("big", oracle.synth_rows(SEED, r0, min(250_000, n - r0), dim, 0, out=buf))
        rows8 = oracle.synth_rows(SEED + 7, 0, n8, dim8, 2)
        b.add_vectors_raw("small_i8", rows8)
        b.build().save(path)
        del b
        assert os.path.getsize(path) > (1 << 32)
        q = oracle.synth_queries(SEED + 1, 64, dim, 0)
        q8 = oracle.synth_queries(SEED + 8, 5, dim8, 2)
        with MvfReader.open(path) as r:
            assert r.file_size() == os.path.getsize(path) and r.num_vector_spaces() == 2
            blocks = r.blocks()
            assert blocks[0].size == n * dim * 4 and blocks[1].offset == n * dim * 4 > (1 << 32)
            big, small = r.vector_space("big"), r.vector_space("small_i8")
            assert big.total_vectors() == n and small.total_vectors() == n8
            # R4 addressing beyond 4 GiB: the last Float32 row and an Int8 row behind it, byte for byte
            assert big.get_vector(n - 1).as_bytes() == oracle.synth_rows(SEED, n - 1, 1, dim, 0).tobytes()
            assert small.get_vector(n8 - 1).as_bytes() == rows8[n8 - 1].tobytes()
            with upload_space(big, verify_checksum=True) as c, G.GpuCorpus.synthetic(n, dim, 0, SEED) as ref:
                assert (c.read_rows(n - 3, 3) == ref.read_rows(n - 3, 3)).all()
                one = find_top_k_similar(big, q[0], k, corpus=c)
                want1 = ref.search(q[0], k, G.COSINE)
                assert [h.index for h in one] == want1.indices[0].tolist()
                assert np.array([h.score for h in one], np.float32).tobytes() == want1.scores[0].tobytes()
                assert (one[0].vector == oracle.synth_rows(SEED, one[0].index, 1, dim, 0)[0]).all()
                many = find_top_k_similar_batch(big, q, k, corpus=c)
                wantb = ref.search(q, k, G.COSINE)
                for qi in range(q.shape[0]):
                    assert [h.index for h in many[qi]] == wantb.indices[qi].tolist()
                    assert np.array([h.score for h in many[qi]], np.float32).tobytes() == wantb.scores[qi].tobytes()
                # and against the oracle over ALL rows, two of the queries
                all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, 0, 2, q[[0, 63]])
                assert_float_topk(2, want1.scores[0], want1.indices[0], all_sc[0], None, q[0], k)
                assert_float_topk(2, wantb.scores[63], wantb.indices[63], all_sc[1], None, q[63], k)
            with upload_space(small, verify_checksum=True) as c8:   # the block at an offset beyond 2^32
                res8 = c8.search(q8, 10, G.INNER_PRODUCT)
            assert_exact(res8, *oracle.search(rows8, 2, 1, q8, 10))
    finally:
        if os.path.exists(path):
            os.remove(path)


# ---------------------------------------------------------------------------------------------------------------------
# Part 3 -- k beyond one pass.  The reference takes any k: usize (examples/similarity_search.rs:143,
# :166-168); a search for more than MVFGPU_K_PER_PASS = 1024 results runs ceil(k / 1024) passes of the streaming kernel, each
# returning the rows ranked strictly behind the last row of the pass before (the floor travels on the device).
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("metric", [G.L2, G.INNER_PRODUCT, G.COSINE])
def test_k_5000_on_200k_rows_vs_the_oracle(oracle, dtype, metric):
    n, dim, k = 200_000, 96, 5000
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, 3, dim, dtype)
    with G.GpuCorpus.from_array(rows, index_base=7_000_000) as c:
        one = c.search(q[0], k, metric)          # one query per pass
        three = c.search(q, k, metric)           # the four-query pass
    osc, oidx, oraw = oracle.search(rows, dtype, metric, q, k, index_base=7_000_000)
    if dtype in (2, 3):                          # integer spaces: bit-exact, ties by row position across the pass boundaries
        assert_exact(three, osc, oidx, oraw)
        assert_exact(one, osc[:1], oidx[:1], oraw[:1])
    else:
        rows32 = rows.astype(np.float32)
        for qi in range(3):
            sc = oracle.scores(rows, dtype, metric, q[qi])[0]
            assert_float_topk(metric, three.scores[qi], three.indices[qi], sc, rows32, q[qi], k, index_base=7_000_000)
        assert (one.indices[0] == three.indices[0]).all() and (one.scores[0].view(np.uint32) == three.scores[0].view(np.uint32)).all()


def test_large_k_ties_deletions_ids_and_exhaustion(oracle):
    """What the pass boundaries must not break: a tie group that straddles rank 1024 (equal keys: the floor is the full
    composite, so the split is by row position), deleted rows, vector ids, k beyond the live rows (the later passes find
    nothing and pad), k = 16384 = MVFGPU_K_BY_PASSES, and batches above the four-query pass."""
    rng = np.random.default_rng(11)
    n, dim = 30_000, 32
    rows = rng.integers(-3, 4, (n, dim)).astype(np.int8)          # few distinct scores: ties everywhere
    q = rng.integers(-3, 4, (6, dim)).astype(np.int8)
    dead = np.zeros(n, bool)
    dead[rng.choice(n, 12_000, replace=False)] = True
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(10**12)
    live = np.nonzero(~dead)[0]
    with G.GpuCorpus.from_array(rows) as c:
        for k in (1025, 2048, 5000, 16384):
            got = c.search(q, k, G.INNER_PRODUCT)
            assert_exact(got, *oracle.search(rows, 2, 1, q, k))
        c.set_tombstones(np.packbits(dead, bitorder="little"))
        c.set_vector_ids(ids)
        k = 16384                                                  # 18 000 live rows: the last passes still find rows
        got = c.search(q, k, G.L2)
        osc, oidx, oraw = oracle.search(rows[live], 2, 0, q, k)
        assert (got.indices == ids[live[oidx.astype(np.int64)]]).all() and (got.raw == oraw).all()
    with G.GpuCorpus.from_array(rows[:3000]) as c:                 # k beyond the rows: 3000 results, then padding
        got = c.search(q[:2], 5000, G.INNER_PRODUCT)
        assert_exact(got, *oracle.search(rows[:3000], 2, 1, q[:2], 5000))
        assert (got.indices[:, 3000:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and (got.scores[:, 3000:] == -np.inf).all()
    with G.GpuCorpus.from_array(rows[:100]) as c:              # beyond what passes serve: the whole-shard sort (test_gpu_large_k.py)
        got = c.search(q[:1], 16385, G.L2)
        assert_exact(got, *oracle.search(rows[:100], 2, 0, q[:1], 16385))


def test_large_k_on_a_batch_the_mfma_path_would_take(oracle):
    """300 queries, k = 1500, Float32 cosine: above one pass the batch is served by passes of the exact streaming kernel
    (four queries each) instead of the MFMA path -- same answers as 300 single searches at k = 1024 on their first 1024."""
    n, dim, nq, k = 50_000, 64, 300, 1500
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    with G.GpuCorpus.from_array(rows) as c:
        got = c.search(q, k, G.COSINE)
        c.set_scan_path(1)
        ref = c.search(q, 1024, G.COSINE)
    assert (got.indices[:, :1024] == ref.indices).all()
    assert (got.scores[:, :1024].view(np.uint32) == ref.scores.view(np.uint32)).all()
    rows32 = rows.astype(np.float32)
    for qi in (0, 150, 299):
        sc = oracle.scores(rows, 0, 2, q[qi])[0]
        assert_float_topk(2, got.scores[qi], got.indices[qi], sc, rows32, q[qi], k)


# ---------------------------------------------------------------------------------------------------------------------
# Part 4 -- the automatic path choice on the device: what the handle reports after a run of searches
# ---------------------------------------------------------------------------------------------------------------------

def test_sane_and_wild_norm_corpora_keep_the_int8_selection(oracle):
    """mvfgpu_corpus_get_info().selection_state: bit 0 = the repair feedback switched the int8-shadow selection off, bit 1 =
    it switched the folded pre-filter off.  Neither may happen on the benchmark's rows nor on rows whose norms span two
    orders of magnitude (cosine / inner product: the multiplicative row term of the bounds carries them) -- after eight
    searches (the feedback consumes samples two searches back) the handle still selects on the int8 shadow."""
    rng = np.random.default_rng(4)
    n, dim, nq, k = 200_000, 64, 300, 25
    wild = (rng.standard_normal((n, dim)) * np.exp(rng.uniform(-2.3, 2.3, n))[:, None]).astype(np.float32)
    sane = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    for rows, metrics in ((sane, (G.COSINE, G.L2, G.INNER_PRODUCT)), (wild, (G.COSINE, G.INNER_PRODUCT))):
        with G.GpuCorpus.from_array(rows) as c:
            c.set_profiling(True)
            for i in range(8):
                c.search(q, k, metrics[i % len(metrics)])
                assert c.last_timing().repaired_queries == 0
            inf = c.info()
            assert inf.selection_state == 0, f"selection_state {inf.selection_state}"
            assert inf.shadows & 1 and c.last_timing().scan_kernel == 6


def test_prefilter_switched_off_by_the_feedback_leaves_the_int8_selection_on(oracle, monkeypatch):
    """ADVICE r3 (medium), on the device.  Tiny candidate regions make nearly every query of a search overflow while the
    folded pre-filter is on (one record per wave region); round 2's epilogue spills a full region into the per-query list
    and needs no repair.  So: the first searches are repaired, the feedback switches the pre-filter off (selection_state
    2) -- and the NEXT sample consumed still comes from a search that ran with it.  Counted against the fresh totals it
    used to switch the int8 selection off as well, one search later and for good (selection_state 3, the f16 selection,
    ~1.6 x slower); it is dropped now: the handle stays at 2, keeps selecting on the int8 shadow and needs no repairs."""
    n, dim, nq, k = 200_000, 64, 300, 20
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, nq, dim, 0)
    monkeypatch.setenv("MVF_K2_REGION_RECORDS", "4096")
    want = oracle.search(rows, 0, 2, q, k)
    with G.GpuCorpus.from_array(rows) as c:
        c.set_profiling(True)
        states, repaired = [], []
        for _ in range(8):
            got = c.search(q, k, G.COSINE)
            states.append(c.info().selection_state)
            repaired.append(c.last_timing().repaired_queries)
            assert (np.sort(got.indices, axis=1) == np.sort(want[1], axis=1)).mean() >= 0.999   # exact on the way, too
        assert repaired[0] > nq // 2, repaired                      # the regions WERE too small for the folded pre-filter
        assert states[0] == 0 and sorted(states) == states and states[-1] == 2, states   # 0 -> 2, never 3
        assert repaired[-1] == 0 and repaired[-2] == 0, repaired
        assert c.last_timing().scan_kernel == 6                     # still the int8 kernel on the int8 shadow


# ---------------------------------------------------------------------------------------------------------------------
# Part 5 -- small searches (BASELINE configs[0] and the reference example's own sizes): the host API reads the query and
# writes the results in pinned host memory (no copy engine), and a small corpus takes the four-query passes of a batch
# in ONE launch (pass = blockIdx.y)
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("metric", [G.L2, G.INNER_PRODUCT, G.COSINE])
def test_small_corpus_batches_take_their_passes_in_one_launch(oracle, dtype, metric):
    """Batches of 2 .. 31 queries on corpora that leave the GPU mostly idle (60 .. 40 000 rows): up to 8 four-query
    passes per launch, the last one padded; against the oracle, with deleted rows and an index base."""
    rng = np.random.default_rng(100 * dtype + metric)
    for (n, dim, k) in ((60, 4, 5), (10_000, 128, 10), (40_000, 40, 100), (3_000, 300, 33)):
        rows = oracle.synth_rows(SEED + n, 0, n, dim, dtype)
        dead = rng.random(n) < 0.1
        live = np.nonzero(~dead)[0]
        with G.GpuCorpus.from_array(rows, index_base=1_000) as c:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
            for nq in (2, 5, 8, 9, 16, 29, 31):
                q = oracle.synth_queries(SEED + 1 + nq, nq, dim, dtype)
                got = c.search(q, k, metric)
                osc, oidx, oraw = oracle.search(rows[live], dtype, metric, q, min(k, len(live)))
                oidx = live[oidx.astype(np.int64)].astype(np.uint64) + np.uint64(1_000)
                kk = osc.shape[1]
                if dtype in (2, 3):
                    assert (got.indices[:, :kk] == oidx).all() and (got.raw[:, :kk] == oraw).all(), (n, nq)
                    assert (got.scores[:, :kk].view(np.uint32) == osc.view(np.uint32)).all(), (n, nq)
                else:
                    rows32 = rows.astype(np.float32)
                    for qi in (0, nq // 2, nq - 1):
                        sc = oracle.scores(rows, dtype, metric, q[qi])[0]
                        sc = np.where(dead, np.inf if metric == G.L2 else -np.inf, sc).astype(np.float32)
                        assert_float_topk(metric, got.scores[qi, :kk], got.indices[qi, :kk], sc, rows32, q[qi], kk, index_base=1_000)
                    one = c.search(q[nq - 1], k, metric)      # the same query through the single-query kernel
                    assert (one.indices[0] == got.indices[nq - 1]).all(), (n, nq)


def test_host_api_in_place_buffers_agree_with_the_copy_path(oracle, monkeypatch):
    """mvfgpu_search with the query read / the results written in pinned host memory (the default for small transfers)
    returns what the staged-copy path returns, bit for bit, on both sides of the size limits; the limits come from the
    handle's tuning (MVF_HOST_ZC_QUERY / MVF_HOST_ZC_RESULTS, bytes)."""
    n, dim = 50_000, 96
    for dtype, metric in ((0, G.COSINE), (1, G.L2), (2, G.INNER_PRODUCT)):
        rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
        with G.GpuCorpus.from_array(rows) as c:
            for nq, k in ((1, 10), (1, 1024), (4, 100), (40, 100), (64, 1024), (200, 10), (1, 5000)):
                q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
                res = []
                for zq, zo in ((None, None), ("0", "0"), ("0", None), (None, "0"), ("100", "100000")):
                    for var, val in (("MVF_HOST_ZC_QUERY", zq), ("MVF_HOST_ZC_RESULTS", zo)):
                        monkeypatch.delenv(var, raising=False) if val is None else monkeypatch.setenv(var, val)
                    c.reload_tuning()
                    res.append(c.search(q, k, metric))
                for r in res[1:]:
                    assert (r.indices == res[0].indices).all() and (r.raw == res[0].raw).all(), (dtype, nq, k)
                    assert (r.scores.view(np.uint32) == res[0].scores.view(np.uint32)).all(), (dtype, nq, k)
            monkeypatch.delenv("MVF_HOST_ZC_QUERY", raising=False)
            monkeypatch.delenv("MVF_HOST_ZC_RESULTS", raising=False)


def test_payload_fetch_in_place_and_by_copy(oracle, monkeypatch):
    """mvfgpu_corpus_gather_rows (the payload of ScoredVector.vector, examples/similarity_search.rs:18): small fetches are
    written straight into pinned host memory, large ones through the device mirror; 16-, 4- and 1-byte copy units."""
    rng = np.random.default_rng(5)
    for dtype, dim in ((0, 96), (1, 6), (2, 7), (3, 64), (0, 771)):
        n = 20_000
        rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
        with G.GpuCorpus.from_array(rows, index_base=500) as c:
            for count in (1, 10, 100, 3000):
                idx = rng.integers(0, n, count).astype(np.uint64)
                for zq, zo in ((None, None), ("0", "0"), ("16", "1000")):
                    for var, val in (("MVF_HOST_ZC_QUERY", zq), ("MVF_HOST_ZC_RESULTS", zo)):
                        monkeypatch.delenv(var, raising=False) if val is None else monkeypatch.setenv(var, val)
                    c.reload_tuning()
                    got = c.gather_rows(idx + np.uint64(500))
                    assert got.dtype == rows.dtype and (got.view(np.uint8) == rows[idx.astype(np.int64)].view(np.uint8)).all(), (dtype, dim, count, zq)
        monkeypatch.delenv("MVF_HOST_ZC_QUERY", raising=False)
        monkeypatch.delenv("MVF_HOST_ZC_RESULTS", raising=False)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
def test_search_fetch_returns_the_rows_the_results_name(oracle, dtype, monkeypatch):
    """mvfgpu_search_fetch = mvfgpu_search + the payload rows (ScoredVector.vector), gathered on the device behind the
    search: same results as mvfgpu_search bit for bit, rows = the stored rows at the reported indices, zero rows behind a
    short list; single queries, the four-query pass, a batch on the MFMA path, k beyond one pass; with an index base,
    deletions and (the host-mapped fallback) vector ids; payloads on both sides of the in-place limit."""
    rng = np.random.default_rng(40 + dtype)
    n, dim = 60_000, 72
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    dead = rng.random(n) < 0.05
    ids = rng.permutation(n).astype(np.uint64) + np.uint64(5 * 10**9)
    pad = np.uint64(0xFFFFFFFFFFFFFFFF)
    for use_ids in (False, True):
        with G.GpuCorpus.from_array(rows, index_base=2_000_000) as c:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
            if use_ids:
                c.set_vector_ids(ids)
            for nq, k in ((1, 10), (3, 100), (40, 50), (1, 3000), (2, 1024)):
                q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
                for limit in (None, "0", "4096"):
                    monkeypatch.delenv("MVF_HOST_ZC_RESULTS", raising=False) if limit is None else monkeypatch.setenv("MVF_HOST_ZC_RESULTS", limit)
                    c.reload_tuning()
                    ref = c.search(q, k, G.COSINE)
                    got, vec = c.search_fetch(q, k, G.COSINE)
                    assert (got.indices == ref.indices).all() and (got.raw == ref.raw).all()
                    assert (got.scores.view(np.uint32) == ref.scores.view(np.uint32)).all()
                    assert vec.shape == (nq, k, dim) and vec.dtype == rows.dtype
                    pos = (np.argsort(ids)[np.searchsorted(np.sort(ids), got.indices)] if use_ids
                           else (got.indices - np.uint64(2_000_000)).astype(np.int64))
                    assert not dead[pos].any()
                    assert (vec.view(np.uint8) == rows[pos].view(np.uint8)).all(), (dtype, use_ids, nq, k, limit)
            monkeypatch.delenv("MVF_HOST_ZC_RESULTS", raising=False)
    with G.GpuCorpus.from_array(rows[:7]) as c:                    # fewer rows than k: zero rows behind the list
        q = oracle.synth_queries(SEED + 1, 2, dim, dtype)
        got, vec = c.search_fetch(q, 12, G.L2)
        assert (got.indices[:, 7:] == pad).all() and not vec[:, 7:].view(np.uint8).any()
        assert (vec[:, :7].view(np.uint8) == rows[got.indices[:, :7].astype(np.int64)].view(np.uint8)).all()


def test_shard_set_in_place_buffers_agree_with_the_copy_path(oracle, monkeypatch):
    """mvfgpu_shardset_search: every shard reads small batches from ONE pinned host buffer and the merge writes the results
    into pinned host memory; same answers as with the staged copies (a set reads MVF_HOST_ZC_* when it is created), and as
    one handle over all the rows."""
    n, dim = 90_000, 64
    for dtype, metric in ((0, G.L2), (2, G.INNER_PRODUCT)):
        rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
        cuts = [0, 20_000, 20_000, 55_001, n]                    # an empty shard among them
        shards = [G.GpuCorpus.from_array(rows[a:b], index_base=a) for a, b in zip(cuts[:-1], cuts[1:])]
        try:
            with G.GpuCorpus.from_array(rows) as whole:
                for nq, k in ((1, 10), (5, 100), (300, 20)):
                    q = oracle.synth_queries(SEED + 1, nq, dim, dtype)
                    ref = whole.search(q, k, metric)
                    res = []
                    for limit in (None, "0"):
                        for var in ("MVF_HOST_ZC_QUERY", "MVF_HOST_ZC_RESULTS"):
                            monkeypatch.delenv(var, raising=False) if limit is None else monkeypatch.setenv(var, limit)
                        with G.ShardSet(shards) as ss:
                            res.append(ss.search(q, k, metric))
                            res.append(ss.search(q, k, metric))    # and again on the warm buffers
                    for r in res:
                        assert (r.indices == res[0].indices).all() and (r.scores.view(np.uint32) == res[0].scores.view(np.uint32)).all()
                    if dtype == 2:
                        assert (res[0].indices == ref.indices).all() and (res[0].raw == ref.raw).all()
                    else:
                        assert (res[0].indices == ref.indices).mean() > 0.99
        finally:
            for var in ("MVF_HOST_ZC_QUERY", "MVF_HOST_ZC_RESULTS"):
                monkeypatch.delenv(var, raising=False)
            for s in shards:
                s.close()
