"""The oracle against the golden vectors and against an
independent numpy float32 model.  CPU only."""
import numpy as np
import pytest


def _rows(golden):
    g = golden["similarity_search_60x4"]
    return np.array(g["rows_bits"], np.uint32).view(np.float32)


def test_known_answers_intended_and_as_written(oracle, golden):
    X = _rows(golden)
    for case in golden["similarity_search_60x4"]["cases"]:
        q = np.array(case["query"], np.float32)
        for farthest, key in ((False, "intended_nearest"), (True, "as_written_farthest")):
            idx, sc = oracle.find_top_k_similar_faithful(X.tobytes(), 60, 4, oracle.F32, q, case["k"], farthest)
            assert idx.tolist() == case[key]["indices"]
            assert sc.view(np.uint32).tolist() == case[key]["score_bits"]


def test_survey_six_digit_values(oracle, golden):
    # the 6-digit values SURVEY.md §8c lists for eyeballing
    X = _rows(golden)
    idx, sc = oracle.find_top_k_similar_faithful(X.tobytes(), 60, 4, 0, np.array([1, 1, 1, 1], np.float32), 5, False)
    assert idx.tolist() == [0, 1, 2, 3, 4]
    np.testing.assert_allclose(sc, [0.0, 0.158114, 0.316228, 0.474342, 0.632456], atol=5e-7)
    idx, sc = oracle.find_top_k_similar_faithful(X.tobytes(), 60, 4, 0, np.array([5, 5, 5, 5], np.float32), 5, True)
    assert idx.tolist() == [19, 43, 42, 41, 40]
    np.testing.assert_allclose(sc, [8.545466, 8.578607, 8.679286, 8.782511, 8.888194], atol=5e-6)
    idx, sc = oracle.find_top_k_similar_faithful(X.tobytes(), 60, 4, 0, np.array([0, 0, 0, 0], np.float32), 5, False)
    np.testing.assert_allclose(sc, [2.0, 2.00624, 2.024846, 2.05548, 2.097618], atol=5e-6)


def test_search_equals_faithful_intended(oracle, golden):
    X = _rows(golden)
    for case in golden["similarity_search_60x4"]["cases"]:
        q = np.array(case["query"], np.float32)
        sc, idx, _ = oracle.search(X, oracle.F32, oracle.L2, q, case["k"])
        assert idx[0].tolist() == case["intended_nearest"]["indices"]
        assert sc[0].view(np.uint32).tolist() == case["intended_nearest"]["score_bits"]


def test_simple_example(oracle, golden):
    g = golden["simple_5x4"]
    rows, q = np.array(g["rows"], np.float32), np.array(g["query"], np.float32)
    sc, _, _ = oracle.scores(rows, oracle.F32, oracle.L2, q)
    assert sc.view(np.uint32).tolist() == g["distance_bits"]
    _, idx, _ = oracle.search(rows, oracle.F32, oracle.L2, q, 1)
    assert int(idx[0, 0]) == g["best_match"]


def test_three_by_four_all_metrics(oracle, golden):
    g = golden["test_space_3x4"]
    rows, q = np.array(g["rows"], np.float32), np.array(g["query"], np.float32)
    for metric, key in ((oracle.L2, "l2_bits"), (oracle.IP, "dot_bits"), (oracle.COS, "cos_bits")):
        sc, _, _ = oracle.scores(rows, oracle.F32, metric, q)
        assert sc.view(np.uint32).tolist() == g[key]


def test_half_conversions(oracle, golden):
    g = golden["half"]
    f32 = np.array(g["f32_bits"], np.uint32).view(np.float32)
    lib = oracle.lib()
    assert [lib.mvfo_f32_to_f16(float(x)) for x in f32] == g["f16_bits"]
    # exhaustive: every f16 bit pattern widens exactly and narrows back
    allh = np.arange(65536, dtype=np.uint16)
    wide = allh.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 7):
        w = lib.mvfo_f16_to_f32(h)
        if np.isnan(wide[h]):
            assert np.isnan(w)
        else:
            assert np.float32(w).view(np.uint32) == wide[h].view(np.uint32)
            assert lib.mvfo_f32_to_f16(w) == h
    rng = np.random.default_rng(3)
    xs = np.concatenate([(rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 6, 4000)).astype(np.float32),
                         np.array([65519.99, 65520.0, 5.9604645e-08, 2.9802322e-08, 2.98024e-08], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.mvfo_f32_to_f16(float(x)) for x in xs], np.uint16)
    assert (got == want).all()


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_c_oracle_matches_numpy_strict_model(oracle, dtype, metric):
    rows = oracle.synth_rows(11, 0, 40, 37, dtype)
    q = oracle.synth_queries(12, 1, 37, dtype)[0]
    sc, _, _ = oracle.scores(rows, dtype, metric, q)
    fn = {0: oracle.np_l2_strict, 1: oracle.np_dot_strict, 2: oracle.np_cos_strict}[metric]
    want = np.array([fn(q, r.astype(np.float32)) for r in rows], np.float32)
    assert (sc.view(np.uint32) == want.view(np.uint32)).all()


@pytest.mark.parametrize("dtype", [2, 3])
def test_integer_scores_exact(oracle, dtype):
    rows = oracle.synth_rows(5, 0, 64, 100, dtype)
    q = oracle.synth_queries(6, 1, 100, dtype)[0]
    r64, q64 = rows.astype(np.int64), q.astype(np.int64)
    _, _, raw = oracle.scores(rows, dtype, oracle.IP, q)
    assert (raw == r64 @ q64).all()
    sc, _, raw = oracle.scores(rows, dtype, oracle.L2, q)
    want = ((r64 - q64) ** 2).sum(1)
    assert (raw == want).all()
    assert (sc == np.sqrt(want.astype(np.float32))).all()
    sc, _, _ = oracle.scores(rows, dtype, oracle.COS, q)
    den = np.sqrt(np.float32((q64 * q64).sum())) * np.sqrt((r64 * r64).sum(1).astype(np.float32))
    assert (sc == ((r64 @ q64).astype(np.float32) / den).astype(np.float32)).all()


def test_order_keys(oracle):
    lib = oracle.lib()
    vals = [-np.inf, -3.5, -0.0, 0.0, 1e-30, 2.0, np.inf]
    keys = [lib.mvfo_key_from_score(float(v), oracle.L2) for v in vals]
    assert keys[2] == keys[3]  # -0.0 == +0.0
    assert keys == sorted(keys)
    assert lib.mvfo_key_from_score(float("nan"), oracle.L2) == 0xFFFFFFFF
    assert lib.mvfo_key_from_score(float("nan"), oracle.IP) == 0xFFFFFFFF
    keys = [lib.mvfo_key_from_score(float(v), oracle.IP) for v in vals]
    assert keys == sorted(keys, reverse=True)
    ints = [-(2 ** 31), -5, 0, 7, 2 ** 31 - 1]
    assert [lib.mvfo_key_from_raw(i, oracle.L2) for i in ints] == sorted(lib.mvfo_key_from_raw(i, oracle.L2) for i in ints)
    assert [lib.mvfo_key_from_raw(i, oracle.IP) for i in ints] == sorted((lib.mvfo_key_from_raw(i, oracle.IP) for i in ints), reverse=True)


def test_ties_break_by_index_and_nan_last(oracle):
    rows = np.zeros((6, 4), np.float32)
    rows[1] = rows[4] = [1, 0, 0, 0]
    rows[2] = [np.nan, 0, 0, 0]
    rows[5] = [0.5, 0, 0, 0]
    q = np.array([1, 0, 0, 0], np.float32)
    sc, idx, _ = oracle.search(rows, oracle.F32, oracle.L2, q, 6)
    assert idx[0].tolist() == [1, 4, 5, 0, 3, 2]
    assert np.isnan(sc[0, 5])
    sc, idx, _ = oracle.search(rows, oracle.F32, oracle.IP, q, 8)
    assert idx[0].tolist()[:6] == [1, 4, 5, 0, 3, 2]
    assert idx[0, 6] == np.uint64(0xFFFFFFFFFFFFFFFF) and sc[0, 6] == -np.inf


def test_cosine_zero_norm_is_zero(oracle):
    rows = np.array([[0, 0, 0], [1, 2, 3]], np.float32)
    sc, _, _ = oracle.scores(rows, oracle.F32, oracle.COS, np.array([1, 1, 1], np.float32))
    assert sc[0] == 0.0 and sc[1] > 0.9
    sc, _, _ = oracle.scores(rows, oracle.F32, oracle.COS, np.zeros(3, np.float32))
    assert (sc == 0.0).all()


def test_faithful_errors_on_int_spaces(oracle):
    with pytest.raises(RuntimeError):  # Vector::as_f32 -> Build error (vector.rs:90)
        oracle.find_top_k_similar_faithful(bytes(16), 4, 4, oracle.I8, np.zeros(4, np.float32), 2, False)


def test_faithful_zip_truncates_to_shorter(oracle):
    rows = np.array([[1, 2, 3, 4], [0, 0, 9, 9]], np.float32)
    idx, sc = oracle.find_top_k_similar_faithful(rows.tobytes(), 2, 4, 0, np.array([0, 0], np.float32), 2, False)
    assert idx.tolist() == [1, 0] and sc[0] == 0.0  # similarity_search.rs:154 zip


def test_merge_equals_global(oracle):
    rng = np.random.default_rng(0)
    for dtype, metric in ((0, 0), (0, 2), (2, 1), (3, 0)):
        rows = oracle.synth_rows(21, 0, 300, 16, dtype)
        q = oracle.synth_queries(22, 3, 16, dtype)
        k = 17
        gs, gi, gr = oracle.search(rows, dtype, metric, q, k)
        cuts = [0, 90, 91, 300]
        parts = [oracle.search(rows[a:b], dtype, metric, q, k, index_base=a) for a, b in zip(cuts[:-1], cuts[1:])]
        ms, mi, mr = oracle.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                                       np.stack([p[2] for p in parts]), metric, dtype)
        assert (mi == gi).all() and (ms.view(np.uint32) == gs.view(np.uint32)).all() and (mr == gr).all()


def test_synth_generator_properties(oracle):
    a = oracle.synth_rows(99, 0, 8, 16, 0)
    b = oracle.synth_rows(99, 3, 5, 16, 0)
    assert (a[3:] == b).all()  # counter-based: any row is recomputable
    assert a.min() >= -1.0 and a.max() < 1.0
    h = oracle.synth_rows(99, 0, 8, 16, 1)
    assert (h == a.astype(np.float16)).all()
    i8, u8 = oracle.synth_rows(99, 0, 8, 16, 2), oracle.synth_rows(99, 0, 8, 16, 3)
    assert (i8.view(np.uint8) == u8).all()
