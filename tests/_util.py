"""Shared comparison helpers for the parity tests."""
import numpy as np

PAD = np.uint64(0xFFFFFFFFFFFFFFFF)

# north_star tolerance: f32 L2 / cosine within 1e-5 relative.  Dot products and
# cosines are compared relative to |q||x| (cancellation makes "relative to the
# value" meaningless near 0) — SURVEY.md §8c.
TOL = 1e-5


_NORM_CACHE = {}


def row_norms(rows_f32):
    """|x| per row in f64, cached per array (the parity loops call this once per query)."""
    key = (id(rows_f32), rows_f32.shape)
    hit = _NORM_CACHE.get(key)
    if hit is None or hit[0] is not rows_f32:
        if len(_NORM_CACHE) > 4:
            _NORM_CACHE.clear()
        hit = (rows_f32, np.sqrt(np.einsum("ij,ij->i", rows_f32, rows_f32, dtype=np.float64)))
        _NORM_CACHE[key] = hit
    return hit[1]


def norms(rows_f32, q_f32, idx=None):
    xn = row_norms(rows_f32)
    return (xn if idx is None else xn[idx]), float(np.linalg.norm(q_f32.astype(np.float64)))


def score_tolerance(metric, oracle_scores, rows_f32, q_f32, idx):
    """Absolute tolerance per returned entry."""
    if metric == 0:
        return TOL * np.maximum(np.abs(oracle_scores), 1e-30)
    if metric == 2:
        return np.full(len(idx), TOL)
    xn, qn = norms(rows_f32, q_f32, idx)
    return TOL * np.maximum(xn * qn, 1e-30)


def assert_float_topk(metric, got_scores, got_idx, all_scores, rows_f32, q_f32, k, index_base=0):
    """GPU top-k of ONE query vs the oracle's score of every row.

    all_scores: oracle f32 score per local row.  Checks: padding, uniqueness,
    per-entry score within tolerance, best-first order, and set equality up to
    rows whose oracle score is within tolerance of the k-th best."""
    n = len(all_scores)
    kk = min(k, n)
    got_idx = np.asarray(got_idx)
    assert (got_idx[kk:] == PAD).all(), "padding indices"
    pad = np.inf if metric == 0 else -np.inf
    assert (got_scores[kk:] == pad).all(), "padding scores"
    li = (got_idx[:kk] - np.uint64(index_base)).astype(np.int64)
    assert ((li >= 0) & (li < n)).all()
    assert len(set(li.tolist())) == kk, "duplicate indices"
    sc = got_scores[:kk].astype(np.float64)
    ref = all_scores[li].astype(np.float64)
    tol = score_tolerance(metric, ref, rows_f32, q_f32, li)
    fin = np.isfinite(ref)
    assert (np.abs(sc[fin] - ref[fin]) <= tol[fin]).all(), f"score mismatch max={np.max(np.abs(sc[fin]-ref[fin]))}"
    sign = 1.0 if metric == 0 else -1.0
    order = sign * sc
    finite_order = order[np.isfinite(order)]
    assert (np.diff(finite_order) >= 0).all(), "not sorted best-first"
    # set equality modulo near-ties at the boundary
    key = sign * all_scores.astype(np.float64)
    key = np.where(np.isnan(key), np.inf, key)
    kth = np.partition(key, kk - 1)[kk - 1] if kk else np.inf
    if metric == 0:
        btol = TOL * max(abs(kth), 1e-30)
    elif metric == 2:
        btol = TOL
    else:
        xn, qn = norms(rows_f32, q_f32)
        xfin = xn[np.isfinite(xn)]           # rows holding Inf / NaN have no meaningful norm
        btol = TOL * float((xfin.max() if xfin.size else 0.0) * qn)
    btol *= 2
    must = set(np.nonzero(key < kth - btol)[0].tolist())
    may = set(np.nonzero(key <= kth + btol)[0].tolist())
    got = set(li.tolist())
    assert must <= got, f"missing clear winners: {sorted(must - got)[:5]}"
    assert got <= may, f"returned clear losers: {sorted(got - may)[:5]}"


def assert_exact(res, osc, oidx, oraw):
    """Integer spaces: bit-exact indices, raw integers and f32 score bits."""
    assert (res.indices == oidx).all()
    assert (res.raw == oraw).all()
    assert (res.scores.view(np.uint32) == osc.view(np.uint32)).all()


def recall_at_k(got_idx, ref_idx):
    """|got ∩ ref| / |ref| over the non-padding entries."""
    got_idx, ref_idx = np.asarray(got_idx), np.asarray(ref_idx)
    hits = total = 0
    for g, r in zip(got_idx, ref_idx):
        rs = set(r[r != PAD].tolist())
        hits += len(set(g[g != PAD].tolist()) & rs)
        total += len(rs)
    return hits / max(total, 1)


# ---------------------------------------------------------------------------------------------------------------------
# The oracle over a WHOLE synthetic corpus: full-size configs are anchored to the oracle's top-k over
# every row, not to sampled windows.  Rows are regenerated on the CPU chunk by chunk (the GPU's generator is checked
# against the same bytes elsewhere), scored by the strict-order oracle (OpenMP over rows) and reduced per query.
# ---------------------------------------------------------------------------------------------------------------------

def oracle_topk_all_rows(oracle, seed, row0, n, dim, dtype, metric, queries, k, chunk=250_000):
    """Exact oracle top-k of every query over rows [row0, row0+n) of the synthetic corpus: chunked oracle.search +
    oracle.merge_topk (merge(top-k per chunk) == top-k(all), selection is by a total order).
    -> (scores f32[nq,k], indices u64[nq,k] (global), raw i32[nq,k])"""
    buf = np.empty((min(chunk, n), dim), oracle.NP_DTYPE[dtype])
    S, I, R = [], [], []
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        rows = oracle.synth_rows(seed, row0 + r0, m, dim, dtype, out=buf)
        sc, idx, raw = oracle.search(rows, dtype, metric, queries, k, index_base=row0 + r0)
        S.append(sc), I.append(idx), R.append(raw)
    return oracle.merge_topk(np.stack(S), np.stack(I), np.stack(R), metric, dtype)


def oracle_scores_all_rows(oracle, seed, row0, n, dim, dtype, metric, queries, chunk=250_000):
    """The oracle's f32 score of EVERY row for every query -> f32[nq, n] (float spaces: what assert_float_topk needs to
    decide which rows are clear winners / clear losers / boundary ties at the k-th rank)."""
    queries = np.atleast_2d(queries)
    out = np.empty((queries.shape[0], n), np.float32)
    buf = np.empty((min(chunk, n), dim), oracle.NP_DTYPE[dtype])
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        rows = oracle.synth_rows(seed, row0 + r0, m, dim, dtype, out=buf)
        for qi, q in enumerate(queries):
            out[qi, r0:r0 + m] = oracle.scores(rows, dtype, metric, q)[0]
    return out
