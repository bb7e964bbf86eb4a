"""ctypes front-end of the CPU ORACLE (oracle/mvf_oracle.c) + an independent
numpy float32 strict-order restatement used to cross-check the C code.

TEST INFRASTRUCTURE ONLY — never imported by metrovector_amd/ (the product).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

PARITY STATUS: "parity unpinned" — the Rust reference cannot be built or run
in this image; the oracle is pinned by hand-derived known answers only
(tests/golden/).

Reference lines restated: examples/similarity_search.rs:140-176,
src/vectors/vector_space.rs:101-142, src/vectors/vector.rs:71-92.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmvf_oracle.so")

F32, F16, I8, U8 = 0, 1, 2, 3
L2, IP, COS = 0, 1, 2
UINT64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)

NP_DTYPE = {F32: np.float32, F16: np.float16, I8: np.int8, U8: np.uint8}
QUERY_DTYPE = {F32: np.float32, F16: np.float32, I8: np.int8, U8: np.uint8}


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "mvf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def cpus_granted() -> dict:
    """What the cgroup / affinity mask grants this process: {"cpus_allowed": affinity mask size, "cgroup_cpu_quota":
    cpu.max quota in CPUs or None, "granted": the smaller of the two (>= 1)}."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, period = open(path).read().split()
            if q != "max":
                quota = int(q) / int(period)
        except (OSError, ValueError):
            pass
    granted = allowed if quota is None else min(allowed, max(1, int(quota)))
    return {"cpus_allowed": allowed, "cgroup_cpu_quota": quota, "granted": max(1, granted)}


def _cpu_budget() -> int:
    """Threads the CHECKER's OpenMP loops use: the CPU quota of this container / box (cgroup cpu.max or the affinity
    mask), at most 16.  Oversubscribing a quota-limited box with one OpenMP team per core turns every parallel region
    into milliseconds of spinning."""
    return max(1, min(16, cpus_granted()["granted"]))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _lib = C.CDLL(_LIB_PATH)
        vp, u64, u32, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_int
        _lib.mvfo_f32_to_f16.restype = C.c_uint16
        _lib.mvfo_f32_to_f16.argtypes = [C.c_float]
        _lib.mvfo_f16_to_f32.restype = C.c_float
        _lib.mvfo_f16_to_f32.argtypes = [C.c_uint16]
        _lib.mvfo_key_from_score.restype = u32
        _lib.mvfo_key_from_score.argtypes = [C.c_float, u8]
        _lib.mvfo_key_from_raw.restype = u32
        _lib.mvfo_key_from_raw.argtypes = [C.c_int32, u8]
        _lib.mvfo_scores.restype = i32
        _lib.mvfo_scores.argtypes = [vp, u64, u32, u8, u64, u8, vp, vp, vp, vp]
        _lib.mvfo_topk_from_keys.restype = i32
        _lib.mvfo_topk_from_keys.argtypes = [vp, u64, u32, vp]
        _lib.mvfo_search.restype = i32
        _lib.mvfo_search.argtypes = [vp, u64, u32, u8, u64, u8, vp, u32, u32, u64, vp, vp, vp]
        _lib.mvfo_merge_topk.restype = i32
        _lib.mvfo_merge_topk.argtypes = [vp, vp, vp, u32, u32, u32, u8, u8, vp, vp, vp]
        _lib.mvfo_find_top_k_similar_faithful.restype = i32
        _lib.mvfo_find_top_k_similar_faithful.argtypes = [vp, u64, u64, u32, u8, vp, u32, u32, i32, vp, vp, vp]
        _lib.mvfo_search_best_effort_f32.restype = i32
        _lib.mvfo_search_best_effort_f32.argtypes = [vp, u64, u32, u8, vp, u32, u64, i32, vp, vp]
        _lib.mvfo_synth_rows.restype = None
        _lib.mvfo_synth_rows.argtypes = [u64, u64, u64, u32, u8, vp]
    return _lib


def set_threads(n: int) -> int:
    """Size of the checker's OpenMP team from now on (`omp_set_num_threads` of the libgomp the oracle is linked against).
    torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks; a rank that checks a whole shard against the oracle
    asks for its share of the host's cores here.  Returns the size set."""
    lib()
    n = max(1, int(n))
    C.CDLL("libgomp.so.1").omp_set_num_threads(n)
    return n


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _rows2d(rows: np.ndarray, dtype: int) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=NP_DTYPE[dtype])
    assert rows.ndim == 2
    return rows


def scores(rows: np.ndarray, dtype: int, metric: int, query: np.ndarray):
    """-> (scores f32[n], keys u32[n], raw i32[n])"""
    rows = _rows2d(rows, dtype)
    n, dim = rows.shape
    q = np.ascontiguousarray(query, dtype=QUERY_DTYPE[dtype])
    assert q.shape == (dim,)
    sc = np.empty(n, np.float32)
    keys = np.empty(n, np.uint32)
    raw = np.empty(n, np.int32)
    rc = lib().mvfo_scores(_ptr(rows), n, dim, dtype, rows.strides[0] if n else dim * rows.itemsize,
                           metric, _ptr(q), _ptr(sc), _ptr(keys), _ptr(raw))
    if rc != 0:
        raise RuntimeError(f"mvfo_scores rc={rc}")
    return sc, keys, raw


def search(rows: np.ndarray, dtype: int, metric: int, queries: np.ndarray, k: int, index_base: int = 0):
    """-> (scores f32[nq,k], idx u64[nq,k], raw i32[nq,k])"""
    rows = _rows2d(rows, dtype)
    n, dim = rows.shape
    q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE[dtype])
    if q.ndim == 1:
        q = q[None, :]
    nq = q.shape[0]
    assert q.shape[1] == dim
    sc = np.empty((nq, k), np.float32)
    idx = np.empty((nq, k), np.uint64)
    raw = np.empty((nq, k), np.int32)
    rc = lib().mvfo_search(_ptr(rows), n, dim, dtype, rows.strides[0] if n else dim * rows.itemsize,
                           metric, _ptr(q), nq, k, index_base, _ptr(sc), _ptr(idx), _ptr(raw))
    if rc != 0:
        raise RuntimeError(f"mvfo_search rc={rc}")
    return sc, idx, raw


def merge_topk(scores_l: np.ndarray, idx_l: np.ndarray, raw_l: np.ndarray | None, metric: int, dtype: int):
    """scores_l/idx_l/raw_l: [nlists, nq, k] -> merged ([nq,k], [nq,k], [nq,k])"""
    scores_l = np.ascontiguousarray(scores_l, np.float32)
    idx_l = np.ascontiguousarray(idx_l, np.uint64)
    nl, nq, k = scores_l.shape
    if raw_l is not None:
        raw_l = np.ascontiguousarray(raw_l, np.int32)
    sc = np.empty((nq, k), np.float32)
    idx = np.empty((nq, k), np.uint64)
    raw = np.empty((nq, k), np.int32)
    rc = lib().mvfo_merge_topk(_ptr(scores_l), _ptr(idx_l), _ptr(raw_l), nl, nq, k, metric, dtype,
                               _ptr(sc), _ptr(idx), _ptr(raw))
    if rc != 0:
        raise RuntimeError(f"mvfo_merge_topk rc={rc}")
    return sc, idx, raw


def search_best_effort_f32(rows: np.ndarray, metric: int, query: np.ndarray, k: int, threads: int, index_base: int = 0):
    """NOT THE CHECKER: the host's best effort (mvf_cpu_best_effort.c) -- bench.py's cpu_baseline_best_effort leg only."""
    rows = np.ascontiguousarray(rows, np.float32)
    n, dim = rows.shape
    q = np.ascontiguousarray(query, np.float32).reshape(-1)
    assert q.size == dim
    sc = np.empty(k, np.float32)
    idx = np.empty(k, np.uint64)
    rc = lib().mvfo_search_best_effort_f32(_ptr(rows), n, dim, metric, _ptr(q), k, index_base, threads, _ptr(sc), _ptr(idx))
    if rc != 0:
        raise RuntimeError(f"mvfo_search_best_effort_f32 rc={rc}")
    return sc, idx


def find_top_k_similar_faithful(block: bytes | np.ndarray, total_vectors: int, dim: int, dtype: int,
                                query: np.ndarray, k: int, farthest: bool):
    """Literal restatement of examples/similarity_search.rs:140-176.
    -> (idx u64[m], scores f32[m]) sorted ascending by score."""
    buf = np.frombuffer(bytes(block), np.uint8) if not isinstance(block, np.ndarray) else block.view(np.uint8).reshape(-1)
    buf = np.ascontiguousarray(buf)
    q = np.ascontiguousarray(query, np.float32)
    idx = np.empty(k, np.uint64)
    sc = np.empty(k, np.float32)
    cnt = C.c_uint32(0)
    rc = lib().mvfo_find_top_k_similar_faithful(_ptr(buf), buf.size, total_vectors, dim, dtype, _ptr(q), q.size,
                                                k, int(farthest), _ptr(idx), _ptr(sc), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"faithful rc={rc}")
    return idx[: cnt.value].copy(), sc[: cnt.value].copy()


def synth_rows(seed: int, row0: int, nrows: int, dim: int, dtype: int, out: np.ndarray | None = None) -> np.ndarray:
    """Rows [row0, row0+nrows) of the synthetic corpus.  `out` (C-contiguous,
    right dtype, >= nrows rows) lets callers reuse one buffer across chunks."""
    if out is None:
        out = np.empty((nrows, dim), NP_DTYPE[dtype])
    else:
        assert out.dtype == NP_DTYPE[dtype] and out.flags.c_contiguous and out.shape[1] == dim and out.shape[0] >= nrows
        out = out[:nrows]
    lib().mvfo_synth_rows(seed, row0, nrows, dim, dtype, _ptr(out))
    return out


def synth_queries(seed: int, nq: int, dim: int, dtype: int) -> np.ndarray:
    """Queries use the generator of the QUERY dtype (f32 for f32/f16 spaces)."""
    qd = F32 if dtype in (F32, F16) else dtype
    return synth_rows(seed, 0, nq, dim, qd)


# --------------------------------------------------------------------------
# Independent numpy float32 restatement (slow; small cases only).  Each numpy
# scalar op on np.float32 rounds to f32, and the loop is the strict j order of
# examples/similarity_search.rs:152-157, so it must agree with the C oracle
# bit for bit.
# --------------------------------------------------------------------------

def np_l2_strict(q: np.ndarray, x: np.ndarray) -> np.float32:
    s = np.float32(0.0)
    for a, b in zip(q.astype(np.float32), x.astype(np.float32)):
        t = np.float32(a - b)
        s = np.float32(s + np.float32(t * t))
    return np.float32(np.sqrt(s))


def np_dot_strict(q: np.ndarray, x: np.ndarray) -> np.float32:
    s = np.float32(0.0)
    for a, b in zip(q.astype(np.float32), x.astype(np.float32)):
        s = np.float32(s + np.float32(a * b))
    return s


def np_cos_strict(q: np.ndarray, x: np.ndarray) -> np.float32:
    den = np.float32(np.sqrt(np_dot_strict(q, q)) * np.sqrt(np_dot_strict(x, x)))
    if not den > 0:
        return np.float32(0.0)
    return np.float32(np_dot_strict(q, x) / den)


def np_find_top_k(rows_f32: np.ndarray, query: np.ndarray, k: int, farthest: bool):
    """numpy model of find_top_k_similar for distinct scores."""
    d = np.array([np_l2_strict(query, r) for r in rows_f32], np.float32)
    order = np.lexsort((np.arange(len(d)), -d if farthest else d))[:k]
    order = order[np.argsort(d[order], kind="stable")]
    return order.astype(np.uint64), d[order]
