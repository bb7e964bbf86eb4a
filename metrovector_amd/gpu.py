"""GpuCorpus — Python face of one HBM-resident row-range shard
(`mvfgpu_corpus`, include/mvf_gpu.h).

The search replaces the reference's `find_top_k_similar`
(examples/similarity_search.rs:140-176).  Everything numeric happens in
libmvf_gpu.so (HIP, gfx950); this module only marshals pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .errors import BuildError, InvalidArgument

# schema/types.fbs codes
FLOAT32, FLOAT16, INT8, UINT8 = 0, 1, 2, 3
L2, INNER_PRODUCT, COSINE = 0, 1, 2

_NP_OF = {FLOAT32: np.float32, FLOAT16: np.float16, INT8: np.int8, UINT8: np.uint8}
_CODE_OF = {np.dtype(np.float32): FLOAT32, np.dtype(np.float16): FLOAT16,
            np.dtype(np.int8): INT8, np.dtype(np.uint8): UINT8}


def query_dtype_code(space_dtype: int) -> int:
    """Float32 queries for Float32/Float16 spaces (Vector::as_f32 widens,
    src/vectors/vector.rs:81-89); Int8/UInt8 spaces take their own type."""
    return FLOAT32 if space_dtype in (FLOAT32, FLOAT16) else space_dtype


def device_count() -> int:
    n = C.c_int(0)
    _lib.gpu_check(_lib.gpu().mvfgpu_device_count(C.byref(n)))
    return n.value


@dataclass
class SearchResult:
    scores: np.ndarray   # f32 [nq, k]
    indices: np.ndarray  # u64 [nq, k]
    raw: np.ndarray      # i32 [nq, k] (exact integer score on Int8/UInt8 spaces)


class GpuCorpus:
    """One shard of a vector space, resident in HBM on one MI355X."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        self._shape = None  # (rows, dimension, data_type): fixed for the life of the handle, asked for once

    # ---- construction ------------------------------------------------------
    @classmethod
    def from_pointer(cls, ptr: int, rows: int, dimension: int, data_type: int, stride_bytes: int,
                     device: int = 0, index_base: int = 0, prepare_batched: bool = False, pinned_staging: bool | None = None,
                     chunk_mib: int = 0) -> "GpuCorpus":
        """What a Rust caller passes: VectorSlice::as_ptr / stride / count
        (src/vectors/mem.rs:75-77, vector_space.rs:155-188).  `prepare_batched` builds the row norms and (Float32
        spaces) the f16 shadow chunk by chunk beside the copy (`mvfgpu_corpus_create_ex`)."""
        h = C.c_void_p()
        flags = _lib.UPLOAD_EAGER_SHADOW if prepare_batched else 0
        if pinned_staging is not None:  # None: the library's choice (pinned staging from 256 MiB up)
            flags |= _lib.UPLOAD_PINNED_STAGING if pinned_staging else _lib.UPLOAD_PAGEABLE
        opts = _lib.UploadOptions(C.sizeof(_lib.UploadOptions), flags, chunk_mib, 0)
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_create_ex(C.c_void_p(ptr), rows, dimension, data_type, stride_bytes,
                                                          device, index_base, C.byref(opts), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_array(cls, rows: np.ndarray, device: int = 0, index_base: int = 0, **upload) -> "GpuCorpus":
        if rows.ndim != 2:
            raise InvalidArgument("rows must be a 2-D array")
        code = _CODE_OF.get(rows.dtype)
        if code is None:
            raise BuildError("Unsupported vector data type")
        if rows.shape[0] and rows.strides[1] != rows.itemsize:
            rows = np.ascontiguousarray(rows)
        stride = rows.strides[0] if rows.shape[0] else rows.shape[1] * rows.itemsize
        return cls.from_pointer(rows.ctypes.data, rows.shape[0], rows.shape[1], code, stride, device, index_base, **upload)

    @classmethod
    def synthetic(cls, rows: int, dimension: int, data_type: int, seed: int, row0: int = 0,
                  device: int = 0) -> "GpuCorpus":
        h = C.c_void_p()
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_create_synthetic(rows, dimension, data_type, seed, row0, device,
                                                                 C.byref(h)))
        return cls(h.value)

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.gpu().mvfgpu_corpus_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- deletions / ids (schema/core.fbs:35-39, :54) ---------------------------
    def set_tombstones(self, bitmap: np.ndarray | None, first_bit: int = 0) -> None:
        """Mask deleted rows: bit (first_bit + r) of `bitmap` (uint8, LSB first) = local row r is deleted."""
        if bitmap is None:
            _lib.gpu_check(_lib.gpu().mvfgpu_corpus_set_tombstones(self._h, None, 0, 0))
            return
        b = np.ascontiguousarray(bitmap, np.uint8)
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_set_tombstones(self._h, b.ctypes.data_as(C.c_void_p), first_bit, b.size * 8))

    def set_vector_ids(self, ids: np.ndarray | None) -> None:
        """Report ids[row] instead of index_base + row."""
        if ids is None:
            _lib.gpu_check(_lib.gpu().mvfgpu_corpus_set_vector_ids(self._h, None, 0))
            return
        a = np.ascontiguousarray(ids, np.uint64)
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_set_vector_ids(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    # ---- introspection -------------------------------------------------------
    def info(self) -> _lib.CorpusInfo:
        out = _lib.CorpusInfo()
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_get_info(self._h, C.byref(out)))
        return out

    def _fixed(self) -> tuple[int, int, int]:
        if self._shape is None:
            inf = self.info()
            self._shape = (inf.rows, inf.dimension, inf.data_type)
        return self._shape

    @property
    def rows(self) -> int:
        return self._fixed()[0]

    @property
    def dimension(self) -> int:
        return self._fixed()[1]

    @property
    def data_type(self) -> int:
        return self._fixed()[2]

    def read_rows(self, first: int, count: int) -> np.ndarray:
        _, dim, dt = self._fixed()
        out = np.empty((count, dim), _NP_OF[dt])
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_read_rows(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def gather_rows(self, indices) -> np.ndarray:
        """Rows by GLOBAL index, in the order given (`mvfgpu_corpus_gather_rows`): the payload of the
        reference's ScoredVector.vector, served from HBM."""
        _, dim, dt = self._fixed()
        idx = np.ascontiguousarray(np.asarray(indices).reshape(-1), np.uint64)
        out = np.empty((idx.size, dim), _NP_OF[dt])
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_gather_rows(self._h, idx.ctypes.data_as(C.c_void_p), idx.size,
                                                            out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- search ----------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int, metric: int = L2) -> SearchResult:
        """Host-buffer search (`mvfgpu_search`)."""
        q = np.asarray(queries)
        if q.ndim == 1:
            q = q[None, :]
        qcode = _CODE_OF.get(q.dtype)
        if qcode is None:
            raise BuildError(f"unsupported query dtype {q.dtype}")
        q = np.ascontiguousarray(q)
        nq, qdim = q.shape
        sc = np.empty((nq, k), np.float32)
        idx = np.empty((nq, k), np.uint64)
        raw = np.empty((nq, k), np.int32)
        _lib.gpu_check(_lib.gpu().mvfgpu_search(self._h, metric, q.ctypes.data_as(C.c_void_p), qcode, qdim, nq, k,
                                                sc.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                raw.ctypes.data_as(C.c_void_p)))
        return SearchResult(sc, idx, raw)

    def search_fetch(self, queries: np.ndarray, k: int, metric: int = L2) -> tuple[SearchResult, np.ndarray]:
        """Search + payload in one call (`mvfgpu_search_fetch`): the results and the rows they name, [nq, k, dimension] in
        the stored type (zero rows behind a short result list)."""
        q = np.asarray(queries)
        if q.ndim == 1:
            q = q[None, :]
        qcode = _CODE_OF.get(q.dtype)
        if qcode is None:
            raise BuildError(f"unsupported query dtype {q.dtype}")
        q = np.ascontiguousarray(q)
        nq, qdim = q.shape
        _, dim, dt = self._fixed()
        sc = np.empty((nq, k), np.float32)
        idx = np.empty((nq, k), np.uint64)
        raw = np.empty((nq, k), np.int32)
        vec = np.zeros((nq, k, dim), _NP_OF[dt])  # zeros, not empty: the library writes min(k, rows) rows per query; untouched pages stay unmapped
        _lib.gpu_check(_lib.gpu().mvfgpu_search_fetch(self._h, metric, q.ctypes.data_as(C.c_void_p), qcode, qdim, nq, k,
                                                      sc.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                      raw.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p)))
        return SearchResult(sc, idx, raw), vec

    def search_device(self, d_queries: int, query_dtype: int, query_dim: int, nq: int, k: int, metric: int,
                      d_scores: int, d_indices: int, d_raw: int = 0, stream: int = 0) -> None:
        """Device-pointer search (`mvfgpu_search_device`), asynchronous on `stream`."""
        _lib.gpu_check(_lib.gpu().mvfgpu_search_device(self._h, metric, C.c_void_p(d_queries), query_dtype, query_dim,
                                                       nq, k, C.c_void_p(d_scores), C.c_void_p(d_indices),
                                                       C.c_void_p(d_raw) if d_raw else None,
                                                       C.c_void_p(stream) if stream else None))

    # ---- profiling -------------------------------------------------------------
    def set_profiling(self, enabled: bool) -> None:
        _lib.gpu_check(_lib.gpu().mvfgpu_set_profiling(self._h, int(enabled)))

    def last_timing(self) -> _lib.Timing:
        t = _lib.Timing()
        _lib.gpu_check(_lib.gpu().mvfgpu_last_timing(self._h, C.byref(t)))
        return t

    def set_scan_path(self, path: int) -> None:
        _lib.gpu_check(_lib.gpu().mvfgpu_set_scan_path(self._h, path))

    def reload_tuning(self) -> None:
        """Re-read the MVF_* tuning switches of the environment (they are read once, when the handle is created)."""
        _lib.gpu_check(_lib.gpu().mvfgpu_corpus_reload_tuning(self._h))


class ShardSet:
    """Several GPUs in ONE process (`mvfgpu_shardset_*`): per-shard searches on every device, one packed RCCL
    all-gather of the top-k lists, merge on the first shard's device.  The shards are borrowed."""

    def __init__(self, shards: list[GpuCorpus]):
        self._shards = list(shards)  # keeps them alive
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        h = C.c_void_p()
        _lib.gpu_check(_lib.gpu().mvfgpu_shardset_create(arr, len(shards), C.byref(h)))
        self._h = h

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.gpu().mvfgpu_shardset_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self) -> _lib.ShardsetInfo:
        out = _lib.ShardsetInfo()
        _lib.gpu_check(_lib.gpu().mvfgpu_shardset_get_info(self._h, C.byref(out)))
        return out

    def last_timing(self) -> _lib.ShardsetTiming:
        out = _lib.ShardsetTiming()
        _lib.gpu_check(_lib.gpu().mvfgpu_shardset_last_timing(self._h, C.byref(out)))
        return out

    def search(self, queries: np.ndarray, k: int, metric: int = L2) -> SearchResult:
        q = np.asarray(queries)
        if q.ndim == 1:
            q = q[None, :]
        qcode = _CODE_OF.get(q.dtype)
        if qcode is None:
            raise BuildError(f"unsupported query dtype {q.dtype}")
        q = np.ascontiguousarray(q)
        nq, qdim = q.shape
        sc = np.empty((nq, k), np.float32)
        idx = np.empty((nq, k), np.uint64)
        raw = np.empty((nq, k), np.int32)
        _lib.gpu_check(_lib.gpu().mvfgpu_shardset_search(self._h, metric, q.ctypes.data_as(C.c_void_p), qcode, qdim, nq, k,
                                                         sc.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                         raw.ctypes.data_as(C.c_void_p)))
        return SearchResult(sc, idx, raw)


def merge_topk_host(scores: np.ndarray, indices: np.ndarray, raw: np.ndarray | None, metric: int,
                    data_type: int) -> SearchResult:
    """Merge per-shard results [nlists, nq, k] (host) — `mvfgpu_merge_topk_host`."""
    scores = np.ascontiguousarray(scores, np.float32)
    indices = np.ascontiguousarray(indices, np.uint64)
    nl, nq, k = scores.shape
    if raw is not None:
        raw = np.ascontiguousarray(raw, np.int32)
    sc = np.empty((nq, k), np.float32)
    idx = np.empty((nq, k), np.uint64)
    rw = np.empty((nq, k), np.int32)
    _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_host(
        scores.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
        raw.ctypes.data_as(C.c_void_p) if raw is not None else None, nl, nq, k, metric, data_type,
        sc.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), rw.ctypes.data_as(C.c_void_p)))
    return SearchResult(sc, idx, rw)
