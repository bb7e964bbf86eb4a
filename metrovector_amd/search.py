"""find_top_k_similar — the GPU drop-in for the reference's hot loop
(examples/similarity_search.rs:140-176), same name and argument meaning.

    reference:  find_top_k_similar(space: &VectorSpace, query: &[f32], k) -> Vec<ScoredVector>
    here:       find_top_k_similar(space, query, k) -> list[ScoredVector]

Differences, all deliberate and documented in DESIGN.md §3:
  * returns the k NEAREST (the reference as written keeps the k farthest,
    SURVEY.md F5; its comments and examples/simple.rs:90 intend nearest);
  * the metric defaults to the space's stored `distance_metric()` (the
    reference ignores it and always computes L2);
  * a query whose length differs from the space's dimension raises
    DimensionMismatch (the reference's zip silently truncates);
  * Int8/UInt8 spaces are searchable (the reference errors in as_f32).
The scan itself runs in libmvf_gpu.so; nothing here computes a distance.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .errors import BuildError, InvalidVectorType
from .gpu import COSINE, INNER_PRODUCT, L2, GpuCorpus, query_dtype_code
from .reader import VectorSpace

_NP_OF = {0: np.float32, 1: np.float16, 2: np.int8, 3: np.uint8}


@dataclass
class ScoredVector:
    """reference examples/similarity_search.rs:14-19"""
    index: int
    score: float
    vector: np.ndarray  # decoded like Vector::as_f32 for float spaces; raw ints for Int8/UInt8


def upload_space(space: VectorSpace, device: int = 0, first: int = 0, count: int | None = None,
                 prepare_batched: bool = False, verify_checksum: bool = False) -> GpuCorpus:
    """HBM-resident copy of rows [first, first+count) of a space, via the
    reference's own hand-off: map_vector_range(..).as_ptr() + stride + count
    (vector_space.rs:155-188, mem.rs:75-77).  The space's deletions and vector ids (schema/core.fbs:35-39, :54) travel
    with it; compressed blocks and Sparse spaces are refused.  `prepare_batched`: norms / f16 shadow are built chunk
    by chunk beside the copy.  `verify_checksum`: the file's CRC32s (what the reference's validate_with_checksum
    leaves as todo!(), src/reader.rs:220) are checked on a second thread WHILE the rows upload; a mismatch raises
    CorruptedData and nothing stays resident."""
    if int(space.vector_type()) != 0:
        raise InvalidVectorType("Invalid vector type: expected Dense, got Sparse")
    total = space.total_vectors()
    if count is None:
        count = total - first
    sl = space.map_vector_range(first, count)
    ids = space.vector_ids()
    tomb = space.tombstone_bitmap()
    check_err: list[BaseException] = []
    th = None
    if verify_checksum:
        import threading

        def _check():
            try:
                space._reader.validate_with_checksum()  # ctypes releases the GIL: runs beside the upload
            except BaseException as e:  # noqa: BLE001 - re-raised below
                check_err.append(e)

        th = threading.Thread(target=_check)
        th.start()
    corpus = None
    try:
        corpus = GpuCorpus.from_pointer(sl.as_ptr(), sl.count, space.dimension(), int(space.data_type()), sl.stride,
                                        device=device, index_base=first, prepare_batched=prepare_batched)
        if tomb is not None:
            corpus.set_tombstones(tomb, first_bit=first)
        if ids is not None:
            corpus.set_vector_ids(ids[first:first + count])
    finally:
        if th is not None:
            th.join()
    if check_err:
        corpus.close()
        raise check_err[0]
    return corpus


def find_top_k_similar(space: VectorSpace, query, k: int, metric: int | None = None, corpus: GpuCorpus | None = None,
                       device: int = 0) -> list[ScoredVector]:
    if metric is None:
        metric = int(space.distance_metric())
    if metric not in (L2, INNER_PRODUCT, COSINE):
        raise BuildError(f"Unsupported distance metric {metric}")
    own = corpus is None
    if own:
        corpus = upload_space(space, device)
    try:
        qd = _NP_OF[query_dtype_code(int(space.data_type()))]
        res, vec = corpus.search_fetch(np.asarray(query, dtype=qd), k, metric)  # the k best + their payload straight from HBM
        valid = res.indices[0] != np.uint64(0xFFFFFFFFFFFFFFFF)  # fewer than k rows: the reference returns fewer items
        rows = vec[0][valid]
    finally:
        if own:
            corpus.close()
    dt = int(space.data_type())
    out = []
    for idx, score, row in zip(res.indices[0][valid], res.scores[0][valid], rows):
        payload = row.astype(np.float32) if dt in (0, 1) else row.copy()  # as Vector::as_f32 for float spaces
        out.append(ScoredVector(int(idx), float(score), payload))
    return out


def find_top_k_similar_batch(space: VectorSpace, queries, k: int, metric: int | None = None,
                             corpus: GpuCorpus | None = None, device: int = 0,
                             with_vectors: bool = False) -> list[list[ScoredVector]]:
    """The batched form: one call for many queries ([nq, dimension]) -- on large spaces two or more queries take the
    MFMA path (include/mvf_gpu.h).  Same semantics per query as `find_top_k_similar`; the row payloads are fetched only
    when `with_vectors` is set (nq * k rows)."""
    if metric is None:
        metric = int(space.distance_metric())
    if metric not in (L2, INNER_PRODUCT, COSINE):
        raise BuildError(f"Unsupported distance metric {metric}")
    dt = int(space.data_type())
    q = np.asarray(queries, dtype=_NP_OF[query_dtype_code(dt)])
    if q.ndim != 2:
        raise BuildError("queries must be a 2-D array [nq, dimension]")
    own = corpus is None
    if own:
        corpus = upload_space(space, device)
    try:
        if with_vectors:
            res, vec = corpus.search_fetch(q, k, metric)
        else:
            res, vec = corpus.search(q, k, metric), None
        valid = res.indices != np.uint64(0xFFFFFFFFFFFFFFFF)
        rows = vec[valid] if with_vectors else None
    finally:
        if own:
            corpus.close()
    out, r = [], 0
    for i in range(q.shape[0]):
        hits = []
        for idx, score in zip(res.indices[i][valid[i]], res.scores[i][valid[i]]):
            payload = None
            if rows is not None:
                payload = rows[r].astype(np.float32) if dt in (0, 1) else rows[r].copy()
                r += 1
            hits.append(ScoredVector(int(idx), float(score), payload))
        out.append(hits)
    return out
