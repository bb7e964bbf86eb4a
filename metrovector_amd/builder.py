"""MvfBuilder / BuiltMvf mirror (reference src/builder.rs:44-559) over
libmvf_host.so.  Used to produce real .mvf files for tests, fixtures and the
bench; the encode rules (f32 LE bits, f16 RNE) live in the C++ library."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .errors import InvalidArgument

QUIRK_TOTAL_VECTORS_DIV4 = 1  # reproduce builder.rs:476


class BuiltMvf:
    """reference src/builder.rs:395-400"""

    def __init__(self, builder: "MvfBuilder", quirks: int = 0):
        self._b, self._quirks = builder, quirks

    def save(self, path) -> None:  # builder.rs:408-411
        _lib.host_check(_lib.host().mvf_builder_save(self._b._h, os.fsencode(path), self._quirks))

    def to_bytes(self) -> bytes:  # builder.rs:417-558
        p, n = C.c_void_p(), C.c_uint64()
        _lib.host_check(_lib.host().mvf_builder_to_bytes(self._b._h, self._quirks, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p.value, n.value)
        finally:
            _lib.host().mvf_free(p)


class MvfBuilder:
    """reference src/builder.rs:44-51"""

    def __init__(self):
        h = C.c_void_p()
        _lib.host_check(_lib.host().mvf_builder_new(C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h is not None and self._h.value:
                _lib.host().mvf_builder_free(self._h)
        except Exception:
            pass
        self._h = None

    def version(self) -> int:  # builder.rs:98-100
        return 1

    def add_vector_space(self, name: str, dimension: int, vector_type: int, distance_metric: int,
                         data_type: int) -> int:  # builder.rs:113-135
        idx = C.c_uint64()
        _lib.host_check(_lib.host().mvf_builder_add_vector_space(self._h, name.encode(), dimension, int(vector_type),
                                                                 int(distance_metric), int(data_type), C.byref(idx)))
        return idx.value

    def add_vectors(self, space_name: str, vectors) -> None:
        """builder.rs:151-196: values are taken as f32 (`T: Into<f32>`) and
        encoded per the space's dtype; Int8/UInt8 spaces raise BuildError."""
        a = np.ascontiguousarray(vectors, dtype=np.float32)
        if a.size == 0:
            a = a.reshape(0, 0)
        if a.ndim != 2:
            raise InvalidArgument("vectors must be a sequence of equal-length vectors")
        _lib.host_check(_lib.host().mvf_builder_add_vectors_f32(self._h, space_name.encode(), a.ctypes.data_as(C.c_void_p),
                                                                a.shape[0], a.shape[1]))

    def add_vectors_raw(self, space_name: str, rows: np.ndarray) -> None:
        """EXTENSION: rows already in the space's storage dtype (the only way to
        write Int8/UInt8 spaces; the reference cannot)."""
        a = np.ascontiguousarray(rows)
        if a.ndim != 2:
            raise InvalidArgument("rows must be 2-D")
        _lib.host_check(_lib.host().mvf_builder_add_vectors_raw(self._h, space_name.encode(), a.ctypes.data_as(C.c_void_p),
                                                                a.shape[0], a.shape[1]))

    def reserve_vectors(self, space_name: str, n_vectors: int) -> None:
        """EXTENSION: room for n_vectors rows up front (a multi-GB block appended in pieces is not re-copied)."""
        _lib.host_check(_lib.host().mvf_builder_reserve_vectors(self._h, space_name.encode(), int(n_vectors)))

    def set_vector_ids(self, space_name: str, ids) -> None:
        """EXTENSION: one u64 id per row (the reference's builder has the field, builder.rs:61, but no setter)."""
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        _lib.host_check(_lib.host().mvf_builder_set_vector_ids(self._h, space_name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def set_tombstones(self, space_name: str, fmt: int, payload: bytes, deleted_count: int) -> None:
        """EXTENSION: a deletion block -- fmt 1 Bitmap (bit per row position, LSB first), 2 SortedList (ascending
        u64 LE ids); TombstoneFormat, schema/types.fbs:35-39."""
        buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload) if payload else None
        _lib.host_check(_lib.host().mvf_builder_set_tombstones(self._h, space_name.encode(), int(fmt), buf, len(payload),
                                                               int(deleted_count)))

    def add_metadata_column(self, name: str, data_type: int, values: bytes) -> None:  # builder.rs:211-236
        buf = (C.c_uint8 * len(values)).from_buffer_copy(values) if values else None
        _lib.host_check(_lib.host().mvf_builder_add_metadata_column(self._h, name.encode(), int(data_type), buf, len(values)))

    def build(self, quirks: int = 0) -> BuiltMvf:  # builder.rs:241-308
        return BuiltMvf(self, quirks)
