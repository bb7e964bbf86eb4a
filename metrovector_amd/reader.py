"""Host-side mirror of the reference's reader API over libmvf_host.so.

  MvfReader    reference src/reader.rs:27-289
  VectorSpace  reference src/vectors/vector_space.rs:34-318
  Vector       reference src/vectors/vector.rs:28-207
  VectorSlice  reference src/vectors/mem.rs:24-187

Same names, argument meaning and error behaviour (errors.py mirrors
MvfError).  All byte-level work — mmap, footer parse, bounds checks — is in
the C++ library; this module holds no format logic of its own.
"""
from __future__ import annotations

import ctypes as C
import enum
import os

import numpy as np

from . import _lib
from .errors import InvalidArgument


class DataType(enum.IntEnum):  # schema/types.fbs:3-11
    Float32 = 0
    Float16 = 1
    Int8 = 2
    UInt8 = 3
    UInt32 = 4
    UInt64 = 5
    StringRef = 6


class VectorType(enum.IntEnum):  # schema/types.fbs:14-17
    Dense = 0
    Sparse = 1


class DistanceMetric(enum.IntEnum):  # schema/types.fbs:20-25
    L2 = 0
    InnerProduct = 1
    Cosine = 2
    Custom = 255


_NP_OF = {0: np.float32, 1: np.float16, 2: np.int8, 3: np.uint8}


def _alive(owner) -> None:
    """Views borrow the reader's mapping (VectorSpace<'a>, Vector<'a> in the reference): once the reader is
    closed they dangle.  Rust rejects that at compile time; here it is a Python error instead of a segfault."""
    if owner is not None and getattr(owner, "_h", None) is None:
        raise InvalidArgument("the MvfReader this view borrows from has been closed")


def _enum(cls, v):
    try:
        return cls(v)
    except ValueError:
        return v


class Vector:
    """A single row, zero-copy (reference src/vectors/vector.rs:28-33)."""

    def __init__(self, ptr: int, nbytes: int, dimension: int, data_type: int, owner):
        self._ptr, self._nbytes, self._dim, self._dt, self._owner = ptr, nbytes, dimension, data_type, owner

    def dimension(self) -> int:  # vector.rs:51
        return self._dim

    def data_type(self):  # vector.rs:56
        return _enum(DataType, self._dt)

    def as_bytes(self) -> bytes:  # vector.rs:61
        _alive(self._owner)
        return C.string_at(self._ptr, self._nbytes)

    def as_f32(self) -> np.ndarray:
        """vector.rs:71-92: Float32/Float16 decode; anything else raises
        BuildError("Cannot convert to f32")."""
        _alive(self._owner)
        n = C.c_uint64(0)
        _lib.host_check(_lib.host().mvf_vector_as_f32(C.c_void_p(self._ptr), self._nbytes, self._dt, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        _lib.host_check(_lib.host().mvf_vector_as_f32(C.c_void_p(self._ptr), self._nbytes, self._dt,
                                                      out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out

    def as_slice(self, dtype) -> np.ndarray:  # vector.rs:104-119 (zero-copy typed view)
        _alive(self._owner)
        dt = np.dtype(dtype)
        if self._nbytes % dt.itemsize:
            from .errors import CorruptedData
            raise CorruptedData("Invalid vector data alignment")
        buf = (C.c_uint8 * self._nbytes).from_address(self._ptr)
        a = np.frombuffer(buf, dtype=dt)
        a.flags.writeable = False
        self._keep = buf
        return a


class VectorSlice:
    """Contiguous multi-row view (reference src/vectors/mem.rs:24-30) — the
    hand-off the GPU boundary consumes: as_ptr / stride / count / dtype."""

    def __init__(self, cs: _lib.CVectorSlice, owner):
        self._cs, self._owner = cs, owner

    def as_ptr(self) -> int:  # mem.rs:75-77
        _alive(self._owner)
        return self._cs.data or 0

    @property
    def stride(self) -> int:
        return self._cs.stride

    @property
    def count(self) -> int:
        return self._cs.count

    @property
    def element_type(self):
        return _enum(DataType, self._cs.data_type)

    def to_numpy(self, dimension: int) -> np.ndarray:
        """Zero-copy [count, dimension] view of the mapped rows (valid while the reader is open)."""
        _alive(self._owner)
        dt = np.dtype(_NP_OF[self._cs.data_type])
        nbytes = self._cs.count * self._cs.stride
        if nbytes == 0:
            return np.empty((0, dimension), dt)
        buf = (C.c_uint8 * nbytes).from_address(self._cs.data)
        a = np.frombuffer(buf, dtype=dt).reshape(self._cs.count, dimension)
        a.flags.writeable = False
        self._keep = buf
        return a


class VectorSpace:
    """A named collection of vectors (reference src/vectors/vector_space.rs:34-39)."""

    def __init__(self, cs: _lib.CVectorSpace, reader: "MvfReader"):
        self._cs, self._reader = cs, reader

    def name(self) -> str:  # :62
        return C.string_at(self._cs.name, self._cs.name_len).decode("utf-8")

    def dimension(self) -> int:  # :67
        return self._cs.dimension

    def total_vectors(self) -> int:  # :72
        return self._cs.total_vectors

    def vector_type(self):  # :77
        return _enum(VectorType, self._cs.vector_type)

    def distance_metric(self):  # :82
        return _enum(DistanceMetric, self._cs.distance_metric)

    def data_type(self):  # :87
        return _enum(DataType, self._cs.data_type)

    def get_vector(self, index: int) -> Vector:  # :101-142
        if index < 0:
            raise InvalidArgument("index must be >= 0")
        _alive(self._reader)
        p, n = C.c_void_p(), C.c_uint64()
        _lib.host_check(_lib.host().mvf_space_get_vector(C.byref(self._cs), index, C.byref(p), C.byref(n)))
        return Vector(p.value, n.value, self._cs.dimension, self._cs.data_type, self._reader)

    def map_vector_range(self, start: int, count: int) -> VectorSlice:  # :155-188
        _alive(self._reader)
        out = _lib.CVectorSlice()
        _lib.host_check(_lib.host().mvf_space_map_vector_range(C.byref(self._cs), start, count, C.byref(out)))
        return VectorSlice(out, self._reader)

    # ---- vector ids / deletions (schema/core.fbs:54, :56, :35-39; layouts: include/mvf_file.h) -------------------
    def vector_ids(self) -> np.ndarray | None:
        """The space's id per row (u64), or None when positions are the ids (vector_ids_block_index = 0)."""
        _alive(self._reader)
        p, n = C.c_void_p(), C.c_uint64()
        _lib.host_check(_lib.host().mvf_space_vector_ids(C.byref(self._cs), C.byref(p), C.byref(n)))
        if not p.value:
            return None
        return np.frombuffer(C.string_at(p.value, n.value * 8), dtype="<u8").copy()

    def deleted_count(self) -> int:
        return self._cs.tombstone_deleted_count if self._cs.has_tombstones else 0

    def tombstone_bitmap(self) -> np.ndarray | None:
        """Deleted row POSITIONS as a bitmap (bit r & 7 of byte r >> 3), whichever on-disk format the space uses;
        None when nothing is deleted."""
        _alive(self._reader)
        if not self._cs.has_tombstones:
            return None
        out = np.zeros((self._cs.total_vectors + 7) // 8, np.uint8)
        dead = C.c_uint64()
        _lib.host_check(_lib.host().mvf_space_tombstone_bitmap(C.byref(self._cs), out.ctypes.data_as(C.c_void_p), out.size,
                                                               C.byref(dead)))
        return out if dead.value else None

    def clone_concurrent(self) -> "VectorSpace":  # :194-201
        cs = _lib.CVectorSpace()
        C.memmove(C.byref(cs), C.byref(self._cs), C.sizeof(cs))
        return VectorSpace(cs, self._reader)


class MvfReader:
    """Reader for MVF files (reference src/reader.rs:27-31); mmap-backed."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    @classmethod
    def open(cls, path) -> "MvfReader":  # reader.rs:45-79
        h = C.c_void_p()
        _lib.host_check(_lib.host().mvf_reader_open(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_bytes(cls, data: bytes) -> "MvfReader":
        h = C.c_void_p()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if len(data) else None
        _lib.host_check(_lib.host().mvf_reader_open_bytes(buf, len(data), C.byref(h)))
        return cls(h.value)

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.host().mvf_reader_close(self._h)
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

    def version(self) -> int:  # :82
        v = C.c_uint16()
        _lib.host_check(_lib.host().mvf_reader_version(self._h, C.byref(v)))
        return v.value

    def num_vector_spaces(self) -> int:  # :87
        n = C.c_uint64()
        _lib.host_check(_lib.host().mvf_reader_num_vector_spaces(self._h, C.byref(n)))
        return n.value

    def vector_space_names(self) -> list[str]:  # :92
        out = []
        for i in range(self.num_vector_spaces()):
            p, n = C.c_void_p(), C.c_uint32()
            _lib.host_check(_lib.host().mvf_reader_vector_space_name(self._h, i, C.byref(p), C.byref(n)))
            out.append(C.string_at(p.value, n.value).decode("utf-8"))
        return out

    def vector_space(self, name: str) -> VectorSpace:  # :104-119
        cs = _lib.CVectorSpace()
        _lib.host_check(_lib.host().mvf_reader_vector_space(self._h, name.encode("utf-8"), C.byref(cs)))
        return VectorSpace(cs, self)

    def file_size(self) -> int:  # :122
        n = C.c_uint64()
        _lib.host_check(_lib.host().mvf_reader_file_size(self._h, C.byref(n)))
        return n.value

    def has_metadata(self) -> bool:  # :127
        v = C.c_int()
        _lib.host_check(_lib.host().mvf_reader_has_metadata(self._h, C.byref(v)))
        return bool(v.value)

    def metadata_column_names(self) -> list[str]:  # :132
        n = C.c_uint64()
        _lib.host_check(_lib.host().mvf_reader_num_metadata_columns(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            p, ln = C.c_void_p(), C.c_uint32()
            _lib.host_check(_lib.host().mvf_reader_metadata_column_name(self._h, i, C.byref(p), C.byref(ln)))
            out.append(C.string_at(p.value, ln.value).decode("utf-8"))
        return out

    def blocks(self) -> list[_lib.DataBlock]:
        n = C.c_uint64()
        _lib.host_check(_lib.host().mvf_reader_num_blocks(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            b = _lib.DataBlock()
            _lib.host_check(_lib.host().mvf_reader_block(self._h, i, C.byref(b)))
            out.append(b)
        return out

    def validate(self) -> None:  # :149
        _lib.host_check(_lib.host().mvf_reader_validate(self._h))

    def validate_with_checksum(self) -> None:  # :172 (the reference ends in todo!())
        _lib.host_check(_lib.host().mvf_reader_validate_with_checksum(self._h))
