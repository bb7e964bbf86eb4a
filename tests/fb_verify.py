"""A FlatBuffers VERIFIER for the MVF footer, restated in Python (test infrastructure).

The Rust reader runs the `flatbuffers` crate's verifier on every footer it opens -- `flatbuffers::root::<FileFooter>`
at /root/reference/src/reader.rs:64 and again in `validate_footer_bounds` (:245) -- before any field is read, so a footer
that our hand-written C++ emitter (metrovector_amd/csrc/mvf_file.cpp) produces is only usable by the reference if it is
verifier-clean.  There is no flatbuffers implementation in this image (no flatc, no Python package), hence this
restatement of the crate's published algorithm (flatbuffers 25.2.10, `verifier.rs`, default `VerifierOptions`:
max_depth 64, max_tables 1 000 000, max_apparent_size 2^31, null terminators required) over the schema of
/root/reference/schema/{mvf,core,types,index,extensions}.fbs (field slots in declaration order; no explicit ids).

Rules checked (positions are offsets into the footer slice, as in the crate -- alignment is relative to the slice):
  root      : u32 at 0 in bounds; table at 0 + uoffset
  table     : position 4-aligned and in bounds; vtable = pos - soffset inside the buffer, 2-aligned; vtable length read,
              vtable END 2-aligned (length even) and the whole vtable in bounds; depth <= 64; tables <= 1e6
  field     : slot beyond the vtable's length or 0 = absent (error if `required`); else position = table + voffset
  scalar    : aligned to its size, in bounds
  uoffset   : u32 aligned 4 in bounds, target = pos + value (no wrap)
  vector    : length u32 aligned 4 in bounds; data at pos + 4 aligned to the element; len * size in bounds
  struct vec: element size 40, alignment 8 (DataBlock, schema/core.fbs:7-13)
  string    : byte vector + valid UTF-8 + a NUL right behind the bytes, inside the buffer
  union     : tag and value both present or both absent; known tags verify their table, unknown ones pass (forward
              compatibility, as the generated Rust does)
Beyond the verifier, `check_reader_expectations` restates what reader.rs itself then demands (format_version == 1).
"""
from __future__ import annotations

import struct

MAX_DEPTH, MAX_TABLES, MAX_APPARENT = 64, 1_000_000, 1 << 31


class VerifyError(Exception):
    pass


# ---- schema ---------------------------------------------------------------------------------------------------------
# field = (name, kind, arg, required); kinds: u8 u16 u32 u64 f32 | str | table:<T> | vec:<scalar> | vecstruct:<size,align>
# | vectable:<T> | vecstr | union:<U> (occupies two slots: tag then value)
S = {
    "FileFooter": [("format_version", "u16"), ("vector_spaces", "vectable:VectorSpace", True), ("block_manifest", "vecstruct:40,8", True),
                   ("metadata_columns", "vectable:MetadataColumn"), ("string_heap_block_index", "u32"), ("extensions", "table:Extensions"),
                   ("compatibility_version", "u16"), ("deprecated_fields", "vecstr")],
    "VectorSpace": [("name", "str", True), ("dimension", "u32"), ("total_vectors", "u64"), ("vector_type", "u8"), ("distance_metric", "u8"),
                    ("data_type", "u8"), ("vectors_block_index", "u32"), ("index_type", "union:Index"), ("vector_ids_block_index", "u32"),
                    ("sparse_metadata", "table:SparseMetadata"), ("tombstones", "table:TombstoneInfo")],
    "MetadataColumn": [("name", "str", True), ("data_type", "u8"), ("data_block_index", "u32"), ("null_count", "u64"),
                       ("min_value", "vec:u8"), ("max_value", "vec:u8")],
    "SparseMetadata": [("indices_block_index", "u32"), ("values_block_index", "u32"), ("max_nnz", "u32")],
    "TombstoneInfo": [("format", "u8"), ("data_block_index", "u32"), ("deleted_count", "u64")],
    "FlatIndex": [],
    "IVFIndex": [("num_lists", "u32"), ("centroids_block_index", "u32"), ("lists_block_index", "u32")],
    "HNSWIndex": [("entry_point", "u64"), ("max_connections", "u32"), ("graph_block_index", "u32")],
    "CustomIndex": [("type_name", "str", True), ("config_block_index", "u32")],
    "Extensions": [("extended_types", "table:ExtendedTypes"), ("quantization", "table:QuantizationInfo"),
                   ("complex_metadata", "table:ComplexMetadata"), ("security", "table:SecurityInfo"),
                   ("performance_hints", "table:PerformanceHints"), ("statistics", "table:FileStatistics"),
                   ("custom_extensions", "vectable:CustomExtension")],
    "ExtendedTypes": [("supported_types", "vecstr"), ("type_mappings", "vec:u8")],
    "QuantizationInfo": [("method", "str", True), ("parameters", "vec:u8"), ("codebooks_block_index", "u32"), ("codes_block_index", "u32")],
    "ComplexMetadata": [("array_columns", "vectable:ArrayColumn"), ("nested_columns", "vectable:NestedColumn"), ("map_columns", "vectable:MapColumn")],
    "ArrayColumn": [("name", "str", True), ("element_type", "u8"), ("data_block_index", "u32"), ("offsets_block_index", "u32")],
    "NestedColumn": [("name", "str", True), ("child_schema", "vec:u8"), ("data_block_index", "u32")],
    "MapColumn": [("name", "str", True), ("key_type", "u8"), ("value_type", "u8"), ("keys_block_index", "u32"), ("values_block_index", "u32"),
                  ("offsets_block_index", "u32")],
    "SecurityInfo": [("encryption_algorithm", "str"), ("encrypted_blocks", "vec:u32"), ("key_derivation", "vec:u8")],
    "PerformanceHints": [("memory_layout", "str"), ("prefetch_strategy", "str"), ("cache_hints", "vec:u8")],
    "FileStatistics": [("creation_timestamp", "u64"), ("last_modified", "u64"), ("total_size", "u64"), ("integrity_hash", "vec:u8"),
                       ("vector_quality_score", "f32"), ("index_quality_metrics", "vec:u8"), ("build_tool", "str"), ("build_version", "str")],
    "CustomExtension": [("name", "str", True), ("version", "u16"), ("data_block_index", "u32"), ("metadata", "vec:u8")],
}
UNIONS = {"Index": {1: "FlatIndex", 2: "IVFIndex", 3: "HNSWIndex", 4: "CustomIndex"}}
SCALAR = {"u8": (1, "<B"), "u16": (2, "<H"), "u32": (4, "<I"), "u64": (8, "<Q"), "f32": (4, "<f")}


class Verifier:
    def __init__(self, buf: bytes):
        self.b = bytes(buf)
        self.depth = self.tables = self.apparent = 0

    # -- primitives (verifier.rs: is_aligned / range_in_buffer / in_buffer / get_*) -------------------------------
    def aligned(self, pos, align, what):
        if pos % align:
            raise VerifyError(f"{what}: position {pos} is not {align}-byte aligned")

    def in_range(self, pos, size, what):
        if pos < 0 or pos + size > len(self.b):
            raise VerifyError(f"{what}: range [{pos}, {pos + size}) is outside the {len(self.b)}-byte buffer")
        self.apparent += size
        if self.apparent > MAX_APPARENT:
            raise VerifyError("apparent size exceeds 2^31")

    def scalar(self, pos, kind, what):
        size, fmt = SCALAR[kind]
        self.aligned(pos, size, what)
        self.in_range(pos, size, what)
        return struct.unpack_from(fmt, self.b, pos)[0]

    def uoffset(self, pos, what):
        off = self.scalar(pos, "u32", what + " (uoffset)")
        tgt = pos + off
        if tgt >= 1 << 32:
            raise VerifyError(f"{what}: uoffset wraps")
        return tgt

    def vector_range(self, pos, esize, ealign, what):
        n = self.scalar(pos, "u32", what + " (length)")
        start = pos + 4
        self.aligned(start, ealign, what + " (elements)")
        self.in_range(start, n * esize, what + " (elements)")
        return start, n

    def string(self, pos, what):
        start, n = self.vector_range(pos, 1, 1, what)
        try:
            self.b[start:start + n].decode("utf-8")
        except UnicodeDecodeError as e:
            raise VerifyError(f"{what}: invalid UTF-8 ({e})")
        if start + n >= len(self.b) or self.b[start + n] != 0:
            raise VerifyError(f"{what}: string is not NUL-terminated inside the buffer")

    # -- tables ----------------------------------------------------------------------------------------------------
    def table(self, pos, tname, what):
        what = f"{what}<{tname}>"
        self.aligned(pos, 4, what)
        self.in_range(pos, 4, what)
        soff = struct.unpack_from("<i", self.b, pos)[0]
        vt = pos - soff
        if vt < 0 or vt > len(self.b):
            raise VerifyError(f"{what}: vtable position {vt} is outside the buffer")
        self.aligned(vt, 2, what + " vtable")
        vlen = self.scalar(vt, "u16", what + " vtable length")
        self.aligned(vt + vlen, 2, what + " vtable end")
        self.in_range(vt, vlen, what + " vtable")
        self.depth += 1
        self.tables += 1
        if self.depth > MAX_DEPTH:
            raise VerifyError("depth limit (64) reached")
        if self.tables > MAX_TABLES:
            raise VerifyError("table limit reached")

        def field_pos(slot):
            voff = 4 + 2 * slot
            if voff + 2 > vlen:
                return None
            o = struct.unpack_from("<H", self.b, vt + voff)[0]  # inside the verified vtable range
            return None if o == 0 else pos + o

        slot = 0
        for f in S[tname]:
            name, kind = f[0], f[1]
            required = len(f) > 2 and f[2]
            w = f"{what}.{name}"
            if kind.startswith("union:"):
                kp, vp = field_pos(slot), field_pos(slot + 1)
                slot += 2
                if kp is None and vp is None:
                    if required:
                        raise VerifyError(f"{w}: required union missing")
                    continue
                if kp is None or vp is None:
                    raise VerifyError(f"{w}: inconsistent union (tag {'absent' if kp is None else 'present'}, value {'absent' if vp is None else 'present'})")
                tag = self.scalar(kp, "u8", w + " tag")
                member = UNIONS[kind[6:]].get(tag)
                tgt = self.uoffset(vp, w)
                if member is not None:
                    self.table(tgt, member, w)
                continue
            fp = field_pos(slot)
            slot += 1
            if fp is None:
                if required:
                    raise VerifyError(f"{w}: required field missing")
                continue
            if kind in SCALAR:
                self.scalar(fp, kind, w)
            elif kind == "str":
                self.string(self.uoffset(fp, w), w)
            elif kind.startswith("table:"):
                self.table(self.uoffset(fp, w), kind[6:], w)
            elif kind.startswith("vec:"):
                size = SCALAR[kind[4:]][0]
                self.vector_range(self.uoffset(fp, w), size, size, w)
            elif kind.startswith("vecstruct:"):
                size, align = (int(x) for x in kind[10:].split(","))
                self.vector_range(self.uoffset(fp, w), size, align, w)
            elif kind.startswith("vectable:") or kind == "vecstr":
                start, n = self.vector_range(self.uoffset(fp, w), 4, 4, w)
                for i in range(n):
                    tgt = self.uoffset(start + 4 * i, f"{w}[{i}]")
                    if kind == "vecstr":
                        self.string(tgt, f"{w}[{i}]")
                    else:
                        self.table(tgt, kind[9:], f"{w}[{i}]")
            else:
                raise AssertionError(kind)
        self.depth -= 1


def verify_footer(footer: bytes) -> None:
    """flatbuffers::root::<FileFooter>(footer) -- raises VerifyError where the crate returns InvalidFlatbuffer."""
    v = Verifier(footer)
    v.table(v.uoffset(0, "root"), "FileFooter", "root")


def footer_of(image: bytes) -> bytes:
    """The footer slice of a whole .mvf image, with reader.rs:225-243's bounds arithmetic."""
    if len(image) < 12 or image[:4] != b"MVF1" or image[-4:] != b"MVF1":
        raise VerifyError("not an MVF image (magic / size)")
    flen = struct.unpack_from("<I", image, len(image) - 8)[0]
    if flen + 8 > len(image) - 4:
        raise VerifyError("Invalid footer length")
    return image[len(image) - 8 - flen:len(image) - 8]


def check_reader_expectations(footer: bytes) -> None:
    """What reader.rs demands after the verifier passed: format_version() == 1 (:250; the schema default is 3, so the
    field must be stored)."""
    root = struct.unpack_from("<I", footer, 0)[0]
    vt = root - struct.unpack_from("<i", footer, root)[0]
    vlen = struct.unpack_from("<H", footer, vt)[0]
    o = struct.unpack_from("<H", footer, vt + 4)[0] if vlen >= 6 else 0
    ver = struct.unpack_from("<H", footer, root + o)[0] if o else 3
    if ver != 1:
        raise VerifyError(f"format_version is {ver}: the reference reader accepts 1 only")


def verify_image(image: bytes) -> None:
    f = footer_of(image)
    verify_footer(f)
    check_reader_expectations(f)
