"""The footers libmvf_host writes pass a FlatBuffers verifier (tests/fb_verify.py: the `flatbuffers` crate's rules,
which the reference reader applies at src/reader.rs:64 and :245 before it reads a single field), and the verifier
restatement itself rejects what the crate rejects.  CPU only.

tests/conftest.py additionally runs `verify_image` over EVERY image a test obtains from `BuiltMvf.to_bytes()` /
`.save()`, so each builder configuration the suite exercises is covered, not only the ones listed here."""
import glob
import os
import struct

import numpy as np
import pytest

import fb_verify as F
from metrovector_amd.builder import MvfBuilder
from metrovector_amd.reader import DataType, DistanceMetric, VectorType


def _image(spaces=1, dtype=DataType.Float32, metadata=0, ids=False, tomb=0, rows=7, dim=5, names=None):
    b = MvfBuilder()
    rng = np.random.default_rng(3)
    for s in range(spaces):
        name = names[s] if names else f"space_{s}"
        b.add_vector_space(name, dim + s, VectorType.Dense, DistanceMetric.Cosine if s % 2 else DistanceMetric.L2, dtype)
        if dtype in (DataType.Int8, DataType.UInt8):
            b.add_vectors_raw(name, rng.integers(0, 100, (rows + s, dim + s)).astype(np.int8 if dtype == DataType.Int8 else np.uint8))
        else:
            b.add_vectors(name, rng.standard_normal((rows + s, dim + s)).astype(np.float32))
        if ids:
            b.set_vector_ids(name, np.arange(rows + s, dtype=np.uint64) * 3 + 1)
        if tomb == 1:
            b.set_tombstones(name, 1, bytes([0b101] + [0] * ((rows + s + 7) // 8 - 1)), 2)
        if tomb == 2:
            b.set_tombstones(name, 2, np.array([1, 4], np.uint64).tobytes(), 2)
    for m in range(metadata):
        b.add_metadata_column(f"col{m}" + "x" * m, DataType.UInt32, np.arange(rows, dtype=np.uint32).tobytes())
    return b.build().to_bytes()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.mvf"))))
def test_golden_files_are_verifier_clean(path):
    F.verify_image(open(path, "rb").read())


@pytest.mark.parametrize("spaces", [1, 2, 5])
@pytest.mark.parametrize("dtype", [DataType.Float32, DataType.Float16, DataType.Int8, DataType.UInt8])
@pytest.mark.parametrize("metadata,ids,tomb", [(0, False, 0), (1, False, 0), (3, True, 0), (0, True, 1), (2, False, 2), (1, True, 2)])
def test_builder_output_is_verifier_clean(spaces, dtype, metadata, ids, tomb):
    img = _image(spaces, dtype, metadata, ids, tomb)
    F.verify_image(img)
    assert len(F.footer_of(img)) % 8 == 0  # a footer that starts 8-aligned inside its own slice ends 8-aligned too


@pytest.mark.parametrize("name", ["a", "ab", "abc", "abcd", "abcde", "späce", "x" * 63, "x" * 64, "x" * 65, ""])
def test_string_lengths_and_utf8(name):
    F.verify_image(_image(2, names=[name, name + "_2"]))


def test_quirk_build_and_empty_spaces_are_verifier_clean():
    b = MvfBuilder()
    b.add_vector_space("h", 8, VectorType.Dense, DistanceMetric.L2, DataType.Float16)
    b.add_vectors("h", np.ones((3, 8), np.float32))
    F.verify_image(b.build(quirks=1).to_bytes())  # builder.rs:476's total_vectors / 4
    e = MvfBuilder()
    e.add_vector_space("empty", 4, VectorType.Dense, DistanceMetric.L2, DataType.Float32)
    F.verify_image(e.build().to_bytes())


# ---- the checker has teeth: what the crate's verifier rejects, this one rejects ---------------------------------------

def _footer_parts(img):
    f = bytearray(F.footer_of(img))
    root = struct.unpack_from("<I", f, 0)[0]
    vt = root - struct.unpack_from("<i", f, root)[0]
    return f, root, vt


def test_rejects_what_the_crate_rejects():
    img = _image(2, metadata=1, ids=True, tomb=1)
    f, root, vt = _footer_parts(img)
    F.verify_footer(bytes(f))

    def broken(mut, match):
        g = bytearray(f)
        mut(g)
        with pytest.raises(F.VerifyError, match=match):
            F.verify_footer(bytes(g))

    broken(lambda g: struct.pack_into("<I", g, 0, len(g) + 8), "outside")                       # root offset past the end
    broken(lambda g: struct.pack_into("<I", g, 0, root + 2), "aligned")                        # table not 4-aligned
    broken(lambda g: struct.pack_into("<i", g, root, -(1 << 30)), "outside")                    # vtable far away
    broken(lambda g: struct.pack_into("<H", g, vt, struct.unpack_from("<H", g, vt)[0] + 1), "aligned")  # odd vtable length
    broken(lambda g: struct.pack_into("<H", g, vt, 60000), "outside")                            # vtable runs off the buffer
    broken(lambda g: struct.pack_into("<H", g, vt + 4 + 2 * 1, 0), "required field missing")     # vector_spaces slot zeroed
    broken(lambda g: struct.pack_into("<H", g, vt + 4 + 2 * 2, 0), "required field missing")     # block_manifest slot zeroed
    # a field offset that points outside the buffer
    broken(lambda g: struct.pack_into("<H", g, vt + 4, 65000), "outside")
    # vector_spaces: length blown up
    vs_field = root + struct.unpack_from("<H", f, vt + 6)[0]
    vs = vs_field + struct.unpack_from("<I", f, vs_field)[0]
    broken(lambda g: struct.pack_into("<I", g, vs, 1 << 28), "outside")
    # first VectorSpace: name loses its NUL / becomes invalid UTF-8 / goes missing
    sp = vs + 4 + struct.unpack_from("<I", f, vs + 4)[0]
    svt = sp - struct.unpack_from("<i", f, sp)[0]
    name_field = sp + struct.unpack_from("<H", f, svt + 4)[0]
    name = name_field + struct.unpack_from("<I", f, name_field)[0]
    nlen = struct.unpack_from("<I", f, name)[0]
    broken(lambda g: g.__setitem__(name + 4 + nlen, 0x41), "NUL")
    broken(lambda g: g.__setitem__(name + 4, 0xFF), "UTF-8")
    broken(lambda g: struct.pack_into("<H", g, svt + 4, 0), "required field missing")
    # union (the builder writes FlatIndex like builder.rs:464-467): a tag without a value, a value without a tag
    assert struct.unpack_from("<H", f, svt)[0] >= 4 + 2 * 9
    tag_slot, val_slot = svt + 4 + 2 * 7, svt + 4 + 2 * 8
    assert struct.unpack_from("<H", f, tag_slot)[0] != 0 and struct.unpack_from("<H", f, val_slot)[0] != 0
    assert f[sp + struct.unpack_from("<H", f, tag_slot)[0]] == 1  # Index::FlatIndex
    broken(lambda g: struct.pack_into("<H", g, val_slot, 0), "inconsistent union")
    broken(lambda g: struct.pack_into("<H", g, tag_slot, 0), "inconsistent union")
    g = bytearray(f)  # both absent: fine (an index-less space), unknown tags pass like in the generated Rust
    struct.pack_into("<H", g, val_slot, 0)
    struct.pack_into("<H", g, tag_slot, 0)
    F.verify_footer(bytes(g))
    g = bytearray(f)
    g[sp + struct.unpack_from("<H", f, tag_slot)[0]] = 77
    F.verify_footer(bytes(g))
    # truncation anywhere is caught, never an IndexError / struct.error
    for cut in range(0, len(f), 3):
        with pytest.raises(F.VerifyError):
            F.verify_footer(bytes(f[:cut]))


def test_reader_expectation_format_version():
    img = _image(1)
    f, root, vt = _footer_parts(img)
    F.check_reader_expectations(bytes(f))
    g = bytearray(f)
    off = struct.unpack_from("<H", g, vt + 4)[0]
    struct.pack_into("<H", g, root + off, 3)
    with pytest.raises(F.VerifyError, match="format_version"):
        F.check_reader_expectations(bytes(g))
