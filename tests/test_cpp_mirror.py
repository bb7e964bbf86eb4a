"""include/mvf.hpp -- the C++ mirror of the reference's host API (MvfReader, VectorSpace, Vector, MvfBuilder, BuiltMvf,
MvfError, ScoredVector, find_top_k_similar) over the two C ABIs -- through examples/cpp/similarity_search.cpp, which follows
the reference example (examples/similarity_search.rs:78-138) line for line in its flow."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "cpp", "similarity_search.cpp")
LIBDIR = os.path.join(ROOT, "metrovector_amd")


def _compile(tmp_path):
    exe = str(tmp_path / "similarity_search_cpp")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-lmvf_gpu", "-lmvf_host", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_cpp_mirror_host_half_and_error_variants(tmp_path):
    """Builder -> file -> reader -> space -> vector, and the reference's error variants by name (src/errors.rs:8-40):
    IndexOutOfBounds past total_vectors (vector_space.rs:102-107), VectorSpaceNotFound (reader.rs:104-119), Io for a
    missing file, DimensionMismatch with the reference's message for ragged input (builder.rs:168-173)."""
    exe = _compile(tmp_path)
    out = subprocess.run([exe, str(tmp_path / "cpp_example.mvf"), "--host-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "space embeddings: 60 vectors x 4, version 1, 1 space(s)" in out.stdout
    assert "vector 25: [5.5, 4.5, 5.25, 4.75]" in out.stdout
    assert "get_vector(60) -> IndexOutOfBounds" in out.stdout
    assert "vector_space(nope) -> VectorSpaceNotFound" in out.stdout
    assert "open(missing) -> Io" in out.stdout
    assert "add_vectors(ragged) -> DimensionMismatch: Dimension mismatch: expected 4, got 3" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_reproduces_the_reference_example(tmp_path):
    """find_top_k_similar(&space, &query, k) of the mirror on the example's dataset and its four queries: the intended
    nearest rows and distances of tests/golden/known_answers.json, payloads fetched from HBM."""
    exe = _compile(tmp_path)
    out = subprocess.run([exe, str(tmp_path / "cpp_example.mvf")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["similarity_search_60x4"]
    lines = {l.split(":")[0]: l for l in out.stdout.splitlines() if l.startswith("query ")}
    for c, case in enumerate(golden["cases"][:4]):
        want = case["intended_nearest"]
        got = lines[f"query {c}"].split(":", 1)[1].split()
        idx = [int(t.split(":")[0]) for t in got]
        sc = [struct.unpack("<f", struct.pack("<I", int(t.split(":")[1], 16)))[0] for t in got]
        assert idx == want["indices"]
        ws = [struct.unpack("<f", struct.pack("<I", b))[0] for b in want["score_bits"]]
        np.testing.assert_allclose(sc, ws, rtol=1e-5, atol=1e-7)
    assert "1. Vector 0 (distance: 0.000): [1, 1, 1, 1]" in out.stdout
    assert "short query -> DimensionMismatch" in out.stdout
