"""A plain-C consumer of the two C ABIs (examples/c/similarity_search.c): the drop-in boundary exercised the way a
foreign-language binding would — `cc -std=c99`, headers only, shared libraries only — on the reference example's
own dataset and queries (examples/similarity_search.rs:42-76, :104-109)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c", "similarity_search.c")
LIBDIR = os.path.join(ROOT, "metrovector_amd")


def _compile(tmp_path):
    exe = str(tmp_path / "similarity_search_c")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-lmvf_gpu", "-lmvf_host", f"-Wl,-rpath,{LIBDIR}", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_c_consumer_compiles_as_c99_and_refuses_to_run_without_a_gpu(tmp_path):
    """The public headers are C99-clean, both libraries link from C, the host half (builder, reader, checksum
    validation, zero-copy slice) works, and the GPU half fails LOUDLY when there is no device — no CPU fallback."""
    import torch
    exe = _compile(tmp_path)
    out = subprocess.run([exe, str(tmp_path / "c_example.mvf")], capture_output=True, text=True, timeout=120)
    assert "space embeddings: 60 vectors x 4, dtype 0, metric 0" in out.stdout
    if not torch.cuda.is_available():
        assert out.returncode != 0
        assert "no HIP device" in out.stderr and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_c_consumer_reproduces_the_reference_example(tmp_path):
    exe = _compile(tmp_path)
    out = subprocess.run([exe, str(tmp_path / "c_example.mvf")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["similarity_search_60x4"]
    lines = {l.split(":")[0]: l for l in out.stdout.splitlines()}
    for c, case in enumerate(golden["cases"][:3]):
        want = case["intended_nearest"]
        got = lines[f"query {c}"].split(":", 1)[1].split()
        idx = [int(t.split(":")[0]) for t in got]
        sc = [struct.unpack("<f", struct.pack("<I", int(t.split(":")[1], 16)))[0] for t in got]
        assert idx == want["indices"]
        ws = [struct.unpack("<f", struct.pack("<I", b))[0] for b in want["score_bits"]]
        np.testing.assert_allclose(sc, ws, rtol=1e-5, atol=1e-7)
        assert f"best {c}: " in out.stdout
    refusal = out.stdout.split("short query -> ")[1].splitlines()[0]
    assert "mismatch" in refusal.lower()          # DimensionMismatch, not a silent truncation
