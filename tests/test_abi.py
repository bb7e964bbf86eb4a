"""The C-ABI libraries load and export every symbol their headers declare;
without a GPU the compute entry points refuse loudly (no CPU fallback).
CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from metrovector_amd import _lib
from metrovector_amd import errors as E
from metrovector_amd import gpu as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_gpu_library_exports_header_symbols():
    lib = _lib.gpu()
    names = _declared("mvf_gpu.h", "mvfgpu_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libmvf_gpu.so does not export {n}"


def test_host_library_exports_header_symbols():
    lib = _lib.host()
    names = _declared("mvf_file.h", "mvf_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libmvf_host.so does not export {n}"


def test_struct_layouts_match_header():
    # sizes the C compiler gives the headers' structs (gcc, x86-64) against their ctypes mirrors
    want = {"mvfgpu_corpus_info": _lib.CorpusInfo, "mvfgpu_timing": _lib.Timing, "mvfgpu_upload_options": _lib.UploadOptions,
            "mvfgpu_shardset_info": _lib.ShardsetInfo, "mvfgpu_shardset_timing": _lib.ShardsetTiming,
            "mvf_data_block": _lib.DataBlock, "mvf_vector_space": _lib.CVectorSpace, "mvf_vector_slice": _lib.CVectorSlice}
    assert C.sizeof(_lib.CorpusInfo) == 56 and C.sizeof(_lib.Timing) == 80 and C.sizeof(_lib.DataBlock) == 40
    import subprocess
    import tempfile
    body = "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in want)
    src = '#include <stdio.h>\n#include "mvf_gpu.h"\n#include "mvf_file.h"\nint main(void){' + body + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")],
                       check=True)
        out = subprocess.run([os.path.join(d, "s")], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        name, size = line.split()
        assert C.sizeof(want[name]) == int(size), f"ctypes mirror of {name}: {C.sizeof(want[name])} bytes, header: {size}"


def test_strerror_covers_reference_variants():
    lib = _lib.gpu()
    msgs = [lib.mvfgpu_strerror(i).decode() for i in range(13)]
    assert msgs[5] == "Index out of bounds" and msgs[6] == "Dimension mismatch" and msgs[10] == "Build error"
    assert len(set(msgs)) == 13


def _no_gpu():
    return G.device_count() == 0


def test_no_cpu_fallback_without_device():
    if not _no_gpu():
        pytest.skip("a GPU is present")
    rows = np.zeros((4, 4), np.float32)
    with pytest.raises(E.DeviceError, match="no CPU fallback"):
        G.GpuCorpus.from_array(rows)
    with pytest.raises(E.DeviceError):
        G.GpuCorpus.synthetic(10, 4, 0, 1)


def test_argument_validation_precedes_device_use():
    rows = np.zeros((4, 4), np.float32)
    with pytest.raises(E.BuildError, match="Unsupported vector data type"):   # vector_space.rs:126
        G.GpuCorpus.from_pointer(rows.ctypes.data, 4, 4, 6, 16)
    with pytest.raises(E.InvalidArgument):
        G.GpuCorpus.from_pointer(rows.ctypes.data, 4, 0, 0, 16)
    with pytest.raises(E.BuildError, match="33025"):
        G.GpuCorpus.from_pointer(rows.ctypes.data, 1, 40000, 2, 40000)
    with pytest.raises(E.CorruptedData):
        G.GpuCorpus.from_pointer(rows.ctypes.data, 4, 4, 0, 8)                # stride < row bytes
    with pytest.raises(E.InvalidArgument, match="1 GiB"):
        G.GpuCorpus.from_pointer(rows.ctypes.data, 1, 1 << 30, 0, 1 << 32)    # dim * 4 would overflow 32 bits


@pytest.mark.parametrize("dtype,metric", [(0, 0), (0, 1), (0, 2), (2, 1), (3, 0), (2, 2)])
def test_merge_topk_host_matches_oracle(oracle, dtype, metric):
    rows = oracle.synth_rows(31, 0, 500, 24, dtype)
    q = oracle.synth_queries(32, 4, 24, dtype)
    k = 33
    cuts = [0, 10, 170, 171, 500]
    parts = [oracle.search(rows[a:b], dtype, metric, q, k, index_base=a) for a, b in zip(cuts[:-1], cuts[1:])]
    S, I, R = (np.stack([p[j] for p in parts]) for j in range(3))
    got = G.merge_topk_host(S, I, R, metric, dtype)
    ws, wi, wr = oracle.search(rows, dtype, metric, q, k)
    assert (got.indices == wi).all()
    assert (got.scores.view(np.uint32) == ws.view(np.uint32)).all()
    assert (got.raw == wr).all()


def test_merge_topk_host_padding(oracle):
    rows = oracle.synth_rows(1, 0, 5, 8, 0)
    q = oracle.synth_queries(2, 1, 8, 0)
    a = oracle.search(rows[:2], 0, 0, q, 4, index_base=0)
    b = oracle.search(rows[2:], 0, 0, q, 4, index_base=2)
    got = G.merge_topk_host(np.stack([a[0], b[0]]), np.stack([a[1], b[1]]), None, 0, 0)
    ws, wi, _ = oracle.search(rows, 0, 0, q, 4)
    assert (got.indices == wi).all()
    got = G.merge_topk_host(a[0][None], a[1][None], None, 0, 0)
    assert got.indices[0, 2] == np.uint64(0xFFFFFFFFFFFFFFFF) and got.scores[0, 2] == np.inf


def test_abi_version_is_exported_and_checked_at_load():
    lib = _lib.gpu()  # raises ImportError on a mismatch
    assert lib.mvfgpu_abi_version() == _lib.ABI_VERSION
    text = open(os.path.join(ROOT, "include", "mvf_gpu.h")).read()
    assert int(re.search(r"#define MVFGPU_ABI_VERSION (\d+)u", text).group(1)) == _lib.ABI_VERSION


def _feedback(samples):
    a = np.asarray(samples, np.uint32).reshape(-1, 4)
    out = np.zeros(4, np.uint32)
    _lib.gpu_check(_lib.gpu().mvfgpu_selftest_feedback(a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.c_void_p)))
    return dict(seen=int(out[0]), redone=int(out[1]), bias_off=bool(out[2]), qs_off=bool(out[3]))


def test_repair_feedback_blames_the_prefilter_first_and_drops_its_stale_sample():
    """The automatic path choice as a pure function of the sample sequence (no GPU): {queries, repaired, ran with the folded
    pre-filter, selected on the int8 shadow}.  ADVICE r3: two searches are in flight, so when the first bad one switches
    the pre-filter off, the NEXT sample consumed still comes from a search that ran with it -- it must not be counted
    against the fresh totals (it used to switch the int8 selection off for good one search later)."""
    bad_with_bias, good_without = [1024, 600, 1, 1], [1024, 0, 0, 1]
    st = _feedback([bad_with_bias])
    assert st == dict(seen=0, redone=0, bias_off=True, qs_off=False)
    st = _feedback([bad_with_bias, bad_with_bias])            # the stale second sample is dropped
    assert st == dict(seen=0, redone=0, bias_off=True, qs_off=False)
    st = _feedback([bad_with_bias, bad_with_bias] + [good_without] * 5)
    assert st["bias_off"] and not st["qs_off"] and st["redone"] == 0 and st["seen"] == 5 * 1024
    # repairs that persist WITHOUT the pre-filter are the int8 bound's: the selection goes, as before
    st = _feedback([bad_with_bias, bad_with_bias, [1024, 600, 0, 1]])
    assert st["bias_off"] and st["qs_off"]
    # a corpus that never used the pre-filter (Float16 rows on the f16 flavour, MVF_K2_BIAS=0): straight to the selection
    assert _feedback([[1024, 600, 0, 1]]) == dict(seen=1024, redone=600, bias_off=False, qs_off=True)
    # sane data: a handful of repairs in thousands of queries changes nothing, and old history fades (halving at 8192)
    st = _feedback([[1024, 1, 1, 1]] * 20)
    assert not st["bias_off"] and not st["qs_off"] and st["seen"] < 8192 + 1024
    # fewer than four repairs never trigger, whatever the ratio (single streamed queries)
    assert _feedback([[1, 1, 0, 1]] * 3) == dict(seen=3, redone=3, bias_off=False, qs_off=False)
    assert _feedback([[1, 1, 0, 1]] * 4)["qs_off"]


def _schedule(rows, nq, k, int8_selection=1):
    b = (C.c_uint64 * 32)()
    n, g, mask = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    _lib.gpu_check(_lib.gpu().mvfgpu_selftest_schedule(rows, nq, k, int8_selection, b, 32, C.byref(n), C.byref(g), C.byref(mask)))
    return [int(b[i]) for i in range(n.value)], g.value, mask.value


def test_the_phase_schedule_of_a_batched_search_is_a_function_of_its_shape():
    """The geometric phases of a batched search, their growth and where the threshold is refined (no GPU).  Constants from the
    in-process A/B runs of profiles/r05_k2_walk_and_phase_costs.txt (8b-8d): a change has to show up here."""
    # cfg3: 10M rows, 1024 queries, top-100 on the int8 selection: growth 4 laid out backwards from the end, the direct phase
    # under the list capacity, the threshold refined in front of the last TWO phases
    b, g, mask = _schedule(10_000_000, 1024, 100)
    assert g == 4 and b == [2560, 9984, 39168, 156416, 625152, 2500096, 10_000_000]
    assert mask == 0b0110000                                     # behind phases 4 and 5 = in front of the last two
    assert all(x % 256 == 0 for x in b[:-1]) and b[0] <= 8192
    # the last phase is always (1 - 1/g) of the rows
    assert abs((b[-1] - b[-2]) / b[-1] - 0.75) < 1e-3
    # batches of up to 128 queries grow by 6 while a phase's records stay near half of the list capacity (k <= 117 on 8192 slots) ...
    assert [_schedule(1_000_000, nq, 100)[1] for nq in (1, 16, 64, 128, 129, 256)] == [6, 6, 6, 6, 4, 4]
    assert [_schedule(1_000_000, 16, k)[1] for k in (10, 100, 117, 118, 146, 147, 409)] == [6, 6, 6, 5, 5, 4, 4]
    # ... on the exact-key lists (4096 slots) only for small k
    assert [_schedule(1_000_000, 16, k, 0)[1] for k in (10, 58, 59, 100)] == [6, 6, 5, 4]
    # large k: the growth that keeps k (g - 1) + k inside half a list
    assert _schedule(1_000_000, 1024, 1024)[1] == 4 and _schedule(1_000_000, 1024, 1024, 0)[1] == 2
    # a refinement is latency whatever the batch: only in front of a phase of >= 256M query x row pairs and >= 200k rows
    assert _schedule(1_000_000, 16, 100)[2] == 0 and _schedule(1_000_000, 64, 100)[2] == 0
    b, g, mask = _schedule(10_000_000, 64, 100)                  # 64 x 8.3M pairs: the last phase only
    assert g == 6 and mask == 1 << (len(b) - 2)
    b, g, mask = _schedule(10_000_000, 16, 100)                  # 16 x 8.3M = 133M pairs: none
    assert mask == 0
    b, g, mask = _schedule(1_200_000, 400, 100)                  # 400 x 900k = 360M: the last; 400 x 225k = 90M: not the one before
    assert g == 4 and mask == 1 << (len(b) - 2)
    assert _schedule(150_000, 1024, 100)[2] == 0                 # phases under 200k rows: never
    assert _schedule(10_000_000, 1024, 100, 0)[2] == 0           # no int8 selection, no refinement
    # small corpora: one direct phase, or a direct phase and one more
    assert _schedule(5000, 1024, 10) == ([5000], 4, 0)
    b, g, mask = _schedule(20_000, 300, 10)
    assert b[-1] == 20_000 and b[0] <= 8192 and len(b) == 2
    # refused
    lib = _lib.gpu()
    bb = (C.c_uint64 * 4)()
    n, gg, mm = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    assert lib.mvfgpu_selftest_schedule(0, 1, 1, 1, bb, 4, C.byref(n), C.byref(gg), C.byref(mm)) != 0
    assert lib.mvfgpu_selftest_schedule(10, 1, 2000, 1, bb, 4, C.byref(n), C.byref(gg), C.byref(mm)) != 0       # k beyond one pass
    assert lib.mvfgpu_selftest_schedule(10_000_000, 1024, 100, 1, bb, 4, C.byref(n), C.byref(gg), C.byref(mm)) != 0  # buffer too short
    assert lib.mvfgpu_selftest_schedule(10, 1, 1, 1, None, 4, C.byref(n), C.byref(gg), C.byref(mm)) != 0


def test_search_entry_points_refuse_null_arguments_before_touching_a_device():
    """mvfgpu_search / mvfgpu_search_fetch / mvfgpu_corpus_gather_rows: a NULL handle or buffer is MVF_ERR_INVALID_ARGUMENT
    (code 12) with a message, on a box without a GPU too (checked before any HIP call)."""
    import ctypes as C
    lib = _lib.gpu()
    buf = (C.c_float * 16)()
    idx = (C.c_uint64 * 4)()
    rc = lib.mvfgpu_search_fetch(None, 0, buf, 0, 4, 1, 4, buf, idx, None, buf)
    assert rc != 0 and b"NULL" in lib.mvfgpu_last_error_message()
    rc2 = lib.mvfgpu_search(None, 0, buf, 0, 4, 1, 4, buf, idx, None)
    assert rc2 == rc
    assert lib.mvfgpu_search_fetch(None, 0, buf, 0, 4, 1, 4, buf, idx, None, None) == rc      # no payload buffer
    assert lib.mvfgpu_corpus_gather_rows(None, idx, 4, buf) == rc
    with pytest.raises(E.InvalidArgument):
        _lib.gpu_check(rc)


def _route(rows, dim, dtype, metric, nq, k):
    out = C.c_uint32(99)
    _lib.gpu_check(_lib.gpu().mvfgpu_selftest_route(rows, dim, dtype, metric, nq, k, C.byref(out)))
    return out.value


def test_the_route_of_a_search_is_a_function_of_its_shape():
    """Which kernels serve a search (no GPU): 0 = the streaming kernel K1, 1 = the batched MFMA route, 2 = passes of K1 behind a
    floor (k > 1024), 3 = K1 as a dump + the whole-shard sort.  The BASELINE configs, the measured crossovers of
    profiles/r04_small_corpora_crossover.txt (round 5: r05_small_corpora_crossover.txt) and profiles/r04_any_k.txt -- a change of a threshold has to show up here."""
    K1, K2, PASSES, SORT = 0, 1, 2, 3
    F32, F16, I8, U8 = 0, 1, 2, 3
    L2, IP, COS = 0, 1, 2
    # BASELINE.json configs
    assert _route(10_000, 128, F32, L2, 1, 10) == K1                 # configs[0]
    assert _route(10_000_000, 768, F32, COS, 1, 100) == K1           # configs[1]: the headline
    assert _route(10_000_000, 768, F32, COS, 1024, 100) == K2        # configs[2]
    assert _route(50_000_000, 768, I8, IP, 256, 100) == K2           # configs[3]
    assert _route(12_500_000, 1024, F16, L2, 1024, 100) == K2        # a shard of configs[4]
    # small batches: K1 takes four queries per pass, the batched route costs 70-100 us before its first row
    assert [_route(10_000, 128, F32, L2, nq, 10) for nq in (4, 16, 32, 33)] == [K1, K1, K1, K2]          # < 8 MiB: up to 32 queries
    assert [_route(30_000, 128, F32, L2, nq, 10) for nq in (16, 31, 32)] == [K1, K1, K2]                 # < 16 MiB: up to 31
    assert [_route(100_000, 128, F32, L2, nq, 10) for nq in (2, 4, 8, 9, 16)] == [K1, K1, K1, K2, K2]    # 51 MB
    assert [_route(300_000, 128, F32, L2, nq, 10) for nq in (4, 8, 9)] == [K1, K1, K2]                   # 154 MB
    assert [_route(1_000_000, 128, F32, L2, nq, 10) for nq in (4, 5, 8)] == [K1, K2, K2]                 # 512 MB: one pass only
    assert [_route(1_000_000, 768, F32, COS, nq, 100) for nq in (1, 2, 4)] == [K1, K2, K2]               # 3 GB: from two queries on
    assert [_route(300_000, 768, F16, COS, nq, 10) for nq in (4, 5)] == [K1, K2]                         # 461 MB
    assert [_route(10_000_000, 768, F32, COS, nq, 100) for nq in (2, 4, 16)] == [K2, K2, K2]
    # ... unless such a small corpus is MANY short rows: K1's passes cost by the row there (round 5,
    # profiles/r05_small_corpora_crossover.txt: 30k x 128 f16, 32 queries 154 -> 73 us; 100k x 128 int8, 16 queries 145 -> 96)
    assert [_route(30_000, 128, F16, COS, nq, 10) for nq in (8, 12, 16, 32)] == [K1, K1, K2, K2]         # 7.7 MB, >= 24k f16 rows: from 16
    assert [_route(20_000, 128, F16, COS, nq, 10) for nq in (16, 32, 33)] == [K1, K1, K2]                # fewer rows: as before
    assert [_route(100_000, 128, I8, IP, nq, 10) for nq in (8, 11, 12, 16)] == [K1, K1, K2, K2]          # 12.8 MB, >= 64k rows: from 12
    assert [_route(100_000, 32, F32, L2, nq, 10) for nq in (8, 12)] == [K1, K2]                          # 12.8 MB of 128-byte f32 rows
    # Int8 / UInt8 rows: K1's v_dot4 pass holds its own longer
    assert [_route(30_000, 768, I8, IP, nq, 10) for nq in (8, 16, 17)] == [K1, K1, K2]                   # 23 MB of 768-byte rows
    assert [_route(300_000, 768, U8, L2, nq, 10) for nq in (4, 8, 9)] == [K1, K1, K2]                    # 230 MB
    assert [_route(1_000_000, 128, I8, IP, nq, 10) for nq in (4, 8)] == [K1, K2]                         # 128 MB of 128-byte rows
    assert [_route(50_000_000, 768, I8, IP, nq, 100) for nq in (1, 2)] == [K1, K2]                       # 38 GB: from two queries on
    # k beyond one pass: the select + sort from the second pass on nearly everywhere, passes for one or two passes over a small
    # corpus (profiles/r05_any_k.txt: 10k x 128, k = 2048: 0.11 against 0.17 ms; four queries 0.12 against 0.19), the sort
    # route alone beyond 16384
    assert _route(10_000_000, 768, F32, COS, 1, 1025) == SORT and _route(10_000_000, 768, F32, COS, 4, 16384) == SORT
    assert _route(10_000, 128, F32, L2, 4, 2048) == PASSES and _route(10_000, 128, F32, L2, 4, 4096) == SORT
    assert _route(10_000, 128, F32, L2, 1, 1025) == PASSES and _route(10_000, 128, F32, L2, 1, 2048) == PASSES
    assert _route(10_000, 128, F32, L2, 1, 4096) == SORT
    assert _route(1_000_000, 128, F32, L2, 1, 2048) == SORT and _route(20_000_000, 64, I8, IP, 1, 1025) == SORT
    assert _route(10_000, 128, F32, L2, 1, 16385) == SORT and _route(60, 4, F32, L2, 300, 1 << 20) == SORT
    # refused shapes
    out = C.c_uint32(0)
    lib = _lib.gpu()
    assert lib.mvfgpu_selftest_route(10, 4, 9, 0, 1, 1, C.byref(out)) != 0          # unknown data type
    assert lib.mvfgpu_selftest_route(10, 4, 0, 7, 1, 1, C.byref(out)) != 0          # unknown metric
    assert lib.mvfgpu_selftest_route(10, 4, 0, 0, 1, 0, C.byref(out)) != 0          # k = 0
    assert lib.mvfgpu_selftest_route(10, 4, 0, 0, 1, 1, None) != 0
