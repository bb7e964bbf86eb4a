"""K1 lane-group width against V = 16-byte vectors per row (Int8 rows of dim 16 V, ~2 GiB corpora, one query, cosine,
top-100): ms per scan at G = 8 / 16 / 32 / 64 and at choose_group's pick.  Prints CSV V,row_bytes,G8,G16,G32,G64,auto (GB/s)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metrovector_amd import _lib, gpu as G

lib = _lib.gpu()
VS = [int(x) for x in os.environ.get("MVF_SWEEP_V", ",".join(str(v) for v in list(range(9, 65)) + [72, 80, 88, 96, 100, 112, 120, 136, 150, 168, 184])).split(",")]
DT = int(os.environ.get("MVF_SWEEP_DT", "2"))
ES = {0: 4, 1: 2, 2: 1, 3: 1}[DT]
print("V,row_bytes,G8_GBps,G16_GBps,G32_GBps,G64_GBps,auto_GBps", flush=True)
for V in VS:
    rb = 16 * V
    dim = rb // ES
    n = (2 << 30) // rb
    out = []
    for g in ("8", "16", "32", "64", ""):
        if g:
            os.environ["MVF_K1_G"] = g
        else:
            os.environ.pop("MVF_K1_G", None)
        c = G.GpuCorpus.synthetic(n, dim, DT, 0x4D564631)
        qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[DT]
        dq = torch.empty((1, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 1, dim, DT, 0x4D564632, 0, None))
        ds = torch.empty((1, 100), dtype=torch.float32, device="cuda:0")
        di = torch.empty((1, 100), dtype=torch.int64, device="cuda:0")
        c.set_scan_path(1)
        c.set_profiling(True)
        for _ in range(8):
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), G.query_dtype_code(DT), dim, 1, 100, ds.data_ptr(),
                                                    di.data_ptr(), None, None))
        torch.cuda.synchronize()
        out.append(n * rb / (c.last_timing().scan_ms_avg * 1e-3) / 1e9)
        c.close()
    print(f"{V},{rb}," + ",".join(f"{x:.0f}" for x in out), flush=True)
