#!/usr/bin/env python3
"""K1 on short rows of the narrow types (the instantiations that ask for 8 waves per SIMD = 64 registers and spill 1-25 of
them OUTSIDE their row loop, profiles/r04_kernel_spill_sites.csv): scan ms / TB/s per shape.  Run once with the shipped
library and once with MVF_GPU_LIB_PATH=scripts/bin/libmvf_gpu_k1w6.so (-DMVF_K1_SHORT_ROW_WAVES=6: 80 registers, no spill)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
print("library:", _lib.GPU_LIB_PATH, flush=True)
for dt, dim in ((2, 64), (2, 128), (3, 64), (1, 32), (1, 64), (1, 8), (2, 16)):
    es = {1: 2, 2: 1, 3: 1}[dt]
    n = min(200_000_000, (4 << 30) // (dim * es))
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qdt = {1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
    dq = torch.empty((1, dim), dtype=qdt, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 1, dim, dt, 0x4D564632, 0, None))
    ds = torch.empty((1, 100), dtype=torch.float32, device="cuda:0")
    di = torch.empty((1, 100), dtype=torch.int64, device="cuda:0")
    c.set_scan_path(1)
    out = []
    for metric in (0, 1, 2):
        for _ in range(5):
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, 1, 100, ds.data_ptr(), di.data_ptr(), None, None))
        c.set_profiling(True)
        for _ in range(30):
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, 1, 100, ds.data_ptr(), di.data_ptr(), None, None))
        torch.cuda.synchronize()
        tm = c.last_timing()
        c.set_profiling(False)
        out.append(f"metric {metric}: {tm.scan_ms_avg:.3f} ms {tm.scan_bytes / tm.scan_ms_avg / 1e9:.2f} TB/s")
    print(f"dtype {dt} dim {dim:4d} ({dim * es:3d}-B rows, {n / 1e6:.0f}M rows)  " + "   ".join(out), flush=True)
    c.close()
