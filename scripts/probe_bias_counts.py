"""Diagnostic: how often the folded pre-filter of scan_mfma16_dma.hip sends a wave tile into its rare path, and what it
finds there.  Needs a library built with -DMVF_DIAG_COUNT (MVF_GPU_LIB_PATH=...).  usage: probe_bias_counts.py [cfg3,cfg5,cfg4]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metrovector_amd import _lib, gpu as G

CFGS = {"cfg4": (50_000_000, 768, 2, 1, 256), "cfg5": (12_500_000, 1024, 1, 0, 1024), "cfg3": (10_000_000, 768, 0, 2, 1024),
        "cfg3l2": (10_000_000, 768, 0, 0, 1024), "cfg3ip": (10_000_000, 768, 0, 1, 1024), "u8": (20_000_000, 768, 3, 0, 256),
        "cfg3q": (2_500_000, 768, 0, 2, 1024), "cfg3s": (625_000, 768, 0, 2, 1024),  # cfg3 minus its last phase / its last two
        "i8ip": (3_000_000, 768, 2, 1, 256), "i8l2": (3_000_000, 768, 2, 0, 256), "i8cos": (3_000_000, 768, 2, 2, 256)}
lib = _lib.gpu()
for name in (sys.argv[1] if len(sys.argv) > 1 else "cfg3,cfg5,cfg4").split(","):
    n, dim, dt, metric, nq = CFGS[name]
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
    dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
    k = 100
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
    di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    args = (c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None)
    _lib.gpu_check(lib.mvfgpu_search_device(*args))
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.mvfgpu_diag_bias_counts(out, 1)
    _lib.gpu_check(lib.mvfgpu_search_device(*args))
    torch.cuda.synchronize()
    lib.mvfgpu_diag_bias_counts(out, 1)
    t, _, g, ip, _, _, _, _ = list(out)
    print(f"{name}: wave tiles {t}, flagged query groups {g} ({g / max(t, 1):.2f} of 8 per wave tile), "
          f"records written {ip} ({ip / max(t, 1):.3f} per wave tile)", flush=True)
    c.close()
