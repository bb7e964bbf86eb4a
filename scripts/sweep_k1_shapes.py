"""K1 (streaming scan) across row shapes: every dtype x dim in 64..2048, each lane-group width G the kernel has
(MVF_K1_G forces it; unset = choose_group's pick).  Prints CSV: dtype,dim,row_bytes,G,chosen,ms,GB/s,frac_of_8TB/s.
Corpora of ~4 GiB (synthetic, on device); single query, cosine (the metric with the most arithmetic), top-100."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metrovector_amd import _lib, gpu as G

DIMS = [int(x) for x in os.environ.get("MVF_SWEEP_DIMS", "32,64,100,128,200,256,384,512,768,1024,1536,2048").split(",")]
ES = {0: 4, 1: 2, 2: 1, 3: 1}
lib = _lib.gpu()
print("dtype,dim,row_bytes,G,chosen_by_default,ms_per_scan,GB_per_s,frac_of_hbm_peak", flush=True)
for dt in (0, 1, 2, 3):
    for dim in DIMS:
        rb = dim * ES[dt]
        n = min(200_000_000, (4 << 30) // rb)
        res = {}
        for g in ("", "64", "32", "16", "8", "4", "1"):
            if g:
                os.environ["MVF_K1_G"] = g
            else:
                os.environ.pop("MVF_K1_G", None)
            V = (rb + 15) // 16
            if g and int(g) > 4 * V and int(g) > 1:
                continue  # absurdly wide for the row: most lanes idle, pointless to time
            c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
            qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
            dq = torch.empty((1, dim), dtype=qdt, device="cuda:0")
            _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 1, dim, dt, 0x4D564632, 0, None))
            ds = torch.empty((1, 100), dtype=torch.float32, device="cuda:0")
            di = torch.empty((1, 100), dtype=torch.int64, device="cuda:0")
            c.set_scan_path(1)
            c.set_profiling(True)
            for _ in range(8):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), G.query_dtype_code(dt), dim, 1, 100, ds.data_ptr(),
                                                        di.data_ptr(), None, None))
            torch.cuda.synchronize()
            tm = c.last_timing()
            res[g] = tm.scan_ms_avg
            c.close()
        best = min(v for k, v in res.items() if k)
        for g, ms in res.items():
            if not g:
                continue
            chosen = abs(ms - res[""]) < 0.02 * ms and ms <= min(v for k, v in res.items() if k and abs(v - res[""]) < 0.02 * v) + 1e-9
            gb = n * rb / (ms * 1e-3) / 1e9
            print(f"{dt},{dim},{rb},{g},{'' if not chosen else 'default'}{'' if ms > best * 1.0001 else ' best'},{ms:.4f},{gb:.1f},{gb / 8000:.3f}", flush=True)
        print(f"{dt},{dim},{rb},auto,default,{res['']:.4f},{n * rb / (res[''] * 1e-3) / 1e9:.1f},{n * rb / (res[''] * 1e-3) / 1e9 / 8000:.3f}", flush=True)
