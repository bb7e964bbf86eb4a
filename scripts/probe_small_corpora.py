"""Crossover between the streaming kernel (scan path 1: one pass per 4 queries) and the batched path (path 5 / 2) on small
corpora: wall us per search."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for dt in (0, 2):
    for (n, dim) in ((10_000, 128), (100_000, 128), (1_000_000, 128), (100_000, 768), (1_000_000, 768)):
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        line = []
        for nq in (4, 5, 8, 16, 32):
            k = 10
            dq = torch.empty((nq, dim), dtype=torch.float32 if dt == 0 else torch.int8, device="cuda:0")
            _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
            ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
            ts = []
            for path in (1, 5 if dt == 0 else 2):
                c.set_scan_path(path)
                for it in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), dt, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e6
                ts.append(t)
            line.append(f"nq={nq}: K1 {ts[0]:6.0f} K2 {ts[1]:6.0f}")
        print(f"dt={dt} {n} x {dim} ({n * dim * (4 if dt == 0 else 1) / 1e6:.0f} MB): " + "  ".join(line), flush=True)
        c.close()
