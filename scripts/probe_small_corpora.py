"""Crossover between the streaming kernel (scan path 1: one pass per 4 queries) and the batched path (the library's default
batched route: scan path 5 for float rows = int8-shadow selection, 2 for the integer types) on small and mid-size corpora: wall
us per search (20 device-pointer searches + synchronise, second round), and what the automatic choice (scan path 0) takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
SHAPES = ((10_000, 128), (30_000, 128), (100_000, 128), (300_000, 128), (1_000_000, 128), (3_000_000, 128), (30_000, 768), (100_000, 768),
          (300_000, 768), (1_000_000, 768))
for dt in (0, 1, 2):
    for (n, dim) in SHAPES:
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        line = []
        for nq in (2, 4, 8, 12, 16, 32):
            k = 10
            dq = torch.empty((nq, dim), dtype=torch.float32 if dt in (0, 1) else torch.int8, device="cuda:0")
            _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
            ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
            ts = []
            for path in (1, 5 if dt in (0, 1) else 2, 0):
                c.set_scan_path(path)
                for it in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), 0 if dt in (0, 1) else dt, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e6
                ts.append(t)
            pick = "K1" if abs(ts[2] - ts[0]) < abs(ts[2] - ts[1]) else "K2"
            line.append(f"nq={nq}: K1 {ts[0]:5.0f} K2 {ts[1]:5.0f} auto {pick}{'' if (pick == 'K1') == (ts[0] <= ts[1]) or abs(ts[0] - ts[1]) < 0.08 * min(ts[0], ts[1]) else ' (!)'}")
        print(f"dt={dt} {n} x {dim} ({n * dim * (4 if dt == 0 else 2 if dt == 1 else 1) / 1e6:.0f} MB): " + "  ".join(line), flush=True)
        c.close()
