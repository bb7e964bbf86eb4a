"""One to four queries: exact K1 (path 1), K1 over the f16 shadow (path 4, f32 rows, nq <= 2), K2 int8 selection on the
64-query tile (path 5), K1 over the int8 shadow (path 6) -- wall ms per search, device-resident queries, and the number
of queries the repair pass had to redo (MVF_DEBUG_REPAIR)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for (n, dim, dt, metric) in ((10_000_000, 768, 0, 2), (10_000_000, 768, 0, 0), (12_500_000, 1024, 1, 0), (12_500_000, 1024, 1, 1)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    for k in (10, 100):
        for nq in (1, 2, 3, 4):
            dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
            _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
            ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
            out = []
            ref = None
            for path in (1, 4, 5, 6, 0):
                if path == 4 and (dt != 0 or nq > 2): continue
                c.set_scan_path(path)
                for it in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(5):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5 * 1e3
                idx = di.cpu().numpy().copy()
                if ref is None: ref = idx
                out.append(f"path{path} {t:6.2f} ms same={bool((idx == ref).all())}")
            print(f"dt={dt} metric={metric} {n}x{dim} k={k} nq={nq}: " + "  ".join(out), flush=True)
    c.close()
