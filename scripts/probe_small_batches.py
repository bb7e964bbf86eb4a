"""Small batches (the 64-query, HBM-bound K2 tile): f16 selection (scan path 3 / native f16 rows) vs int8-shadow selection
(scan path 5) vs the streaming kernel, wall ms per search with device-resident queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for (n, dim, dt, metric) in ((10_000_000, 768, 0, 2), (12_500_000, 1024, 1, 0)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    for nq in (2, 4, 8, 16, 64, 128, 256, 512):
        dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((nq, 100), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, 100), dtype=torch.int64, device="cuda:0")
        out = []
        ref = None
        for path in (1, 3, 5):
            c.set_scan_path(path)
            for it in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 0, dim, nq, 100, ds.data_ptr(), di.data_ptr(), None, None))
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3 * 1e3
            idx = di.cpu().numpy().copy()
            if ref is None: ref = idx
            out.append(f"path{path} {t:7.2f} ms same={bool((idx == ref).all())}")
        print(f"dt={dt} {n}x{dim} nq={nq:4d}: " + "   ".join(out), flush=True)
    c.close()
