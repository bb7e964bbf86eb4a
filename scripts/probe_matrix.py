"""Eyeball sweep for performance cliffs: Float32 / Int8 corpora of ~3 GB, dims 32..4096, 1..512 queries, k = 10 / 1000,
default path; wall ms per search and the repair count (MVF_DEBUG_REPAIR prints it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for dt in (0, 2):
    es = 4 if dt == 0 else 1
    for dim in (32, 128, 768, 2048, 4096):
        n = int(3e9 // (dim * es))
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        for k in (10, 1000):
            out = []
            for nq in (1, 8, 64, 512):
                dq = torch.empty((nq, dim), dtype=torch.float32 if dt == 0 else torch.int8, device="cuda:0")
                _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
                ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
                for it in range(2):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(3):
                        _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), dt, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3 * 1e3
                out.append(f"nq={nq}: {t:7.2f} ms")
            print(f"dt={dt} {n} x {dim} k={k}: " + "   ".join(out), flush=True)
        c.close()
