"""k = 10 / 100 / 1000 on 10M x 768 f32 cosine: one query (stored rows, int8 shadow streamed), 16 and 256 queries (default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
n, dim = 10_000_000, 768
c = G.GpuCorpus.synthetic(n, dim, 0, 0x4D564631)
for k in (10, 100, 1000):
    out = []
    for path, nq in ((1, 1), (6, 1), (0, 16), (0, 256)):
        dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, 0, 0x4D564632, 0, None))
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        c.set_scan_path(path)
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5 * 1e3
        out.append(f"path{path} nq={nq}: {t:6.2f} ms")
    print(f"k={k}: " + "   ".join(out), flush=True)
c.close()
