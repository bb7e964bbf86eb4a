import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for (n, dim) in ((1_000_000, 128), (2_000_000, 128), (1_000_000, 768), (700_000, 256), (3_000_000, 96)):
    c = G.GpuCorpus.synthetic(n, dim, 0, 0x4D564631)
    dq = torch.empty((1, dim), dtype=torch.float32, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 1, dim, 0, 0x4D564632, 0, None))
    ds = torch.empty((1, 10), dtype=torch.float32, device="cuda:0"); di = torch.empty((1, 10), dtype=torch.int64, device="cuda:0")
    out = []
    for b in ("0", "1", "0", "1"):
        os.environ["MVF_K1_BALANCE"] = b
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), 0, dim, 1, 10, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50 * 1e6
        out.append(f"bal={b} {t:6.1f} us")
    print(f"{n} x {dim}: " + "  ".join(out), flush=True)
    c.close()
