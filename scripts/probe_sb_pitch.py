"""Small-batch K2 on Int8 rows of several widths: the streaming MFMA kernel (MVF_K2_SB=1) against the 64-query tile shape
(MVF_K2_SB=0), wall ms per search of 8 queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for dim in (512, 768, 1008, 1024, 1040, 1280, 2048):
    n = int(8e9 // dim)
    c = G.GpuCorpus.synthetic(n, dim, 2, 0x4D564631)
    nq, k = 8, 100
    dq = torch.empty((nq, dim), dtype=torch.int8, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, 2, 0x4D564632, 0, None))
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    out = []
    c.set_scan_path(2)
    for sb in ("0", "1"):
        os.environ["MVF_K2_SB"] = sb
        c.reload_tuning()
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 1, dq.data_ptr(), 2, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5 * 1e3
        out.append(f"sb={sb} {t:6.2f} ms ({n * dim / t / 1e9:5.2f} TB/s)")
    print(f"int8 {n} x {dim}: " + "   ".join(out), flush=True)
    c.close()
