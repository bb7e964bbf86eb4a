import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/metrovector_amd") else os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for (n, dim, dt) in ((10_000_000, 768, 0), (10_000_000, 1024, 0), (12_500_000, 1024, 1), (12_500_000, 768, 1)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    nq, k = 8, 100
    dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    c.set_scan_path(5)
    for metric in (0, 1, 2):
        out = []
        for sb in ("0", "1"):
            os.environ["MVF_K2_SB"] = sb
            c.reload_tuning()
            for it in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5 * 1e3
            out.append(f"sb={sb} {t:6.2f} ms")
        print(f"dt={dt} {n} x {dim} metric={metric}: " + "   ".join(out), flush=True)
    c.close()
