import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for dt, dim, n in ((0, 768, 10_000_000), (1, 1024, 12_500_000), (2, 768, 50_000_000), (0, 128, 40_000_000)):
    for g in ("64", "16"):
        os.environ["MVF_K1_G"] = g
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8}[dt]
        for nq in (1, 4):
            dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
            _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
            ds = torch.empty((nq, 100), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, 100), dtype=torch.int64, device="cuda:0")
            c.set_scan_path(1); c.set_profiling(True)
            for _ in range(20):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, 100, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); tm = c.last_timing(); c.set_profiling(False)
            print(f"dt={dt} dim={dim} n={n} nq={nq} G={g}: {tm.scan_ms_avg:.3f} ms  {tm.scan_bytes/tm.scan_ms_avg/1e6:.0f} GB/s", flush=True)
        c.close()
