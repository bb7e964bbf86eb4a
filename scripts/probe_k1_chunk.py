import os, sys
sys.path.insert(0, os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
ES = {0: 4, 1: 2, 2: 1, 3: 1}
for dt, dim in ((0, 32), (0, 64), (2, 64), (2, 128), (1, 64), (0, 128)):
    rb = dim * ES[dt]
    n = (4 << 30) // rb
    for cr in ("", "1024", "2048", "4096"):
        if cr: os.environ["MVF_K1_CHUNK_ROWS"] = cr
        else: os.environ.pop("MVF_K1_CHUNK_ROWS", None)
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
        dq = torch.empty((1, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 1, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((1, 100), dtype=torch.float32, device="cuda:0")
        di = torch.empty((1, 100), dtype=torch.int64, device="cuda:0")
        c.set_scan_path(1); c.set_profiling(True)
        for _ in range(8):
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), G.query_dtype_code(dt), dim, 1, 100, ds.data_ptr(), di.data_ptr(), None, None))
        torch.cuda.synchronize()
        ms = c.last_timing().scan_ms_avg
        print(f"dt={dt} dim={dim} row={rb}B chunk_rows={cr or 'default'}: {ms:.4f} ms {n*rb/ms/1e6:.0f} GB/s", flush=True)
        c.close()
