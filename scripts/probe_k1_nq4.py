"""K1's four-query pass on 4-GiB corpora of short rows: wall ms per search (scan path 1, cosine, top-100).  MVF_K1_LONG4 was
the switch of the experiment recorded in profiles/r03_k1_piece_length.txt; the library now always takes long pieces."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
ES = {0: 4, 1: 2, 2: 1, 3: 1}
for dt, dim in ((0, 32), (2, 64), (2, 128), (0, 64), (1, 64), (0, 128), (2, 768)):
    rb = dim * ES[dt]; n = (4 << 30) // rb
    for mode in ("", "1"):
        if mode: os.environ["MVF_K1_LONG4"] = "1"
        else: os.environ.pop("MVF_K1_LONG4", None)
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
        dq = torch.empty((4, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), 4, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((4, 100), dtype=torch.float32, device="cuda:0"); di = torch.empty((4, 100), dtype=torch.int64, device="cuda:0")
        c.set_scan_path(1)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), G.query_dtype_code(dt), dim, 4, 100, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
        print(f"dt={dt} dim={dim} row={rb}B nq=4 long_pieces={'yes' if mode else 'no '}: {best:.3f} ms {n*rb/best/1e6:.0f} GB/s", flush=True)
        c.close()
