"""Whole-search wall ms on ANN-benchmark-like shapes (SIFT / GIST / GloVe / DEEP sized corpora), default path, device queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
SHAPES = [("sift1m f32 L2", 1_000_000, 128, 0, 0), ("sift1m u8 L2", 1_000_000, 128, 3, 0), ("sift10m u8 L2", 10_000_000, 128, 3, 0),
          ("gist1m f32 L2", 1_000_000, 960, 0, 0), ("glove1.2m-100 f32 cos", 1_183_514, 100, 0, 2), ("deep10m-96 f32 cos", 10_000_000, 96, 0, 2),
          ("sift100m u8 L2", 100_000_000, 128, 3, 0)]
for name, n, dim, dt, metric in SHAPES:
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
    for nq, k in ((1, 10), (100, 10), (1000, 10), (10000, 10), (10000, 100)):
        dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 2 * 1e3)
        tm = c.last_timing()
        print(f"{name:24s} {n}x{dim} nq={nq:6d} k={k:4d}: {best:9.3f} ms  {nq / best * 1e3:12.0f} queries/s  {2.0 * nq * n * dim / best / 1e9:9.1f} Top/s  repaired={tm.repaired_queries}", flush=True)
    c.close()
