"""K2 (default batched path) effective rate against the row length: nq = 4096 queries, corpora of 3.84e9 elements
(5M x 768 ... 60M x 64), Int8 dot / UInt8 L2 / Float32 cosine (int8-shadow selection), top-10; Top/s = 2 nq n dim / t."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
nq, k = int(os.environ.get("MVF_NQ", "4096")), 10
for dt, metric, label in ((2, 1, "int8 dot"), (3, 0, "uint8 L2"), (0, 2, "f32 cosine")):
    for dim in (64, 128, 256, 384, 768):
        n = 3_840_000_000 // dim
        if dt == 0: n //= 2
        c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
        qdt = {0: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
        dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 2 * 1e3)
        print(f"{label:10s} {n:9d} x {dim:4d} nq={nq}: {best:8.2f} ms  {2.0 * nq * n * dim / best / 1e9:8.1f} Top/s  repaired={c.last_timing().repaired_queries}", flush=True)
        c.close()
