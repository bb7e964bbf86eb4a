"""Whole-search wall ms against the batch size on the benchmark corpora, default path, device-resident queries:
10M x 768 f32 cosine (configs[1]/[2]), 12.5M x 1024 f16 L2 (a shard of configs[4]), 50M x 768 int8 dot (configs[3])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
NQS = [int(x) for x in os.environ.get("MVF_NQS", "1,2,4,8,16,32,64,65,96,128,129,192,256,257,384,512,768,1024,1536,2048,4096").split(",")]
which = os.environ.get("MVF_CFGS", "cfg3,cfg5,cfg4").split(",")
CFG = {"cfg3": (10_000_000, 768, 0, 2), "cfg5": (12_500_000, 1024, 1, 0), "cfg4": (50_000_000, 768, 2, 1)}
for name in which:
    n, dim, dt, metric = CFG[name]
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qdt = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}[dt]
    for nq in NQS:
        dq = torch.empty((nq, dim), dtype=qdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((nq, 100), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, 100), dtype=torch.int64, device="cuda:0")
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), G.query_dtype_code(dt), dim, nq, 100, ds.data_ptr(), di.data_ptr(), None, None))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 3 * 1e3)
        print(f"{name} {n}x{dim} dt={dt} nq={nq:5d}: {best:8.3f} ms  {best / nq * 1e3:9.2f} us/query  {nq * n / best / 1e6:10.1f} M distance-ops/ms", flush=True)
    c.close()
