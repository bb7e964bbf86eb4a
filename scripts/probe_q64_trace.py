"""One batch (nq queries, default 64) on 10M x 768 f32 cosine (or "cfg4": 50M x 768 int8 dot, "cfg5": 12.5M x 1024 f16 L2) through the default path, a few searches: run under
rocprofv3 --kernel-trace to see where a 64-query search's time goes (phases, compactions, re-scoring)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, dim, dt, metric, k = 10_000_000, 768, 0, 2, 100
if len(sys.argv) > 2 and sys.argv[2] == "cfg4":
    n, dim, dt, metric = 50_000_000, 768, 2, 1
if len(sys.argv) > 2 and sys.argv[2] == "cfg5":
    n, dim, dt, metric = 12_500_000, 1024, 1, 0
if len(sys.argv) > 2 and sys.argv[2].startswith("shape:"):  # shape:n,dim,dtype,metric,k
    n, dim, dt, metric, k = (int(x) for x in sys.argv[2][6:].split(","))
lib = _lib.gpu()
c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
dq = torch.empty((nq, dim), dtype=torch.int8 if dt == 2 else torch.uint8 if dt == 3 else torch.float32, device="cuda:0")
_lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
for _ in range(4):
    _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), dt if dt >= 2 else 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
torch.cuda.synchronize()
c.close()
