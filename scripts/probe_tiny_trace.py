"""A few single-query searches of a small corpus (run under rocprofv3 --kernel-trace): the fixed costs of a search."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
n, dim, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = _lib.gpu()
c = G.GpuCorpus.synthetic(n, dim, 0, 0x4D564631)
k = 10
dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
_lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, 0, 0x4D564632, 0, None))
ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
for _ in range(5):
    _lib.gpu_check(lib.mvfgpu_search_device(c._h, 2, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
torch.cuda.synchronize()
c.close()
