"""Three 8-query searches of 10M x 768 f32 through the int8 shadow with one metric (argv[1]); for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
metric = int(sys.argv[1])
lib = _lib.gpu()
n, dim, nq, k = 10_000_000, 768, 8, 100
c = G.GpuCorpus.synthetic(n, dim, 0, 0x4D564631)
dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
_lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, 0, 0x4D564632, 0, None))
ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
c.set_scan_path(5)
for _ in range(3):
    _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
torch.cuda.synchronize()
c.close()
