"""Wall time of single 1024-query searches, one after another (development aid): does any search take much longer than the
rest?  usage: probe_wall_jitter.py [dtype=1] [profiling=0] [searches=24] [n,dim,metric,nq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
dt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n, dim, metric, nq, k = 10_000_000, 768, 2, 1024, 100
if len(sys.argv) > 4:  # n,dim,metric,nq
    n, dim, metric, nq = (int(x) for x in sys.argv[4].split(","))
lib = _lib.gpu()
c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
qd = dt if dt >= 2 else 0
dq = torch.empty((nq, dim), dtype={0: torch.float32, 2: torch.int8, 3: torch.uint8}[qd], device="cuda:0")
_lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, qd, 0x4D564632, 0, None))
ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
lib.mvfgpu_set_profiling(c._h, prof)
out = []
for i in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), qd, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    if i % 3 == 2:
        _ = di.cpu()
print(f"dtype {dt} profiling {prof}: enqueue ms / done ms per search")
print(" ".join(f"{a:.2f}/{b:.2f}" for a, b in out))
done = sorted(b for _, b in out[2:])
print(f"median {done[len(done) // 2]:.3f} ms  min {done[0]:.3f} ms  (MVF_K2_GROWTH={os.environ.get('MVF_K2_GROWTH', '-')})")
c.close()
