"""Host-side cost of one mvfgpu_search_device call (enqueue only, no synchronisation) against the device time of the
search, for small and large corpora: how much of a search is launch overhead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
lib = _lib.gpu()
for (n, dim, dt, metric) in ((100_000, 128, 0, 2), (1_000_000, 128, 0, 2), (1_000_000, 768, 0, 2), (10_000_000, 768, 0, 2)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    for nq in (1, 4, 64, 1024):
        k = 10
        dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        def call():
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), 0, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
        for _ in range(3): call()
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps): call()
        t_enq = (time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps * 1e6
        # latency of ONE search from idle
        lat = []
        for _ in range(10):
            torch.cuda.synchronize(); t1 = time.perf_counter(); call(); torch.cuda.synchronize(); lat.append((time.perf_counter() - t1) * 1e6)
        print(f"{n}x{dim} nq={nq:4d}: enqueue {t_enq:7.1f} us/call   throughput {t_all:8.1f} us/search   single-search latency {min(lat):8.1f} us", flush=True)
    c.close()
