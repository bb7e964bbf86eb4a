"""Where the host API's latency goes on small searches (cfg1: 10k x 128 f32 L2, top-10).  Per corpus and batch size, the
median wall time of one blocking mvfgpu_search call with pageable host buffers:
  copies   -- MVF_HOST_ZC_QUERY=0 MVF_HOST_ZC_RESULTS=0: staged hipMemcpyAsync H2D of the query, D2H of the three result arrays
  zc-out   -- results written in place into pinned host memory, the query still copied
  zc       -- the default: the query read in place too
  with rows -- the k best AND their payload rows: mvfgpu_search_fetch (one call) against mvfgpu_search + mvfgpu_corpus_gather_rows
  device   -- mvfgpu_search_device on device buffers + a stream synchronise (no transfer at all), and `enqueue`, the CPU time
              of that call alone."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from metrovector_amd import _lib, gpu as G

lib = _lib.gpu()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def med(f, reps=REPS):
    for _ in range(20):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


MODES = (("copies", "0", "0"), ("zc-out", "0", None), ("zc", None, None))
for (n, dim, dt, metric, k) in ((60, 4, 0, 0, 5), (10_000, 128, 0, 0, 10), (10_000, 128, 2, 1, 10), (100_000, 128, 0, 0, 10),
                                (1_000_000, 128, 0, 0, 10), (1_000_000, 768, 0, 2, 100), (10_000_000, 768, 0, 2, 100)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    tdt = torch.float32 if dt == 0 else torch.int8
    for nq in (1, 4, 16, 64):
        if n >= 10_000_000 and nq > 16:
            continue
        dq = torch.empty((nq, dim), dtype=tdt, device="cuda:0")
        _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dt, 0x4D564632, 0, None))
        hq = dq.cpu().numpy().copy()
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0")
        di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
        sc = np.empty((nq, k), np.float32)
        ix = np.empty((nq, k), np.uint64)

        def A():
            _lib.gpu_check(lib.mvfgpu_search(c._h, metric, hq.ctypes.data_as(C.c_void_p), dt, dim, nq, k,
                                             sc.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), None))

        def B():
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, C.c_void_p(dq.data_ptr()), dt, dim, nq, k, C.c_void_p(ds.data_ptr()),
                                                    C.c_void_p(di.data_ptr()), None, None))
            torch.cuda.synchronize()

        res, ref = [], None
        for name, zq, zo in MODES:
            for var, val in (("MVF_HOST_ZC_QUERY", zq), ("MVF_HOST_ZC_RESULTS", zo)):
                if val is None:
                    os.environ.pop(var, None)
                else:
                    os.environ[var] = val
            c.reload_tuning()
            sc[:] = 0; ix[:] = 0
            t = med(A)
            if ref is None:
                ref = (sc.copy(), ix.copy())
            same = np.array_equal(sc, ref[0]) and np.array_equal(ix, ref[1])
            res.append(f"{name} {t:7.1f}{'' if same else ' DIFF'}")
        b = med(B)
        vec = np.empty((nq, k, dim), hq.dtype)

        def F():  # the k best and their rows: one call ...
            _lib.gpu_check(lib.mvfgpu_search_fetch(c._h, metric, hq.ctypes.data_as(C.c_void_p), dt, dim, nq, k, sc.ctypes.data_as(C.c_void_p),
                                                   ix.ctypes.data_as(C.c_void_p), None, vec.ctypes.data_as(C.c_void_p)))

        def S():  # ... or two
            A()
            _lib.gpu_check(lib.mvfgpu_corpus_gather_rows(c._h, ix.ctypes.data_as(C.c_void_p), nq * k, vec.ctypes.data_as(C.c_void_p)))

        fetch = (med(F), med(S))
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, C.c_void_p(dq.data_ptr()), dt, dim, nq, k, C.c_void_p(ds.data_ptr()),
                                                    C.c_void_p(di.data_ptr()), None, None))
            ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        ts.sort()
        print(f"n={n:>9} dim={dim:>4} dt={dt} k={k:>3} nq={nq:>3}:  " + "   ".join(res) + f"   device {b:7.1f}   enqueue {ts[100] * 1e6:5.1f}   with rows: fetch {fetch[0]:7.1f}  search+gather {fetch[1]:7.1f}", flush=True)
    c.close()
