"""In-process A/B of one tuning variable on the batched benchmark searches (development aid): ONE corpus per config, the variable
switched with mvfgpu_corpus_reload_tuning between rounds, rounds interleaved; wall ms of a search (enqueue to synchronize) and the library's own event time, results
compared across the modes.  usage: probe_tune_ab.py VAR v1,v2[,v3] [cfg3,cfg5,cfg4] [rounds=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrovector_amd import _lib, gpu as G
var, modes = sys.argv[1], sys.argv[2].split(",")
names = (sys.argv[3] if len(sys.argv) > 3 else "cfg3,cfg5,cfg4").split(",")
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 10
CFGS = {"cfg3": (10_000_000, 768, 0, 2, 1024), "cfg5": (12_500_000, 1024, 1, 0, 1024), "cfg4": (50_000_000, 768, 2, 1, 256),
        "cfg3f16": (10_000_000, 768, 1, 2, 1024), "q256": (10_000_000, 768, 0, 2, 256), "q4096": (4_000_000, 768, 0, 2, 4096),
        "q16_1m": (1_000_000, 768, 0, 2, 16), "q64_1m": (1_000_000, 768, 0, 2, 64), "q16": (10_000_000, 768, 0, 2, 16), "q64": (10_000_000, 768, 0, 2, 64),
        "q128": (10_000_000, 768, 0, 2, 128), "q512": (10_000_000, 768, 0, 2, 512), "q256_c5": (12_500_000, 1024, 1, 0, 256),
        "q128_3m": (3_000_000, 768, 0, 1, 128), "q384": (10_000_000, 768, 0, 2, 384), "q256_d128": (20_000_000, 128, 0, 0, 256), "q32_3m": (3_000_000, 768, 0, 2, 32)}
lib = _lib.gpu()
k = 100
for name in names:
    n, dim, dt, metric, nq = CFGS[name]
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    qd = dt if dt >= 2 else 0
    dq = torch.empty((nq, dim), dtype={0: torch.float32, 2: torch.int8, 3: torch.uint8}[qd], device="cuda:0")
    _lib.gpu_check(lib.mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, qd, 0x4D564632, 0, None))
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); di = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    res = {m: [] for m in modes}
    dev = {m: [] for m in modes}
    lib.mvfgpu_set_profiling(c._h, 1)
    ref = None
    same = True
    for rnd in range(rounds + 1):
        order = modes if rnd == 0 else modes[(rnd - 1) % len(modes):] + modes[:(rnd - 1) % len(modes)]
        for m in order:
            os.environ[var] = m
            _lib.gpu_check(lib.mvfgpu_corpus_reload_tuning(c._h))
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _lib.gpu_check(lib.mvfgpu_search_device(c._h, metric, dq.data_ptr(), qd, dim, nq, k, ds.data_ptr(), di.data_ptr(), None, None))
                torch.cuda.synchronize()
                if rnd and rep:
                    res[m].append((time.perf_counter() - t0) * 1e3)
                    tm = _lib.Timing()
                    lib.mvfgpu_last_timing(c._h, _lib.C.byref(tm))
                    dev[m].append(tm.search_ms)
            idx = di.cpu()
            if ref is None:
                ref = idx
            same &= bool((idx == ref).all())
    for m in modes:
        x = sorted(res[m])
        d = sorted(dev[m])
        print(f"== {name} {var}={m:6s} wall median {x[len(x) // 2]:8.3f} ms  min {x[0]:8.3f}   device (first to last kernel, HIP events) median {d[len(d) // 2]:8.3f}  min {d[0]:8.3f}  "
              f"(n={len(x)})  identical across modes: {same}", flush=True)
    c.close()
