"""Ad-hoc GPU probe: parity of every dtype x metric vs the oracle on small
shapes, then a first timing of the streaming scan.  (Development aid; the
real tests live in tests/.)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mvf_oracle as O
from metrovector_amd import gpu as G

def check(dt, metric, n, dim, nq, k, seed=7):
    rows = O.synth_rows(seed, 0, n, dim, dt)
    q = O.synth_queries(seed + 1, nq, dim, dt)
    c = G.GpuCorpus.from_array(rows)
    r = c.search(q, k, metric)
    osc, oidx, oraw = O.search(rows, dt, metric, q, k)
    c.close()
    if dt in (2, 3):
        ok = (r.indices == oidx).all() and (r.raw == oraw).all() and (r.scores.view(np.uint32) == osc.view(np.uint32)).all()
        return ok, 0.0
    # float: compare rank-wise scores, and index sets modulo near ties
    fin = np.isfinite(osc)
    err = np.max(np.abs(r.scores[fin] - osc[fin]) / np.maximum(np.abs(osc[fin]), 1e-6)) if fin.any() else 0.0
    if not (r.indices[~fin] == oidx[~fin]).all(): err = 1.0
    same = (r.indices == oidx).mean()
    return (err < 1e-5 and same > 0.98), err

fails = 0
for dt in (0, 1, 2, 3):
    for metric in (0, 1, 2):
        for (n, dim, nq, k) in ((1000, 128, 1, 10), (5000, 768, 3, 100), (777, 4, 1, 5), (3000, 13, 5, 7), (100, 100, 2, 128)):
            ok, err = check(dt, metric, n, dim, nq, k)
            print(f"dt={dt} metric={metric} n={n} d={dim} nq={nq} k={k}: {'OK' if ok else 'FAIL'} err={err:.2e}", flush=True)
            fails += (not ok)
print("FAILS", fails, flush=True)

# timing
import ctypes as C
for (n, dim, dt, metric) in ((10_000_000, 768, 0, 2), (10_000_000, 768, 0, 0)):
    t0 = time.time()
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    print("synth", time.time() - t0, "s", flush=True)
    q = O.synth_queries(0x4D564632, 1, dim, dt)
    c.set_profiling(True)
    for it in range(5):
        t0 = time.time()
        r = c.search(q, 100, metric)
        dtm = time.time() - t0
        tm = c.last_timing()
        gbs = tm.scan_bytes / (tm.scan_ms * 1e-3) / 1e9 if tm.scan_ms > 0 else 0
        print(f"metric={metric} wall={dtm*1e3:.3f} ms scan={tm.scan_ms:.3f} ms select={tm.select_ms:.3f} ms  {gbs:.0f} GB/s", flush=True)
    print(r.indices[0, :5], r.scores[0, :5])
    # verify top-k rows by re-scoring with the oracle
    idx = r.indices[0].astype(np.int64)
    rows = np.stack([O.synth_rows(0x4D564631, int(i), 1, dim, dt)[0] for i in idx])
    sc, _, _ = O.scores(rows, dt, metric, q[0])
    print("rescore max rel err", np.max(np.abs(sc - r.scores[0]) / np.abs(sc)))
    c.close()
