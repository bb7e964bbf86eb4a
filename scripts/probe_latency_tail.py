"""Latency distribution of repeated searches on the cfg2 corpus (development aid): per-call wall times, percentiles."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
for nq, reps in ((1, 300), (16, 300), (256, 300), (1024, 100)):
    q = O.synth_queries(0x4D564632, nq, 768, 0)
    c.search(q, 100, 2)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); c.search(q, 100, 2); t.append((time.perf_counter() - t0) * 1e3)
    t = np.array(t)
    big = np.nonzero(t > 2 * np.median(t))[0]
    print(f"nq={nq:5d}: median {np.median(t):7.2f} ms  p99 {np.percentile(t, 99):7.2f}  max {t.max():7.2f}  "
          f"calls > 2x median: {len(big)} at {big[:10].tolist()}", flush=True)
c.close()
