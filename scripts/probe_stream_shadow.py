"""Single-query latency on cfg2 (10M x 768 f32): K1 on the stored rows vs K1 on the f16 shadow, scan path 4 (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
c.set_profiling(True)
for metric in (2, 0, 1):
    for nq in (1, 2):
        q = O.synth_queries(0x4D564632, nq, 768, 0)
        for path in (1, 4):
            c.set_scan_path(path)
            r = c.search(q, 100, metric)
            t0 = time.time()
            for _ in range(20):
                r = c.search(q, 100, metric)
            w = (time.time() - t0) / 20 * 1e3
            tm = c.last_timing()
            if path == 1: ref = r
            same = (r.indices == ref.indices).mean()
            print(f"metric={metric} nq={nq} path={path} kernel={tm.scan_kernel}: wall {w:6.2f} ms  scan {tm.scan_ms:5.2f} ms  "
                  f"({tm.scan_bytes/(tm.scan_ms*1e-3)/1e9:6.0f} GB/s)  indices equal to path 1: {same:.4f}", flush=True)
c.close()
