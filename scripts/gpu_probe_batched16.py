"""Ad-hoc timing of the f16 / int8 MFMA batched path (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
for (n, dim, dt, metric, nq) in ((50_000_000, 768, 2, 1, 256), (12_500_000, 1024, 1, 2, 1024), (12_500_000, 1024, 1, 0, 1024)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    q = O.synth_queries(0x4D564632, nq, dim, dt)
    c.set_profiling(True)
    for it in range(3):
        t0 = time.time(); r = c.search(q, 100, metric); dt_s = time.time() - t0
        tm = c.last_timing()
        tf = tm.scan_flops / (tm.scan_ms * 1e-3) / 1e12 if tm.scan_ms > 0 else 0
        gb = tm.scan_bytes / (tm.scan_ms * 1e-3) / 1e9 if tm.scan_ms > 0 else 0
        print(f"dt={dt} nq={nq} wall={dt_s*1e3:.1f} ms last-phase scan={tm.scan_ms:.2f} ms ({tf:.1f} Tops/s, {gb:.0f} GB/s) launches={tm.scan_launches} kernel={tm.scan_kernel}", flush=True)
    c.set_profiling(False)
    c.set_scan_path(1); ref = c.search(q[:2], 100, metric); c.set_scan_path(0)
    print("  vs streaming path: idx equal", (r.indices[:2] == ref.indices).mean(), "raw equal", (r.raw[:2] == ref.raw).all())
    c.close()
