"""K1 (streaming, 4 queries per pass) vs K2 (MFMA batched) wall time by batch size (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
for (n, dim, dt, metric) in ((10_000_000, 768, 0, 2), (12_500_000, 1024, 1, 0), (50_000_000, 768, 2, 1)):
    c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
    for nq in (2, 3, 4, 5, 8, 16, 64, 65, 128, 129):
        q = O.synth_queries(0x4D564632, nq, dim, dt)
        out = []
        for path in (1, 3):
            c.set_scan_path(path)
            c.search(q, 100, metric)
            t0 = time.time()
            for _ in range(2):
                c.search(q, 100, metric)
            out.append((time.time() - t0) / 2 * 1e3)
        print(f"dt={dt} nq={nq:4d}  K1 {out[0]:8.2f} ms   K2 {out[1]:8.2f} ms", flush=True)
    c.close()
