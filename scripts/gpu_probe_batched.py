"""Ad-hoc timing of the MFMA batched path (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mvf_oracle as O
from metrovector_amd import gpu as G
n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768
c = G.GpuCorpus.synthetic(n, dim, 0, 0x4D564631)
for nq in (1024, 256, 32):
    q = O.synth_queries(0x4D564632, nq, dim, 0)
    c.set_profiling(True)
    for it in range(3):
        t0 = time.time(); r = c.search(q, 100, 2); dt = time.time() - t0
        tm = c.last_timing()
        tf = tm.scan_flops / (tm.scan_ms * 1e-3) / 1e12 if tm.scan_ms > 0 else 0
        print(f"nq={nq} wall={dt*1e3:.1f} ms  last-phase scan={tm.scan_ms:.2f} ms ({tf:.1f} TFLOP/s, {tm.scan_flops/1e12:.2f} TF) launches={tm.scan_launches} kernel={tm.scan_kernel}", flush=True)
    c.set_profiling(False)
    # spot check 3 queries vs K1
    c.set_scan_path(1); ref = c.search(q[:3], 100, 2); c.set_scan_path(0)
    print("  overlap with streaming path:", [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(r.indices[:3], ref.indices)])
c.close()
