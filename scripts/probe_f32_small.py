"""f32 K2 at small batch sizes: wall vs last-phase kernel time (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
path = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c.set_scan_path(path)
c.set_profiling(True)
for nq in (4, 16, 128, 256, 512, 1024, 4096):
    q = O.synth_queries(0x4D564632, nq, 768, 0)
    c.search(q, 100, 2)
    t0 = time.time(); c.search(q, 100, 2); w = (time.time() - t0) * 1e3
    tm = c.last_timing()
    print(f"path={path} kernel={tm.scan_kernel} nq={nq:5d} wall {w:7.2f} ms  last-phase scan {tm.scan_ms:7.2f} ms  {tm.scan_flops/(tm.scan_ms*1e-3)/1e12:6.1f} TFLOP/s (algorithmic)  launches {tm.scan_launches}", flush=True)
c.close()
