"""Two batched searches of cfg3 (10M x 768 f32 cosine, 1024 queries) for profiling (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
c.set_scan_path(path)
q = O.synth_queries(0x4D564632, nq, 768, 0)
for _ in range(3):
    c.search(q, 100, 2)
c.close()
