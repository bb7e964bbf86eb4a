"""Two batched searches of cfg4 (50M x 768 int8 dot, 256 queries; argv[1] = 2, default) or of a cfg5 shard
(12.5M x 1024 f16 L2, 1024 queries; argv[1] = 1), quiet, for rocprofv3 (the profiles/r01_cfg4_* / r01_cfg5_* files)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
dt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, dim, nq, metric = (50_000_000, 768, 256, 1) if dt == 2 else (12_500_000, 1024, 1024, 2)
c = G.GpuCorpus.synthetic(n, dim, dt, 0x4D564631)
q = O.synth_queries(0x4D564632, nq, dim, dt)
for _ in range(2):
    c.search(q, 100, metric)
c.close()
