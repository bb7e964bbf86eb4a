"""One query through a shard set of S handles on one device (rehearsal: device copies instead of the all-gather), wall us
per blocking mvfgpu_shardset_search call, small transfers in place (default) against staged copies (MVF_HOST_ZC_*=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metrovector_amd import gpu as G

for (n, dim, dt, k) in ((10_000, 128, 0, 10), (1_000_000, 128, 0, 10), (10_000_000, 768, 0, 100)):
    for S in (1, 2, 4):
        per = n // S
        shards = [G.GpuCorpus.synthetic(per, dim, dt, 0x4D564631, row0=s * per) for s in range(S)]
        q = np.random.default_rng(1).random((1, dim), dtype=np.float32) * 2 - 1
        out = []
        for mode in ("0", None):
            for var in ("MVF_HOST_ZC_QUERY", "MVF_HOST_ZC_RESULTS"):
                os.environ.pop(var, None) if mode is None else os.environ.__setitem__(var, mode)
            with G.ShardSet(shards) as ss:
                for _ in range(20):
                    ss.search(q, k, G.COSINE)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter(); ss.search(q, k, G.COSINE); ts.append(time.perf_counter() - t0)
                ts.sort()
                out.append(ts[100] * 1e6)
        print(f"n={n} dim={dim} k={k} shards={S}: copies {out[0]:8.1f} us   in place {out[1]:8.1f} us", flush=True)
        for s in shards:
            s.close()
