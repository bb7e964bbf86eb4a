"""Full-size equivalence of the f16-shadow paths with the exact ones on cfg2/cfg3's corpus (development aid).

10M x 768 f32, all three metrics: (a) 1024 batched queries, scan path 3 (f16 MFMA on the shadow + exact re-score) vs
scan path 2 (exact f32 MFMA); (b) 200 single queries, scan path 4 (K1 on the shadow + exact re-score) vs scan path 1
(K1 on the f32 rows).  Reports index equality and the largest relative score difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
c = G.GpuCorpus.synthetic(10_000_000, 768, 0, 0x4D564631)
q = O.synth_queries(0x4D564632, 1024, 768, 0)
TOL = 1e-5

def tie_only(a_idx, a_sc, b_idx, b_sc):
    """True if the two top-k lists differ only by rows scoring within TOL (relative) of the k-th score."""
    kth = a_sc[-1]
    odd = [(a_sc[i]) for i in range(len(a_idx)) if a_idx[i] not in set(b_idx.tolist())] + \
          [(b_sc[i]) for i in range(len(b_idx)) if b_idx[i] not in set(a_idx.tolist())]
    return all(abs(v - kth) <= TOL * max(abs(kth), 1e-30) for v in odd)

bad = 0
for metric, name in ((2, "cosine"), (0, "L2"), (1, "dot")):
    c.set_scan_path(2); exact = c.search(q, 100, metric)
    c.set_scan_path(3); shadow = c.search(q, 100, metric)
    eq = (exact.indices == shadow.indices)
    # rows may swap inside exact ties of the two summation orders: compare as sets per query too
    sets = np.mean([set(a.tolist()) == set(b.tolist()) for a, b in zip(exact.indices, shadow.indices)])
    rel = np.max(np.abs(exact.scores - shadow.scores) / np.maximum(np.abs(exact.scores), 1e-30))
    diff = [i for i, (a, b) in enumerate(zip(exact.indices, shadow.indices)) if set(a.tolist()) != set(b.tolist())]
    ties = all(tie_only(exact.indices[i], exact.scores[i], shadow.indices[i], shadow.scores[i]) for i in diff)
    print(f"batched 1024q {name:6s}: positions equal {eq.mean():.6f}, per-query sets equal {sets:.6f}, max rel score diff {rel:.2e}; "
          f"{len(diff)} differing sets, all boundary ties within {TOL:g}: {ties}", flush=True)
    bad += not ties
    c.set_scan_path(1); e1 = [c.search(q[i], 100, metric) for i in range(200)]
    c.set_scan_path(4); s4 = [c.search(q[i], 100, metric) for i in range(200)]
    eq1 = np.mean([(a.indices == b.indices).mean() for a, b in zip(e1, s4)])
    sets1 = np.mean([set(a.indices[0].tolist()) == set(b.indices[0].tolist()) for a, b in zip(e1, s4)])
    rel1 = max(float(np.max(np.abs(a.scores - b.scores) / np.maximum(np.abs(a.scores), 1e-30))) for a, b in zip(e1, s4))
    diff1 = [i for i, (a, b) in enumerate(zip(e1, s4)) if set(a.indices[0].tolist()) != set(b.indices[0].tolist())]
    ties1 = all(tie_only(e1[i].indices[0], e1[i].scores[0], s4[i].indices[0], s4[i].scores[0]) for i in diff1)
    print(f"single  200q  {name:6s}: positions equal {eq1:.6f}, per-query sets equal {sets1:.6f}, max rel score diff {rel1:.2e}; "
          f"{len(diff1)} differing sets, all boundary ties within {TOL:g}: {ties1}", flush=True)
    bad += not ties1
c.close()
sys.exit(1 if bad else 0)
