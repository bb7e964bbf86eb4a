import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mvf_oracle as O
from metrovector_amd import gpu as G
SEED = 0x4D564631
n, dim, nq, k = 12000, 96, 36, 25
rng = np.random.default_rng(21)
rows = O.synth_rows(SEED, 0, n, dim, 0)
rows *= (10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)[:, None]
rows[::11, ::3] *= 1e-7
rows[5] = 0.0
rows[17, 3] = np.inf
rows[23, 0] = np.nan
rows[29] = 3.0e38
q = O.synth_queries(SEED + 1, nq, dim, 0)
for metric in (2, 1, 0):
    sc, _, _ = O.scores(rows, 0, metric, q[0])
    key = sc.astype(np.float64) * (1 if metric == 0 else -1)
    key = np.where(np.isnan(key), np.inf, key)
    order = np.argsort(key, kind="stable")[:k]
    print("metric", metric, "oracle top:", order[:8], sc[order[:8]])
    for path in (1, 2, 3):
        with G.GpuCorpus.from_array(rows) as c:
            c.set_scan_path(path)
            res = c.search(q, k, metric)
        gi = res.indices[0].astype(np.int64)
        same = len(set(gi.tolist()) & set(order.tolist()))
        print(f"  path {path}: overlap {same}/{k}; first idx {gi[:6]} scores {res.scores[0][:6]} oracle-at-those {sc[gi[:6]]}")
