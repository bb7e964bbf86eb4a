#!/usr/bin/env python3
"""Two 8M-row shards of a 16M x 1024 f16 corpus (the 2-rank rehearsal of bench.py's cfg5_strong leg), merged on the host,
four of the 1024 queries against the oracle's score of EVERY row with the tolerance-aware criterion of tests/_util.py:
is the one row in 400 that differed from the oracle's list in that rehearsal a boundary tie or a miss?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from metrovector_amd import gpu as G
from oracle import mvf_oracle as O
from _util import assert_float_topk, oracle_scores_all_rows
SEED, n, dim, nq, k = 0x4D564631, 16_000_000, 1024, 1024, 100
q = O.synth_queries(SEED + 1, nq, dim, 1)
sel = [0, nq // 3, 2 * nq // 3, nq - 1]
parts = []
for lo in (0, 8_000_000):
    with G.GpuCorpus.synthetic(8_000_000, dim, 1, SEED, row0=lo) as c:
        parts.append(c.search(q, k, G.L2))
m = G.merge_topk_host(np.stack([p.scores for p in parts]), np.stack([p.indices for p in parts]), None, G.L2, 1)
all_sc = oracle_scores_all_rows(O, SEED, 0, n, dim, 1, 0, q[sel])
for j, qi in enumerate(sel):
    key = all_sc[j].astype(np.float64)
    order = np.lexsort((np.arange(n), key))[:k + 3]
    got = m.indices[qi].astype(np.int64)
    odd = sorted(set(got.tolist()) ^ set(order[:k].tolist()))
    print(f"query {qi}: differing rows {odd}; oracle scores around rank k: {key[order[k-3:k+3]]}; "
          f"oracle scores of the differing rows: {[float(key[r]) for r in odd]}", flush=True)
    assert_float_topk(0, m.scores[qi], m.indices[qi], all_sc[j], None, q[qi], k)
print("all four lists are the oracle's up to boundary ties within 1e-5")
