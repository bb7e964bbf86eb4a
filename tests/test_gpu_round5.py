"""-m gpu tests added in round 5.

Part 1 -- the WHOLE of BASELINE.json configs[4] on the one GPU a box has: the 100M x 1024 Float16 corpus (204.8 GB) as EIGHT
12.5M-row handles in one `mvfgpu_shardset` (the 8-way split north_star names; the lists travel by device copies instead of
RCCL because the shards share a device) and as ONE 100M-row handle -- the two must return the same rows and the same score
bits for all 1024 queries (merge(top-k per shard) == top-k(global)) -- and two of the queries against the
oracle's score of ALL 100M rows.  The shards select on their stored Float16 rows (MVF_I8_SHADOW=0): eight int8 shadows do not
fit beside 204.8 GB of rows, and the single handle has no room for one either.
"""
import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import assert_float_topk, oracle_scores_all_rows

pytestmark = pytest.mark.gpu
SEED = 0x4D564631


def test_cfg5_100m_x_1024_f16_l2_eight_shards_vs_one_handle_vs_the_oracle_over_all_rows(oracle, monkeypatch):
    n_shards, shard_rows, dim, nq, k = 8, 12_500_000, 1024, 1024, 100
    n = n_shards * shard_rows
    monkeypatch.setenv("MVF_I8_SHADOW", "0")  # read once per handle, at creation
    q = oracle.synth_queries(SEED + 1, nq, dim, 1)
    shards = [G.GpuCorpus.synthetic(shard_rows, dim, 1, SEED, row0=i * shard_rows) for i in range(n_shards)]
    try:
        with G.ShardSet(shards) as ss:
            info = ss.info()
            assert info.n_shards == n_shards and info.rows == n
            merged = ss.search(q, k, G.L2)
    finally:
        for s in shards:
            s.close()
    with G.GpuCorpus.synthetic(n, dim, 1, SEED) as c:
        assert c.info().rows == n
        one = c.search(q, k, G.L2)
    assert (merged.indices == one.indices).all(), "merge(top-k per shard) != top-k(one handle)"
    assert (merged.scores.view(np.uint32) == one.scores.view(np.uint32)).all()
    assert merged.indices.max() < n and (np.sort(merged.indices, axis=1)[:, 1:] != np.sort(merged.indices, axis=1)[:, :-1]).all()
    hit_shards = np.unique((merged.indices // shard_rows).astype(np.int64))
    assert len(hit_shards) == n_shards, "1024 x 100 results that miss a whole shard of a uniform corpus"
    sel = [5, 1000]
    all_sc = oracle_scores_all_rows(oracle, SEED, 0, n, dim, 1, 0, q[sel], chunk=500_000)
    for j, qi in enumerate(sel):
        assert_float_topk(0, one.scores[qi], one.indices[qi], all_sc[j], None, q[qi], k)
