"""Seeded random sweep over every scan path: random shapes, batch sizes, k, metrics, storage types, deletions, vector ids
and index bases against the oracle ("filter, then search" on the CPU).  Each case is small; the point is the combinations
no hand-written case names -- batch sizes that are not a multiple of any tile, k next to the live row count, a scan path
forced on a shape it was not tuned for, several searches on one handle in a row (state carried between calls)."""
import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import PAD, assert_exact, assert_float_topk, recall_at_k

pytestmark = pytest.mark.gpu
SEED = 0x4D564631


def _dims(rng):
    return int(rng.choice([1, 2, 3, 7, 8, 16, 31, 64, 100, 128, 200, 257, 384, 768, 1000]))


@pytest.mark.parametrize("case", range(300))
def test_random_case_against_the_oracle(oracle, case):
    rng = np.random.default_rng(1000 + case)
    dtype = int(rng.integers(0, 4))
    metric = int(rng.integers(0, 3))
    dim = _dims(rng)
    n = int(rng.choice([1, 2, 17, 255, 256, 257, 1000, 4097, 9000, 20000, 50000, 150000]))
    if n * dim > 12_000_000:
        n = max(1, 12_000_000 // dim)
    rows = oracle.synth_rows(SEED + case, 0, n, dim, dtype)
    dead = None
    if rng.random() < 0.5:
        dead = rng.random(n) < rng.choice([0.01, 0.3, 0.9, 1.0])
    ids = rng.permutation(np.arange(10_000, 10_000 + n)).astype(np.uint64) if rng.random() < 0.3 else None
    index_base = int(rng.choice([0, 5, 1 << 33]))
    with G.GpuCorpus.from_array(rows, index_base=index_base) as c:
        if dead is not None:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
        if ids is not None:
            c.set_vector_ids(ids)
        for step in range(3):  # several searches on the same handle: different paths, batch sizes and k
            path = int(rng.choice([0, 1, 2, 3, 4, 5, 6] if dtype < 2 else [0, 1, 2]))
            nq = int(rng.choice([1, 2, 3, 4, 5, 16, 63, 64, 65, 130, 257]))
            k = int(min(rng.choice([1, 2, 10, 33, 100, 256, 1000]), 1024))
            q = oracle.synth_queries(SEED + 7 * case + step, nq, dim, dtype)
            c.set_scan_path(path)
            res = c.search(q, k, metric)
            live = np.ones(n, bool) if dead is None else ~dead
            sub = rows[live]
            pos = np.nonzero(live)[0]
            if len(pos) == 0:
                assert (res.indices == PAD).all()
                continue
            osc, oidx, oraw = oracle.search(sub, dtype, metric, q, k)
            want = np.full(oidx.shape, PAD, np.uint64)
            ok = oidx != PAD
            p = pos[oidx[ok].astype(np.int64)]
            want[ok] = ids[p] if ids is not None else p.astype(np.uint64) + np.uint64(index_base)
            tag = f"case {case} step {step}: dtype {dtype} metric {metric} n {n} dim {dim} nq {nq} k {k} path {path}"
            if dtype >= 2:
                assert (res.indices == want).all(), tag
                assert (res.raw == oraw).all(), tag
                assert (res.scores.view(np.uint32) == osc.view(np.uint32)).all(), tag
            else:
                # positions in the live sub-corpus for the tolerance-aware comparison
                inv = {int(v): i for i, v in enumerate(ids[pos] if ids is not None else pos.astype(np.uint64) + np.uint64(index_base))}
                rf = sub.astype(np.float32)
                for qi in sorted(set([0, nq // 2, nq - 1])):
                    got = res.indices[qi]
                    kk = min(k, len(pos))
                    assert (got[kk:] == PAD).all(), tag
                    local = np.array([inv[int(g)] for g in got[:kk]], np.uint64)
                    sc, _, _ = oracle.scores(sub, dtype, metric, q[qi])
                    padded = np.concatenate([local, got[kk:]])
                    assert_float_topk(metric, res.scores[qi], padded, sc, rf, q[qi], k)
                if dim >= 16:  # tiny dimensions: crowds of scores within the tolerance of each other (checked above)
                    assert recall_at_k(res.indices, want) >= 0.99, tag
