"""Seeded random sweep over every scan path: random shapes, batch sizes, k, metrics, storage types, deletions, vector ids
and index bases against the oracle ("filter, then search" on the CPU).  Each case is small; the point is the combinations
no hand-written case names -- batch sizes that are not a multiple of any tile, k next to the live row count, a scan path
forced on a shape it was not tuned for, several searches on one handle in a row (state carried between calls)."""
import os

import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import PAD, assert_exact, assert_float_topk, recall_at_k

pytestmark = pytest.mark.gpu
SEED = 0x4D564631
OFFSET = int(os.environ.get("MVF_FUZZ_OFFSET", "0"))  # soak runs: other cases than the 300 + 24 + 12 the suite pins


def _dims(rng):
    return int(rng.choice([1, 2, 3, 7, 8, 16, 31, 64, 100, 128, 200, 257, 384, 768, 1000]))


@pytest.mark.parametrize("case", range(300))
def test_random_case_against_the_oracle(oracle, case):
    rng = np.random.default_rng(1000 + case + OFFSET)
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
                for qi in (range(nq) if nq <= 8 else sorted(set([0, nq // 2, nq - 1]))):  # small batches: every query
                    got = res.indices[qi]
                    kk = min(k, len(pos))
                    assert (got[kk:] == PAD).all(), tag
                    local = np.array([inv[int(g)] for g in got[:kk]], np.uint64)
                    sc, _, _ = oracle.scores(sub, dtype, metric, q[qi])
                    padded = np.concatenate([local, got[kk:]])
                    assert_float_topk(metric, res.scores[qi], padded, sc, rf, q[qi], k)
                if dim >= 16:  # tiny dimensions: crowds of scores within the tolerance of each other (checked above)
                    # one boundary tie ranked the other way (legitimate within the tolerance; the check above is the
                    # criterion) is 2 % of a 5 x 10 answer: allow one such row on small answers
                    # (answers of one or two entries: a single boundary tie is the whole answer -- the criterion above decides)
                    nres = nq * min(k, len(pos))
                    if nres >= 3:
                        assert recall_at_k(res.indices, want) >= min(0.99, max(1.0 - 1.5 / nres, 1.0 - 1.0 / 3.0)), tag


@pytest.mark.parametrize("case", range(24))
def test_random_shard_splits_merge_to_the_unsharded_answer(oracle, case):
    """Row-range shards (random cut points, empty-ish and one-row shards included), per-shard deletions and ids, through
    the single-process shard set: the merged top-k equals the unsharded corpus' (ties by global position)."""
    rng = np.random.default_rng(5000 + case + OFFSET)
    dtype = int(rng.integers(0, 4))
    metric = int(rng.integers(0, 3))
    dim = int(rng.choice([8, 33, 64, 200]))
    n = int(rng.choice([300, 5000, 40000]))
    nshards = int(rng.integers(2, 6))
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, n), nshards - 1, replace=False)) + [n]
    rows = oracle.synth_rows(SEED + 99 + case, 0, n, dim, dtype)
    if dtype >= 2:
        rows[rng.choice(n, n // 10, replace=False)] = rows[0]  # integer ties across shard boundaries
    dead = rng.random(n) < 0.2 if rng.random() < 0.5 else None
    ids = rng.permutation(np.arange(7_000_000, 7_000_000 + n)).astype(np.uint64) if rng.random() < 0.4 else None
    nq = int(rng.choice([1, 3, 40, 130]))
    k = int(rng.choice([1, 10, 100, 400]))
    q = oracle.synth_queries(SEED + 3 * case, nq, dim, dtype)

    def prepare(c, a, b):
        if dead is not None:
            c.set_tombstones(np.packbits(dead[a:b], bitorder="little"))
        if ids is not None:
            c.set_vector_ids(ids[a:b])

    shards = []
    try:
        for a, b in zip(cuts[:-1], cuts[1:]):
            c = G.GpuCorpus.from_array(rows[a:b], index_base=a)
            prepare(c, a, b)
            shards.append(c)
        with G.ShardSet(shards) as ss:
            got = ss.search(q, k, metric)
        with G.GpuCorpus.from_array(rows) as whole:
            prepare(whole, 0, n)
            want = whole.search(q, k, metric)
    finally:
        for s in shards:
            s.close()
    tag = f"case {case}: dtype {dtype} metric {metric} n {n} dim {dim} cuts {cuts} nq {nq} k {k}"
    if dtype >= 2:
        assert (got.indices == want.indices).all(), tag
        assert (got.raw == want.raw).all(), tag
    else:
        assert recall_at_k(got.indices, want.indices) >= 0.995, tag
        assert ((got.indices == PAD) == (want.indices == PAD)).all(), tag
        live = want.indices != PAD
        gs, ws = np.where(live, got.scores, 0.0), np.where(live, want.scores, 0.0)
        scale = max(1.0, float(np.abs(ws).max(initial=1.0)))
        assert np.abs(np.sort(gs, axis=1) - np.sort(ws, axis=1)).max() <= 2e-5 * scale, tag


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
def test_every_combination_of_the_tuning_switches_returns_the_same_rows(oracle, dtype):
    """The environment switches choose kernels, tiles, shadows and schedules -- never results."""
    import itertools
    import os
    switches = {"MVF_K2_PP": ("0", "1"), "MVF_K2_DMA": ("0", "1"), "MVF_I8_SHADOW": ("0", "1"), "MVF_F16_SHADOW": ("0", "1"),
                "MVF_K2_TILE": ("64", "256"), "MVF_K2_GROWTH": ("2", "8"), "MVF_QS_REFINE": ("0", "1"),
                "MVF_K2_PERSISTENT16": ("0", "1"), "MVF_K2_SB": ("0", "1")}
    n, dim, nq, k = 90_000, 72, 140, 25
    metric = dtype % 3
    rows = oracle.synth_rows(SEED + dtype, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 50 + dtype, nq, dim, dtype)
    rng = np.random.default_rng(77 + dtype)
    combos = list(itertools.product(*switches.values()))
    picks = [combos[i] for i in rng.choice(len(combos), 48, replace=False)]
    with G.GpuCorpus.from_array(rows) as c:
        want = c.search(q, k, metric)
        want_small = c.search(q[:40], k, metric)   # one tile of <= 64 queries: the streaming MFMA kernel's range
        for combo in picks:
            env = dict(zip(switches.keys(), combo))
            os.environ.update(env)
            try:
                got = c.search(q, k, metric)
                got_small = c.search(q[:40], k, metric)
            finally:
                for name in env:
                    os.environ.pop(name, None)
            if dtype >= 2:
                assert (got.indices == want.indices).all() and (got.raw == want.raw).all(), env
                assert (got_small.indices == want_small.indices).all() and (got_small.raw == want_small.raw).all(), env
            else:  # exact f32 MFMA selection (both shadows off) sums in another order: rows within 1e-5 may swap
                assert recall_at_k(got.indices, want.indices) >= 0.999, env
                assert recall_at_k(got_small.indices, want_small.indices) >= 0.999, env
                assert np.abs(np.sort(got.scores, axis=1) - np.sort(want.scores, axis=1)).max() <= 2e-5 * max(1.0, float(np.abs(want.scores).max())), env


@pytest.mark.parametrize("case", range(12))
def test_random_short_rows_in_long_pieces(oracle, case):
    """The streaming kernel's long guarded pieces are only in play from a few million short rows on (smaller corpora get
    one short chunk per block): random shapes of <= 256-byte rows at 2.2M..5M rows, one to four queries per search,
    deletions and ids at random, against "filter, then search" on the oracle."""
    rng = np.random.default_rng(5000 + case + OFFSET)
    dtype = int(rng.integers(0, 4))
    metric = int(rng.integers(0, 3))
    es = {0: 4, 1: 2, 2: 1, 3: 1}[dtype]
    dim = int(rng.choice([16, 24, 32, 40, 50, 64, 100, 128, 200, 256])) // es
    dim = max(dim, 4)
    n = int(rng.integers(2_200_000, 5_000_000))
    rows = oracle.synth_rows(SEED + 31 * case, 0, n, dim, dtype)
    dead = (rng.random(n) < rng.choice([0.001, 0.2])) if rng.random() < 0.5 else None
    ids = rng.permutation(np.arange(7, 7 + n)).astype(np.uint64) if rng.random() < 0.3 else None
    with G.GpuCorpus.from_array(rows) as c:
        if dead is not None:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
        if ids is not None:
            c.set_vector_ids(ids)
        c.set_scan_path(1)
        live = np.ones(n, bool) if dead is None else ~dead
        sub, pos = rows[live], np.nonzero(live)[0]
        for nq, k in ((1, int(rng.choice([1, 10, 100, 1000]))), (int(rng.integers(2, 5)), int(rng.choice([5, 64])))):
            q = oracle.synth_queries(SEED + 11 * case + nq, nq, dim, dtype)
            res = c.search(q, k, metric)
            osc, oidx, oraw = oracle.search(sub, dtype, metric, q, k)
            p = pos[oidx.astype(np.int64)]
            want = ids[p] if ids is not None else p.astype(np.uint64)
            tag = f"case {case}: dtype {dtype} metric {metric} n {n} dim {dim} nq {nq} k {k}"
            if dtype >= 2:
                assert (res.indices == want).all(), tag
                assert (res.raw == oraw).all(), tag
            else:
                assert recall_at_k(res.indices, want) >= 0.99, tag
                assert np.abs(res.scores - osc).max() <= 1e-5 * max(1.0, float(np.abs(osc).max())), tag


@pytest.mark.parametrize("case", range(48))
def test_random_large_k_against_the_oracle(oracle, case):
    """The same sweep for k beyond one pass (MVFGPU_K_PER_PASS = 1024; round 4): random k up to 16384 next to, at and beyond
    the live row count, batch sizes on both sides of the four-query pass, deletions, ids, index bases, and a forced scan
    path -- which must not matter: above one pass every request is served by passes of the exact streaming kernel."""
    rng = np.random.default_rng(9000 + case + OFFSET)
    dtype = int(rng.integers(0, 4))
    metric = int(rng.integers(0, 3))
    dim = int(rng.choice([1, 3, 8, 16, 33, 64, 100, 256]))
    n = int(rng.choice([17, 1023, 1024, 1025, 2047, 3000, 9000, 40000]))
    rows = oracle.synth_rows(SEED + 500 + case, 0, n, dim, dtype)
    if dtype >= 2 and rng.random() < 0.5:
        rows[rng.choice(n, n // 3, replace=False)] = rows[0]   # big tie groups across the pass boundaries
    dead = rng.random(n) < rng.choice([0.05, 0.5]) if rng.random() < 0.4 else None
    ids = rng.permutation(np.arange(10_000, 10_000 + n)).astype(np.uint64) if rng.random() < 0.3 else None
    index_base = int(rng.choice([0, 5, 1 << 33]))
    with G.GpuCorpus.from_array(rows, index_base=index_base) as c:
        if dead is not None:
            c.set_tombstones(np.packbits(dead, bitorder="little"))
        if ids is not None:
            c.set_vector_ids(ids)
        for step in range(2):
            nq = int(rng.choice([1, 2, 4, 5, 9, 70]))
            k = int(rng.choice([1025, 1500, 2048, 2049, 4000, 16384]))
            c.set_scan_path(int(rng.choice([0, 1, 2, 3, 5, 6] if dtype < 2 else [0, 1, 2])))
            q = oracle.synth_queries(SEED + 11 * case + step, nq, dim, dtype)
            res = c.search(q, k, metric)
            live = np.ones(n, bool) if dead is None else ~dead
            sub, pos = rows[live], np.nonzero(live)[0]
            osc, oidx, oraw = oracle.search(sub, dtype, metric, q, k)
            ok = oidx != PAD
            want = np.full(oidx.shape, PAD, np.uint64)
            p = pos[oidx[ok].astype(np.int64)]
            want[ok] = ids[p] if ids is not None else p.astype(np.uint64) + np.uint64(index_base)
            tag = f"case {case} step {step}: dtype {dtype} metric {metric} n {n} dim {dim} nq {nq} k {k}"
            assert ((res.indices == PAD) == (want == PAD)).all(), tag
            if dtype >= 2:
                assert (res.indices == want).all(), tag
                assert (res.raw == oraw).all(), tag
                assert (res.scores.view(np.uint32) == osc.view(np.uint32)).all(), tag
            else:
                inv = {int(v): i for i, v in enumerate(ids[pos] if ids is not None else pos.astype(np.uint64) + np.uint64(index_base))}
                rf = sub.astype(np.float32)
                for qi in sorted(set([0, nq // 2, nq - 1])):
                    kk = min(k, len(pos))
                    local = np.array([inv[int(g)] for g in res.indices[qi][:kk]], np.uint64)
                    sc = oracle.scores(sub, dtype, metric, q[qi])[0]
                    assert_float_topk(metric, res.scores[qi], np.concatenate([local, res.indices[qi][kk:]]), sc, rf, q[qi], k)
