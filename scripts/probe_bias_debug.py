"""Development aid: the batched path (folded pre-filter, scan_mfma16_bias.inc) against the streaming kernel on integer rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _synth as O  # the library's own generator (scripts/_synth.py)
from metrovector_amd import gpu as G
os.environ["MVF_DEBUG_REPAIR"] = "1"
for n, dim, nq in ((20011, 96, 300), (300_000, 96, 300), (3_000_000, 768, 256)):
    for dtype in (2, 3):
        rows = O.synth_rows(5, 0, n, dim, dtype)
        q = O.synth_queries(6, nq, dim, dtype)
        with G.GpuCorpus.from_array(rows) as c:
            for metric in (0, 1, 2):
                c.set_scan_path(1)
                want = c.search(q[:40], 33, metric)
                c.set_scan_path(0)
                c.search(q, 33, metric)
                t0 = time.perf_counter()
                got = c.search(q, 33, metric)
                dt = (time.perf_counter() - t0) * 1e3
                bad = int((got.indices[:40] != want.indices).any(axis=1).sum())
                print(f"n={n} dim={dim} nq={nq} dtype={dtype} metric={metric}: {dt:.2f} ms, queries differing from K1: {bad}/40", flush=True)

# float rows whose norms / scales differ wildly from row to row (per-lane bounds are loose there), int8-shadow selection
rng = np.random.default_rng(1)
for n, dim, nq in ((2_000_000, 256, 1024),):
    base = rng.standard_normal((n, dim), dtype=np.float32)
    for name, scale in (("equal norms", np.ones(n, np.float32)), ("norms x [0.1, 10) lognormal", np.exp(rng.uniform(-2.3, 2.3, n)).astype(np.float32)),
                        ("heavy-tailed elements", None)):
        rows = base * scale[:, None] if scale is not None else (base ** 3).astype(np.float32)
        q = rng.standard_normal((nq, dim), dtype=np.float32)
        with G.GpuCorpus.from_array(rows) as c:
            for metric in (2, 1, 0):
                c.set_scan_path(1)
                want = c.search(q[:8], 50, metric)
                c.set_scan_path(0)
                for _ in range(3):
                    t0 = time.perf_counter()
                    got = c.search(q, 50, metric)
                    dt = (time.perf_counter() - t0) * 1e3
                bad = int((got.indices[:8] != want.indices).any(axis=1).sum())
                print(f"f32 {name}: n={n} dim={dim} nq={nq} metric={metric}: {dt:.2f} ms (3rd search), queries differing from K1: {bad}/8", flush=True)
