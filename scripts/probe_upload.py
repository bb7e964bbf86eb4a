"""Upload pipeline: wall time of `upload + first batched search` on a cfg2-sized host corpus
(10M x 768 f32 = 30.72 GB, pageable memory), with and without the per-chunk norms / f16-shadow build beside the copy,
and with pinned double-buffered staging instead of the runtime's own bounce buffers.
usage: python scripts/probe_upload.py [rows] [dim]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from metrovector_amd import gpu as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rng = np.random.default_rng(1)
tile = rng.uniform(-1, 1, (100_000, dim)).astype(np.float32)
host = np.empty((n, dim), np.float32)
for r0 in range(0, n, len(tile)):
    h = min(len(tile), n - r0)
    host[r0:r0 + h] = tile[:h] * np.float32(1.0 + (r0 // len(tile)) * 1e-3)  # touched, distinct pages
q = rng.uniform(-1, 1, (1024, dim)).astype(np.float32)
gb = host.nbytes / 1e9
print(f"host corpus {n} x {dim} f32 = {gb:.2f} GB (pageable)", flush=True)
ref = None
for name, kw in (("round 1: pageable source, one pass; shadow + norms built by the first batched search", {"pinned_staging": False}),
                 ("pageable source in 256-MiB chunks, norms + shadow per chunk beside the copy", {"pinned_staging": False, "prepare_batched": True}),
                 ("DEFAULT: pinned 64-MiB double buffer (8 memcpy threads)", {}),
                 ("pinned 64-MiB double buffer, norms + shadow per chunk beside the copy", {"prepare_batched": True}),
                 ("pinned 256-MiB double buffer, norms + shadow per chunk", {"prepare_batched": True, "chunk_mib": 256}),
                 ("pinned 16-MiB double buffer, norms + shadow per chunk", {"prepare_batched": True, "chunk_mib": 16}),
                 ("round 1 again", {"pinned_staging": False})):
    t0 = time.perf_counter()
    c = G.GpuCorpus.from_array(host, **kw)
    t1 = time.perf_counter()
    r = c.search(q, 100, G.COSINE)
    t2 = time.perf_counter()
    r2 = c.search(q, 100, G.COSINE)
    t3 = time.perf_counter()
    if ref is None:
        ref = r.indices
    print(f"{name:100s} upload {t1 - t0:6.3f} s ({gb / (t1 - t0):5.1f} GB/s)  first batched search {1e3 * (t2 - t1):7.1f} ms  "
          f"second {1e3 * (t3 - t2):6.1f} ms  upload+first {t2 - t0:6.3f} s  same_results={bool((r.indices == ref).all())}", flush=True)
    c.close()
