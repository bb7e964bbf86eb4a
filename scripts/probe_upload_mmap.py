#!/usr/bin/env python3
"""Upload straight off an mmap'd .mvf: a real file of ROWS x 768 f32 (default 6M rows = 18.4 GB) is
written by the C++ builder, opened with MvfReader, and its vector space uploaded through map_vector_range -> as_ptr ->
mvfgpu_corpus_create_ex with the page cache COLD (fsync + POSIX_FADV_DONTNEED) and WARM, for several copy-thread counts,
with and without the MADV_WILLNEED readahead, beside the same bytes in anonymous memory.

    python scripts/probe_upload_mmap.py [ROWS] [DIR]
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from metrovector_amd import gpu as G
from metrovector_amd.builder import MvfBuilder
from metrovector_amd.reader import MvfReader
from metrovector_amd.search import upload_space

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
where = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
dim, SEED = 768, 0x4D564631
nbytes = rows * dim * 4
path = os.path.join(where, f"probe_upload_{os.getpid()}.mvf")


def evict():
    fd = os.open(path, os.O_RDONLY)
    os.fsync(fd)
    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    os.close(fd)


try:
    ref = G.GpuCorpus.synthetic(rows, dim, 0, SEED)
    t0 = time.perf_counter()
    b = MvfBuilder()
    b.add_vector_space("big", dim, 0, 2, 0)
    b.reserve_vectors("big", rows)
    for r0 in range(0, rows, 500_000):
        b.add_vectors_raw("big", ref.read_rows(r0, min(500_000, rows - r0)))
    t1 = time.perf_counter()
    b.build().save(path)
    t2 = time.perf_counter()
    del b
    print(f"file: {nbytes / 1e9:.2f} GB in {where}; built in memory {t1 - t0:.2f} s, crc + save {t2 - t1:.2f} s "
          f"({nbytes / (t2 - t1) / 1e9:.1f} GB/s)", flush=True)
    q = np.random.default_rng(1).standard_normal((3, dim)).astype(np.float32)
    want = ref.search(q, 10, G.COSINE)
    ref.close()

    def upload(label, **env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            r = MvfReader.open(path)
            sp = r.vector_space("big")
            t = time.perf_counter()
            c = upload_space(sp, device=0)
            dt = time.perf_counter() - t
            got = c.search(q, 10, G.COSINE)
            ok = bool((got.indices == want.indices).all())
            c.close()
            r.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
        print(f"{label:58s} {dt:7.3f} s  {nbytes / dt / 1e9:6.1f} GB/s  results {'ok' if ok else 'WRONG'}", flush=True)

    for threads in ("8", "16", "32"):
        for adv in ("0",):  # MADV_WILLNEED ahead of the copy was measured here in round 4: no effect, removed
            evict()
            upload(f"COLD  threads={threads} advise={adv}", MVF_UPLOAD_THREADS=threads, MVF_UPLOAD_ADVISE=adv)
    for threads in ("4", "8", "16", "32", "64"):
        upload(f"WARM  threads={threads} advise=1", MVF_UPLOAD_THREADS=threads)
    upload("WARM  threads=16 advise=0", MVF_UPLOAD_THREADS="16", MVF_UPLOAD_ADVISE="0")
    upload("WARM  default")
    # the same bytes in anonymous memory
    r = MvfReader.open(path)
    anon = np.array(r.vector_space("big").map_vector_range(0, rows).to_numpy(dim))
    r.close()
    for threads in ("8", "16", "32"):
        os.environ["MVF_UPLOAD_THREADS"] = threads
        t = time.perf_counter()
        c = G.GpuCorpus.from_array(anon)
        dt = time.perf_counter() - t
        c.close()
        print(f"{'ANON  threads=' + threads:58s} {dt:7.3f} s  {nbytes / dt / 1e9:6.1f} GB/s", flush=True)
    os.environ.pop("MVF_UPLOAD_THREADS", None)
    t = time.perf_counter()
    r = MvfReader.open(path)
    r.validate_with_checksum()
    dt = time.perf_counter() - t
    r.close()
    print(f"{'checksum (warm)':58s} {dt:7.3f} s  {nbytes / dt / 1e9:6.1f} GB/s", flush=True)
finally:
    if os.path.exists(path):
        os.remove(path)
