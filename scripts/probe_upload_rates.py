"""Upload-rate probe (development aid): host rows -> HBM through mvfgpu_corpus_create."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metrovector_amd import gpu as G
n, dim = 1_500_000, 768
rows = np.random.default_rng(0).standard_normal((n, dim), dtype=np.float32)
gb = rows.nbytes / 1e9
for it in range(3):
    t = time.time(); c = G.GpuCorpus.from_array(rows); dt = time.time() - t
    print(f"contiguous f32 {gb:.2f} GB: {dt:.3f} s = {gb/dt:.1f} GB/s", flush=True)
    c.close()
# strided source (pitch conversion path)
wide = np.zeros((n // 2, dim + 5), np.float32)
view = wide[:, :dim]
t = time.time(); c = G.GpuCorpus.from_pointer(view.ctypes.data, n // 2, dim, 0, wide.strides[0]); dt = time.time() - t
print(f"strided f32 {view.shape[0]*dim*4/1e9:.2f} GB: {dt:.3f} s = {view.shape[0]*dim*4/1e9/dt:.1f} GB/s")
c.close()
# tightly packed but unaligned rows (dim not a multiple of 4 floats): repack path
odd = np.random.default_rng(1).standard_normal((n, 767), dtype=np.float32)
for it in range(2):
    t = time.time(); c = G.GpuCorpus.from_array(odd); dt = time.time() - t
    print(f"unaligned f32 dim 767 {odd.nbytes/1e9:.2f} GB: {dt:.3f} s = {odd.nbytes/1e9/dt:.1f} GB/s", flush=True)
    assert (c.read_rows(n - 3, 3) == odd[-3:]).all() and (c.read_rows(0, 2) == odd[:2]).all()
    c.close()
odd8 = np.random.default_rng(2).integers(-128, 128, (3_000_000, 777), dtype=np.int8)
t = time.time(); c = G.GpuCorpus.from_array(odd8); dt = time.time() - t
print(f"unaligned i8 dim 777 {odd8.nbytes/1e9:.2f} GB: {dt:.3f} s = {odd8.nbytes/1e9/dt:.1f} GB/s", flush=True)
assert (c.read_rows(2_999_990, 10) == odd8[-10:]).all()
c.close()
# mmap'd file source (the reference's situation)
import tempfile
from metrovector_amd.builder import MvfBuilder
from metrovector_amd.reader import MvfReader
from metrovector_amd.search import upload_space
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    p = os.path.join(d, "big.mvf")
    b = MvfBuilder(); b.add_vector_space("s", dim, 0, 2, 0); b.add_vectors("s", rows[:500_000]); b.build().save(p)
    r = MvfReader.open(p); sp = r.vector_space("s")
    for it in range(2):
        t = time.time(); c = upload_space(sp); dt = time.time() - t
        print(f"mmap'd .mvf {500_000*dim*4/1e9:.2f} GB: {dt:.3f} s = {500_000*dim*4/1e9/dt:.1f} GB/s")
        c.close()
