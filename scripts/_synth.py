"""Synthetic queries / rows for the development probes, from the LIBRARY's own generator (the oracle is test
infrastructure: nothing outside tests/, smoke() and bench.py's CPU legs uses it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from metrovector_amd import _lib, gpu as G

_QDT = {0: torch.float32, 1: torch.float32, 2: torch.int8, 3: torch.uint8}


def synth_queries(seed, nq, dim, dtype):
    """[nq][dim] queries in the query type of `dtype` (f32 for Float32 / Float16 spaces, else the integer type)."""
    dq = torch.empty((nq, dim), dtype=_QDT[dtype], device="cuda:0")
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dtype, seed, 0, None))
    return dq.cpu().numpy()


def synth_rows(seed, row0, n, dim, dtype):
    """Rows row0 .. row0 + n - 1 of the synthetic corpus of `seed`, in the storage type."""
    with G.GpuCorpus.synthetic(n, dim, dtype, seed, row0=row0) as c:
        return c.gather_rows(np.arange(n, dtype=np.uint64))
