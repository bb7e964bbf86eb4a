"""Row-range sharding of one vector space over the GPUs of a node — one
process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference has no distributed code; this is the multi-GPU
form of its scan (examples/similarity_search.rs:140-176): GPU g holds rows
[g*ceil(N/G), min(N,(g+1)*ceil(N/G))) and searches them with global indices;
the only exchange step is ONE all-gather of the per-shard top-k lists, packed
as {u64 indices | f32 scores | i32 raw} = 16 bytes per result (nq*k results per
rank — latency-bound, 1.6 MB at nq=1024,k=100), after which every rank merges
G sorted lists:  merge(top-k per shard) == top-k(global), because selection
is by a total order (score key, global index).

torch is plumbing here (device buffers, streams, the collective); the scan
and the merge run in libmvf_gpu.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable

import numpy as np

from . import _lib
from .gpu import GpuCorpus, SearchResult, merge_topk_host, query_dtype_code


def shard_range(total_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row range of `rank`."""
    per = -(-total_rows // world_size)
    lo = min(total_rows, rank * per)
    return lo, min(total_rows, lo + per)


def _all_gather(t, group):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = t.contiguous().view(-1)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices):
        # stage the lists through the host; the production backend is nccl (= RCCL)
        host = torch.empty(world * flat.numel(), dtype=t.dtype)
        dist.all_gather_into_tensor(host, flat.cpu(), group=group)
        return host.to(t.device).view((world,) + tuple(t.shape)), None
    out = torch.empty(world * flat.numel(), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, flat, group=group, async_op=True)
    return out.view((world,) + tuple(t.shape)), work


class ShardedSearcher:
    """GPU path: local `mvfgpu_search_device` -> one RCCL all-gather of the packed list ->
    `mvfgpu_merge_topk_packed_device`, all on torch's current stream; results stay on the device.

    always_exchange=True runs the collective and the merge at world size 1 as well (a 1-rank RCCL group):
    it exists so the exchange path can be exercised on a single GPU."""

    def __init__(self, corpus: GpuCorpus, group=None, always_exchange: bool = False):
        import torch
        self.corpus = corpus
        self.group = group
        self.always_exchange = always_exchange
        inf = corpus.info()
        self.dtype = inf.data_type
        self.dim = inf.dimension
        self.device = torch.device("cuda", inf.device)
        self._bufs = {}
        self.timing = False  # record CUDA events around the local search and the exchange step (bench.py)
        self._ev = []

    def _buffers(self, nq: int, k: int, world: int):
        import torch
        key = (nq, k, world)
        if key not in self._bufs:
            d = self.device
            n = nq * k
            # this rank's list in the packed layout of include/mvf_gpu.h (MVFGPU_PACKED_LIST_BYTES): int64 words
            # [0, n) = indices, then n f32 scores, then n i32 raw; the three search outputs are views into it
            mine = torch.empty(2 * n, dtype=torch.int64, device=d)
            tail = mine[n:].view(torch.int32)
            self._bufs[key] = dict(
                mine=mine, i=mine[:n].view(nq, k), s=tail[:n].view(torch.float32).view(nq, k), r=tail[n:].view(nq, k),
                all=torch.empty(world * 2 * n, dtype=torch.int64, device=d),
                os=torch.empty((nq, k), dtype=torch.float32, device=d), oi=torch.empty((nq, k), dtype=torch.int64, device=d),
                orr=torch.empty((nq, k), dtype=torch.int32, device=d))
        return self._bufs[key]

    def search(self, d_queries, k: int, metric: int):
        """d_queries: torch tensor on the corpus' device, [nq, dim], f32 (float spaces) or int8/uint8.
        Returns (scores f32[nq,k], indices i64[nq,k] (bit pattern of u64), raw i32[nq,k]) device tensors,
        identical on every rank."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        nq = d_queries.shape[0]
        b = self._buffers(nq, k, world)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.timing else None
        if ev:
            ev[0].record()
        self.corpus.search_device(d_queries.data_ptr(), query_dtype_code(self.dtype), d_queries.shape[1], nq, k, metric,
                                  b["s"].data_ptr(), b["i"].data_ptr(), b["r"].data_ptr(), stream)
        if ev:
            ev[1].record()
        if world == 1 and not (self.always_exchange and dist.is_initialized()):
            if ev:
                ev[2].record()
                self._ev.append(ev)
            return b["s"], b["i"], b["r"]
        if dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices):
            # stage the list through the host; the production backend is nccl (= RCCL)
            host = torch.empty(b["all"].numel(), dtype=torch.int64)
            dist.all_gather_into_tensor(host, b["mine"].cpu(), group=self.group)
            b["all"].copy_(host)
        else:
            dist.all_gather_into_tensor(b["all"], b["mine"], group=self.group)  # enqueued on the current stream
        _lib.gpu_check(_lib.gpu().mvfgpu_merge_topk_packed_device(
            b["all"].data_ptr(), world, nq, k, metric, self.dtype,
            b["os"].data_ptr(), b["oi"].data_ptr(), b["orr"].data_ptr(), self.device.index or 0, C.c_void_p(stream)))
        if ev:
            ev[2].record()
            self._ev.append(ev)
        return b["os"], b["oi"], b["orr"]

    def take_timings(self):
        """(mean local-search ms, mean exchange+merge ms) over the searches since timing was switched on; device
        time between events on torch's current stream.  Synchronises."""
        import torch
        torch.cuda.synchronize(self.device)
        evs, self._ev = self._ev, []
        if not evs:
            return 0.0, 0.0
        a = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
        b = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)
        return a, b


def sharded_search_host(local_search: Callable[[], SearchResult], metric: int, data_type: int, group=None) -> SearchResult:
    """Host-buffer variant of the same protocol (any torch.distributed backend,
    e.g. gloo): `local_search()` returns this rank's SearchResult with GLOBAL
    indices; the per-shard lists are all-gathered and merged with
    `mvfgpu_merge_topk_host`.  Used by the CPU tests of the N>1 path, where the
    local search is stubbed (there is no CPU search in the product)."""
    import torch
    import torch.distributed as dist
    local = local_search()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    ts = torch.from_numpy(np.ascontiguousarray(local.scores))
    ti = torch.from_numpy(np.ascontiguousarray(local.indices).view(np.int64))
    tr = torch.from_numpy(np.ascontiguousarray(local.raw))
    gs, w1 = _all_gather(ts, group)
    gi, w2 = _all_gather(ti, group)
    gr, w3 = _all_gather(tr, group)
    for w in (w1, w2, w3):
        w.wait()
    return merge_topk_host(gs.numpy(), gi.numpy().view(np.uint64), gr.numpy(), metric, data_type)
