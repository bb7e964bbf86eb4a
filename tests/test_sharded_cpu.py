"""The N>1 path on CPU: world_size-2 gloo processes run the shard / gather /
merge protocol of metrovector_amd.sharded with the local search stubbed by the
oracle (the product has no CPU search).  Checks merge(top-k per shard) ==
top-k(global) and the shard ranges."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_rows_without_overlap():
    from metrovector_amd.sharded import shard_range
    for total in (0, 1, 7, 8, 9, 1000, 10_000_000, 100_000_001):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
                assert b == c and a <= b
            per = -(-total // world)
            assert all(b - a <= per for a, b in ranges)


def _worker(rank, world, port, dtype, metric, n, dim, nq, k, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from metrovector_amd.gpu import SearchResult
    from metrovector_amd.sharded import shard_range, sharded_search_host
    from oracle import mvf_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, world, rank)
        rows = O.synth_rows(123, lo, hi - lo, dim, dtype)      # this rank's shard only
        queries = O.synth_queries(124, nq, dim, dtype)

        def local():
            sc, idx, raw = O.search(rows, dtype, metric, queries, k, index_base=lo)
            return SearchResult(sc, idx, raw)

        res = sharded_search_host(local, metric, dtype)
        q.put((rank, res.scores, res.indices, res.raw))
    finally:
        dist.destroy_process_group()


def _run_world(oracle, world, dtype, metric, n, k, dim=32, nq=3):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype, metric, n, dim, nq, k, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    rows = oracle.synth_rows(123, 0, n, dim, dtype)
    queries = oracle.synth_queries(124, nq, dim, dtype)
    ws, wi, wr = oracle.search(rows, dtype, metric, queries, k)
    assert sorted(o[0] for o in outs) == list(range(world))
    for _, sc, idx, raw in outs:   # every rank holds the same, global answer
        assert (idx == wi).all()
        assert (sc.view(np.uint32) == ws.view(np.uint32)).all()
        assert (raw == wr).all()


@pytest.mark.parametrize("dtype,metric,n,k", [(0, 2, 5000, 100), (2, 1, 3001, 64), (1, 0, 40, 100), (3, 0, 999, 10),
                                             (2, 0, 30_000, 12_000)])   # any k: 2 x 12 000 entries per query through the host merge
def test_two_rank_gloo_merge_equals_global(oracle, dtype, metric, n, k):
    _run_world(oracle, 2, dtype, metric, n, k)


@pytest.mark.parametrize("world,dtype,metric,n,k", [
    (3, 0, 2, 5000, 100),   # N not divisible by the world (1667 + 1667 + 1666)
    (3, 2, 1, 2, 5),        # N < world: shard_range gives the last rank lo == hi (an EMPTY shard), k > N pads
    (8, 1, 0, 1003, 50),    # eight ranks (the node's shape), ragged last shard
    (8, 3, 0, 5, 8),        # eight ranks, five rows: three empty tail shards, k > rows
])
def test_wider_worlds_ragged_and_empty_shards(oracle, world, dtype, metric, n, k):
    """The N>1 protocol at the world sizes the node has: shard_range when N is no multiple of the
    world and when it is smaller (empty tail shards contribute padding-only lists), merged == global on every rank."""
    from metrovector_amd.sharded import shard_range
    ranges = [shard_range(n, world, r) for r in range(world)]
    if n < world:
        assert any(lo == hi for lo, hi in ranges)
    _run_world(oracle, world, dtype, metric, n, k)
