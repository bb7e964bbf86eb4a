"""-m gpu tests of the blocking host call's completion wait (round 4, second session).

A small mvfgpu_search (results <= 256 KiB, written in place into pinned host memory) does not wait on its stream: the final
select stores a sequence number behind its results and the host spins on that word -- hipStreamSynchronize learns of a
finished kernel ~5 us later (profiles/r04_flag_wait.txt).  MVF_HOST_FLAG_WAIT=0 brings the stream wait back.  Same answers
either way; searches in a row, batches (several select blocks: the last one to finish signals), threads sharing a handle
and device-API searches interleaved with host ones must all see complete results."""
import threading

import numpy as np
import pytest

from metrovector_amd import gpu as G

from _util import assert_exact

pytestmark = pytest.mark.gpu
SEED = 0x4D564631


def same(a, b):
    return (a.indices == b.indices).all() and (a.raw == b.raw).all() and \
        (a.scores.view(np.uint32) == b.scores.view(np.uint32)).all()


@pytest.mark.parametrize("dtype,metric", [(0, G.L2), (1, G.COSINE), (2, G.INNER_PRODUCT), (3, G.L2)])
def test_flag_wait_and_stream_wait_return_the_same_results(oracle, monkeypatch, dtype, metric):
    n, dim = 20_000, 96
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, 24, dim, dtype)
    with G.GpuCorpus.from_array(rows) as c:
        for nq, k in ((1, 10), (1, 1000), (3, 7), (4, 100), (17, 33), (24, 1)):
            monkeypatch.setenv("MVF_HOST_FLAG_WAIT", "1")
            c.reload_tuning()
            a = [c.search(q[:nq], k, metric) for _ in range(3)]
            monkeypatch.setenv("MVF_HOST_FLAG_WAIT", "0")
            c.reload_tuning()
            b = c.search(q[:nq], k, metric)
            assert all(same(x, b) for x in a), (nq, k)
            if dtype in (2, 3):
                assert_exact(b, *oracle.search(rows, dtype, metric, q[:nq], k))


def test_two_thousand_small_searches_in_a_row_never_read_early(oracle):
    """Every call gets another query: a host that stopped waiting one search early would return the previous answer."""
    n, dim, k = 50_000, 32, 20
    rows = oracle.synth_rows(SEED, 0, n, dim, 2)
    q = oracle.synth_queries(SEED + 1, 2000, dim, 2)
    osc, oidx, oraw = oracle.search(rows, 2, 1, q, k)
    with G.GpuCorpus.from_array(rows) as c:
        for i in range(2000):
            got = c.search(q[i], k, G.INNER_PRODUCT)
            assert (got.indices[0] == oidx[i]).all() and (got.raw[0] == oraw[i]).all(), f"search {i}"
        for i in range(0, 2000, 4):                     # four-query calls: four select blocks, one signal
            got = c.search(q[i:i + 4], k, G.INNER_PRODUCT)
            assert (got.indices == oidx[i:i + 4]).all(), f"batch at {i}"


def test_threads_sharing_a_handle_and_device_searches_in_between(oracle):
    import torch
    n, dim, k = 30_000, 64, 15
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, 64, dim, 0)
    with G.GpuCorpus.from_array(rows) as c:
        want = [c.search(q[i], k, G.COSINE) for i in range(64)]
        bad = []

        def host_worker(t):
            for r in range(150):
                i = (t * 17 + r) % 64
                if not same(c.search(q[i], k, G.COSINE), want[i]):
                    bad.append((t, r))

        def device_worker():
            dq = torch.from_numpy(q).cuda()
            ds = torch.empty((64, k), dtype=torch.float32, device="cuda")
            di = torch.empty((64, k), dtype=torch.int64, device="cuda")
            st = torch.cuda.Stream()
            for r in range(60):
                c.search_device(dq.data_ptr(), 0, dim, 64, k, G.COSINE, ds.data_ptr(), di.data_ptr(), 0, st.cuda_stream)
                st.synchronize()
                if not (di.cpu().numpy().view(np.uint64)[5] == want[5].indices[0]).all():
                    bad.append(("device", r))

        ts = [threading.Thread(target=host_worker, args=(t,)) for t in range(4)] + [threading.Thread(target=device_worker)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not bad, bad[:5]


@pytest.mark.parametrize("dtype,dim", [(0, 128), (1, 100), (2, 64), (3, 33)])
def test_search_fetch_rows_copied_by_the_final_select(oracle, monkeypatch, dtype, dim):
    """mvfgpu_search_fetch on small results: the final select copies the k payload rows behind the results (no third kernel)
    and the call waits on the flag.  Same results and rows as the three-kernel path (MVF_HOST_FLAG_WAIT=0); the rows ARE the
    corpus' rows; padding results (k beyond the rows) come with zero rows; row sizes that are no multiple of 16 / 4 bytes."""
    n = 3000
    rows = oracle.synth_rows(SEED, 0, n, dim, dtype)
    q = oracle.synth_queries(SEED + 1, 5, dim, dtype)
    with G.GpuCorpus.from_array(rows, index_base=40) as c:
        for nq, k in ((1, 10), (1, 1), (4, 7), (5, 100)):
            monkeypatch.setenv("MVF_HOST_FLAG_WAIT", "1")
            c.reload_tuning()
            res, vec = c.search_fetch(q[:nq], k, G.L2)
            monkeypatch.setenv("MVF_HOST_FLAG_WAIT", "0")
            c.reload_tuning()
            res0, vec0 = c.search_fetch(q[:nq], k, G.L2)
            assert same(res, res0) and (vec.view(np.uint8) == vec0.view(np.uint8)).all()
            li = (res.indices - np.uint64(40)).astype(np.int64)
            assert (vec.view(np.uint8) == rows[li].view(np.uint8)).all()
            assert same(res, c.search(q[:nq], k, G.L2))
    with G.GpuCorpus.from_array(rows[:6]) as c:           # k beyond the rows: padding results carry zero rows
        res, vec = c.search_fetch(q[0], 9, G.L2)
        assert (res.indices[0, 6:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and not vec[0, 6:].view(np.uint8).any()
        assert (vec[0, :6].view(np.uint8) == rows[:6][res.indices[0, :6].astype(np.int64)].view(np.uint8)).all()
