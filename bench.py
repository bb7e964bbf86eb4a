#!/usr/bin/env python3
"""bench.py — headline benchmark of the MVF brute-force similarity-search path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

`value` (the contract line's workload, BASELINE.json configs[1]): 10M x 768 Float32, cosine, ONE query, top-100 per
GPU -- the HBM-bound streaming scan.  A "step" = one search of the whole resident corpus: query already on the device,
kernels + on-device top-k, results left on the device.  Corpus upload / generation is outside the timed region (done
once; DESIGN.md §6 gives the PCIe-inclusive figure).  N > 1 is WEAK scaling: every rank holds its own 10M-row shard of
an N*10M-row corpus (row-range sharding, metrovector_amd/sharded.py); a step adds the RCCL all-gather of the per-shard
top-k and the merge; value = all rows of all ranks * nq / time (max over ranks, barriers on both sides).

Second leg at EVERY N, `cfg5_sharded` (BASELINE.json configs[4], the config north_star names for 8 GPUs): every rank
holds a 12.5M x 1024 Float16 shard (N = 8: the 100M-row corpus), 1024 batched queries, L2, top-100 -- MFMA path on each
shard, one packed RCCL all-gather, merge; per-rank scan ms, exchange ms and the RCCL ranks the process group reports
are printed with it.

N = 1 only: recall@k against an exact oracle top-k over all rows, the metric's second leg on the same corpus (1024
batched queries: default path = int8-shadow selection, f16-shadow selection, exact f32 MFMA), `host_api` (the same search through the host-buffer entry point
mvfgpu_search: query H2D + kernels + results D2H), `cpu_baseline` (the oracle's faithful single-thread restatement of the
reference loop) and `cpu_baseline_best_effort` (OpenMP over rows, all host cores, no per-row allocation).

One JSON line on rank 0.  `roofline` prices the dominant kernel of the `value` workload against 8 TB/s HBM3E from HIP
events recorded on the kernel's own stream during the timed steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4D564631  # "MVF1"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3  # same guide: dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TF = 2500.0  # same guide: ~2.5 PF dense bf16/f16
MFMA_I8_PEAK_TOPS = 5000.0  # same guide: int8 = 2x the bf16 rate per clock
DT_NAME = {0: "f32", 1: "f16", 2: "i8", 3: "u8"}
M_NAME = {0: "L2", 1: "dot", 2: "cosine"}


def mfma_roofline(tm, dtype):
    """Roofline of the batched path from the live HIP-event timing: the dominant launch (the LAST, largest phase) and
    the WHOLE search (every phase, compactions, re-scoring and repair launches included).
    tm.scan_kernel: 2 = f32 MFMA kernel on Float32 rows, 3 = f16/int8 kernel on the stored rows, 4 = f16 kernel on
    the scaled-f16 shadow of a Float32 corpus, 6 = int8 kernel on the int8 shadow of a Float32 / Float16 corpus (4 and 6:
    selection only; the kept rows are re-scored exactly from the stored rows).  The int8-shadow leg is priced against the
    INT8 MFMA peak: that is the pipe the kernel runs on."""
    ach = tm.scan_flops / (tm.scan_ms_avg * 1e-3) / 1e12
    if tm.scan_kernel == 2:
        peak, unit, kernel = MFMA_F32_PEAK_TF, "TFLOP/s", "scan_mfma_f32_kernel (last phase)"
    elif tm.scan_kernel == 6:
        peak, unit, kernel = MFMA_I8_PEAK_TOPS, "TOP/s", "scan_mfma16 kernel <int8> (last phase) on the int8 shadow of the float rows"
    elif tm.scan_kernel == 3 and dtype in (2, 3):
        peak, unit, kernel = MFMA_I8_PEAK_TOPS, "TOP/s", "scan_mfma16 kernel <int8> (last phase)"
    else:
        peak, unit = MFMA_F16_PEAK_TF, "TFLOP/s"
        kernel = "scan_mfma16 kernel <f16> (last phase)" + (" on the f16 shadow of the f32 rows" if tm.scan_kernel == 4 else "")
    out = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
           "kernel": kernel, "kernel_ms_avg": tm.scan_ms_avg, "launches_timed": tm.samples,
           "scan_launches_per_search": tm.scan_launches, "algorithmic_flops_per_launch": float(tm.scan_flops),
           "algorithmic_bytes_per_launch": float(tm.scan_bytes)}
    if tm.search_ms_avg > 0 and tm.search_flops:
        whole = tm.search_flops / (tm.search_ms_avg * 1e-3) / 1e12
        out["whole_search"] = {"device_ms_avg": tm.search_ms_avg, "achieved": whole, "unit": unit, "frac": whole / peak,
                               "algorithmic_flops": float(tm.search_flops),
                               "covers": "first to last kernel of the search on its stream (query prep, all phases, "
                                         "compactions, re-scoring, repair launches)"}
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", type=int, default=0, help="schema DataType code (0 f32, 1 f16, 2 i8, 3 u8)")
    ap.add_argument("--metric", type=int, default=2, help="schema DistanceMetric code (0 L2, 1 dot, 2 cosine)")
    ap.add_argument("--queries", type=int, default=1)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra q=1024 legs of the default N=1 run")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the cfg5_sharded leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget per leg")
    return ap.parse_args()


def host_description():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "cpus_allowed": aff}


def cpu_baseline(args, oracle):
    """Faithful single-thread restatement of the reference loop (examples/similarity_search.rs:140-176: per-row
    address math, as_f32 alloc+decode, strict serial f32 L2 + sqrt, BinaryHeap) on a bounded sample of the same corpus.
    The reference computes L2 only, so the CPU leg is L2 whatever --metric says; its cost per row is the same."""
    if args.dtype not in (0, 1):
        return None
    chunk = 250_000
    buf = np.empty((chunk, args.dim), oracle.NP_DTYPE[args.dtype])
    q = oracle.synth_queries(SEED + 1, 1, args.dim, args.dtype)[0]
    rows_done, spent = 0, 0.0
    while spent < args.cpu_seconds and rows_done < args.rows:
        n = min(chunk, args.rows - rows_done)
        rows = oracle.synth_rows(SEED, rows_done, n, args.dim, args.dtype, out=buf)
        t0 = time.perf_counter()
        oracle.find_top_k_similar_faithful(rows, n, args.dim, args.dtype, q, args.k, False)
        spent += time.perf_counter() - t0
        rows_done += n
    return {"value": rows_done / spent, "unit": "distance-ops/s", "cores": 1, "kind": "port",
            "sample": f"first {rows_done} rows of the same synthetic corpus, dim {args.dim}, L2 (the only metric the "
                      f"reference computes), k={args.k}, oracle faithful restatement single-threaded, {spent:.1f} s of CPU work"}


def cpu_baseline_best_effort(args, oracle):
    """What the host can do when it tries: OpenMP over rows on every core the box grants, no per-row allocation, the
    requested metric -- the oracle's search (same strict-order f32 arithmetic per row as the reference).  Reported so
    that the GPU / CPU ratio is not inflated by the reference's single thread."""
    threads = oracle._cpu_budget()
    chunk = 500_000
    buf = np.empty((chunk, args.dim), oracle.NP_DTYPE[args.dtype])
    q = oracle.synth_queries(SEED + 1, 1, args.dim, args.dtype)
    rows_done, spent = 0, 0.0
    while spent < args.cpu_seconds and rows_done < args.rows:
        n = min(chunk, args.rows - rows_done)
        rows = oracle.synth_rows(SEED, rows_done, n, args.dim, args.dtype, out=buf)
        t0 = time.perf_counter()
        oracle.search(rows, args.dtype, args.metric, q, args.k, index_base=rows_done)
        spent += time.perf_counter() - t0
        rows_done += n
    out = {"value": rows_done / spent, "unit": "distance-ops/s", "cores": threads, "kind": "port",
           "sample": f"first {rows_done} rows of the same synthetic corpus, dim {args.dim}, {M_NAME[args.metric]}, k={args.k}, "
                     f"oracle search: OpenMP over rows ({threads} threads), strict-order f32 per row, {spent:.1f} s wall"}
    out.update(host_description())
    return out


def oracle_topk_full(args, oracle, q):
    """Exact oracle top-k over the whole N=1 corpus, streamed in chunks (OpenMP)."""
    chunk = 250_000
    buf = np.empty((chunk, args.dim), oracle.NP_DTYPE[args.dtype])
    S, I = [], []
    for r0 in range(0, args.rows, chunk):
        n = min(chunk, args.rows - r0)
        rows = oracle.synth_rows(SEED, r0, n, args.dim, args.dtype, out=buf)
        sc, idx, _ = oracle.search(rows, args.dtype, args.metric, q, args.k, index_base=r0)
        S.append(sc)
        I.append(idx)
    sc, idx, _ = oracle.merge_topk(np.stack(S), np.stack(I), None, args.metric, 0 if args.dtype in (0, 1) else args.dtype)
    return sc, idx


def timed_steps(step, warmup, steps, world, dist, torch):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def max_over_ranks(x, world, dist, torch, device):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cfg5_sharded_leg(args, rank, local_rank, world, backend, dist, torch, G, ShardedSearcher, _lib):
    """BASELINE.json configs[4] per rank: 12.5M x 1024 f16 L2, 1024 batched queries, top-100; N = 8 is the 100M-row
    corpus north_star names.  Returns the leg's dict on rank 0 (None elsewhere)."""
    rows, dim, dtype, metric, nq, k = 12_500_000, 1024, 1, 0, 1024, args.k
    steps, warmup = max(3, args.steps // 5), 2
    dev = f"cuda:{local_rank}"
    corpus = G.GpuCorpus.synthetic(rows, dim, dtype, SEED, row0=rank * rows, device=local_rank)
    searcher = ShardedSearcher(corpus)
    dq = torch.empty((nq, dim), dtype=torch.float32, device=dev)
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dtype, SEED + 1, local_rank, None))
    torch.cuda.synchronize()

    def step():
        return searcher.search(dq, k, metric)

    for _ in range(warmup):
        step()
    corpus.set_profiling(True)
    searcher.timing = True
    elapsed, out = timed_steps(step, 0, steps, world, dist, torch)
    tm = corpus.last_timing()
    corpus.set_profiling(False)
    scan_ms, exch_ms = searcher.take_timings()
    searcher.timing = False
    elapsed = max_over_ranks(elapsed, world, dist, torch, dev if backend == "nccl" else "cpu")
    per_rank = [(scan_ms, exch_ms, tm.scan_ms_avg, tm.search_ms_avg)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    leg = None
    if rank == 0:
        idx = out[1].cpu().numpy().view(np.uint64)
        leg = {"workload": f"{world} x (12.5M x 1024 f16) rows = {world * rows / 1e6:g}M x 1024 f16 L2, {nq} batched queries, "
                           f"top-{k}, row-range sharded x{world} (BASELINE.json configs[4]{'' if world == 8 else ': its per-GPU shard at every N'})",
               "value": float(nq) * rows * world * steps / elapsed, "unit": "distance-ops/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "dtype": "f16",
               "rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else 0),
               "process_group_backend": (dist.get_backend() if world > 1 else None),
               "per_rank": [{"rank": r, "local_search_ms": a, "exchange_merge_ms": b, "last_phase_scan_ms": c_, "search_device_ms": d}
                            for r, (a, b, c_, d) in enumerate(per_rank)],
               "result_check": {"indices_in_range": bool(idx.max() < world * rows), "unique_per_query": bool(
                   all(len(set(r.tolist())) == k for r in idx[:8]))}}
        if tm.samples and tm.scan_ms_avg > 0 and tm.scan_kernel >= 2:
            leg["roofline"] = mfma_roofline(tm, dtype)
    corpus.close()
    return leg


def vendor_gemm_reference(torch, dev):
    """torch.matmul / torch._int_mm (hipBLASLt) on an 8192^3 GEMM: the rate a tuned library sustains on this box under
    its power limit -- the MFMA legs' `roofline.peak` is the nominal dense peak, this is the practical one."""
    out = {"shape": "8192 x 8192 x 8192", "what": "torch.matmul (f16) / torch._int_mm (int8), 3 warm-up + 10 timed calls"}
    try:
        n = 8192
        a = torch.randn(n, n, device=dev, dtype=torch.float16)
        b = torch.randn(n, n, device=dev, dtype=torch.float16)
        a8 = torch.randint(-127, 127, (n, n), device=dev, dtype=torch.int8)
        b8 = torch.randint(-127, 127, (n, n), device=dev, dtype=torch.int8)
        for name, fn in (("f16_tflops", lambda: torch.matmul(a, b.t())), ("int8_tops", lambda: torch._int_mm(a8, b8.t()))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            out[name] = 2.0 * n ** 3 * 10 / (time.perf_counter() - t0) / 1e12
    except Exception as e:  # a torch build without these ops: the reference is context, not a measurement of ours
        out["error"] = str(e)[:200]
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from metrovector_amd import _lib, gpu as G
    from metrovector_amd.sharded import ShardedSearcher

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: metrovector_amd has no CPU fallback")
    backend = os.environ.get("MVF_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of N>1 on fewer GPUs than ranks
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- resident corpus shard (untimed) -----------------------------------------
    row0 = rank * args.rows
    t0 = time.perf_counter()
    corpus = G.GpuCorpus.synthetic(args.rows, args.dim, args.dtype, SEED, row0=row0, device=local_rank)
    gen_s = time.perf_counter() - t0
    searcher = ShardedSearcher(corpus)
    qcode = G.query_dtype_code(args.dtype)
    qdt = {0: torch.float32, 2: torch.int8, 3: torch.uint8}[qcode]
    dq = torch.empty((args.queries, args.dim), dtype=qdt, device=dev)
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), args.queries, args.dim, args.dtype, SEED + 1,
                                                          local_rank, None))
    torch.cuda.synchronize()

    def step():
        return searcher.search(dq, args.k, args.metric)

    for _ in range(args.warmup):
        step()
    corpus.set_profiling(True)  # records HIP events around the scan kernel; never waits inside a step
    searcher.timing = True
    elapsed, out = timed_steps(step, 0, args.steps, world, dist, torch)
    tm = corpus.last_timing()
    corpus.set_profiling(False)
    local_ms, exch_ms = searcher.take_timings()
    searcher.timing = False
    elapsed = max_over_ranks(elapsed, world, dist, torch, dev if backend == "nccl" else "cpu")

    result = None
    if rank == 0:
        total_rows = args.rows * world
        ops = float(args.queries) * total_rows * args.steps
        es = {0: 4, 1: 2, 2: 1, 3: 1}[args.dtype]
        dtname, mname = DT_NAME[args.dtype], M_NAME[args.metric]
        result = {
            "metric": "distance-ops/sec",
            "value": ops / elapsed,
            "unit": "distance-ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtname,
            "data": "synthetic",
            "config": {"workload": f"{args.rows // 1_000_000}M x {args.dim} {dtname} {mname}, {args.queries} query, "
                                   f"top-{args.k}, per GPU (BASELINE.json configs[1] at --gpus 1)",
                       "second_workload": None if args.no_cfg5 else
                       "cfg5_sharded: 12.5M x 1024 f16 L2 per GPU, 1024 batched queries, top-100 (BASELINE.json configs[4])",
                       "rows_per_gpu": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k,
                       "metric": mname, "sharding": f"row-range x{world}" if world > 1 else "none"},
            "rank0_local_search_ms": local_ms, "rank0_exchange_merge_ms": exch_ms,
            "rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else 0),
        }
        # ---- roofline of the dominant kernel (rank 0's shard) ---------------------------
        alg_bytes = float(args.rows) * args.dim * es  # N*d*es per launch
        if tm.samples and tm.scan_ms_avg > 0 and tm.scan_kernel >= 2:
            result["roofline"] = mfma_roofline(tm, args.dtype)
        elif tm.samples and tm.scan_ms_avg > 0:
            ach = alg_bytes / (tm.scan_ms_avg * 1e-3) / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                  "kernel": "scan_stream_kernel", "kernel_ms_avg": tm.scan_ms_avg,
                                  "select_ms_avg": tm.select_ms_avg, "search_device_ms_avg": tm.search_ms_avg,
                                  "launches_timed": tm.samples, "algorithmic_bytes_per_launch": alg_bytes}
        # HBM bytes per launch from the PMC counters: they need their own rocprofv3 --pmc passes (FETCH_SIZE x2 for
        # wide streaming reads on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM), so the figure is the committed
        # summary of those passes for this exact workload, not a live measurement.
        if "roofline" in result:
            for name in ("r02_bench_n1_hbm_traffic.json", "r01_bench_n1_hbm_traffic.json", "r01_bench_n1_q1024_hbm_traffic.json"):
                tp = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(tp) or result["roofline"].get("traffic"):
                    continue
                prof = json.load(open(tp))
                w = prof.get("workload", {})
                if (w.get("rows"), w.get("dim"), w.get("dtype"), w.get("metric"), w.get("queries"), w.get("k")) == (
                        args.rows, args.dim, args.dtype, args.metric, args.queries, args.k):
                    result["roofline"]["traffic"] = prof["roofline_traffic_bytes_per_launch"]
                    result["roofline"]["traffic_source"] = "profiles/" + name
        result["corpus_generation_s"] = gen_s

    if rank == 0 and world == 1:
        from oracle import mvf_oracle as oracle
        oracle.build()
        q = dq.cpu().numpy()
        # ---- the same search through the HOST-buffer entry point: query H2D + kernels + results D2H (SURVEY.md §8d's
        # timed region); blocking per call
        hq = q.copy()
        for _ in range(args.warmup):
            corpus.search(hq, args.k, args.metric)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            corpus.search(hq, args.k, args.metric)
        eh = time.perf_counter() - t0
        result["host_api"] = {"entry_point": "mvfgpu_search (host buffers: query H2D + search + results D2H, blocking)",
                              "ms_per_step": eh / args.steps * 1e3, "value": float(args.queries) * args.rows * args.steps / eh,
                              "unit": "distance-ops/s", "steps": args.steps}
        result["host_api_ms_per_step"] = eh / args.steps * 1e3
        if not args.no_recall:
            t1 = time.perf_counter()
            sel = sorted(set([0, args.queries // 3, 2 * args.queries // 3, args.queries - 1]))  # <= 4 sampled queries
            osc, oidx = oracle_topk_full(args, oracle, q[sel])
            gi = out[1].cpu().numpy().view(np.uint64)[sel]
            hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, oidx))
            result["recall_at_k"] = hits / oidx.size
            result["recall_queries_checked"] = len(sel)
            result["recall_oracle_s"] = time.perf_counter() - t1
        if not args.no_cpu_baseline:
            cb = cpu_baseline(args, oracle)
            if cb:
                result["cpu_baseline"] = cb
            result["cpu_baseline_best_effort"] = cpu_baseline_best_effort(args, oracle)
        # ---- opt-in single-query paths: K1 streams a SHADOW of the rows -- the scaled-f16 one (scan path 4, half the
        # bytes) or the int8 one (scan path 6, a quarter) -- and every candidate within a proven margin is re-scored from
        # the f32 rows: same results, 1.8x / 3x sooner.  Not the default (extra device memory, a shadow to build):
        # `value` above is the scan of the stored f32 rows.
        if args.queries == 1 and args.dtype == 0 and not args.no_batched:
            for name, path, code, what in (("single_query_f16_shadow_stream", 4, 5, "f16 shadow rows x per-row scale"),
                                           ("single_query_int8_shadow_stream", 6, 7, "int8 shadow rows x per-row scale")):
                corpus.set_scan_path(path)
                for _ in range(max(args.warmup, 1)):
                    step()
                torch.cuda.synchronize()
                corpus.set_profiling(True)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    outs = step()
                torch.cuda.synchronize()
                es_ = time.perf_counter() - t0
                tms = corpus.last_timing()
                corpus.set_profiling(False)
                corpus.set_scan_path(0)
                leg = {"workload": result["config"]["workload"],
                       "scan_path": f"{path} (K1 streams the {'f16' if path == 4 else 'int8'} shadow; exact re-score)",
                       "value": float(args.rows) * args.steps / es_, "unit": "distance-ops/s", "steps": args.steps,
                       "ms_per_step": es_ / args.steps * 1e3}
                if tms.samples and tms.scan_ms_avg > 0 and tms.scan_kernel == code:
                    achs = tms.scan_bytes / (tms.scan_ms_avg * 1e-3) / 1e9
                    leg["roofline"] = {"bound": "hbm", "achieved": achs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": achs / HBM_PEAK_GBS, "traffic": None,
                                       "kernel": f"scan_stream_kernel ({what})",
                                       "kernel_ms_avg": tms.scan_ms_avg, "launches_timed": tms.samples,
                                       "algorithmic_bytes_per_launch": float(tms.scan_bytes)}
                if not args.no_recall:
                    gi = outs[1].cpu().numpy().view(np.uint64)[sel]
                    leg["recall_at_k"] = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, oidx)) / oidx.size
                result[name] = leg
        # ---- the metric's second leg: the same resident corpus, 1024 batched queries (MFMA path) ----------
        # Three ways, same results: the default (int8 MFMA kernel selecting on the int8 shadow of the rows, every row inside
        # a proven bound of the k-th best re-scored exactly from the f32 rows), the f16 MFMA kernel on the scaled-f16
        # shadow (round 1's default), and the exact f32 MFMA kernel on the rows themselves.
        if args.queries == 1 and args.dtype == 0 and not args.no_batched:
            nqb, bsteps = 1024, 5
            dqb = torch.empty((nqb, args.dim), dtype=qdt, device=dev)
            _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dqb.data_ptr(), nqb, args.dim, args.dtype, SEED + 1,
                                                                  local_rank, None))
            sel = [0, nqb // 3, 2 * nqb // 3, nqb - 1]
            oidx = None
            if not args.no_recall:
                osc, oidx = oracle_topk_full(args, oracle, dqb.cpu().numpy()[sel])
            for name, path in (("batched_q1024", 0), ("batched_q1024_f16_shadow", 3), ("batched_q1024_f32_mfma", 2)):
                corpus.set_scan_path(path)
                searcher.search(dqb, args.k, args.metric)  # warm-up (builds the row norms / the shadow once)
                torch.cuda.synchronize()
                corpus.set_profiling(True)
                t0 = time.perf_counter()
                for _ in range(bsteps):
                    outb = searcher.search(dqb, args.k, args.metric)
                torch.cuda.synchronize()
                eb = time.perf_counter() - t0
                tmb = corpus.last_timing()
                corpus.set_profiling(False)
                leg = {"workload": f"{args.rows // 1_000_000}M x {args.dim} {dtname} {mname}, {nqb} batched queries, top-{args.k}",
                       "scan_path": {0: "automatic (int8-shadow selection + exact re-scoring)",
                                     3: "3 (f16-shadow selection + exact re-scoring; round 1's default)",
                                     2: "2 (exact f32 MFMA on the stored rows)"}[path],
                       "value": float(nqb) * args.rows * bsteps / eb, "unit": "distance-ops/s", "steps": bsteps,
                       "ms_per_step": eb / bsteps * 1e3}
                if tmb.samples and tmb.scan_ms_avg > 0 and tmb.scan_kernel >= 2:
                    leg["roofline"] = mfma_roofline(tmb, args.dtype)
                    # HBM bytes per launch from the committed PMC passes of this exact workload and kernel
                    tp = os.path.join(ROOT, "profiles", {2: "r02_bench_n1_q1024_hbm_traffic.json",
                                                         4: "r02_bench_n1_q1024_shadow_hbm_traffic.json",
                                                         6: "r02_bench_n1_q1024_i8_shadow_hbm_traffic.json"}.get(tmb.scan_kernel, "-"))
                    if os.path.exists(tp):
                        leg["roofline"]["traffic"] = json.load(open(tp))["roofline_traffic_bytes_per_launch"]
                        leg["roofline"]["traffic_source"] = "profiles/" + os.path.basename(tp)
                if oidx is not None:
                    gi = outb[1].cpu().numpy().view(np.uint64)[sel]
                    leg["recall_at_k"] = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, oidx)) / oidx.size
                    leg["recall_queries_checked"] = len(sel)
                result[name] = leg
            corpus.set_scan_path(0)
            # ---- small batches on the same corpus: one tile of queries, HBM-bound (the streaming MFMA kernel on the int8
            # shadow; every candidate inside the proven bound re-scored exactly) -- wall time per search, default path
            small = {}
            for nqs in (4, 16, 64):
                for _ in range(3):
                    searcher.search(dqb[:nqs], args.k, args.metric)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    outs = searcher.search(dqb[:nqs], args.k, args.metric)
                torch.cuda.synchronize()
                es_ = (time.perf_counter() - t0) / 20
                small[f"q{nqs}"] = {"ms_per_step": es_ * 1e3, "value": float(nqs) * args.rows / es_, "unit": "distance-ops/s"}
            # the first rows of the 1024-query result (exact f32 MFMA leg) are these queries' answers
            a, b = outs[1][:4].cpu().numpy(), outb[1][:4].cpu().numpy()
            small["overlap_with_batched_q1024_f32_mfma_first_4_queries"] = float(
                sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(a, b)) / b.size)
            result["small_batches"] = small

    corpus.close()
    del searcher
    torch.cuda.empty_cache()

    # ---- second workload, every N: BASELINE.json configs[4] per rank ----------------------------------------------
    if not args.no_cfg5:
        leg = cfg5_sharded_leg(args, rank, local_rank, world, backend, dist, torch, G, ShardedSearcher, _lib)
        if rank == 0:
            result["cfg5_sharded"] = leg

    # ---- context for the MFMA fractions above: what the vendor GEMM library holds on THIS box (best case, 8192^3) --------
    if rank == 0 and world == 1 and not args.no_batched:
        result["vendor_gemm_reference"] = vendor_gemm_reference(torch, dev)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
