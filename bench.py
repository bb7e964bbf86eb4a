#!/usr/bin/env python3
"""bench.py — headline benchmark of the MVF brute-force similarity-search path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 10M x 768 Float32, cosine, ONE query,
top-100 — the HBM-bound streaming scan.  A "step" = one search of the whole
resident corpus: query already on the device, kernels + on-device top-k,
results left on the device.  Corpus upload / generation is outside the timed
region (it is done once; DESIGN.md §7 gives the PCIe-inclusive figure).

N > 1 is WEAK scaling: every rank holds its own 10M-row shard of an N*10M-row
corpus (row range sharding, metrovector_amd/sharded.py); a step adds the RCCL
all-gather of the per-shard top-k and the merge.  value = all rows of all
ranks * nq / time.

One JSON line on rank 0; `roofline` prices the streaming-scan kernel against
8 TB/s HBM3E using HIP events recorded on the kernel's own stream during the
timed steps; `cpu_baseline` times the oracle's faithful single-thread
restatement of the reference loop on a bounded sample.
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
MFMA_F16_PEAK_TF = 2500.0  # same guide: ~2.5 PF dense bf16/f16 (v_mfma_f32_32x32x16_f16)
MFMA_I8_PEAK_TOPS = 5000.0  # same guide: int8 = 2x the bf16 rate per clock (v_mfma_i32_32x32x32_i8)


def mfma_roofline(tm, dtype):
    """Roofline of the batched path's dominant launch (the LAST, largest phase) from the live HIP-event timing.
    tm.scan_kernel: 2 = f32 MFMA kernel on Float32 rows, 3 = f16/int8 kernel on the stored rows, 4 = f16 kernel on
    the scaled-f16 shadow of a Float32 corpus (selection; the kept rows are re-scored exactly)."""
    ach = tm.scan_flops / (tm.scan_ms_avg * 1e-3) / 1e12
    if tm.scan_kernel == 2:
        peak, unit, kernel = MFMA_F32_PEAK_TF, "TFLOP/s", "scan_mfma_f32_kernel (last phase)"
    elif tm.scan_kernel == 3 and dtype in (2, 3):
        peak, unit, kernel = MFMA_I8_PEAK_TOPS, "TOP/s", "scan_mfma16_kernel<int8> (last phase)"
    else:
        peak, unit = MFMA_F16_PEAK_TF, "TFLOP/s"
        kernel = "scan_mfma16_kernel<f16> (last phase)" + (" on the f16 shadow of the f32 rows" if tm.scan_kernel == 4 else "")
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
            "kernel": kernel, "kernel_ms_avg": tm.scan_ms_avg, "launches_timed": tm.samples,
            "scan_launches_per_search": tm.scan_launches, "algorithmic_flops_per_launch": float(tm.scan_flops),
            "algorithmic_bytes_per_launch": float(tm.scan_bytes)}


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
    ap.add_argument("--no-batched", action="store_true", help="skip the extra q=1024 leg of the default N=1 run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget")
    return ap.parse_args()


def cpu_baseline(args, oracle):
    """Faithful single-thread restatement of the reference loop
    (examples/similarity_search.rs:140-176: per-row address math, as_f32
    alloc+decode, strict serial f32 L2 + sqrt, BinaryHeap) on a bounded
    sample of the same corpus.  The reference computes L2 only, so the CPU
    leg is L2 whatever --metric says; its cost per row is the same."""
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
    from metrovector_amd import gpu as G
    from metrovector_amd.sharded import ShardedSearcher

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: metrovector_amd has no CPU fallback")
    backend = os.environ.get("MVF_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of N>1 on fewer GPUs than ranks
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
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
    dq = torch.empty((args.queries, args.dim), dtype=qdt, device=f"cuda:{local_rank}")
    from metrovector_amd import _lib
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), args.queries, args.dim, args.dtype, SEED + 1,
                                                          local_rank, None))
    torch.cuda.synchronize()

    def step():
        return searcher.search(dq, args.k, args.metric)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    corpus.set_profiling(True)  # records HIP events around the scan kernel; never waits inside a step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = corpus.last_timing()
    corpus.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_rows = args.rows * world
        ops = float(args.queries) * total_rows * args.steps
        es = {0: 4, 1: 2, 2: 1, 3: 1}[args.dtype]
        dtname = {0: "f32", 1: "f16", 2: "i8", 3: "u8"}[args.dtype]
        mname = {0: "L2", 1: "dot", 2: "cosine"}[args.metric]
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
                       "rows_per_gpu": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k,
                       "metric": mname, "sharding": f"row-range x{world}" if world > 1 else "none"},
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
                                  "select_ms_avg": tm.select_ms_avg, "launches_timed": tm.samples,
                                  "algorithmic_bytes_per_launch": alg_bytes}
        # HBM bytes per launch from the PMC counters: they need their own rocprofv3 --pmc passes (FETCH_SIZE x2 for
        # wide streaming reads on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM), so the figure is the committed
        # summary of those passes for this exact workload, not a live measurement.
        if "roofline" in result:
            for name in ("r01_bench_n1_hbm_traffic.json", "r01_bench_n1_q1024_hbm_traffic.json"):
                tp = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(tp):
                    continue
                prof = json.load(open(tp))
                w = prof.get("workload", {})
                if (w.get("rows"), w.get("dim"), w.get("dtype"), w.get("metric"), w.get("queries"), w.get("k")) == (
                        args.rows, args.dim, args.dtype, args.metric, args.queries, args.k):
                    result["roofline"]["traffic"] = prof["roofline_traffic_bytes_per_launch"]
                    result["roofline"]["traffic_source"] = "profiles/" + name
        result["corpus_generation_s"] = gen_s

        if world == 1:
            from oracle import mvf_oracle as oracle
            oracle.build()
            q = dq.cpu().numpy()
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
            # ---- opt-in single-query path (scan path 4): K1 streams the scaled-f16 shadow of the rows (half the bytes),
            # candidates within a proven margin are re-scored from the f32 rows -- same results, ~1.8x sooner.  Not the
            # default: `value` above is the scan of the stored f32 rows.
            if args.queries == 1 and args.dtype == 0 and not args.no_batched:
                corpus.set_scan_path(4)
                for _ in range(args.warmup):
                    step()
                torch.cuda.synchronize()
                corpus.set_profiling(True)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    outs = step()
                torch.cuda.synchronize()
                es = time.perf_counter() - t0
                tms = corpus.last_timing()
                corpus.set_profiling(False)
                corpus.set_scan_path(0)
                leg = {"workload": result["config"]["workload"], "scan_path": "4 (K1 streams the f16 shadow; exact re-score)",
                       "value": float(args.rows) * args.steps / es, "unit": "distance-ops/s", "steps": args.steps,
                       "ms_per_step": es / args.steps * 1e3}
                if tms.samples and tms.scan_ms_avg > 0 and tms.scan_kernel == 5:
                    achs = tms.scan_bytes / (tms.scan_ms_avg * 1e-3) / 1e9
                    leg["roofline"] = {"bound": "hbm", "achieved": achs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": achs / HBM_PEAK_GBS, "traffic": None,
                                       "kernel": "scan_stream_kernel (f16 shadow rows x per-row scale)",
                                       "kernel_ms_avg": tms.scan_ms_avg, "launches_timed": tms.samples,
                                       "algorithmic_bytes_per_launch": float(tms.scan_bytes)}
                if not args.no_recall:
                    gi = outs[1].cpu().numpy().view(np.uint64)[sel]
                    leg["recall_at_k"] = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, oidx)) / oidx.size
                result["single_query_f16_shadow_stream"] = leg
            # ---- the metric's second leg: the same resident corpus, 1024 batched queries (MFMA path) ----------
            # Two ways, same results: the default (f16 MFMA kernel selecting on the scaled-f16 shadow of the rows,
            # kept rows re-scored exactly from the f32 rows) and the exact f32 MFMA kernel on the rows themselves.
            if args.queries == 1 and args.dtype == 0 and not args.no_batched:
                nqb, bsteps = 1024, 5
                dqb = torch.empty((nqb, args.dim), dtype=qdt, device=f"cuda:{local_rank}")
                _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dqb.data_ptr(), nqb, args.dim, args.dtype, SEED + 1,
                                                                      local_rank, None))
                sel = [0, nqb // 3, 2 * nqb // 3, nqb - 1]
                oidx = None
                if not args.no_recall:
                    osc, oidx = oracle_topk_full(args, oracle, dqb.cpu().numpy()[sel])
                for name, path in (("batched_q1024", 0), ("batched_q1024_f32_mfma", 2)):
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
                           "scan_path": "automatic" if path == 0 else "2 (exact f32 MFMA on the stored rows)",
                           "value": float(nqb) * args.rows * bsteps / eb, "unit": "distance-ops/s", "steps": bsteps,
                           "ms_per_step": eb / bsteps * 1e3}
                    if tmb.samples and tmb.scan_ms_avg > 0 and tmb.scan_kernel >= 2:
                        leg["roofline"] = mfma_roofline(tmb, args.dtype)
                        # HBM bytes per launch from the committed PMC passes of this exact workload and kernel
                        tp = os.path.join(ROOT, "profiles", {2: "r01_bench_n1_q1024_hbm_traffic.json",
                                                             4: "r01_bench_n1_q1024_shadow_hbm_traffic.json"}.get(tmb.scan_kernel, "-"))
                        if os.path.exists(tp):
                            leg["roofline"]["traffic"] = json.load(open(tp))["roofline_traffic_bytes_per_launch"]
                            leg["roofline"]["traffic_source"] = "profiles/" + os.path.basename(tp)
                    if oidx is not None:
                        gi = outb[1].cpu().numpy().view(np.uint64)[sel]
                        leg["recall_at_k"] = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, oidx)) / oidx.size
                        leg["recall_queries_checked"] = len(sel)
                    result[name] = leg
                corpus.set_scan_path(0)
        print(json.dumps(result), flush=True)

    corpus.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
