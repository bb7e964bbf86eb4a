#!/usr/bin/env python3
"""bench.py — headline benchmark of the MVF brute-force similarity-search path.

    python bench.py --gpus N --steps K --warmup W
    N > 1 from a plain shell: bench.py starts its N ranks itself (a child `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py --gpus N ...`, launched before this process touches the GPU; rank 0's JSON line and
    the child's exit code are passed through).  Already under torch.distributed.run (RANK / WORLD_SIZE set): runs as a
    rank.  Fewer visible GPUs than ranks: the ranks share the GPUs and exchange through gloo -- a REHEARSAL of the
    protocol, flagged as such in the line, not a scaling measurement.

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

Third leg at EVERY N, `shardset` (what a Rust host binds, include/mvf_gpu.h mvfgpu_shardset_*): ONE process, N corpus
handles on N devices, the same cfg5 shard shape per device, per-shard searches + one grouped RCCL all-gather + merge
inside the library; run by rank 0 in a child process after the ranks have released their GPUs.

Fourth leg at EVERY N, `cfg5_strong`: STRONG scaling of the same config -- ONE fixed 100M x 1024 Float16 corpus (204.8 GB:
it fits one MI355X) split by row range over the N ranks, value = 1024 * 100M / t at every N, so value(N) / value(1) is
north_star's ">= 6x at 8 GPUs vs 1".  A 204.8 GB shard has no room for the int8 shadow of ALL its rows -- since round 5 it
shadows the ~95M-row prefix that fits and searches the corpus as two row ranges (int8 selection / f16 kernels) whose lists
are merged -- so the leg also runs with the f16 selection forced on every rank (`f16_selection_at_every_n`): that pair of
numbers compares like with like.  At N = 1 the same corpus is also searched as eight handles in one shard set.

N = 1 also: `mvf_file_e2e` (a real 4.6 GB two-space .mvf: write, open, upload off the mmap cold and warm, checksum,
search), `cfg4_int8` (BASELINE.json configs[3]: 50M x 768 Int8 dot, 256 batched queries, top-100; both the HBM and the
int8-MFMA fraction of the whole search) and `cfg1` (configs[0]: 10k x 128 f32 L2 top-10, the faithful CPU restatement
timed in full beside the GPU's time for the same search).

N = 1 only: recall@k against an exact oracle top-k over all rows, the metric's second leg on the same corpus (1024
batched queries: default path = int8-shadow selection, f16-shadow selection, exact f32 MFMA), `host_api` (the same search through the host-buffer entry point
mvfgpu_search: query H2D + kernels + results D2H), `cpu_baseline` (the oracle's faithful single-thread restatement of the
reference loop) and `cpu_baseline_best_effort` (every CPU the cgroup grants, no per-row allocation, 16 partial sums per
row so the fold vectorises -- not bit-exact with the reference's strict left fold, and not used as a checker).

One JSON line on rank 0.  `roofline` prices the dominant kernel of the `value` workload against 8 TB/s HBM3E from HIP
events recorded on the kernel's own stream during the timed steps.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.perf_counter()

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
    ap.add_argument("--no-shardset", action="store_true", help="skip the single-process shard-set leg")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg (cfg5_strong)")
    ap.add_argument("--strong-rows", type=int, default=100_000_000,
                    help="rows of the FIXED 1024-dim f16 corpus the strong-scaling leg splits over the ranks (BASELINE configs[4]: 100M)")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4_int8 leg (N = 1)")
    ap.add_argument("--no-cfg1", action="store_true", help="skip the cfg1 block (N = 1)")
    ap.add_argument("--no-file", action="store_true", help="skip the mvf_file_e2e leg (N = 1)")
    ap.add_argument("--file-rows", type=int, default=1_500_000, help="rows of the 768-dim f32 space of the mvf_file_e2e leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget per leg")
    ap.add_argument("--shardset-only", action="store_true",
                    help="internal: run ONLY the single-process shard-set leg over --gpus devices and print its JSON")
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
    out = {"value": rows_done / spent, "unit": "distance-ops/s", "cores": 1, "threads": 1, "kind": "port",
           "sample": f"first {rows_done} rows of the same synthetic corpus, dim {args.dim}, L2 (the only metric the "
                     f"reference computes), k={args.k}, oracle faithful restatement single-threaded, {spent:.1f} s of CPU work"}
    out.update(host_description())
    return out


def cpu_baseline_best_effort(args, oracle):
    """What the host can do when it tries (oracle/mvf_cpu_best_effort.c -- a baseline, never a checker): every CPU the
    cgroup grants, no per-row allocation, the requested metric, 16 partial sums per row so the fold vectorises (AVX-512 /
    AVX2) -- hence NOT bit-exact with the reference's strict left fold.  Reported so that the GPU / CPU ratio is not
    inflated by the reference's single thread.  Float32 rows only; other types fall back to the checker's OpenMP search
    (strict order) and say so."""
    grant = oracle.cpus_granted()
    threads = max(1, min(grant["granted"], 256))
    chunk = 500_000
    buf = np.empty((chunk, args.dim), oracle.NP_DTYPE[args.dtype])
    q = oracle.synth_queries(SEED + 1, 1, args.dim, args.dtype)
    rows_done, spent = 0, 0.0
    fast = args.dtype == 0
    while spent < args.cpu_seconds and rows_done < args.rows:
        n = min(chunk, args.rows - rows_done)
        rows = oracle.synth_rows(SEED, rows_done, n, args.dim, args.dtype, out=buf)
        t0 = time.perf_counter()
        if fast:
            oracle.search_best_effort_f32(rows, args.metric, q[0], args.k, threads, index_base=rows_done)
        else:
            oracle.search(rows, args.dtype, args.metric, q, args.k, index_base=rows_done)
        spent += time.perf_counter() - t0
        rows_done += n
    how = (f"OpenMP over rows on {threads} threads (every CPU the cgroup grants), 16 partial sums per row (vectorised; "
           f"not bit-exact with the reference's strict fold), per-thread bounded heaps") if fast else \
          f"the checker's OpenMP search ({oracle._cpu_budget()} threads), strict-order f32 per row"
    out = {"value": rows_done / spent, "unit": "distance-ops/s", "cores": threads if fast else oracle._cpu_budget(),
           "threads": threads if fast else oracle._cpu_budget(), "kind": "port", "bit_exact_with_reference": not fast,
           "cgroup_cpu_quota": grant["cgroup_cpu_quota"],
           "sample": f"first {rows_done} rows of the same synthetic corpus, dim {args.dim}, {M_NAME[args.metric]}, k={args.k}, "
                     f"{how}, {spent:.1f} s wall"}
    out.update(host_description())
    return out


def oracle_topk_rows(oracle, row0, rows, dim, dtype, metric, q, k, chunk=250_000):
    """Exact oracle top-k of every query in q over ALL rows [row0, row0 + rows) of the synthetic corpus, streamed in
    chunks (rows regenerated on the CPU, strict-order scoring, OpenMP over rows; merge(top-k per chunk) == top-k(all)
    because selection is by a total order).  -> (scores [nq,k], global indices [nq,k], raw [nq,k])"""
    buf = np.empty((min(chunk, rows), dim), oracle.NP_DTYPE[dtype])
    S, I, R = [], [], []
    for r0 in range(0, rows, chunk):
        n = min(chunk, rows - r0)
        block = oracle.synth_rows(SEED, row0 + r0, n, dim, dtype, out=buf)
        sc, idx, raw = oracle.search(block, dtype, metric, q, k, index_base=row0 + r0)
        S.append(sc), I.append(idx), R.append(raw)
    return oracle.merge_topk(np.stack(S), np.stack(I), np.stack(R), metric, dtype)


def recall_of(got_idx, oracle_idx, got_scores=None, oracle_scores=None, metric=None):
    """recall@k = |GPU top-k ∩ oracle top-k| / k, averaged over the queries; SURVEY.md §8d: "ties at the k-th score
    counted as hits" -- with the scores given, a returned row that is not in the oracle's list still counts when its score
    is within the tolerance of the oracle's k-th score (1e-5 relative for L2, 1e-5 absolute for cosine, exact equality
    for dot products / integer spaces): two rows that tie there may be ranked either way by two summation orders."""
    hits = 0
    for qi, (a, b) in enumerate(zip(got_idx, oracle_idx)):
        common = set(a.tolist()) & set(b.tolist())
        hits += len(common)
        if got_scores is not None and len(common) < len(b):
            kth = float(oracle_scores[qi][-1])
            tol = 1e-5 * abs(kth) if metric == 0 else 1e-5 if metric == 2 else 0.0
            hits += sum(1 for r, sc in zip(a.tolist(), got_scores[qi].tolist()) if r not in common and abs(sc - kth) <= tol)
    return hits / oracle_idx.size


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


def cfg5_sharded_leg(args, rank, local_rank, world, backend, dist, torch, G, ShardedSearcher, _lib, total_rows=None):
    """BASELINE.json configs[4]: 1024 f16 dimensions, L2, 1024 batched queries, top-100, row-range sharded over the ranks.
    WEAK (total_rows None): every rank holds 12.5M rows, N = 8 is the 100M-row corpus north_star names.
    STRONG (total_rows given): ONE fixed corpus of total_rows rows split by shard_range over the N ranks -- the
    reference's serial loop over one corpus (examples/similarity_search.rs:147) sharded by row range; value =
    nq * total_rows / t at every N, so value(N) / value(1) is the speed-up north_star's ">= 6x at 8 GPUs" asks for.
    The selection path a shard takes depends on what fits beside it in HBM (the int8 shadow is +50 % of Float16 rows: a
    204.8 GB shard holds the shadow of a ~95M-row prefix and runs its rest on the stored f16 rows), so the strong leg ALSO
    runs with the f16 selection forced (scan path 3) at every N: that pair of numbers compares like with like.
    Returns the leg's dict on rank 0 (None elsewhere)."""
    from metrovector_amd.sharded import shard_range
    dim, dtype, metric, nq, k = 1024, 1, 0, 1024, args.k
    strong = total_rows is not None
    if strong:
        lo, hi = shard_range(total_rows, world, rank)
    else:
        lo, hi = rank * 12_500_000, (rank + 1) * 12_500_000
        total_rows = world * 12_500_000
    rows = hi - lo
    steps, warmup = max(3, args.steps // 5), 2
    dev = f"cuda:{local_rank}"
    corpus = G.GpuCorpus.synthetic(rows, dim, dtype, SEED, row0=lo, device=local_rank)
    searcher = ShardedSearcher(corpus)
    dq = torch.empty((nq, dim), dtype=torch.float32, device=dev)
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dtype, SEED + 1, local_rank, None))
    torch.cuda.synchronize()

    def step():
        return searcher.search(dq, k, metric)

    def timed(path):
        corpus.set_scan_path(path)
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
        mine = (scan_ms, exch_ms, tm.scan_ms_avg, tm.search_ms_avg, int(tm.scan_kernel), rows)
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        corpus.set_scan_path(0)
        return elapsed, out, tm, per_rank

    elapsed, out, tm, per_rank = timed(0)
    shadow_bits = int(corpus.info().shadows)
    forced = None
    if strong:
        e3, out3, tm3, pr3 = timed(3)
        same = bool((out3[1] == out[1]).all().item())
        forced = {"scan_path": "3 (f16 MFMA selection on the stored rows, forced on every rank)",
                  "value": float(nq) * total_rows * steps / e3, "unit": "distance-ops/s", "ms_per_step": e3 / steps * 1e3,
                  "indices_identical_to_default_path": same,
                  "per_rank": [{"rank": r, "rows": p[5], "local_search_ms": p[0], "exchange_merge_ms": p[1], "scan_kernel": p[4]}
                               for r, p in enumerate(pr3)]}
    # recall@k against the oracle's top-k over ALL rows of the corpus, four of the 1024 queries: every rank runs the
    # oracle over ITS shard on its share of the host's cores, the lists are gathered and merged on rank 0
    recall = None
    if not args.no_recall:
        from oracle import mvf_oracle as oracle
        if rank == 0:
            oracle.build()
        if world > 1:
            dist.barrier()
        threads = oracle.set_threads(max(1, min(16, oracle.cpus_granted()["granted"] // world)))
        # (the 100M-row corpus on ONE GPU -- the strong leg's N = 1 point -- is checked with two queries: ~80 s on 16 threads)
        if (rows <= 12_500_000 and (world == 1 or threads >= 8)) or (world == 1 and threads >= 8):
            t1 = time.perf_counter()
            sel = [0, nq // 3, 2 * nq // 3, nq - 1] if rows <= 12_500_000 else [nq // 3, nq - 1]
            mine = oracle_topk_rows(oracle, lo, rows, dim, dtype, metric, dq.cpu().numpy()[sel], k)
            parts = [mine]
            if world > 1:
                parts = [None] * world
                dist.all_gather_object(parts, mine)
            if rank == 0:
                osc, oidx, _ = oracle.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                                                 None, metric, dtype)
                gi = out[1].cpu().numpy().view(np.uint64)[sel]
                gs = out[0].cpu().numpy()[sel]
                same = gi == oidx
                recall = {"recall_at_k": recall_of(gi, oidx, gs, osc, metric), "recall_queries_checked": len(sel),
                          "rows_identical_to_the_oracle_list": int(same.sum()), "rows_checked": int(same.size),
                          "recall_vs": f"the oracle's exact top-k over ALL {total_rows / 1e6:g}M rows (every rank its shard, "
                                       f"{threads} threads each, merged on rank 0)",
                          "max_rel_score_diff_on_identical_ranks": float(np.max(np.abs(gs - osc)[same] / np.maximum(osc[same], 1e-30))) if same.any() else None,
                          "recall_oracle_s": time.perf_counter() - t1}
        elif rank == 0:
            recall = {"recall_at_k": None, "recall_skipped": (
                f"{rows / 1e6:g}M rows per rank on {threads} host threads: the strict-order oracle over the shard would take minutes "
                "(the 12.5M-row shards of N = 8 and of the weak leg are checked)")}
    leg = None
    if rank == 0:
        idx = out[1].cpu().numpy().view(np.uint64)
        what = (f"ONE fixed corpus of {total_rows / 1e6:g}M x 1024 f16 rows split by row range over {world} GPU(s) "
                f"({rows / 1e6:g}M rows on rank 0)") if strong else \
               f"{world} x (12.5M x 1024 f16) rows = {total_rows / 1e6:g}M x 1024 f16"
        leg = {"workload": f"{what}, L2, {nq} batched queries, top-{k}, row-range sharded x{world} "
                           f"(BASELINE.json configs[4]{'' if world == 8 or strong else ': its per-GPU shard at every N'})",
               "value": float(nq) * total_rows * steps / elapsed, "unit": "distance-ops/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": "strong" if strong else "weak", "dtype": "f16",
               "total_rows": total_rows,
               "rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else 0),
               "process_group_backend": (dist.get_backend() if world > 1 else None),
               "selection_path": ("int8 MFMA kernel on the int8 shadow of a PREFIX of the rows (all rows' shadow does not fit beside them) + the f16 MFMA "
                                  "kernel on the stored rows of the rest, the two row ranges' exact lists merged; exact re-scoring" if shadow_bits & 4 else
                                  {3: "f16 MFMA kernel on the stored rows (no room or no use for an int8 shadow)",
                                   6: "int8 MFMA kernel on the int8 shadow + exact re-scoring"}.get(int(tm.scan_kernel), int(tm.scan_kernel))),
               "rank0_shadow_bits": shadow_bits,
               "per_rank": [{"rank": r, "rows": p[5], "local_search_ms": p[0], "exchange_merge_ms": p[1], "last_phase_scan_ms": p[2],
                             "search_device_ms": p[3], "scan_kernel": p[4]} for r, p in enumerate(per_rank)],
               "result_check": {"indices_in_range": bool(idx.max() < total_rows), "unique_per_query": bool(
                   all(len(set(r.tolist())) == k for r in idx[:8]))}}
        if forced:
            leg["f16_selection_at_every_n"] = forced
        if tm.samples and tm.scan_ms_avg > 0 and tm.scan_kernel >= 2:
            leg["roofline"] = mfma_roofline(tm, dtype)
            tp = os.path.join(ROOT, "profiles", "r05_cfg5_hbm_traffic.json")  # a 12.5M-row shard through the int8 shadow: HBM bytes of a whole search
            if rows == 12_500_000 and int(tm.scan_kernel) == 6 and os.path.exists(tp) and "whole_search" in leg["roofline"]:
                prof = json.load(open(tp))
                leg["roofline"]["whole_search"]["traffic"] = prof.get("search_traffic_bytes")
                leg["roofline"]["whole_search"]["algorithmic_bytes"] = prof.get("algorithmic_bytes_per_search")
                leg["roofline"]["whole_search"]["traffic_source"] = ("profiles/r05_cfg5_hbm_traffic.json (separate --pmc passes, FETCH_SIZE x2 + WRITE_SIZE over the "
                                                                     "search's kernels: the shadow rows once, the re-scored f16 rows, the candidate records)")
        if recall:
            leg.update(recall)
    ref_idx = out[1].cpu().numpy().view(np.uint64) if (strong and world == 1 and rank == 0 and not args.no_shardset) else None
    corpus.close()
    if ref_idx is not None:
        try:
            leg["as_eight_handles_in_one_shard_set"] = strong_as_shard_set(args, torch, G, dq, total_rows, dim, dtype, metric, nq, k, local_rank, ref_idx)
        except Exception as e:  # the extra must not cost the leg
            leg["as_eight_handles_in_one_shard_set"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return leg


def strong_as_shard_set(args, torch, G, dq, total_rows, dim, dtype, metric, nq, k, device, ref_idx):
    """The strong leg's N = 1 corpus once more, as the 8-way split north_star names held by ONE GPU: eight row-range handles
    in one mvfgpu_shardset (lists by device copies).  Eight 25.6-GB handles build their int8 selection shadows one after
    the other until HBM runs out -- the shards that got one select at the int8 MFMA rate, the rest on their stored f16
    rows -- and the merged result must be the single handle's, bit for bit."""
    n_sh = 8
    rows = total_rows // n_sh
    if rows * n_sh != total_rows:
        return {"skipped": "the corpus does not split into eight equal shards"}
    hq = dq.cpu().numpy()
    shards = []
    try:
        for i in range(n_sh):
            c = G.GpuCorpus.synthetic(rows, dim, dtype, SEED, row0=i * rows, device=device)
            shards.append(c)
            c.search(hq[:8], k, metric)  # a batched search: norms, and the int8 shadow if it still fits (one shard at a time: no race for the last GB)
        with_shadow = [bool(c.info().shadows & 1) for c in shards]
        steps, warmup = 3, 1
        with G.ShardSet(shards) as ss:
            for _ in range(warmup):
                res = ss.search(hq, k, metric)
            t0 = time.perf_counter()
            for _ in range(steps):
                res = ss.search(hq, k, metric)
            elapsed = time.perf_counter() - t0
            tm = ss.last_timing()
        return {"what": "mvfgpu_shardset_search over eight 12.5M-row handles on ONE device (host queries in, merged host results out; lists by device copies)",
                "ms_per_step": elapsed / steps * 1e3, "value": float(nq) * total_rows * steps / elapsed, "unit": "distance-ops/s",
                "shards_with_int8_shadow": int(sum(with_shadow)), "shard_search_ms": [float(tm.shard_search_ms[i]) for i in range(n_sh)],
                "indices_identical_to_the_single_handle": bool((res.indices == ref_idx).all())}
    finally:
        for c in shards:
            c.close()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) from a plain shell: start the N ranks as a child torch.distributed.run BEFORE
    this process touches the GPU (a process that has initialised HIP must never exec another program), pass rank 0's
    JSON line through on stdout and return the child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # dmabuf IPC.  The pool's environment notes say its host driver supports only dmabuf IPC and that RCCL / device-tensor
    # sharing across processes fails with "hipIpcGetMemHandle: invalid argument" without this setting, and export it on
    # every box already; kept here (setdefault: a no-op there) for a launch from a bare shell.  NOT verified by this repo:
    # a one-GPU box cannot run two RCCL processes.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:  # the contract is ONE JSON line on stdout: anything else the ranks print goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def shardset_child(args):
    """ONE process, N corpus handles on N devices, mvfgpu_shardset_search on cfg5's shard shape (BASELINE.json configs[4]:
    12.5M x 1024 f16 L2 per GPU, 1024 batched queries, top-100): what a Rust host binds (include/mvf_gpu.h).  Prints one
    JSON object.  Fewer devices than shards: the shards share devices and the lists travel by device copies (rehearsal)."""
    import torch
    from metrovector_amd import _lib, gpu as G
    n_sh = args.gpus
    ndev = G.device_count()
    if ndev < 1:
        sys.exit("bench.py needs a GPU: metrovector_amd has no CPU fallback")
    rows, dim, dtype, metric, nq, k = 12_500_000, 1024, 1, 0, 1024, args.k
    steps, warmup = max(3, args.steps // 5), 2
    devs = [i % ndev for i in range(n_sh)]
    shards = [G.GpuCorpus.synthetic(rows, dim, dtype, SEED, row0=i * rows, device=d) for i, d in enumerate(devs)]
    torch.cuda.set_device(0)
    dq = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dtype, SEED + 1, 0, None))
    torch.cuda.synchronize()
    hq = dq.cpu().numpy()
    with G.ShardSet(shards) as ss:
        inf = ss.info()
        for _ in range(warmup):
            res = ss.search(hq, k, metric)
        tms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            res = ss.search(hq, k, metric)
            tms.append(ss.last_timing())
        elapsed = time.perf_counter() - t0
    mean = lambda f: float(sum(f(t) for t in tms) / len(tms))
    idx = res.indices
    leg = {"workload": f"{n_sh} x (12.5M x 1024 f16) rows = {n_sh * rows / 1e6:g}M x 1024 f16 L2, {nq} batched queries, top-{k}: "
                       f"ONE process, {n_sh} corpus handles, mvfgpu_shardset_search (host queries in, merged host results out)",
           "value": float(nq) * rows * n_sh * steps / elapsed, "unit": "distance-ops/s", "n_gpus": n_sh, "devices": devs,
           "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "dtype": "f16",
           "rccl_ranks": int(inf.rccl_ranks), "n_shards": int(inf.n_shards),
           "exchange": ("RCCL ncclAllGather (grouped, one communicator rank per device)" if inf.rccl_ranks else
                        "device-to-device copies (shards share a device or MVF_SHARDSET_NO_RCCL is set)"),
           "total_ms": mean(lambda t: t.total_ms), "enqueue_ms": mean(lambda t: t.enqueue_ms),
           "search_ms": mean(lambda t: t.search_ms), "exchange_merge_ms": mean(lambda t: t.exchange_merge_ms),
           "shard_search_ms": [mean(lambda t, i=i: t.shard_search_ms[i]) for i in range(n_sh)],
           "timing_note": "total/enqueue: host wall clock per call; search: slowest shard's query upload + local search "
                          "(HIP events on its stream); exchange_merge: first shard's stream from the end of its own local "
                          "search to the end of the merge (includes waiting for the slowest shard)",
           "result_check": {"indices_in_range": bool(idx.max() < n_sh * rows),
                            "unique_per_query": bool(all(len(set(r.tolist())) == k for r in idx[:8])),
                            "shards_represented_in_first_query": int(len(set((idx[0] // rows).tolist())))}}
    if ndev < n_sh:
        leg["rehearsal"] = f"{n_sh} shards share {ndev} GPU(s): the protocol runs, the timing says nothing about scaling"
    for c in shards:
        c.close()
    print(json.dumps(leg), flush=True)


def shardset_leg(args, world):
    """Run shardset_child in a CHILD process (this one keeps its HIP context; every rank has released its corpus):
    a failure there must not cost the line its other legs."""
    cmd = [sys.executable, os.path.abspath(__file__), "--shardset-only", "--gpus", str(world), "--steps", str(args.steps),
           "--k", str(args.k)]
    env = {k: v for k, v in os.environ.items() if k not in (
        "RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR",
        "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "OMP_NUM_THREADS")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "the shard-set child did not finish within 240 s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"shard-set child rc={r.returncode}", "stderr_tail": r.stderr[-600:]}
    leg = json.loads(lines[-1])
    leg["child_wall_s"] = time.perf_counter() - t0
    return leg


def cfg4_leg(args, torch, G, _lib, oracle, local_rank):
    """BASELINE.json configs[3]: 50M x 768 Int8 dot, 256 batched queries, top-100 on one GPU (38.4 GB resident).
    SURVEY.md §8(d): nominally HBM-bound (4.8 ms at 8 TB/s) just above the int8-MFMA floor (3.9 ms at 5 POP/s): BOTH
    fractions of the WHOLE search are reported, plus the last (largest) phase's."""
    rows, dim, dtype, metric, nq, k = 50_000_000, 768, 2, 1, 256, 100
    steps = max(5, args.steps // 4)
    dev = f"cuda:{local_rank}"
    corpus = G.GpuCorpus.synthetic(rows, dim, dtype, SEED, device=local_rank)
    dq = torch.empty((nq, dim), dtype=torch.int8, device=dev)
    _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), nq, dim, dtype, SEED + 1, local_rank, None))
    ds = torch.empty((nq, k), dtype=torch.float32, device=dev)
    di = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dr = torch.empty((nq, k), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        corpus.search_device(dq.data_ptr(), dtype, dim, nq, k, metric, ds.data_ptr(), di.data_ptr(), dr.data_ptr(), stream)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    corpus.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = corpus.last_timing()
    corpus.set_profiling(False)
    alg_bytes, alg_ops = float(rows) * dim, 2.0 * nq * rows * dim
    ms = tm.search_ms_avg if tm.search_ms_avg > 0 else el / steps * 1e3
    leg = {"workload": "50M x 768 int8 dot, 256 batched queries, top-100 (BASELINE.json configs[3])",
           "value": float(nq) * rows * steps / el, "unit": "distance-ops/s", "steps": steps, "warmup": 2,
           "ms_per_step": el / steps * 1e3, "dtype": "i8", "search_device_ms_avg": ms,
           "algorithmic_bytes": alg_bytes, "algorithmic_ops": alg_ops,
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "covers": "the WHOLE search (first to last kernel on its stream), HIP events, mean of the timed searches"},
           "roofline_mfma": {"bound": "mfma", "achieved": alg_ops / (ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                             "frac": alg_ops / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                             "covers": "the WHOLE search; SURVEY.md §8(d) asks for both fractions (AI 512 op/B < ridge ~625)"}}
    if tm.samples and tm.scan_ms_avg > 0:
        leg["last_phase"] = {"kernel": "scan_mfma16_dma_kernel<int8> (last, largest phase)", "kernel_ms_avg": tm.scan_ms_avg,
                             "rows_bytes": float(tm.scan_bytes), "ops": float(tm.scan_flops),
                             "hbm_frac": tm.scan_bytes / (tm.scan_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "mfma_frac": tm.scan_flops / (tm.scan_ms_avg * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                             "scan_launches_per_search": tm.scan_launches}
    tp = os.path.join(ROOT, "profiles", "r05_cfg4_hbm_traffic.json")
    if not os.path.exists(tp):
        tp = os.path.join(ROOT, "profiles", "r03_cfg4_hbm_traffic.json")
    if os.path.exists(tp):
        prof = json.load(open(tp))
        leg["roofline"]["traffic"] = prof.get("search_traffic_bytes")
        leg["roofline"]["traffic_source"] = "profiles/" + os.path.basename(tp) + " (separate --pmc passes, FETCH_SIZE x2 + WRITE_SIZE, summed over the search's kernels)"
    # bit-exactness of what came back: every returned row of three queries regenerated on the CPU, exact i64 dot
    if oracle is not None:
        q = dq.cpu().numpy()
        gi, gr, gs = di.cpu().numpy().view(np.uint64), dr.cpu().numpy(), ds.cpu().numpy()
        ok, checked = True, 0
        for qi in (0, nq // 2, nq - 1):
            for j in range(k):
                row = oracle.synth_rows(SEED, int(gi[qi, j]), 1, dim, dtype)[0]
                want = int(np.dot(row.astype(np.int64), q[qi].astype(np.int64)))
                ok = ok and want == int(gr[qi, j]) and float(np.float32(want)) == float(gs[qi, j])
                checked += 1
            ok = ok and bool(np.all(gr[qi, :-1] >= gr[qi, 1:]))
        leg["result_check"] = {"returned_rows_bit_exact_vs_cpu": bool(ok), "rows_checked": checked}
        # recall@k and bit-exactness against the oracle's top-k over ALL 50M rows, four of the 256 queries
        t1 = time.perf_counter()
        sel = [0, nq // 3, 2 * nq // 3, nq - 1]
        osc, oidx, oraw = oracle_topk_rows(oracle, 0, rows, dim, dtype, metric, q[sel], k)
        leg["recall_at_k"] = recall_of(gi[sel], oidx)
        leg["recall_queries_checked"] = len(sel)
        leg["recall_vs"] = "the oracle's exact top-k over ALL 50M rows (chunked)"
        leg["bit_exact_vs_oracle_topk"] = bool((gi[sel] == oidx).all() and (gr[sel] == oraw).all() and
                                               (gs[sel].view(np.uint32) == osc.view(np.uint32)).all())
        leg["recall_oracle_s"] = time.perf_counter() - t1
    corpus.close()
    return leg


def cfg1_block(args, torch, G, oracle, local_rank):
    """BASELINE.json configs[0]: 10k x 128 f32 Euclidean, single query, top-10 -- the reference's own CPU-runnable case.
    The faithful CPU restatement of find_top_k_similar timed IN FULL, and the GPU's time for the same
    search through the host-buffer entry point (query H2D + kernels + results D2H) and on the device alone."""
    n, dim, k = 10_000, 128, 10
    rows = oracle.synth_rows(SEED, 0, n, dim, 0)
    q = oracle.synth_queries(SEED + 1, 1, dim, 0)
    cpu = []
    for _ in range(25):
        t0 = time.perf_counter()
        ci, cs = oracle.find_top_k_similar_faithful(rows, n, dim, 0, q[0], k, False)
        cpu.append(time.perf_counter() - t0)
    cpu.sort()
    with G.GpuCorpus.from_array(rows, device=local_rank) as c:
        for _ in range(10):
            res = c.search(q, k, G.L2)
        host = []
        for _ in range(200):
            t0 = time.perf_counter()
            res = c.search(q, k, G.L2)
            host.append(time.perf_counter() - t0)
        host.sort()
        fetch = []
        for _ in range(210):  # the hits WITH their rows, as the reference's ScoredVector carries them
            t0 = time.perf_counter()
            res_v, vec = c.search_fetch(q, k, G.L2)
            fetch.append(time.perf_counter() - t0)
        fetch = sorted(fetch[10:])
        rows_ok = bool((vec[0] == rows[res_v.indices[0].astype(np.int64)]).all() and (res_v.indices == res.indices).all())
        dq = torch.from_numpy(q).to(f"cuda:{local_rank}")
        ds = torch.empty((1, k), dtype=torch.float32, device=dq.device)
        di = torch.empty((1, k), dtype=torch.int64, device=dq.device)
        stream = torch.cuda.current_stream().cuda_stream
        c.set_profiling(True)
        for _ in range(50):
            c.search_device(dq.data_ptr(), 0, dim, 1, k, G.L2, ds.data_ptr(), di.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        tm = c.last_timing()
        c.set_profiling(False)
    same = bool((res.indices[0] == ci).all())
    rel = float(np.max(np.abs(res.scores[0] - cs) / np.maximum(np.abs(cs), 1e-30)))
    cpu_ms, host_ms = cpu[len(cpu) // 2] * 1e3, host[len(host) // 2] * 1e3
    return {"workload": "10k x 128 f32 L2, single query, top-10 (BASELINE.json configs[0]; examples/similarity_search.rs scaled)",
            "cpu_reference_port": {"ms_per_search_median": cpu_ms, "ms_per_search_min": cpu[0] * 1e3, "runs": len(cpu),
                                   "value": n / (cpu_ms * 1e-3), "unit": "distance-ops/s", "cores": 1, "kind": "port",
                                   "what": "oracle faithful restatement of find_top_k_similar, single thread, all 10k rows per run"},
            "gpu_host_api": {"ms_per_search_median": host_ms, "ms_per_search_min": host[0] * 1e3, "runs": len(host),
                             "value": n / (host_ms * 1e-3), "unit": "distance-ops/s",
                             "what": "mvfgpu_search: query in, scan + top-k, results out, blocking (latency-bound at this size; small transfers go through pinned host memory in place, profiles/r04_host_api_latency.txt)"},
            "gpu_host_api_with_vectors": {"ms_per_search_median": fetch[len(fetch) // 2] * 1e3, "ms_per_search_min": fetch[0] * 1e3, "runs": len(fetch),
                                          "value": n / fetch[len(fetch) // 2], "unit": "distance-ops/s", "rows_match_the_corpus": rows_ok,
                                          "what": "mvfgpu_search_fetch: the same search returning the k rows as well (ScoredVector.vector, "
                                                  "examples/similarity_search.rs:18) -- what the CPU figure beside it includes"},
            "gpu_device_ms": {"search_ms_avg": tm.search_ms_avg, "scan_kernel_ms_avg": tm.scan_ms_avg, "select_ms_avg": tm.select_ms_avg},
            "gpu_matches_cpu": {"indices_identical": same, "max_rel_score_diff": rel, "tolerance": 1e-5}}


def mvf_file_leg(args, G, oracle, local_rank):
    """The drop-in on a REAL multi-GB file: a two-space .mvf whose first block
    is > 4 GiB is written by the C++ builder, then MvfReader::open (O(footer), src/reader.rs:45-79) ->
    map_vector_range(0, total) (src/vectors/vector_space.rs:155-188) -> upload straight off the mmap (page cache cold,
    then warm) -> find_top_k_similar{,_batch}; CRC32 validation alone and beside the upload.  Rows come from the
    library's device generator (read back), so the answers must equal a synthetic corpus' bit for bit."""
    import shutil
    from metrovector_amd.builder import MvfBuilder
    from metrovector_amd.reader import MvfReader
    from metrovector_amd.search import upload_space
    n, dim, k, n8, dim8 = args.file_rows, 768, 100, 20_000, 64
    nbytes = n * dim * 4
    scratch = None
    for d in (os.environ.get("MVF_BENCH_FILE_DIR"), "/tmp", "/dev/shm"):
        if d and os.path.isdir(d) and shutil.disk_usage(d).free >= nbytes + (2 << 30):
            scratch = d
            break
    if scratch is None:
        return {"skipped": f"no scratch directory with {(nbytes + (2 << 30)) / 2**30:.1f} GiB free"}
    path = os.path.join(scratch, f"mvf_bench_{os.getpid()}.mvf")
    leg = {"workload": f"{n / 1e6:g}M x {dim} f32 cosine space ({nbytes / 1e9:.2f} GB block) + a {n8} x {dim8} int8 space behind it "
                       f"(block offset > 2^32) in one .mvf; open -> map_vector_range -> upload -> search",
           "scratch_dir": scratch, "scratch_fs": "tmpfs (memory)" if scratch.startswith("/dev/shm") else "disk"}
    ref = G.GpuCorpus.synthetic(n, dim, 0, SEED, device=local_rank)
    try:
        t0 = time.perf_counter()
        b = MvfBuilder()
        b.add_vector_space("big", dim, 0, 2, 0)
        b.add_vector_space("small_i8", dim8, 0, 1, 2)
        b.reserve_vectors("big", n)
        for r0 in range(0, n, 250_000):
            b.add_vectors_raw("big", ref.read_rows(r0, min(250_000, n - r0)))
        with G.GpuCorpus.synthetic(n8, dim8, 2, SEED + 7, device=local_rank) as c8:
            rows8 = c8.read_rows(0, n8)
        b.add_vectors_raw("small_i8", rows8)
        t1 = time.perf_counter()
        b.build().save(path)
        t2 = time.perf_counter()
        del b
        fd = os.open(path, os.O_RDONLY)
        os.fsync(fd)
        evicted = True
        try:
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)  # clean pages leave the page cache: the next read is cold
        except OSError:
            evicted = False
        os.close(fd)
        size = os.path.getsize(path)
        leg["file_bytes"] = size
        leg["write"] = {"build_in_memory_s": t1 - t0, "crc_and_save_s": t2 - t1, "save_gb_per_s": size / (t2 - t1) / 1e9}

        def timed_open():
            t = time.perf_counter()
            r = MvfReader.open(path)
            return r, (time.perf_counter() - t) * 1e3

        r, open_cold_ms = timed_open()
        r.close()
        opens = []
        for _ in range(20):
            r, ms = timed_open()
            opens.append(ms)
            r.close()
        opens.sort()
        leg["open_ms"] = {"first": open_cold_ms, "median_of_20": opens[10],
                          "what": "mvf_reader_open: mmap + magic / footer-length checks + footer parse; independent of the file size"}
        r = MvfReader.open(path)
        big, small = r.vector_space("big"), r.vector_space("small_i8")
        ups = {}
        for name, kw in (("cold_page_cache" if evicted and not scratch.startswith("/dev/shm") else "first", {}),
                         ("warm_page_cache", {}), ("warm_with_checksum_thread", {"verify_checksum": True}),
                         ("warm_prepare_batched", {"prepare_batched": True})):
            t = time.perf_counter()
            c = upload_space(big, device=local_rank, **kw)
            dt = time.perf_counter() - t
            ups[name] = {"s": dt, "gb_per_s": nbytes / dt / 1e9}
            if name != "warm_prepare_batched":
                c.close()
        leg["upload_from_mmap"] = ups
        t = time.perf_counter()
        r.validate_with_checksum()
        dt = time.perf_counter() - t
        leg["checksum"] = {"s": dt, "gb_per_s": size / dt / 1e9, "what": "mvf_reader_validate_with_checksum over both blocks (CRC-32, threaded slicing-by-8), page cache warm"}
        anon = np.empty((n, dim), np.float32)
        for r0 in range(0, n, 250_000):
            anon[r0:r0 + 250_000] = ref.read_rows(r0, min(250_000, n - r0))
        t = time.perf_counter()
        with G.GpuCorpus.from_array(anon, device=local_rank):
            dt = time.perf_counter() - t
        del anon
        leg["upload_from_anonymous_memory"] = {"s": dt, "gb_per_s": nbytes / dt / 1e9}
        # searches on the uploaded space: one query and 64, host-buffer API; answers against the synthetic corpus' own
        from metrovector_amd import _lib
        import torch
        dq = torch.empty((64, dim), dtype=torch.float32, device=f"cuda:{local_rank}")
        _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq.data_ptr(), 64, dim, 0, SEED + 1, local_rank, None))
        hq = dq.cpu().numpy()
        res = {}
        for nm, q in (("q1", hq[:1]), ("q64", hq)):
            for _ in range(3):
                got = c.search(q, k, G.COSINE)
            t = time.perf_counter()
            for _ in range(10):
                got = c.search(q, k, G.COSINE)
            ms = (time.perf_counter() - t) / 10 * 1e3
            want = ref.search(q, k, G.COSINE)
            res[nm] = {"ms_per_search": ms, "identical_to_synthetic_corpus": bool(
                (got.indices == want.indices).all() and (got.scores.view(np.uint32) == want.scores.view(np.uint32)).all())}
            if oracle is not None and nm == "q64":
                sel = [0, 21, 42, 63]
                _, oidx, _ = oracle_topk_rows(oracle, 0, n, dim, 0, 2, q[sel], k)
                res[nm]["recall_at_k"] = recall_of(got.indices[sel], oidx)
                res[nm]["recall_queries_checked"] = len(sel)
        c.close()
        with upload_space(small, device=local_rank, verify_checksum=True) as c8:
            got8 = c8.search(rows8[:4], 10, G.INNER_PRODUCT)  # rows as queries: each finds itself or a larger-norm row
        res["small_i8_space_behind_4gib"] = {"rows": n8, "block_offset": int(r.blocks()[1].offset),
                                             "searched": bool((got8.raw[:, 0] >= (rows8[:4].astype(np.int64) ** 2).sum(1)).all())}
        leg["search"] = res
        r.close()
    finally:
        ref.close()
        if os.path.exists(path):
            os.remove(path)
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
    if args.shardset_only:
        shardset_child(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))  # nothing above has touched the GPU (no torch, no libmvf_gpu yet)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist
    from metrovector_amd import _lib, gpu as G
    from metrovector_amd.sharded import ShardedSearcher

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: metrovector_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    # one rank per GPU over RCCL; with fewer GPUs than ranks the ranks share them and exchange through gloo (RCCL refuses
    # duplicate devices) -- a rehearsal of the protocol, flagged in the line.  MVF_BENCH_BACKEND forces a backend.
    backend = os.environ.get("MVF_BENCH_BACKEND") or ("nccl" if ndev >= world else "gloo")
    rehearsal = None
    if backend != "nccl":
        local_rank = local_rank % ndev
        if world > 1:
            rehearsal = (f"{world} ranks share {ndev} GPU(s) and exchange through {backend} (host-staged): the N>1 code path "
                         "runs end to end, the numbers say nothing about scaling")
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
            "value_timed_region": "query already on the device -> kernels + on-device top-k (+ all-gather and merge at N > 1) -> "
                                  "results left on the device; the query's H2D copy and the results' D2H copy of SURVEY.md "
                                  "§8(d)'s region are NOT in `value` -- `host_api` is the same search including both (its ms_per_step is in the "
                                  "line: within 0.5 % either way since the host call moves small queries / results through pinned memory in place)",
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
            "process_group_backend": (dist.get_backend() if world > 1 else None),
        }
        if rehearsal:
            result["rehearsal"] = rehearsal
        if world > 1:
            result["cpu_baseline_note"] = "cpu_baseline, recall of the headline and the N = 1 legs are reported by the --gpus 1 run only"
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
            for name in ("r05_bench_n1_hbm_traffic.json", "r04_bench_n1_hbm_traffic.json", "r03_bench_n1_hbm_traffic.json"):
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
            if args.queries == 1:
                # 16 single-query searches on the headline's path (query 0 is the timed one), each against the oracle's
                # top-k over ALL rows
                nchk = 16
                dq16 = torch.empty((nchk, args.dim), dtype=qdt, device=dev)
                _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dq16.data_ptr(), nchk, args.dim, args.dtype, SEED + 1,
                                                                      local_rank, None))
                gi, gsc = [], []
                for j in range(nchk):
                    o = searcher.search(dq16[j:j + 1], args.k, args.metric)
                    gsc.append(o[0].cpu().numpy()[0].copy())
                    gi.append(o[1].cpu().numpy().view(np.uint64)[0].copy())
                gi, gsc = np.stack(gi), np.stack(gsc)
                osc, oidx, _ = oracle_topk_rows(oracle, 0, args.rows, args.dim, args.dtype, args.metric, dq16.cpu().numpy(), args.k)
                sel = [0]
            else:
                sel = sorted(set([0, args.queries // 3, 2 * args.queries // 3, args.queries - 1]))  # <= 4 sampled queries
                osc, oidx, _ = oracle_topk_rows(oracle, 0, args.rows, args.dim, args.dtype, args.metric, q[sel], args.k)
                gi, gsc = out[1].cpu().numpy().view(np.uint64)[sel], out[0].cpu().numpy()[sel]
            result["recall_at_k"] = recall_of(gi, oidx, gsc, osc, args.metric)
            result["rows_identical_to_the_oracle_list"] = int((gi == oidx).sum())
            result["rows_checked"] = int(oidx.size)
            result["recall_queries_checked"] = int(oidx.shape[0])
            result["recall_vs"] = "the oracle's exact top-k over ALL rows of the corpus (chunked, strict-order f32)"
            result["recall_oracle_s"] = time.perf_counter() - t1
            oidx, osc = oidx[:len(sel)], osc[:len(sel)]  # the legs below search the timed query again
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
                    leg["recall_at_k"] = recall_of(gi, oidx, outs[0].cpu().numpy()[sel], osc, args.metric)
                result[name] = leg
        # ---- k beyond one pass (the reference takes any k: usize, examples/similarity_search.rs:143): the same query for its
        # 16384 best rows, by 16 passes of the streaming kernel and by the whole-shard sort (one dump pass + a device-wide
        # radix sort of the 10M order keys), and k = 1M by the sort (the only formulation beyond 16384)
        if args.queries == 1 and not args.no_batched:
            leg = {"workload": result["config"]["workload"].replace(f"top-{args.k}", "top-16384 / top-1000000")}
            ref = None
            for name, env, kk in (("k16384_passes", "1", 16384), ("k16384_sort", "2", 16384), ("k1000000_sort", None, 1_000_000)):
                if kk > args.rows:
                    continue
                if env is None:
                    os.environ.pop("MVF_LARGE_K", None)
                else:
                    os.environ["MVF_LARGE_K"] = env
                corpus.reload_tuning()
                ds = torch.empty((1, kk), dtype=torch.float32, device=dev)
                di = torch.empty((1, kk), dtype=torch.int64, device=dev)

                def big():
                    corpus.search_device(dq.data_ptr(), qcode, args.dim, 1, kk, args.metric, ds.data_ptr(), di.data_ptr())

                big()
                torch.cuda.synchronize()
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    big()
                torch.cuda.synchronize()
                leg[name] = {"ms_per_search": (time.perf_counter() - t0) / reps * 1e3}
                if kk == 16384:
                    if ref is None:
                        ref = (di.clone(), ds.clone())
                    else:
                        leg[name]["identical_to_passes"] = bool((ref[0] == di).all().item() and (ref[1] == ds).all().item())
                else:
                    leg[name]["first_16384_identical"] = bool((ref[0][0] == di[0, :16384]).all().item()) if ref is not None else None
                    leg[name]["sorted_best_first"] = bool(((ds[0, 1:] - ds[0, :-1]) * (1 if args.metric == 0 else -1) >= 0).all().item())
            os.environ.pop("MVF_LARGE_K", None)
            corpus.reload_tuning()
            if not args.no_recall and ref is not None:
                t1 = time.perf_counter()
                osc_k, oidx_k, _ = oracle_topk_rows(oracle, 0, args.rows, args.dim, args.dtype, args.metric, q[:1], 16384)
                gi_k = ref[0].cpu().numpy().view(np.uint64)
                leg["recall_at_16384"] = recall_of(gi_k, oidx_k, ref[1].cpu().numpy(), osc_k, args.metric)
                leg["rows_identical_to_the_oracle_list"] = int((gi_k == oidx_k).sum())
                leg["recall_oracle_s"] = time.perf_counter() - t1
            result["any_k"] = leg
        # ---- the metric's second leg: the same resident corpus, 1024 batched queries (MFMA path) ----------
        # Three ways, same results: the default (int8 MFMA kernel selecting on the int8 shadow of the rows, every row inside
        # a proven bound of the k-th best re-scored exactly from the f32 rows), the f16 MFMA kernel on the scaled-f16
        # shadow (round 1's default), and the exact f32 MFMA kernel on the rows themselves.
        if args.queries == 1 and args.dtype == 0 and not args.no_batched:
            nqb, bsteps = 1024, 5
            dqb = torch.empty((nqb, args.dim), dtype=qdt, device=dev)
            _lib.gpu_check(_lib.gpu().mvfgpu_synth_queries_device(dqb.data_ptr(), nqb, args.dim, args.dtype, SEED + 1,
                                                                  local_rank, None))
            sel = sorted(set(int(x) for x in np.linspace(0, nqb - 1, 16)))  # 16 of the 1024 queries vs the oracle over ALL rows
            oidx = None
            if not args.no_recall:
                osc, oidx, _ = oracle_topk_rows(oracle, 0, args.rows, args.dim, args.dtype, args.metric, dqb.cpu().numpy()[sel], args.k)
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
                    tp = os.path.join(ROOT, "profiles", {2: "r05_bench_n1_q1024_hbm_traffic.json",
                                                         4: "r05_bench_n1_q1024_shadow_hbm_traffic.json",
                                                         6: "r05_bench_n1_q1024_i8_shadow_hbm_traffic.json"}.get(tmb.scan_kernel, "-"))
                    if not os.path.exists(tp):
                        tp = tp.replace("r05_", "r04_")
                    if os.path.exists(tp):
                        leg["roofline"]["traffic"] = json.load(open(tp))["roofline_traffic_bytes_per_launch"]
                        leg["roofline"]["traffic_source"] = "profiles/" + os.path.basename(tp)
                if oidx is not None:
                    gi = outb[1].cpu().numpy().view(np.uint64)[sel]
                    leg["recall_at_k"] = recall_of(gi, oidx, outb[0].cpu().numpy()[sel], osc, args.metric)
                    leg["rows_identical_to_the_oracle_list"] = int((gi == oidx).sum())
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
            # ---- ... and a small batch on a MID-SIZE corpus (VERDICT r4 item 6's case: 1M x 768 f32, 16 queries, top-100): every
            # kernel between the scans is latency here -- the library's own first-to-last-kernel time per search beside the wall time
            if world == 1:
                import ctypes as C
                mrows, mq = 1_000_000, 16
                with G.GpuCorpus.synthetic(mrows, args.dim, args.dtype, SEED, device=local_rank) as mc:
                    mds = torch.empty((mq, args.k), dtype=torch.float32, device=dev)
                    mdi = torch.empty((mq, args.k), dtype=torch.int64, device=dev)
                    lib = _lib.gpu()
                    call = lambda: _lib.gpu_check(lib.mvfgpu_search_device(mc._h, args.metric, dqb.data_ptr(), 0, args.dim, mq, args.k,
                                                                           mds.data_ptr(), mdi.data_ptr(), None, None))
                    for _ in range(5):
                        call()
                    lib.mvfgpu_set_profiling(mc._h, 1)
                    devms = []
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        call()
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t0) / 30
                    tmm = _lib.Timing()
                    lib.mvfgpu_last_timing(mc._h, C.byref(tmm))
                    lib.mvfgpu_set_profiling(mc._h, 0)
                    mid = {"workload": f"{mrows} x {args.dim} f32 {M_NAME[args.metric]}, {mq} batched queries, top-{args.k} (device pointers)",
                           "ms_per_search_wall": wall * 1e3, "ms_per_search_device": float(tmm.search_ms_avg),
                           "device_covers": "first to last kernel of a search on its stream (HIP events), mean of the profiled searches",
                           "value": float(mq) * mrows / wall, "unit": "distance-ops/s"}
                    if not args.no_recall:
                        msel = [0, 5, 10, 15]
                        mosc, moidx, _ = oracle_topk_rows(oracle, 0, mrows, args.dim, args.dtype, args.metric, dqb[:mq].cpu().numpy()[msel], args.k)
                        gi = mdi.cpu().numpy().view(np.uint64)[msel]
                        mid["recall_at_k"] = recall_of(gi, moidx, mds.cpu().numpy()[msel], mosc, args.metric)
                        mid["recall_queries_checked"] = len(msel)
                result["small_batch_mid_corpus"] = mid

    corpus.close()
    del searcher
    torch.cuda.empty_cache()

    # ---- second workload, every N: BASELINE.json configs[4] per rank ----------------------------------------------
    if not args.no_cfg5:
        leg = cfg5_sharded_leg(args, rank, local_rank, world, backend, dist, torch, G, ShardedSearcher, _lib)
        if rank == 0:
            if rehearsal:
                leg["rehearsal"] = rehearsal
            result["cfg5_sharded"] = leg

    # ---- the same config as STRONG scaling, every N: ONE fixed 100M x 1024 f16 corpus split by row range over the ranks --
    if not args.no_strong:
        torch.cuda.empty_cache()
        try:
            leg = cfg5_sharded_leg(args, rank, local_rank, world, backend, dist, torch, G, ShardedSearcher, _lib,
                                   total_rows=args.strong_rows)
        except Exception as e:  # e.g. a device too small for its share: the line keeps its other legs
            if world > 1:
                raise
            leg = {"error": f"{type(e).__name__}: {e}"[:400]}
        if rank == 0:
            if rehearsal:
                leg["rehearsal"] = rehearsal
            result["cfg5_strong"] = leg

    # ---- context for the MFMA fractions above: what the vendor GEMM library holds on THIS box (best case, 8192^3) --------
    if rank == 0 and world == 1 and not args.no_batched:
        result["vendor_gemm_reference"] = vendor_gemm_reference(torch, dev)

    # ---- N = 1: BASELINE.json configs[3] and configs[0] ------------------------------------------------------------------
    if rank == 0 and world == 1:
        from oracle import mvf_oracle as oracle
        if not args.no_cfg4:
            torch.cuda.empty_cache()
            result["cfg4_int8"] = cfg4_leg(args, torch, G, _lib, None if args.no_recall else oracle, local_rank)
        if not args.no_cfg1:
            result["cfg1"] = cfg1_block(args, torch, G, oracle, local_rank)
        if not args.no_file:
            torch.cuda.empty_cache()
            try:
                result["mvf_file_e2e"] = mvf_file_leg(args, G, None if args.no_recall else oracle, local_rank)
            except Exception as e:  # a full disk must not cost the line its other legs
                result["mvf_file_e2e"] = {"error": f"{type(e).__name__}: {e}"[:400]}

    # every rank has released its corpora: the process group is done; rank 0 goes on alone
    torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- third workload, every N: the single-process shard set over the same N devices (child process) ------------------
    if not args.no_shardset:
        result["shardset"] = shardset_leg(args, world)

    result["bench_wall_s"] = time.perf_counter() - T_START  # rank 0, from interpreter start to the line (all legs, oracle checks included)
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
