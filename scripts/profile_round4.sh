#!/bin/bash
# Round 4: collect the bench line and the rocprofv3 summaries on the GPU box, in stages (a gpurun call is at most 20 min):
#   bash scripts/profile_round4.sh bench|pmc|legs|misc
# Writes under gpurun_out/prof_r03/ ; copy what should be judged into profiles/.
# The profiled program goes directly after `--` (python3 ...): no wrapper, no exec hop behind the profiler.
set -o pipefail
STAGE=${1:-bench}
R=r04
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
S=scripts/rocprof_summarize.py
K8='scan_mfma16_dma_kernel<2, 2, false, true, 256, true>'   # int8-shadow selection, cosine (cfg3 default)
case $STAGE in
bench)
  echo "== bench (unprofiled)"; timeout -k 10 900 python3 bench.py > $O/${R}_bench_n1.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  echo "== kernel trace of the default bench"
  sleep 30
  timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt_bench -- python3 bench.py --no-cpu-baseline --no-recall --no-shardset --no-file --no-strong > $O/kt_bench.log 2>&1 || { tail -5 $O/kt_bench.log; exit 1; }
  python3 $S stats $O/kt_bench $O/${R}_bench_n1_kernel_stats.csv "rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-recall --no-shardset --no-file --no-strong (q=1 leg 5+50 searches, host-API leg, f16- and int8-shadow stream legs, three batched legs 1+5 each, cfg5 shard leg, cfg4 leg 2+12, cfg1 block, two 8192^3 library GEMMs); durations in us"
  python3 $S launches $O/kt_bench $O/${R}_k2_scan_launches.csv "scan_mfma16" "per-launch durations of the K2 kernels for the narrow types in the default bench run: cfg3 through the int8 shadow (<2, 2, ., true, 256, true>: folded pre-filter) and the f16 shadow (<1, 2, ...>), the cfg5 shard leg through the int8 shadow (<2, 0, ., true, 256, true>), cfg4 (<2, 1, ., false, 256, true>)"
  rm -rf $O/kt_bench ;;
pmc)
  echo "== PMC passes for the single-query scan (FETCH_SIZE, WRITE_SIZE: separate runs)"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
  done
  python3 $S traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/${R}_bench_n1_hbm_traffic.json 10000000 768 0 2 1 100 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong, MI355X, round 4"
  for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1); grep -E "Correlation_Id|scan_stream_kernel|select_final" "$f" | head -40 > $O/${R}_bench_n1_pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $O/pmc_$c; done
  echo "== PMC passes over the batched legs"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/pmcb_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong > $O/pmcb_$c.log 2>&1 || { tail -5 $O/pmcb_$c.log; exit 1; }
  done
  CM="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong: the last (largest) phase of the leg's K2 kernel"
  python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_i8_shadow_hbm_traffic.json "$K8" 5759926272 "$CM (int8-shadow selection, the default; 7.5M rows x 768 B)"
  python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_shadow_hbm_traffic.json "scan_mfma16_pp_kernel<1, 2, false, true" 11519852544 "$CM (f16-shadow selection, scan path 3)"
  python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_hbm_traffic.json "scan_mfma_f32_kernel<2>" 23039705088 "$CM (exact f32 MFMA, scan path 2)"
  rm -rf $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE
  true ;;
legs)
  echo "== cfg3 / cfg4 / cfg5: kernel traces of single searches, folded pre-filter vs round 2's epilogue"
  for cfg in "" cfg4 cfg5; do
    n=${cfg:-cfg3}
    bash scripts/trace_search.sh ${R}_$n $cfg > /dev/null || exit 1
    cp gpurun_out/${R}_${n}_kernels.txt $O/${R}_${n}_search_kernels.txt
  done
  timeout -k 10 500 python3 scripts/probe_k2_ab.py cfg3,cfg5,cfg4,u8 3 default,oldepi > $O/k2_ab.log 2>&1 || { tail -5 $O/k2_ab.log; exit 1; }
  { echo "# scripts/probe_k2_ab.py cfg3,cfg5,cfg4,u8 3 default,oldepi -- one process, variants interleaved, 3 rounds x 3 searches: wall ms per search, last-phase scan ms"; echo "# default = folded pre-filter (scan_mfma16_bias.inc), oldepi = MVF_K2_BIAS=0 (round 2's epilogue in the same library)"; grep -E "^==" $O/k2_ab.log; } > $O/${R}_k2_ab.txt
  MVF_GPU_LIB_PATH=scripts/bin/libmvf_gpu_count.so timeout -k 10 300 python3 scripts/probe_bias_counts.py cfg3,cfg5,cfg4 2>/dev/null | grep -v amdgpu.ids > $O/${R}_k2_bias_counts.txt || true
  cat $O/${R}_k2_ab.txt $O/${R}_k2_bias_counts.txt ;;
misc)
  echo "== cfg4: HBM bytes of a whole search"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc4_$c -- python3 scripts/probe_q64_trace.py 256 cfg4 > $O/pmc4_$c.log 2>&1 || { tail -5 $O/pmc4_$c.log; exit 1; }
  done
  python3 $S searchtraffic $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE $O/${R}_cfg4_hbm_traffic.json 38400000000 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/probe_q64_trace.py 256 cfg4: 50M x 768 int8 dot, 256 queries, top-100; four searches (the first also builds the row norms)"
  rm -rf $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE
  echo "== K1 shape sweep"
  timeout -k 10 600 python3 scripts/sweep_k1_shapes.py > $O/${R}_k1_shape_sweep.csv 2> $O/sweep.err || { tail -5 $O/sweep.err; exit 1; }
  echo "== full-width equality of the selection paths (the -m gpu tests, with their report lines)"
  timeout -k 10 400 python3 -m pytest tests/test_gpu_round3.py -q -m gpu -s -k "equal" 2>&1 | grep -oE "metric [0-9] path.*|[0-9]+ passed.*|[0-9]+ failed.*" > $O/${R}_shadow_vs_exact_equivalence.txt
  cat $O/${R}_shadow_vs_exact_equivalence.txt ;;
esac
ls -la $O
