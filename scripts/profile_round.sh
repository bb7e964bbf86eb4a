#!/bin/bash
# Collect the round's bench line and rocprofv3 summaries on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r02
# Writes everything under gpurun_out/prof_$1/ ; copy the summaries you want judged into profiles/.
# The profiled program goes directly after `--` (python3 ...): no wrapper, no exec hop behind the profiler.
set -o pipefail
R=${1:-r02}
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
S=scripts/rocprof_summarize.py
echo "== bench (unprofiled)"; timeout -k 10 900 python3 bench.py > $O/${R}_bench_n1.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "== kernel trace of the default bench (after a pause: straight behind the MFMA-heavy legs of the run above the HBM-bound kernel measured 3-4 % slower)"
sleep 60
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt_bench -- python3 bench.py --no-cpu-baseline --no-recall > $O/kt_bench.log 2>&1 || { tail -5 $O/kt_bench.log; exit 1; }
python3 $S stats $O/kt_bench $O/${R}_bench_n1_kernel_stats.csv "rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-recall (q=1 leg 5+50 searches, host-API leg 5+50, f16-shadow and int8-shadow stream legs 5+50 each, three batched legs 1+5 each, cfg5 shard leg 2+10, two 8192^3 library GEMMs); durations in us"
python3 $S launches $O/kt_bench $O/${R}_cfg3_shadow_q1024_scan_launches.csv "scan_mfma16" "per-launch durations of the K2 kernels for the narrow types in the default bench run: cfg3 (10M x 768 f32 cosine, 1024 queries) through the int8 shadow (scan_mfma16_dma_kernel<2, 2, ., true, 256>) and through the f16 shadow (<1, 2, ., true>), and the cfg5 shard leg (12.5M x 1024 f16 L2) through the int8 shadow (<2, 0, ., true, 256>)"
rm -rf $O/kt_bench
echo "== PMC passes for the single-query scan (FETCH_SIZE, WRITE_SIZE: separate runs)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5 > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
done
python3 $S traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/${R}_bench_n1_hbm_traffic.json 10000000 768 0 2 1 100 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5, MI355X, round ${R#r}"
for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1); grep -E "Correlation_Id|scan_stream_kernel|select_final" "$f" | head -40 > $O/${R}_bench_n1_pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $O/pmc_$c; done
echo "== PMC passes over the batched legs + FETCH_SIZE calibration"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/pmcb_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-cfg5 > $O/pmcb_$c.log 2>&1 || { tail -5 $O/pmcb_$c.log; exit 1; }
done
CM="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-cfg5: the last (largest) phase of the leg's K2 kernel"
python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_i8_shadow_hbm_traffic.json "scan_mfma16_dma_kernel<2, 2, false, true" 0 "$CM (int8-shadow selection, the default)"
python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_shadow_hbm_traffic.json "scan_mfma16_pp_kernel<1, 2, false, true" 0 "$CM (f16-shadow selection, scan path 3)"
python3 $S k2traffic $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE $O/${R}_bench_n1_q1024_hbm_traffic.json "scan_mfma_f32_kernel<2>" 0 "$CM (exact f32 MFMA, scan path 2)"
rm -rf $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal -- python3 scripts/calibrate_fetch_size.py > $O/cal.log 2>&1 || { tail -5 $O/cal.log; exit 1; }
python3 $S calibrate $O/cal $O/cal.log $O/${R}_fetch_size_calibration.json
rm -rf $O/cal
echo "== cfg4 / cfg5 kernel traces"
for cfg in cfg4 cfg5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$cfg -- python3 scripts/probe_k2_ab.py $cfg 1 lockstep,pingpong,default > $O/kt_$cfg.log 2>&1 || { tail -5 $O/kt_$cfg.log; exit 1; }
  python3 $S launches $O/kt_$cfg $O/${R}_${cfg}_scan_launches.csv "scan_mfma16" "rocprofv3 --kernel-trace -- python3 scripts/probe_k2_ab.py $cfg 1: every K2 launch of 2 x 3 searches per variant (lockstep = scan_mfma16_dma_kernel on the stored rows, pingpong = scan_mfma16_pp_kernel on the stored rows, default = the library's choice: int8-shadow selection on float rows), interleaved in one process"
  python3 $S stats $O/kt_$cfg $O/${R}_${cfg}_kernel_stats.csv "same run: per-kernel totals"
  rm -rf $O/kt_$cfg
done
ls -la $O
