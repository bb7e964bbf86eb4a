#!/bin/bash
# Round 5: evidence stages on the GPU box (one gpurun call each, the profiled program directly after `--`):
#   bash scripts/profile_round5.sh ladder [cfgs] [tags] [probe variants]   -- unprofiled in-process ladder (scripts/k2_ladder.py)
#   bash scripts/profile_round5.sh ladderpmc [cfg] [tags] [probe variants] -- the same under two --pmc passes (+ kernel trace)
#   bash scripts/profile_round5.sh bench | pmc | traffic45 | legs            -- the bench line + its kernel-trace stats; HBM traffic of the
#        headline and the batched legs; of whole cfg4 / cfg5-shard searches; kernel traces of single searches
# Writes under gpurun_out/prof_r05/ ; copy what should be judged into profiles/.
set -o pipefail
STAGE=${1:-ladder}
R=r05
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
S=scripts/rocprof_summarize.py
K8='scan_mfma16_dma_kernel<2, 2, false, true, 256, true>'   # int8-shadow selection, cosine (cfg3 default)
case $STAGE in
ladder)
  CFGS=${2:-cfg3,cfg5}; TAGS=${3:-main,r4,noepi,nobias}; PV=${4:-8,8n}
  timeout -k 10 900 python3 scripts/k2_ladder.py $CFGS ${ROUNDS:-12} $TAGS $PV > $O/ladder_${CFGS//,/_}.log 2>&1 || { tail -5 $O/ladder_${CFGS//,/_}.log; exit 1; }
  grep "^==" $O/ladder_${CFGS//,/_}.log ;;
ladderpmc)
  CFG=${2:-cfg3}; TAGS=${3:-main,r4,noepi,nobias}; PV=${4:-8,8n}
  for pass in "a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
    p=($pass)
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc ${p[@]:1} --output-format csv -d $O/lp_${CFG}_${p[0]} -- python3 scripts/k2_ladder.py $CFG 1 $TAGS $PV > $O/lp_${CFG}_${p[0]}.log 2>&1 || { tail -5 $O/lp_${CFG}_${p[0]}.log; exit 1; }
    python3 scripts/pmc_summary.py $O/lp_${CFG}_${p[0]} "" > $O/${R}_k2_ladder_${CFG}_pmc_${p[0]}.json
    rm -rf $O/lp_${CFG}_${p[0]}
  done ;;
bench)
  echo "== bench (unprofiled)"; timeout -k 10 900 python3 bench.py > $O/${R}_bench_n1.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  echo "== kernel trace of the default bench"
  sleep 20
  timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt_bench -- python3 bench.py --no-cpu-baseline --no-recall --no-shardset --no-file --no-strong > $O/kt_bench.log 2>&1 || { tail -5 $O/kt_bench.log; exit 1; }
  python3 $S stats $O/kt_bench $O/${R}_bench_n1_kernel_stats.csv "rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-recall --no-shardset --no-file --no-strong (q=1 leg 5+50 searches, host-API leg, f16- and int8-shadow stream legs, three batched legs 1+5 each, cfg5 shard leg, cfg4 leg 2+12, cfg1 block, two 8192^3 library GEMMs); durations in us"
  python3 $S launches $O/kt_bench $O/${R}_k2_scan_launches.csv "scan_mfma16" "per-launch durations of the K2 kernels for the narrow types in the default bench run: cfg3 through the int8 shadow (<2, 2, ., true, 256, true>) and the f16 shadow (<1, 2, ...>), the cfg5 shard leg through the int8 shadow (<2, 0, ., true, 256, true>), cfg4 (<2, 1, ., false, 256, true>)"
  rm -rf $O/kt_bench ;;
pmc)
  echo "== PMC passes for the single-query scan (FETCH_SIZE, WRITE_SIZE: separate runs)"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
  done
  python3 $S traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/${R}_bench_n1_hbm_traffic.json 10000000 768 0 2 1 100 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-batched --no-cfg5 --no-cfg4 --no-cfg1 --no-shardset --no-file --no-strong, MI355X, round 5"
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
traffic45)
  echo "== cfg4 and the cfg5 shard: HBM bytes of a whole search (VERDICT r4 item 4)"
  for cfg in cfg4 cfg5; do
    nq=256; alg=38400000000; what="50M x 768 int8 dot, 256 queries"
    [ $cfg = cfg5 ] && { nq=1024; alg=12800000000; what="12.5M x 1024 f16 L2 through the int8 shadow (12.8 GB of shadow rows), 1024 queries"; }
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${cfg}_$c -- python3 scripts/probe_q64_trace.py $nq $cfg > $O/pmc_${cfg}_$c.log 2>&1 || { tail -5 $O/pmc_${cfg}_$c.log; exit 1; }
    done
    python3 $S searchtraffic $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE $O/${R}_${cfg}_hbm_traffic.json $alg "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/probe_q64_trace.py $nq $cfg: $what, top-100; four searches (the first also builds the row norms / the shadow); round 5's tree"
    rm -rf $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE
  done ;;
legs)
  echo "== cfg3 / cfg4 / cfg5: kernel traces of single searches"
  for cfg in "" cfg4 cfg5; do
    n=${cfg:-cfg3}
    bash scripts/trace_search.sh ${R}_$n $cfg > /dev/null || exit 1
    cp gpurun_out/${R}_${n}_kernels.txt $O/${R}_${n}_search_kernels.txt
  done
  bash scripts/trace_search.sh ${R}_q16_1m shape:1000000,768,0,2,100 16 > /dev/null || exit 1
  cp gpurun_out/${R}_q16_1m_kernels.txt $O/${R}_q16_1m_search_kernels.txt
  cat $O/${R}_*_search_kernels.txt ;;
esac
ls -la $O
