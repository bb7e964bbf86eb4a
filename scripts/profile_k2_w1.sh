#!/bin/bash
# Counters for the K2 wave-structure probe (scripts/probe_k2_w1.hip): effective clock, matrix-pipe and LDS-array busy
# fractions, wave-cycle split for W8 / W4 / W4P on the cfg4 and cfg3 shapes.  One gpurun call.
set -o pipefail
O=gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for shape in "cfg4 50000000 768 256" "cfg3 10000000 768 1024"; do
  set -- $shape
  for pass in "a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
    set -- $shape; tag=$1; n=$2; dim=$3; nq=$4
    p=($pass)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${p[@]:1} --output-format csv -d $O/w1_${tag}_${p[0]} -- scripts/bin/probe_k2_w1 $n $dim $nq 3 8,4,4p > $O/w1_${tag}_${p[0]}.log 2>&1 || { tail -5 $O/w1_${tag}_${p[0]}.log; exit 1; }
    python3 scripts/pmc_summary.py $O/w1_${tag}_${p[0]} kloop_kernel > $O/r04_k2_w1_${tag}_pmc_${p[0]}.json
    rm -rf $O/w1_${tag}_${p[0]}
  done
done
ls -la $O
