#!/bin/bash
# Round 5: evidence stages on the GPU box (one gpurun call each, the profiled program directly after `--`):
#   bash scripts/profile_round5.sh ladder [cfgs] [tags] [probe variants]   -- unprofiled in-process ladder (scripts/k2_ladder.py)
#   bash scripts/profile_round5.sh ladderpmc [cfg] [tags] [probe variants] -- the same under two --pmc passes (+ kernel trace)
# Writes under gpurun_out/prof_r05/ ; copy what should be judged into profiles/.
set -o pipefail
STAGE=${1:-ladder}
R=r05
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
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
esac
ls -la $O
