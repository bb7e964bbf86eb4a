#!/bin/bash
# Instruction / wait counters of the streaming kernel on a small corpus, 1 vs 4 queries per pass (development aid)
set -o pipefail
N=${1:-10000}; D=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for NQ in 1 4; do
 for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  O=gpurun_out/pmc_tiny
  rm -rf $O
  timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O -- python3 scripts/probe_tiny_trace.py $N $D $NQ > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  python3 - $O $NQ <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
disp = defaultdict(dict); name = {}
for f in cc:
    for r in csv.DictReader(open(f)):
        disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
last = {}
for did in sorted(name, key=int):
    if "scan_stream" in name[did] or "select_final" in name[did]:
        last[name[did][:60]] = disp[did]
for n, c in last.items():
    print(f"nq={sys.argv[2]} {n}: " + "  ".join(f"{k}={v:.0f}" for k, v in sorted(c.items())))
PY
 done
done
rm -rf gpurun_out/pmc_tiny
