#!/bin/bash
# Kernel trace of single / small-batch searches of a small corpus (development aid): bash scripts/trace_tiny.sh <n> <dim> -> stdout
set -o pipefail
N=${1:-10000}; D=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for NQ in 1 4 16; do
  O=gpurun_out/kt_tiny_$NQ
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 scripts/probe_tiny_trace.py $N $D $NQ > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  F=$(find $O -name '*kernel_trace.csv' | head -1)
  python3 - "$F" $NQ <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "scan_stream" in r["Kernel_Name"] or "select_final" in r["Kernel_Name"]]
per = len(rows) // 5
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
print(f"nq={sys.argv[2]}: last search, {per} launches")
for r in last:
    n = re.sub(r"\(mvf.*", "", r["Kernel_Name"].replace("void ", "").replace("mvf::(anonymous namespace)::", ""))
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:7.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  grid {r['Grid_Size_X']}x{r.get('Grid_Size_Y', '?')} wg {r['Workgroup_Size_X']} lds {r['LDS_Block_Size']}  {n}")
PY
  rm -rf $O
done
