#!/bin/bash
# Kernel trace of a few 1024-query searches (development aid): bash scripts/trace_search.sh <tag> [cfg4|cfg5|shape:n,dim,dtype,metric,k] [queries] -> gpurun_out/<tag>_kernels.txt
set -o pipefail
TAG=${1:-trace}; CFG=${2:-}
NQ=${3:-1024}; [ "$CFG" = cfg4 ] && [ -z "$3" ] && NQ=256
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/kt_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 scripts/probe_q64_trace.py $NQ $CFG > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
F=$(find $O -name '*kernel_trace.csv' | head -1)
python3 - "$F" > gpurun_out/${TAG}_kernels.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last search: from the last query-preparation kernel on
idx = [i for i, r in enumerate(rows) if "prep_queries" in r["Kernel_Name"]]
s = idx[-1]
t0 = int(rows[s]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows[s:])
print(f"last search: {(t1 - t0) / 1e3:.1f} us first kernel start to last kernel end, {len(rows) - s} launches")
tot = {}
for r in rows[s:]:
    n = r["Kernel_Name"].replace("void ", "").replace("mvf::(anonymous namespace)::", "").replace("mvf::", "")
    n = re.sub(r"\(mvf.*|\(unsigned.*|\(void.*|\(float.*|\(uint.*", "", n)[:80]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot.setdefault(n, [0.0, 0]); tot[n][0] += d; tot[n][1] += 1
busy = sum(v[0] for v in tot.values())
print(f"sum of kernel durations {busy:.1f} us (gaps {(t1 - t0) / 1e3 - busy:.1f} us)")
for k, v in sorted(tot.items(), key=lambda x: -x[1][0]):
    print(f"{v[0]:9.1f} us  x{v[1]:<3d} {k}")
print("scan launches (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows[s:] if "scan_mfma" in r["Kernel_Name"]])
for pat in ("scatter_cand", "compact_margin", "rescore_", "refine_tau"):
    print(pat, "launches (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows[s:] if pat in r["Kernel_Name"]])
PY
rm -rf $O; cat gpurun_out/${TAG}_kernels.txt
