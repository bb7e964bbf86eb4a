"""Summarise a rocprofv3 --pmc --kernel-trace run: for every kernel name, the counters of its LONGEST dispatch and the
figures derived from them (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ counters over all CUs).
usage: python scripts/pmc_summary.py <rocprof output dir> [name filter] > summary.json"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "scan_"
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
disp = defaultdict(dict)
name = {}
for f in cc:
    for r in csv.DictReader(open(f)):
        disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
best = {}
for did, n in name.items():
    if did not in dur or (flt not in n if flt else not ("scan_mfma16_dma_kernel" in n or "kloop_kernel" in n)):
        continue
    if n not in best or dur[did] > dur[best[n]]:
        best[n] = did
out = []
for n, did in best.items():
    c, ms = disp[did], dur[did]
    e = {"kernel": n, "duration_ms": ms, "counters": c}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        e["effective_clock_GHz"] = cyc / (ms * 1e-3) / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in c:
            e["lds_array_busy_fraction"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in c:
                e[k.lower() + "_fraction_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    out.append(e)
json.dump(out, sys.stdout, indent=1)
