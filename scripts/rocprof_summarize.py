"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

  stats    <dir> <out.csv> <comment>             per-kernel calls / total / average duration (us) from the kernel trace
  launches <dir> <out.csv> <name filter> <comment>  one line per matching dispatch: id, kernel, grid, duration (us)
  traffic  <fetch dir> <write dir> <out.json> rows dim dtype metric queries k <comment>
           per-kernel FETCH_SIZE / WRITE_SIZE per launch (KiB) and, for the streaming scan kernel, the corrected HBM
           bytes per launch (gfx950: FETCH_SIZE reports half of a wide coalesced streaming read -> x2; WRITE_SIZE exact;
           MI355X_MICROARCH.md §HBM)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def _trace(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def stats(d, out, comment):
    agg = defaultdict(lambda: [0, 0.0])
    for r in _trace(d):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values()) or 1.0
    with open(out, "w", newline="") as fh:
        fh.write('"# %s"\n' % comment.replace('"', "'"))
        w = csv.writer(fh)
        w.writerow(["name", "total_calls", "total_duration_us", "average_us", "percentage"])
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, n, round(t, 3), round(t / n, 3), round(100 * t / tot, 4)])


def launches(d, out, flt, comment):
    rows = [r for r in _trace(d) if flt in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(out, "w", newline="") as fh:
        fh.write('"# %s"\n' % comment.replace('"', "'"))
        w = csv.writer(fh)
        w.writerow(["dispatch_id", "kernel", "grid_size", "workgroup_size", "vgpr", "lds_bytes", "start_us_rel", "duration_us"])
        t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
        for r in rows:
            grid = r.get("Grid_Size") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if r.get(k))
            wg = r.get("Workgroup_Size") or "x".join(r.get(k, "") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z") if r.get(k))
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"], grid, wg,
                        r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), round((int(r["Start_Timestamp"]) - t0) / 1e3, 3),
                        round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 3)])


def _counter(d, name):
    agg = defaultdict(lambda: [0, 0.0])
    per_dispatch = defaultdict(float)
    kname = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kname[r["Dispatch_Id"]] = r["Kernel_Name"]
    for did, v in per_dispatch.items():
        a = agg[kname[did]]
        a[0] += 1
        a[1] += v
    return agg


def traffic(fd, wd, out, rows, dim, dtype, metric, queries, k, comment):
    f, w = _counter(fd, "FETCH_SIZE"), _counter(wd, "WRITE_SIZE")
    es = {0: 4, 1: 2, 2: 1, 3: 1}[dtype]
    kernels, roof = {}, None
    for name in sorted(set(f) | set(w)):
        e = {}
        if name in f:
            e["fetch_size_kib_per_launch"] = f[name][1] / f[name][0]
            e["fetch_launches"] = f[name][0]
        if name in w:
            e["write_size_kib_per_launch"] = w[name][1] / w[name][0]
            e["write_launches"] = w[name][0]
        if "scan_stream_kernel" in name and name in f and name in w:
            e["hbm_bytes_per_launch_corrected"] = 2 * e["fetch_size_kib_per_launch"] * 1024 + e["write_size_kib_per_launch"] * 1024
            e["algorithmic_bytes_per_launch"] = rows * dim * es
            if roof is None or f[name][0] > roof[1]:
                roof = (e["hbm_bytes_per_launch_corrected"], f[name][0])
        kernels[name] = e
    json.dump({"source": comment,
               "correction": "gfx950: FETCH_SIZE reports exactly half of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM) -> x2; "
                             "WRITE_SIZE exact; both counters are in KiB",
               "kernels": kernels,
               "workload": {"rows": rows, "dim": dim, "dtype": dtype, "metric": metric, "queries": queries, "k": k},
               "roofline_traffic_bytes_per_launch": roof[0] if roof else None}, open(out, "w"), indent=1)


def k2traffic(fd, wd, out, kname, alg_bytes, comment):
    """HBM bytes of the LAST phase (largest FETCH_SIZE dispatch) of one K2 kernel: 2 x FETCH_SIZE + WRITE_SIZE of the same
    dispatch ordinal in the separate WRITE_SIZE pass."""
    def series(d, name):
        out_ = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name), key=lambda r: int(r["Dispatch_Id"]))
            per_disp = defaultdict(float)
            order = []
            for r in rows:
                if kname in r["Kernel_Name"]:
                    if r["Dispatch_Id"] not in per_disp:
                        order.append(r["Dispatch_Id"])
                    per_disp[r["Dispatch_Id"]] += float(r["Counter_Value"])
            return [per_disp[dd] for dd in order]
        return []
    f, w = series(fd, "FETCH_SIZE"), series(wd, "WRITE_SIZE")
    if not f:
        return
    i = max(range(len(f)), key=lambda j: f[j])
    wv = w[i] if i < len(w) else 0.0
    json.dump({"source": comment, "kernel_name_contains": kname, "dispatches_of_the_kernel": len(f), "last_phase_dispatch_ordinal": i,
               "fetch_size_kib": f[i], "write_size_kib": wv,
               "correction": "x2 on FETCH_SIZE (gfx950, 16-byte-per-lane streaming reads, global_load and global_load_lds alike; "
                             "profiles/r02_fetch_size_calibration.json); WRITE_SIZE exact",
               "algorithmic_bytes_per_launch": float(alg_bytes),
               "roofline_traffic_bytes_per_launch": 2 * f[i] * 1024 + wv * 1024}, open(out, "w"), indent=1)


def searchtraffic(fd, wd, out, alg_bytes, comment):
    """HBM bytes of a WHOLE search: 2 x FETCH_SIZE + WRITE_SIZE summed over every dispatch of the search's kernels,
    divided by the number of searches in the run (= dispatches of the query-preparation kernel)."""
    names = ("scan_mfma", "scatter_cand", "compact", "scan_stream", "select_final", "prep_queries", "flag_compact", "rescore", "refine_tau")

    def total(d, counter):
        t, searches, per = 0.0, 0, defaultdict(float)
        seen = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter or not any(n in r["Kernel_Name"] for n in names):
                    continue
                t += float(r["Counter_Value"])
                per[next((n for n in names if n in r["Kernel_Name"]), "other")] += float(r["Counter_Value"])
                if "prep_queries" in r["Kernel_Name"] and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    searches += 1
        return t, max(searches, 1), per
    f, nf, pf = total(fd, "FETCH_SIZE")
    w, nw, pw = total(wd, "WRITE_SIZE")
    json.dump({"source": comment, "searches_in_the_fetch_pass": nf, "searches_in_the_write_pass": nw,
               "fetch_size_kib_per_search": f / nf, "write_size_kib_per_search": w / nw,
               "fetch_size_kib_per_search_by_kernel": {k: v / nf for k, v in sorted(pf.items(), key=lambda kv: -kv[1])},
               "correction": "x2 on FETCH_SIZE for every kernel (gfx950: exact for the 16-byte-per-lane streaming reads of the scan, "
                             "which is 99 % of the bytes; an upper bound for the small kernels); WRITE_SIZE exact",
               "algorithmic_bytes_per_search": float(alg_bytes),
               "search_traffic_bytes": 2 * f / nf * 1024 + w / nw * 1024}, open(out, "w"), indent=1)


def calibrate(d, log, out):
    """Pairs the `expect <kernel> <bytes>` lines of scripts/calibrate_fetch_size.py with the largest FETCH_SIZE dispatch of
    that kernel: known bytes / (FETCH_SIZE KiB x 1024) = the factor the counter has to be multiplied by."""
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for line in open(log):
        if not line.startswith("expect "):
            continue
        _, rest = line.split(" ", 1)
        kname, want = rest.rsplit(" ", 1)
        hits = sorted(((k, max(v)) for k, v in per.items() if kname in k), key=lambda kv: -kv[1])[:1]  # the instantiation that ran the scan
        for k, v in hits:
            rows.append({"kernel": k, "known_hbm_read_bytes_of_the_launch": int(want), "FETCH_SIZE_KiB": v,
                         "FETCH_SIZE_bytes": v * 1024, "factor_known_over_counter": int(want) / (v * 1024)})
    json.dump({"what": "FETCH_SIZE calibration on launches whose HBM read traffic is known (every block reads its rows once, "
                       "no sharing between blocks); MI355X_MICROARCH.md says x2 for wide coalesced streaming reads on gfx950",
               "rows": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif mode == "launches":
        launches(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif mode == "calibrate":
        calibrate(sys.argv[2], sys.argv[3], sys.argv[4])
    elif mode == "k2traffic":
        k2traffic(*sys.argv[2:])
    elif mode == "searchtraffic":
        searchtraffic(*sys.argv[2:])
    elif mode == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], *[int(x) for x in sys.argv[5:11]], sys.argv[11])
