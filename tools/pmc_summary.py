#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files into per-kernel sums (bytes, with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md section HBM).  usage: pmc_summary.py fetch.csv write.csv steps out.json [model batch size]"""
import csv, json, sys, collections
fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0][:70]
        d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
    return d
f = agg(fetch_csv, "FETCH_SIZE"); w = agg(write_csv, "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[0] + w.get(k, [0, 0])[0])):
    fk, wk = f.get(k, [0.0, 0]), w.get(k, [0.0, 0])
    res[k] = {"launches_per_step": max(fk[1], wk[1]) / steps,
              "fetch_GB_per_step_corrected": round(fk[0] * 1024 * 2 / steps / 1e9, 3),   # KB -> B, x2 on gfx950
              "write_GB_per_step": round(wk[0] * 1024 / steps / 1e9, 3)}
# the workload the passes ran (bench.py's defaults unless given: model batch size) -- bench.py uses the summary for that workload only
wl = sys.argv[5:8] if len(sys.argv) >= 8 else ["yolov5s", "64", "640"]
res["_workload"] = {"model": wl[0], "batch": int(wl[1]), "size": int(wl[2])}
json.dump(res, open(out, "w"), indent=1)
for k, v in [kv for kv in res.items() if kv[0] != "_workload"][:14]:
    print(f"{k[:60]:60s} {v}")
