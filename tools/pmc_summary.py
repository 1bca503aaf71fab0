#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files into per-kernel sums (bytes, with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md section HBM).  usage: pmc_summary.py fetch.csv write.csv steps out.json"""
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
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res.items())[:14]:
    print(f"{k[:60]:60s} {v}")
