#!/usr/bin/env python3
"""How many runtime copy kernels (__amd_rocclr_copyBuffer) run INSIDE a steady-state step?  Reads a rocprofv3 --kernel-trace csv and
counts, for the last iterations, the kernels between two consecutive launches of a marker kernel (default k_pack_input: one per
forward).  usage: python tools/copy_in_steady_state.py kernel_trace.csv [marker substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_pack_input"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
print("marker launches:", len(idx))
for a, b in list(zip(idx, idx[1:]))[-3:]:
    c = collections.Counter()
    t = collections.Counter()
    for r in rows[a:b]:
        n = r["Kernel_Name"][:170]
        c[n] += 1
        t[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    print(f"--- iteration of {span / 1e6:.3f} ms, {b - a} kernels")
    for n, k in c.most_common():
        if "copyBuffer" in n or "at::native" in n or "fill" in n.lower():
            print(f"   {k:4d} x {t[n] / k / 1e3:7.1f} us  {n}")
