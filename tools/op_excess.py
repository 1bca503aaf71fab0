#!/usr/bin/env python3
"""Ranks the ops of a tools/op_table.py table by their time above a simple roofline: max(bytes / 4.5 TB/s, FLOP / 1 PFLOP/s) +
4 us per launch.  This is how the 64-channel stem weight gradient's register spill was found (YOLOv5l: 2.2 ms for one op).
usage: python tools/op_excess.py gpurun_out/op_table.txt [top-N]"""
import collections
import re
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows, sec = [], None
for line in open(path):
    if line.startswith("----"):
        sec = line.split()[1]
        continue
    m = re.search(r"^\s*(\d+)\s+(OP_\w+)\s+(.*?)\s+([\d.]+) us\s+([\d.]+) MB\s+(\d+) GB/s\s+(\d+) TF/s", line)
    if not m:
        continue
    i, op, desc, us, mb, gbs, tf = m.groups()
    us, mb, tf = float(us), float(mb), float(tf)
    if us <= 0:
        continue
    ideal = max(mb / 4.5, tf * us / 1000.0) + 4.0          # MB / (4.5 MB/us);  TFLOP/s * us / (1000 TFLOP/s)
    rows.append((us - ideal, us, ideal, sec, int(i), op, desc.strip(), float(gbs), tf))
rows.sort(reverse=True)
for r in rows[:top]:
    print(f"+{r[0]:7.1f} us  t={r[1]:7.1f} ideal={r[2]:6.1f} {r[3]:9s}{r[4]:4d} {r[5]:17s} {r[6]:38s} {r[7]:5.0f} GB/s {r[8]:4.0f} TF/s")
fam = collections.defaultdict(float)
for r in rows:
    fam[r[5]] += r[0]
print("excess by family (us):", {k: round(v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
print(f"sum of op times {sum(r[1] for r in rows) / 1e3:.2f} ms, of which above the roofline {sum(r[0] for r in rows) / 1e3:.2f} ms")
