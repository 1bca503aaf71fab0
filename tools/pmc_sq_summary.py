#!/usr/bin/env python3
"""Per-kernel sums of an SQ --pmc pass (rocprofv3 counter_collection.csv).  usage: pmc_sq_summary.py sq.csv steps out.json
Units (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over SIMDs, GRBM_GUI_ACTIVE is the kernel's wall cycles SUMMED OVER THE 8 XCDs
(checked against launch durations: 71 launches x 45 us = 7.7 M cycles vs 66.9 M counted).
  mfma_busy      = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs)    share of all SIMD-cycles with the matrix pipe busy
  parked/stalled/issuing = WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY over WAVE_CYCLES  (disjoint, ~1 in sum)
  lds_conflict   = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE           share of LDS-array cycles lost to bank conflicts
  lds_busy       = LDS_IDX_ACTIVE / (GUI_ACTIVE / 8 * 256 CUs)  how much of the kernel the LDS array is active AT ALL: a high
                   conflict share over a tiny lds_busy (the BatchNorm kernels read six per-channel constants per thread once, in
                   the prologue) costs nothing"""
import collections
import csv
import json
import sys

path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
res = {}
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    gui, wave = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, c.get("SQ_WAVE_CYCLES", 0.0)
    if gui <= 0 or wave <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    res[k] = {"launches_per_step": round(len(calls[k]) / steps, 1),
              "gpu_cycles_per_step_M": round(gui / steps / 1e6, 3),
              "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 4),
              "parked": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3),
              "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
              "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3),
              "lds_conflict": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds, 4) if lds > 0 else None,
              "lds_busy": round(lds / (gui * 256), 4)}
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res.items())[:14]:
    print(f"{k[:58]:58s} {v}")
