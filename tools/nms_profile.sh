#!/bin/bash
# kernel-level breakdown of one non_max_suppression call on the bench's NMS workload (8 x 100800 x 85)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_nms
cat > /tmp/nms_only.py <<PY
import sys, json, torch
sys.path.insert(0, "$root")
import bench
for _ in range(3):
    r = bench.nms_extra(torch.device("cuda"))
print(json.dumps(r))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nms -o nms -- python /tmp/nms_only.py > $out/nms_prof.json 2> $out/nms_prof.err
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_nms/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
calls = 3 * 8          # nms_extra: 1 + 2 warm-up + 5 timed calls, three times
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    ms = float(r["TotalDurationNs"]) / calls / 1e6
    tot += ms
    print("%7.3f ms/call %6.1f launches/call  min %7.1f us  %s" % (ms, int(r["Calls"]) / calls, float(r.get("MinNs", 0)) / 1e3, r["Name"][:90]))
print("sum of all kernels per call: %.3f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e6))
PY
tail -1 $out/nms_prof.json | cut -c1-200
