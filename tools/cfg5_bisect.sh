#!/bin/bash
# usage (GPU box, repo root): bash tools/cfg5_bisect.sh <sha> [<sha> ...]   (trees + built libraries under ab/bisect/<sha>/, HEAD = the repo itself)
# For every tree: tools/cfg5_time.py (YOLOv5x 1280^2 batch 8 fuse().eval() fp16 forward) under rocprofv3 --kernel-trace --stats;
# prints wall ms and the heaviest kernels' average duration -> gpurun_out/cfg5_bisect.txt
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
: > $out/cfg5_bisect.txt
for s in "$@"; do
    tree=$root/ab/bisect/$s
    [ "$s" = HEAD ] && tree=$root
    rm -rf /tmp/bis_$s
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bis_$s -o ks -- python $tree/tools/cfg5_time.py > /tmp/bis_$s.log 2> /tmp/bis_$s.err)
    f=$(find /tmp/bis_$s -name '*kernel_stats.csv' | head -1)
    echo "== $s  $(grep 'cfg5 ms' /tmp/bis_$s.log)" >> $out/cfg5_bisect.txt
    [ -n "$f" ] && cp $f $out/cfg5_bisect_${s}_kernel_stats.csv && python - "$f" >> $out/cfg5_bisect.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("   total kernel ms per forward (13 calls): %.3f" % (tot / 13 / 1e6))
for r in rows[:12]:
    name = re.sub(r"\(.*", "", r["Name"])
    name = re.sub(r"^void ", "", name)[:70]
    print("   %-70s calls %5s avg %9.1f us total %8.3f ms" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
    tail -2 /tmp/bis_$s.err | cut -c1-200 >> $out/cfg5_bisect.txt
done
cat $out/cfg5_bisect.txt
