#!/bin/bash
# GPU busy time per train step: sum of kernel durations of a train-only bench run / (steps + warmup)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_gb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gb -o gb -- python $root/bench.py --steps 10 --warmup 3 --no-extras > $out/gb_bench.json 2> $out/gb.err
cp $(find /tmp/prof_gb -name '*kernel_stats.csv' | head -1) $out/gb_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/gb_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/13/1e6
n=sum(int(r["Calls"]) for r in rows)/13
print("gpu busy per step: %.2f ms, %d launches/step"%(tot,n))
ay=[r for r in rows if "k_" in r["Name"][:14]]
print("ayolo kernels: %.2f ms"%(sum(float(r["TotalDurationNs"]) for r in ay)/13/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:22]:
    print("%7.3f ms %5.0f x  %s"%(float(r["TotalDurationNs"])/13/1e6,int(r["Calls"])/13,r["Name"][:90]))
PY
cut -c1-160 $out/gb_bench.json
