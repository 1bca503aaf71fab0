#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh r01
# the default bench line, then the same command under rocprofv3 --kernel-trace --stats (kernel stats CSV + its bench line)
tag=${1:-r01}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python $root/bench.py > $out/${tag}_bench_n1.json 2> $out/${tag}_bench.err
cd /tmp
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python $root/bench.py --steps 10 --warmup 3 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
cp $(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
tail -2 $out/${tag}_bench.err; tail -2 $out/${tag}_prof.err
