#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02
# 1) train-only step (bench.py --no-extras) under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_kernel_stats_train_only.csv
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same run          -> gpurun_out/<tag>_pmc_hbm_traffic.json
# (counter passes carry --kernel-trace only: no other trace domain next to --pmc)
tag=${1:-r02}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python $root/bench.py --steps 10 --warmup 3 --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
cp $(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats_train_only.csv
STEPS=6
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python $root/bench.py --steps 4 --warmup 2 --no-extras > /dev/null 2>> $out/${tag}_prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python $root/bench.py --steps 4 --warmup 2 --no-extras > /dev/null 2>> $out/${tag}_prof.err
python $root/tools/pmc_summary.py $(find /tmp/prof_f -name '*counter_collection.csv' | head -1) $(find /tmp/prof_w -name '*counter_collection.csv' | head -1) $STEPS $out/${tag}_pmc_hbm_traffic.json > $out/${tag}_pmc_top.txt 2>> $out/${tag}_prof.err
tail -3 $out/${tag}_prof.err
