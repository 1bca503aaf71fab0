#!/bin/bash
# usage (GPU box, repo root): bash tools/nms_profile.sh  -> gpurun_out/nms_kernel_stats.csv (rocprofv3 kernel stats of the NMS leg)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_nms
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nms -o nms -- python $root/tools/nms_time.py > $out/nms_time_under_rocprof.txt 2>$out/nms_prof.err
cp $(find /tmp/prof_nms -name '*kernel_stats.csv' | head -1) $out/nms_kernel_stats.csv
head -30 $out/nms_kernel_stats.csv | cut -c1-200
