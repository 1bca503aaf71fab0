#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/ddp_timeline.sh r02
# The train step with the gradient exchange forced on ONE GPU (AYOLO_FORCE_DDP=1: single-rank RCCL group, the same bucketed
# FlatGradDDP path the 8-GPU run takes) under rocprofv3 --kernel-trace; tools/ddp_timeline.py turns the trace into a
# per-step table: where every RCCL kernel starts / ends relative to the backward window of its step.
tag=${1:-r02}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ddp
AYOLO_FORCE_DDP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -o ddp -- python $root/bench.py --steps 6 --warmup 3 --no-extras > $out/${tag}_ddp_bench.json 2> $out/${tag}_ddp_prof.err
python $root/tools/ddp_timeline.py $(find /tmp/prof_ddp -name '*kernel_trace.csv' | head -1) > $out/${tag}_ddp_overlap_timeline.txt 2>> $out/${tag}_ddp_prof.err
tail -30 $out/${tag}_ddp_overlap_timeline.txt
