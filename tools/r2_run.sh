#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "maxpool" 2>&1 | tail -3
bash tools/ab_bench.sh ab/libayolo_base.so ayolov2_amd/libayolo_hip.so 2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o pk -- python $GRAFT_REPO_ROOT/bench.py --no-extras --steps 10 --warmup 3 > /dev/null 2>&1; grep -E "maxpool|loss_grad" $(find /tmp/pk -name '*kernel_stats.csv' | head -1) | cut -c1-120
