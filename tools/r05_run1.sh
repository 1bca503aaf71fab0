#!/bin/bash
# round 5, GPU run 1: conv parity with k_wgrad3 routed, per-layer sweep with / without it, same-box A/B against round 4's tree
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -15 > gpurun_out/r05_run1_conv_tests.txt
AYOLO_WGRAD3=0 timeout 300 python tools/conv_sweep.py > gpurun_out/r05_sweep_w3off.txt 2>&1
AYOLO_WGRAD3=1 timeout 300 python tools/conv_sweep.py > gpurun_out/r05_sweep_w3on.txt 2>&1
timeout 600 bash tools/ab_trees.sh 2 30 "r04=python ab/base_r04/bench.py" "new=python bench.py" > gpurun_out/r05_ab1.txt 2>&1
tail -5 gpurun_out/r05_run1_conv_tests.txt; tail -4 gpurun_out/r05_ab1.txt
