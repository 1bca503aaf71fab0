#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -6 > gpurun_out/r05_run10_conv_tests.txt
AYOLO_WGRAD3=1 timeout 300 python tools/conv_sweep.py > gpurun_out/r05_sweep_w3v5b.txt 2>&1
rm -f gpurun_out/r05_w3_probe.txt
for cfg in "64 64 64 1 80 80" "64 128 128 1 40 40" "64 32 32 1 160 160" "64 256 256 1 20 20"; do
  echo "=== $cfg" >> gpurun_out/r05_w3_probe.txt
  AYOLO_LIB=$PWD/ab/libayolo_probe.so timeout 120 python tools/w3_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "^  step  [2-9]\|^  step 1" >> gpurun_out/r05_w3_probe.txt
done
timeout 600 bash tools/ab_trees.sh 2 30 "r04=python ab/base_r04/bench.py" "new=python bench.py" "new_w3off=AYOLO_WGRAD3=0 python bench.py" > gpurun_out/r05_ab6.txt 2>&1
tail -3 gpurun_out/r05_run10_conv_tests.txt; grep "^===\|^mean\|launch" gpurun_out/r05_w3_probe.txt; tail -4 gpurun_out/r05_ab6.txt; cut -c1-24,64-90 gpurun_out/r05_sweep_w3v5b.txt | grep " 3 1 "
