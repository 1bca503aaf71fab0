#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" 2>&1 | tail -3 > gpurun_out/r05_run18_tests.txt
rm -f gpurun_out/r05_w3_probe.txt
for cfg in "64 64 64 1 80 80" "64 256 256 1 20 20" "64 32 32 1 160 160"; do
  echo "=== $cfg" >> gpurun_out/r05_w3_probe.txt
  AYOLO_WGRAD3_MINHW=0 AYOLO_LIB=$PWD/ab/libayolo_probe.so timeout 120 python tools/w3_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "^  step\|^--- work" >> gpurun_out/r05_w3_probe.txt
done
AYOLO_WGRAD3_MINHW=0 timeout 300 python tools/conv_sweep.py > gpurun_out/r05_sweep_w3v6.txt 2>&1
timeout 1500 bash tools/ab_trees.sh 3 30 "hw80=python bench.py" "all=AYOLO_WGRAD3_MINHW=0 python bench.py" "hw40=AYOLO_WGRAD3_MINHW=40 python bench.py" "off=AYOLO_WGRAD3=0 python bench.py" > gpurun_out/r05_ab_w3v6.txt 2>&1
cat gpurun_out/r05_run18_tests.txt; grep "^===\|^mean\|launch\|epilogue" gpurun_out/r05_w3_probe.txt; tail -5 gpurun_out/r05_ab_w3v6.txt; cut -c1-24,64-90 gpurun_out/r05_sweep_w3v6.txt | grep " 3 1 "
