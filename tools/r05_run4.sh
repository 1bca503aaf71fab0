#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r05_w3_probe.txt
for cfg in "64 64 64 1 80 80" "64 128 128 1 40 40" "64 32 32 1 160 160" "64 256 256 1 20 20"; do
  echo "=== $cfg" >> gpurun_out/r05_w3_probe.txt
  AYOLO_LIB=$PWD/ab/libayolo_probe.so timeout 120 python tools/w3_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "^  step  [2-9]\|^  step 1" >> gpurun_out/r05_w3_probe.txt
done
cat gpurun_out/r05_w3_probe.txt
