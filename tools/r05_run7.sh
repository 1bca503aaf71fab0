#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r05_w3_probe_exp.txt
for v in e1 e2; do
for cfg in "64 64 64 1 80 80"; do
  echo "=== experiment $v: $cfg" >> gpurun_out/r05_w3_probe_exp.txt
  AYOLO_LIB=$PWD/ab/libprobe_$v.so timeout 120 python tools/w3_probe.py $cfg 2>&1 | grep "^mean\|launch\|^  step  [01]" >> gpurun_out/r05_w3_probe_exp.txt
done
done
cat gpurun_out/r05_w3_probe_exp.txt
