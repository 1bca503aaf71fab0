#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r05_w3_probe_var.txt
for v in 0 1 2; do
for cfg in "64 64 64 1 80 80" "64 32 32 1 160 160"; do
  echo "=== variant $v: $cfg" >> gpurun_out/r05_w3_probe_var.txt
  AYOLO_LIB=$PWD/ab/libprobe_v$v.so timeout 120 python tools/w3_probe.py $cfg 2>&1 | grep "^mean\|launch" >> gpurun_out/r05_w3_probe_var.txt
done
done
cat gpurun_out/r05_w3_probe_var.txt
