#!/bin/bash
mkdir -p gpurun_out
export AYOLO_LIB=$PWD/ab/libayolo_probe.so
for cfg in "64 64 64 1 80 80" "64 128 128 1 40 40" "64 128 256 2 80 80" "64 32 32 1 160 160"; do
  echo "=== $cfg" >> gpurun_out/r05_w3_probe.txt
  timeout 120 python tools/w3_probe.py $cfg >> gpurun_out/r05_w3_probe.txt 2>&1
done
tail -60 gpurun_out/r05_w3_probe.txt
