#!/bin/bash
# A/B timing of two builds of libayolo_hip.so on the SAME box (box-to-box variance is ~2.5 %, larger than most kernel tweaks).
# usage (GPU box, repo root):  bash tools/ab_bench.sh ayolov2_amd/libayolo_hip_a.so ayolov2_amd/libayolo_hip.so [rounds]
#   build A first:  make -C ayolov2_amd/csrc && cp ayolov2_amd/libayolo_hip.so ayolov2_amd/libayolo_hip_a.so ; then edit + make
# Alternates A, B, A, B ... and prints ms_per_step of every run.
a=$1; b=$2; rounds=${3:-2}
for i in $(seq 1 $rounds); do
  for lib in "$a" "$b"; do
    ms=$(AYOLO_LIB=$(realpath "$lib") python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$lib  $ms ms/step"
  done
done
