#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" | grep -v "amdgpu.ids" | tail -80 > gpurun_out/r05_gpu_tests_full.txt
rm -f gpurun_out/r05_cfg3_ab.txt
for r in 1 2; do for v in 1 0; do echo "AYOLO_WGRAD3=$v" >> gpurun_out/r05_cfg3_ab.txt; AYOLO_WGRAD3=$v timeout 400 python tools/config_bench.py 3 2>/dev/null | tail -1 >> gpurun_out/r05_cfg3_ab.txt; done; done
tail -40 gpurun_out/r05_gpu_tests_full.txt; cat gpurun_out/r05_cfg3_ab.txt
