#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q 2>&1 | tail -6 > gpurun_out/r05_run11_tests.txt
timeout 600 bash tools/ab_trees.sh 3 30 "lin_off=AYOLO_GCONV_LIN=0 python bench.py" "lin_on=python bench.py" "r04=python ab/base_r04/bench.py" > gpurun_out/r05_ab_lin.txt 2>&1
rm -f gpurun_out/r05_cfg5_lin.txt
for v in 0 1; do echo "AYOLO_GCONV_LIN=$v" >> gpurun_out/r05_cfg5_lin.txt; AYOLO_GCONV_LIN=$v timeout 300 python tools/cfg5_time.py 2>&1 | tail -2 >> gpurun_out/r05_cfg5_lin.txt; done
tail -3 gpurun_out/r05_run11_tests.txt; tail -4 gpurun_out/r05_ab_lin.txt; cat gpurun_out/r05_cfg5_lin.txt
