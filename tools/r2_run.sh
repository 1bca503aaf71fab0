#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_loss.py -m gpu -q -x 2>&1 | tail -2
bash tools/ab_bench.sh ab/libayolo_base.so ayolov2_amd/libayolo_hip.so 3
