#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
echo "v1   $(AYOLO_LIB=$(realpath ab/libayolo_g3v1.so) python tools/cfg5_time.py 2>/dev/null | tail -1)"
echo "deep $(python tools/cfg5_time.py 2>/dev/null | tail -1)"
done
bash tools/ab_bench.sh ab/libayolo_g3v1.so ayolov2_amd/libayolo_hip.so 2
python tools/conv_sweep.py yolov5s 64 640 2>/dev/null | grep -E " 3 1 |total" 
