#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
echo "row3=0 $(AYOLO_GCONV_ROW3=0 python tools/cfg5_time.py 2>/dev/null | tail -1)"
echo "row3=1 $(python tools/cfg5_time.py 2>/dev/null | tail -1)"
done
for i in 1 2; do
for r in 0 1; do
  echo "row3=$r $(AYOLO_GCONV_ROW3=$r python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done; done
python tools/conv_sweep.py yolov5s 64 640 2>/dev/null | grep -E " 3 1 |total" 
AYOLO_GCONV_ROW3=0 python tools/conv_sweep.py yolov5s 64 640 2>/dev/null | grep -E " 3 1 |total"
