#!/bin/bash
AYOLO_LIB=$(realpath ab/libayolo_mi2_64.so) timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -2
cat > /tmp/ev.py <<PY
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from ayolov2_amd import YOLOModel
torch.manual_seed(0)
m = YOLOModel("ayolov2_amd/configs/yolov5x.yaml").cuda().fuse().eval()
x = torch.rand(8, 3, 1280, 1280, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); print("cfg5 ms", (time.perf_counter() - t) * 100)
PY
for i in 1 2; do
echo "hip ragged64=0 $(AYOLO_GCONV_RAGGED64=0 python /tmp/ev.py 2>/dev/null | tail -1)"
echo "hip ragged64=1 $(python /tmp/ev.py 2>/dev/null | tail -1)"
echo "mi2_64 ragged64=1 $(AYOLO_LIB=$(realpath ab/libayolo_mi2_64.so) python /tmp/ev.py 2>/dev/null | tail -1)"
echo "mi2_64 ragged64=0 $(AYOLO_GCONV_RAGGED64=0 AYOLO_LIB=$(realpath ab/libayolo_mi2_64.so) python /tmp/ev.py 2>/dev/null | tail -1)"
done
bash tools/ab_bench.sh ayolov2_amd/libayolo_hip.so ab/libayolo_mi2_64.so 2
bash tools/ddp_timeline.sh r02 | tail -32
