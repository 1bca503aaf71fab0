#!/usr/bin/env python3
"""cfg 5 forward leg, wall time per batch: YOLOv5x 1280x1280 batch 8 fuse().eval() fp16 through the inference executor."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel
torch.manual_seed(0)
m = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", "yolov5x.yaml")).cuda().fuse().eval()
x = torch.rand(8, 3, 1280, 1280, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); print("cfg5 ms", (time.perf_counter() - t) * 100)
