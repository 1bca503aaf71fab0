#!/usr/bin/env python3
"""NMS leg of the benchmark alone (bench.nms_extra): ms per 8 x 100 800 x 85 batch, fast route on / off.
usage (GPU box): python tools/nms_time.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ayolov2_amd import metrics as M  # noqa: E402

dev = torch.device("cuda", 0)
for fast in (True, False, True):
    M.NMS_FAST = fast
    r = bench.nms_extra(dev)
    print("fast" if fast else "general", json.dumps({k: r[k] for k in ("nms_ms_per_batch", "nms_boxes_per_s", "nms_candidates", "fixed_nms_ms_per_batch")}))
