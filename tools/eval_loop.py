#!/usr/bin/env python3
"""cfg 5 forward leg in a loop (for rocprofv3 --kernel-trace --stats): YOLOv5x 1280x1280 batch 8 fuse().eval() fp16 through
the inference executor.  Every kernel in the trace should be one of this repo's (k_gconv, k_pack_input, k_maxpool_fwd,
k_upsample_fwd, k_head_decode) -- no torch glue."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel
name, size, batch = (sys.argv[1:] + ["yolov5x", "1280", "8"])[:3]
torch.manual_seed(0)
m = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"{name}.yaml")).cuda().fuse().eval()
x = torch.rand(int(batch), 3, int(size), int(size), device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    torch.cuda.cudart().cudaProfilerStart() if hasattr(torch.cuda, "cudart") else None
    for _ in range(10):
        m(x)
    torch.cuda.synchronize()
print("done")
