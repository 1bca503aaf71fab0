"""k_candidates back-to-back (no host syncs in between): steady-state duration and HBM rate of the candidate filter on the
bench's NMS workload (8 x 100800 x 85 fp32).  Usage: python tools/cand_bench.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ayolov2_amd import _lib                                   # noqa: E402
from ayolov2_amd.ops import call, _stream                      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(0)
B, N, nc, img = 8, 100800, 80, 1280
pred = torch.cat((torch.rand(B, N, 2, generator=g) * img, torch.rand(B, N, 2, generator=g) ** 3 * img / 2 + 2,
                  torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 - 9.5),
                  torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 4)), 2).cuda()
cap = B * N * 2
det = torch.empty((cap, 6), dtype=torch.float32, device="cuda")
keys = torch.empty(cap, dtype=torch.int64, device="cuda")
counters = torch.zeros(1 + B, dtype=torch.int32, device="cuda")


def launch():
    call("ayolo_nms_candidates", pred.data_ptr(), B, N, nc + 5, float(np.float32(0.001)), 1, 1, None, None, N, det.data_ptr(),
         keys.data_ptr(), counters.data_ptr(), cap, 0, _stream())


for _ in range(5):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("k_candidates back-to-back: %.4f ms  %.2f TB/s (%d hits/launch)" % (ms, pred.numel() * 4 / ms / 1e9,
                                                                          int(counters[0].item()) // (reps + 5)))
