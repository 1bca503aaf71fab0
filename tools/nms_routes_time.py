import os, sys, time, json, torch
sys.path.insert(0, os.getcwd())
import bench
from ayolov2_amd import metrics as M
from ayolov2_amd.metrics import non_max_suppression
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B, N, nc, img = 8, 100800, 80, 1280
pred = torch.cat((torch.rand(B, N, 2, generator=g) * img, torch.rand(B, N, 2, generator=g) ** 3 * img / 2 + 2,
                  torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 - 9.5),
                  torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 4)), 2).to(dev)
for fast in (False, True, False):
    M.NMS_FAST = fast
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        non_max_suppression(pred, 0.001, 0.65, multi_label=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("fast" if fast else "staged", [round(t, 3) for t in ts])
