#!/usr/bin/env python3
"""The 3x3 / stride-2 layers of YOLOv5s at batch 64 (the family furthest above its HBM time): forward, dgrad, weight gradient alone
on the chip.  AYOLO_DGRAD_S2_MAXC selects which of them take k_dgrad_s2 (run once per value)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import ops, functional as F_  # noqa: E402
from tools.conv_sweep import timeit  # noqa: E402

dt, dev, B = torch.float16, "cuda", 64
print("AYOLO_DGRAD_S2_MAXC =", os.environ.get("AYOLO_DGRAD_S2_MAXC", "64 (default)"))
for cin, cout, H in ((32, 64, 320), (64, 128, 160), (128, 256, 80), (128, 128, 80), (256, 512, 40), (256, 256, 40)):
    Ho = H // 2
    x = torch.randn(B, cin, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, Ho, Ho, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, 3, 3, cin, device=dev) / (9 * cin) ** 0.5).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    d = ops.make_desc(dt, B, H, H, cin, cin, cout, cout, (3, 3), (2, 2), (1, 1), Ho, Ho)
    y = ops.new_act(B, cout, Ho, Ho, dt, dev)
    dx = ops.new_act(B, cin, H, H, dt, dev)
    stats = torch.zeros((ops.STAT_REPS, 2 * cout), dtype=torch.float64, device=dev)
    dw = torch.zeros((cout, 9 * cin), dtype=torch.float32, device=dev)
    t_f = timeit(lambda: ops.conv_fwd(d, x, w, y, 0, stats=stats), 10)
    t_d = timeit(lambda: ops.conv_dgrad(d, dy, wt, dx), 10)
    t_w = timeit(lambda: ops.conv_wgrad(d, x, dy, dw), 10)
    byt = 2.0 * (x.numel() + dy.numel())
    print(f"{cin:4d}->{cout:4d} {H:3d}->{Ho:3d} | fwd {t_f:6.1f} us ({byt / t_f / 1e6:5.2f} TB/s) | dgrad {t_d:6.1f} us ({byt / t_d / 1e6:5.2f} TB/s) | "
          f"wgrad {t_w:6.1f} us | HBM time at 4.5 TB/s {byt / 4.5e6:6.1f} us")
