#!/usr/bin/env python3
"""Parity of the 256-channel x 256-pixel (8-wavefront) k_gconv tile: run with AYOLO_GCONV_TM=256 in the environment.
Forward / dgrad / wgrad of convs with >= 256 output or input channels against torch fp32 on fp16-rounded operands."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import functional as F_  # noqa: E402

shapes = [(2, 256, 256, 1, 1, 0, 20, 20), (3, 512, 512, 1, 1, 0, 20, 20), (2, 128, 320, 1, 1, 0, 24, 20), (2, 320, 256, 1, 1, 0, 24, 20),
          (2, 128, 256, 3, 2, 1, 40, 40), (2, 256, 512, 3, 2, 1, 20, 24), (1, 1024, 512, 1, 1, 0, 20, 20), (2, 64, 264, 1, 1, 0, 17, 13),
          (8, 256, 256, 1, 1, 0, 40, 40)]
bad = 0
for shape in shapes:
    B, Cin, Cout, k, s, p, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(B, Cin, H, W, generator=g)).half().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half().float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(yr.shape, generator=g).half().float()
    yr.backward(gy)
    xg = x.cuda().half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        yg = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
    yg.backward(gy.cuda().half())
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    ef, ed, ew = rel(yg, yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad)
    ok = max(ef, ed, ew) < 5e-3
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {shape} fwd {ef:.2e} dgrad {ed:.2e} wgrad {ew:.2e}", flush=True)
print("FAILED" if bad else "all ok")
