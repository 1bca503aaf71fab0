#!/usr/bin/env python3
"""Quick conv kernel check: every (shape, dtype) of tests/test_gpu_conv.py without stopping at the first failure."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import functional as F_  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_conv import SHAPES  # noqa: E402

extra = [(1, 32, 32, 1, 1, 0, 16, 16), (1, 32, 128, 1, 1, 0, 16, 16), (1, 64, 64, 1, 1, 0, 16, 16), (4, 8, 32, 3, 1, 1, 32, 32),
         (2, 96, 96, 1, 1, 0, 20, 20), (64, 64, 64, 1, 1, 0, 80, 80)]
for dt in (torch.float16, torch.float32):
    for shape in SHAPES + extra:
        B, Cin, Cout, k, s, p, H, W = shape
        g = torch.Generator().manual_seed(sum(shape))
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        if dt == torch.float16:
            x, w = x.half().float(), w.half().float()
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, s, p)
        gy = torch.randn(yr.shape, generator=g)
        if dt == torch.float16:
            gy = gy.half().float()
        yr.backward(gy)
        xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wg = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        if dt == torch.float16:
            with torch.autocast("cuda", dtype=torch.float16):
                yg = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
        else:
            yg = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
        yg.backward(gy.cuda().to(dt))
        torch.cuda.synchronize()
        def rel(a, b):
            return float((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12))
        ef, ed, ew = rel(yg, yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad)
        tol = 1e-4 if dt == torch.float32 else 2e-2
        flag = "ok " if max(ef, ed, ew) < tol else "BAD"
        print(f"{flag} {str(dt)[6:]:8s} {shape}  fwd {ef:.2e} dgrad {ed:.2e} wgrad {ew:.2e}", flush=True)
