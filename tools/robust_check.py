#!/usr/bin/env python3
"""Robustness probes on the GPU: other model sizes through the plan, batch 1 / odd sizes, and the > 2 GiB batch-split path."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel, functional as F_
from ayolov2_amd.losses import ComputeLoss
HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
for name, B, H, W in [("m", 2, 128, 160), ("l", 1, 96, 96), ("x", 2, 64, 96), ("s", 1, 32, 32), ("n", 3, 160, 96)]:
    torch.manual_seed(0)
    m = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"yolov5{name}.yaml")).cuda().train()
    m.hyp, m.gr = dict(HYP), 1.0
    x = torch.rand(B, 3, H, W).cuda()
    t = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.3]])
    with torch.autocast("cuda", dtype=torch.float16):
        out = m(x)
        loss, items = ComputeLoss(m)(out, t.cuda())
    loss.backward()
    gn = sum(float(p.grad.float().norm()) for p in m.parameters())
    ok = all(torch.isfinite(p.grad).all() for p in m.parameters())
    print(f"yolov5{name} B={B} {H}x{W}: loss {float(loss):.4f} grad-norm-sum {gn:.3e} finite {bool(ok)} plan {'_plans' in m.__dict__ and any(v not in (None, False) for v in m.__dict__['_plans'].values())}", flush=True)
    del m
    torch.cuda.empty_cache()
# > 2 GiB activation (44 x 64 x 640 x 640 fp16 = 2.15 GiB): batch split inside the C ABI
B, C, H, W = 44, 64, 640, 640
x = torch.randn(B, C, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(32, C, 1, 1, device="cuda") / 8).contiguous(memory_format=torch.channels_last).requires_grad_(True)
print("x GiB", x.numel() * 2 / 2**30)
with torch.autocast("cuda", dtype=torch.float16):
    y = F_.ConvFn.apply(x, w, (1, 1), (0, 0), F_._WeightCache())
gy = torch.randn_like(y)
y.backward(gy)
torch.cuda.synchronize()
# reference in slices (fp32 matmul on the GPU)
err_f = err_d = 0.0
dw_ref = torch.zeros(32, C, device="cuda")
w2 = w.detach().half().float().view(32, C)
for b in range(0, B, 8):
    xs = x.detach()[b:b + 8].float()
    yr = torch.einsum("oc,bchw->bohw", w2, xs)
    err_f = max(err_f, float((y.detach()[b:b + 8].float() - yr).abs().max() / yr.abs().max()))
    gs = gy[b:b + 8].float()
    dxr = torch.einsum("oc,bohw->bchw", w2, gs)
    err_d = max(err_d, float((x.grad[b:b + 8].float() - dxr).abs().max() / dxr.abs().max()))
    dw_ref += torch.einsum("bohw,bchw->oc", gs, xs)
err_w = float((w.grad.view(32, C) - dw_ref).abs().max() / dw_ref.abs().max())
print(f"batch-split conv (2.1 GB input): fwd {err_f:.2e} dgrad {err_d:.2e} wgrad {err_w:.2e}")
