#!/usr/bin/env python3
"""Training-mode (batch-stat BN) fp32 forward of the HIP model vs the CPU oracle, module path, per layer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_model import _pair
m, r = _pair("n")
m.train(); r.train()
x = torch.rand(2, 3, 128, 160)
for use_plan in (False, True):
    m.use_plan = use_plan
    og, orf = {}, {}
    hs = []
    for i, (a, b) in enumerate(zip(m.model, r.model)):
        hs.append(a.register_forward_hook(lambda mod, inp, out, i=i: og.__setitem__(i, out)))
        hs.append(b.register_forward_hook(lambda mod, inp, out, i=i: orf.__setitem__(i, out)))
    with torch.no_grad() if False else torch.enable_grad():
        rr = r(x); gg = m(x.cuda())
    for h in hs:
        h.remove()
    print("use_plan", use_plan)
    for i in sorted(og):
        if i not in orf or isinstance(og[i], (tuple, list)):
            continue
        a, b = og[i].detach().float().cpu(), orf[i].detach()
        print(f"  layer {i:2d} {type(m.model[i]).__name__:10s} {tuple(b.shape)} rel err {float((a - b).abs().max() / (b.abs().max() + 1e-12)):.2e}")
    for k, (a, b) in enumerate(zip(gg, rr)):
        a = a.detach().float().cpu(); b = b.detach()
        print(f"  raw {k} {tuple(b.shape)} rel err {float((a - b).abs().max() / (b.abs().max() + 1e-12)):.2e}  max abs {float((a-b).abs().max()):.3e}")
