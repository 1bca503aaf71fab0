#!/usr/bin/env python3
"""Per-layer output comparison of the HIP model (eval, fp32) against the CPU oracle: finds the first diverging layer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_model import _pair
m, r = _pair("s", seed=3)
m.eval(); r.eval()
x = torch.rand(2, 3, 256, 320)
og, orf = {}, {}
for i, (a, b) in enumerate(zip(m.model, r.model)):
    a.register_forward_hook(lambda mod, inp, out, i=i: og.__setitem__(i, out))
    b.register_forward_hook(lambda mod, inp, out, i=i: orf.__setitem__(i, out))
with torch.no_grad():
    r(x); m(x.cuda())
for i in sorted(og):
    if i not in orf:
        continue
    a, b = og[i], orf[i]
    if isinstance(a, (tuple, list)):
        for k, (ra, rb) in enumerate(zip(a[1], b[1])):
            e = float((ra.float().cpu() - rb).abs().max() / (rb.abs().max() + 1e-12))
            print(f"   head raw level {k} shape {tuple(rb.shape)} rel err {e:.2e}")
        a, b = a[0], b[0]
    a = a.float().cpu()
    err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
    print(f"layer {i:2d} {type(m.model[i]).__name__:10s} shape {tuple(b.shape)}  rel err {err:.2e}", flush=True)
