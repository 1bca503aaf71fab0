#!/usr/bin/env python3
"""How well conditioned is the gradient the fp16 parity tests compare?  One train step of a random-init model in the exact-fp32
mode, re-run (a) unchanged, (b) with the input image perturbed by relative 1e-6 / 1e-5 / 3e-4, (c) in fp16 autocast at three
loss scales; prints the cosine of each gradient against the unperturbed fp32 one (whole model and a few layers).
usage: python tools/grad_conditioning.py {n,s,m,l,x} BATCH        (measured: DESIGN.md section 5)"""
import os, sys, copy, torch
sys.path.insert(0, os.getcwd())
import tests.test_gpu_infer as T
from tests.test_gpu_infer import _targets, HYP, CFG
from ayolov2_amd import YOLOModel
from ayolov2_amd.losses import ComputeLoss
name, batch = sys.argv[1], int(sys.argv[2])
torch.manual_seed(28)
m = YOLOModel(os.path.join(CFG, f"yolov5{name}.yaml")).cuda().train()
m.hyp, m.gr, m.nc = dict(HYP), 1.0, 80
x, t = torch.rand(batch, 3, 640, 640).cuda(), _targets(batch, 29).cuda()
sd = copy.deepcopy(m.state_dict())
def step(x, amp, scale):
    m.load_state_dict(sd)
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        raws = m(x)
        loss, _ = ComputeLoss(m)(raws, t)
    (loss * scale).backward()
    return float(loss), {k: p.grad.detach().float().clone() / scale for k, p in m.named_parameters()}
def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))
def report(tag, g, g32):
    ks = [k for k in g32 if g32[k].dim() == 4]
    glob = cos(torch.cat([g[k].flatten() for k in g32]), torch.cat([g32[k].flatten() for k in g32]))
    sel = ["model.0.conv.weight", "model.8.cv3.conv.weight", "model.17.cv3.conv.weight", "model.20.cv3.conv.weight", "model.23.cv1.conv.weight", "model.23.m.0.cv1.conv.weight", "model.23.cv3.conv.weight", "model.24.conv.2.weight"]
    print(tag, "glob %.4f" % glob, " ".join("%s:%.3f" % (k.split("model.")[1].replace(".conv.weight", ""), cos(g[k], g32[k])) for k in sel if k in g32))
l32, g32 = step(x, False, 1.0)
l32b, g32b = step(x, False, 1.0)
report("fp32 rerun", g32b, g32)
for eps in (1e-6, 1e-5, 3e-4):
    xp = x * (1 + eps * torch.randn_like(x))
    l, g = step(xp, False, 1.0)
    report("fp32 input perturbed %.0e (loss %.6f vs %.6f)" % (eps, l, l32), g, g32)
for sc in (64.0, 4096.0, 65536.0):
    l, g = step(x, True, sc)
    report("fp16 scale %g (loss %.6f)" % (sc, l), g, g32)
