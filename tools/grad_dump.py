#!/usr/bin/env python3
"""Dump / compare the fp16 train step of YOLOv5s at 4 x 320^2 (the shapes of tests/test_gpu_infer.py) under different kernel
switches: a route that computes the same arithmetic as the default differs only by fp16 roundings flipped by the BatchNorm
statistics' summation order; a wrong tile shows as a layer whose gradient leaves the others' band.
  python tools/grad_dump.py dump /tmp/g_base.pt         (environment switches select the route)
  python tools/grad_dump.py cmp /tmp/g_base.pt /tmp/g_x.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dump(path, name="s", size=320, batch=4):
    from test_gpu_infer import HYP, _pair, _targets, _train_step
    m, _ = _pair(name, seed=26)
    m.hyp, m.gr, m.nc = dict(HYP), 1.0, 80
    m = m.cuda().train()
    x, t = torch.rand(batch, 3, size, size), _targets(batch, 27)
    acts = {}
    l16, raws, g = _train_step(m, x.cuda(), t.cuda(), amp=True)
    plan = [p for p in m._plans.values() if p][0]
    torch.cuda.synchronize()
    zs = [(f"{k:02d} C={P['co']:3d} npix={P['npix']}", P["z"].detach().float().cpu().clone()) for k, P in enumerate(plan._producers)]
    dzs = [(f"{k:02d} n={t.numel()}", t.detach().float().cpu().clone()) for k, t in enumerate(plan.dz_list)]
    bn = []
    for k, L in enumerate(plan._bn_layers[:12]):
        bn.append((f"{k:02d} da C={L['C']}", L["a"].grad().detach().float().cpu().clone()))
        bn.append((f"{k:02d} sums", L["sums"].detach().double().sum(0).float().cpu().clone() if L["sums"].dim() == 2 else
                   L["sums"].detach().double().view(L["R"], -1).sum(0).float().cpu().clone()))
        bn.append((f"{k:02d} mean|invstd", L["sm"].detach().float().cpu().clone()))
    torch.save(dict(loss=l16, raws=[r.cpu() for r in raws], grads={k: v.cpu() for k, v in g.items()}, zs=zs, dzs=dzs, bn=bn), path)
    print("loss", l16)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    print(f"loss {A['loss']:.6f} vs {B['loss']:.6f}")
    for k, (x, y) in enumerate(zip(A["raws"], B["raws"])):
        print(f"raw {k}: max diff {float((x - y).abs().max()):.3e} of range {float(x.max() - x.min()):.2f}; differing elements {int((x != y).sum())} / {x.numel()}")
    for what in ("zs", "dzs", "bn"):
        print(f"-- {what} (forward / backward order): differing elements, max diff / max, cosine")
        for (ka, x), (kb, y) in zip(A.get(what, []), B.get(what, [])):
            if x.numel() != y.numel():
                print(f"  {what} {ka}: sizes differ"); continue
            x, y = x.flatten().double(), y.flatten().double()
            print(f"  {what} {ka:28s} differ {float((x != y).double().mean()):.4f}  max diff {float((x - y).abs().max() / (x.abs().max() + 1e-300)):.2e}"
                  f"  cosine {float((x @ y) / (x.norm() * y.norm() + 1e-300)):.6f}")
    # the loss gradient w.r.t. the logits of either run, by torch autograd on the CPU oracle: is the loss itself this sensitive?
    try:
        from test_gpu_infer import HYP, _pair, _targets
        from ayolov2_amd.losses import ComputeLoss
        _, r = _pair("s", seed=26)
        r.hyp, r.gr, r.nc = dict(HYP), 1.0, 80
        t = _targets(A["raws"][0].shape[0], 27)
        gs = []
        for D in (A, B):
            raws = [x.clone().requires_grad_(True) for x in D["raws"]]
            loss, _ = ComputeLoss(r)(raws, t)
            loss.backward()
            gs.append([x.grad.flatten().double() for x in raws])
        for k, (x, y) in enumerate(zip(*gs)):
            print(f"loss gradient (torch autograd on the dumped logits) level {k}: cosine {float((x @ y) / (x.norm() * y.norm())):.6f}, "
                  f"max diff / max {float((x - y).abs().max() / x.abs().max()):.3e}")
    except Exception as e:                                   # noqa: BLE001
        print("loss-gradient check skipped:", repr(e))
    rows = []
    for k in A["grads"]:
        x, y = A["grads"][k].flatten().double(), B["grads"][k].flatten().double()
        cos = float((x @ y) / (x.norm() * y.norm() + 1e-300))
        rows.append((k, cos, float((x - y).abs().max() / (x.abs().max() + 1e-300))))
    fa = torch.cat([v.flatten().double() for v in A["grads"].values()])
    fb = torch.cat([v.flatten().double() for v in B["grads"].values()])
    print(f"whole-gradient cosine {float((fa @ fb) / (fa.norm() * fb.norm())):.6f}")
    for k, cos, rel in rows:
        if k.endswith("conv.weight") or ".conv." in k:
            print(f"  {k:40s} cosine {cos:.5f}  max diff / max {rel:.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], *(sys.argv[3:4] or ["s"]), *(int(v) for v in sys.argv[4:6]))
    else:
        cmp(sys.argv[2], sys.argv[3])
