#!/usr/bin/env python3
"""Model-level check of the BatchNorm-backward sums in the dgrad epilogue: the fp16 train step's parameter gradients with
plan.BN_REDUCE_IN_DGRAD on vs off (same weights, same input), next to the run-to-run noise of the off path (fp32 atomics).
usage (GPU box): python tools/bnr_check.py [model] [batch] [size]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ayolov2_amd import plan as P  # noqa: E402


def grads(model, loss_fn, imgs, targets, bnr):
    P.BN_REDUCE_IN_DGRAD = bnr
    model.__dict__.pop("_plans", None)
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss, _ = loss_fn(model(imgs), targets)
    loss.backward()
    plan = [p for p in model._plans.values() if p][0]
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}, plan.bn_in_dgrad


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov5s"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 320
    dev = torch.device("cuda", 0)
    model, _, _, loss_fn, _ = bench.build_train_objects(name, dev, 1)
    gen = torch.Generator().manual_seed(0)
    imgs = torch.rand(batch, 3, size, size, generator=gen).to(dev)
    targets = bench.synth_targets(batch, 8, gen).to(dev)
    # BN running statistics change with every forward; gradients do not depend on them
    l0, g0, n0 = grads(model, loss_fn, imgs, targets, False)
    l1, g1, n1 = grads(model, loss_fn, imgs, targets, False)
    l2, g2, n2 = grads(model, loss_fn, imgs, targets, True)
    print(f"loss off {l0:.6f} off {l1:.6f} on {l2:.6f}; BN layers folded into dgrad epilogues: {n0} / {n1} / {n2}")

    def cmp(a, b):
        worst = []
        for k in a:
            sc = float(a[k].abs().max()) + 1e-20
            worst.append((float((a[k] - b[k]).abs().max()) / sc, k))
        worst.sort(reverse=True)
        fa = torch.cat([v.flatten() for v in a.values()]); fb = torch.cat([v.flatten() for v in b.values()])
        cos = float(torch.dot(fa, fb) / (fa.norm() * fb.norm()))
        return cos, worst[:5]
    print("off vs off (noise):", cmp(g0, g1))
    print("off vs on         :", cmp(g0, g2))


if __name__ == "__main__":
    main()
