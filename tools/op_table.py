#!/usr/bin/env python3
"""Per-op table of one YOLOv5 train step on the plan executor: every launch of the forward / backward list with its
in-situ duration (ayolo_run_ops_timed: HIP events on the op's own stream), algorithmic bytes and FLOP, GB/s and TF/s.
usage (GPU box): python tools/op_table.py [model] [batch] [size] > gpurun_out/op_table.txt
Environment switches of the executor apply (AYOLO_WGRAD_STREAM=0 gives every op the chip to itself)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ayolov2_amd import plan as P  # noqa: E402

NAMES = {v: k for k, v in vars(P).items() if k.startswith("OP_") and isinstance(v, int) and k not in ("OP_SIDE",)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov5s"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    dev = torch.device("cuda", 0)
    model, run_model, opt, loss_fn, scaler = bench.build_train_objects(name, dev, 1)
    gen = torch.Generator().manual_seed(0)
    imgs = torch.rand(batch, 3, size, size, generator=gen).to(dev)
    tc = bench.synth_targets(batch, 8, gen)
    targets = tc.to(dev)
    head = model.model[-1]
    shapes = [(batch, head.na, size // int(s), size // int(s), head.no) for s in head._strides_py]
    from ayolov2_amd.trainer import training_step

    def step():
        prep = loss_fn.prepare(tc, shapes, dev)
        return training_step(run_model, lambda pred, tg: loss_fn(pred, tg, prepared=prep), opt, scaler, imgs, targets,
                             world_size=1, amp=True, ema=None)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    plan = [p for p in model._plans.values() if p][0]
    plan.collect_times, plan.op_times = True, {}
    reps = 4
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    plan.collect_times = False
    tot = {}
    for what in ("forward", "backward"):
        t = np.mean(np.stack(plan.op_times[what]), axis=0)
        ops = plan.fwd if what == "forward" else plan.bwd
        print(f"---- {what}: {len(ops)} ops, sum {t.sum():.3f} ms")
        for k, (o, (fam, byts, flop), ms) in enumerate(zip(ops, plan.op_costs(what), t)):
            d = o.conv
            kind = o.kind & 0xff
            side = "S" if o.kind & P.OP_SIDE else " "
            shape = ""
            if kind in (P.OP_CONV_FWD, P.OP_CONV_DGRAD, P.OP_CONV_WGRAD):
                shape = f"{d.Cin:4d}->{d.Cout:4d} k{d.kh} s{d.sh} {d.H:3d}->{d.Ho:3d}"
            elif kind == P.OP_WGRAD_GROUP and what == "backward":
                shape = f"{plan.wgroup_costs[k][2]} layers in one grouped launch"
            elif kind in (P.OP_BN_TRAIN_ACT, P.OP_BN_BWD_REDUCE):
                shape = f"C={o.i[3]:4d} npix={o.l[0]}"
            elif kind == P.OP_BN_BWD_APPLY:
                shape = f"C={o.i[4]:4d} npix={o.l[0]}"
            elif kind == P.OP_BN_BWD_APPLY2:
                shape = f"C={o.i[5]:3d}|{o.i[6]:3d} npix={o.l[0]}"
            gbs = byts / ms / 1e6 if ms > 0 else 0.0
            tfs = flop / ms / 1e9 if ms > 0 else 0.0
            print(f"{k:4d} {side} {str(NAMES.get(kind, kind)):18s} {shape:34s} {ms * 1e3:8.1f} us {byts / 1e6:8.1f} MB {gbs:7.0f} GB/s {tfs:6.0f} TF/s")
            f = tot.setdefault(fam, [0.0, 0.0, 0])
            f[0] += ms; f[1] += byts; f[2] += 1
    print("---- families")
    for fam, (ms, byts, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"{fam:16s} {n:4d} launches {ms:7.3f} ms {byts / 1e9:7.3f} GB {byts / ms / 1e6 if ms else 0:7.0f} GB/s")


if __name__ == "__main__":
    main()
