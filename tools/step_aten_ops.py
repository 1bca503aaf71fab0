#!/usr/bin/env python3
"""Which torch (aten) ops does one train step still launch next to the executor's own kernels, and from where?
torch.profiler over 3 warm steps of the bench's step, grouped by Python stack.  usage (GPU box): python tools/step_aten_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    batch, size = 64, 640
    model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
    gen = torch.Generator().manual_seed(0)
    imgs = torch.rand(batch, 3, size, size, generator=gen).to(dev)
    tc = bench.synth_targets(batch, 8, gen)
    targets = tc.to(dev)
    head = model.model[-1]
    shapes = [(batch, head.na, size // int(s), size // int(s), head.no) for s in head._strides_py]
    from ayolov2_amd.trainer import ModelEMA, training_step
    ema = ModelEMA(model)

    def step():
        prep = loss_fn.prepare(tc, shapes, dev)
        return training_step(run_model, lambda pred, tg: loss_fn(pred, tg, prepared=prep), opt, scaler, imgs, targets,
                             world_size=1, amp=True, ema=ema)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    ev = prof.key_averages(group_by_stack_n=8)
    rows = [e for e in ev if e.key.startswith("aten::") and e.device_time_total > 0]
    rows.sort(key=lambda e: -e.count)
    print(f"aten ops with device time, per step (of {n} steps)")
    for e in rows[:60]:
        stack = [s for s in e.stack if "ayolov2_amd" in s or "bench.py" in s or "torch/amp" in s or "grad_scaler" in s][:3]
        print(f"{e.count / n:7.1f} x {e.key:28s} dev {e.device_time_total / n:8.1f} us  cpu {e.cpu_time_total / n:8.1f} us  {' <- '.join(s.strip()[-90:] for s in stack)}")
    print("---- device kernels by name (per step)")
    ks = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    ks.sort(key=lambda e: -e.count)
    for e in ks[:25]:
        print(f"{e.count / n:7.1f} x {e.key[:90]:90s} {e.device_time_total / n:8.1f} us")


if __name__ == "__main__":
    main()
