#!/usr/bin/env python3
"""Which torch ops (copies, fills, tiny elementwise kernels) does one train step still issue around the executor's two C
calls?  torch.profiler over 3 steps of bench.py's step, grouped by (op, Python call site)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ayolov2_amd.trainer import ModelEMA, training_step  # noqa: E402


def main():
    dev = torch.device("cuda")
    model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
    gen = torch.Generator().manual_seed(1234)
    imgs = torch.rand(64, 3, 640, 640, generator=gen).to(dev)
    tc = bench.synth_targets(64, 8, gen)
    tg = tc.to(dev)
    head = model.model[-1]
    shapes = [(64, head.na, 640 // int(s), 640 // int(s), head.no) for s in head._strides_py]
    ema = ModelEMA(model)

    def step():
        prep = loss_fn.prepare(tc, shapes, dev)
        return training_step(run_model, lambda p, t: loss_fn(p, t, prepared=prep), opt, scaler, imgs, tg, world_size=1, amp=True, ema=ema)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.name in (
                "aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::mul", "aten::mul_", "aten::clone", "aten::cat",
                "aten::_foreach_add_", "aten::_amp_foreach_non_finite_check_and_unscale_", "aten::_amp_update_scale_", "aten::to",
                "aten::zeros", "aten::full_like", "aten::empty_like", "aten::sum", "aten::stack", "aten::detach", "aten::div", "aten::reciprocal"):
            site = "?"
            for fr in (e.stack or []):
                if "ayolov2_amd" in fr or "bench.py" in fr or "grad_scaler" in fr:
                    site = fr.strip()[-70:]
                    break
            cnt[(e.name, site)] += 1
    for (name, site), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
        print(f"{n / 3:7.1f} per step  {name:48s} {site}")
    kc = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            kc[e.name[:70]] += 1
    print("--- device activities per step")
    for name, n in sorted(kc.items(), key=lambda kv: -kv[1])[:14]:
        print(f"{n / 3:7.1f}  {name}")


if __name__ == "__main__":
    main()
