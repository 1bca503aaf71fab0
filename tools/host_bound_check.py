#!/usr/bin/env python3
"""Is the train step bound by the GPU or by the host that feeds it?  The bench's step with parts of the HOST work removed while the
GPU work stays the same: target assignment (loss_fn.prepare, numpy) done once instead of every step; ModelEMA's host bookkeeping
skipped (the kernel still runs through a cached table).  If the step time does not move, the GPU is the bottleneck."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ayolov2_amd.trainer import ModelEMA, training_step  # noqa: E402

dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
gen = torch.Generator().manual_seed(1234)
imgs = torch.rand(64, 3, 640, 640, generator=gen).to(dev)
tc = bench.synth_targets(64, 8, gen)
tg = tc.to(dev)
head = model.model[-1]
shapes = [(64, head.na, 640 // int(s), 640 // int(s), head.no) for s in head._strides_py]
ema = ModelEMA(model)
prep0 = loss_fn.prepare(tc, shapes, dev)


def run(label, reuse_prep, use_ema, n=60):
    def step():
        prep = prep0 if reuse_prep else loss_fn.prepare(tc, shapes, dev)
        return training_step(run_model, lambda p, t: loss_fn(p, t, prepared=prep), opt, scaler, imgs, tg, world_size=1, amp=True,
                             ema=ema if use_ema else None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(n):
        h0 = time.perf_counter()
        step()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    print(f"{label:44s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms/step   host time in step() {host / n * 1e3:6.2f} ms")


run("bench step (prepare every step, EMA)", False, True)
run("prepare once", True, True)
run("prepare once, no EMA", True, False)
run("prepare every step, no EMA", False, False)
run("bench step again", False, True)
