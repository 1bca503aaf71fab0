#!/usr/bin/env python3
"""Per-layer timing of the conv kernels on the YOLOv5 shapes (HIP events on the current stream).
usage: python tools/conv_sweep.py [model] [batch] [size]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel, ops, functional as F_  # noqa: E402
from ayolov2_amd.modules import Conv  # noqa: E402


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov5s"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    dev = torch.device("cuda")
    model = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"{name}.yaml")).to(dev)
    shapes = []
    hs = [m.register_forward_hook(lambda mod, i, o: shapes.append((mod, tuple(i[0].shape)))) for m in model.modules()
          if isinstance(m, Conv)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.eval()
        model.use_plan = False          # per-module path: the forward hooks below see every Conv (the executor bypasses them)
        model(torch.rand(1, 3, size, size, device=dev))
    for h in hs:
        h.remove()
    seen = {}
    for mod, xs in shapes:
        conv = mod.conv
        key = (conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], xs[2])
        seen.setdefault(key, [mod, 0])[1] += 1
    dt = torch.float16
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totflop = 0.0
    print(f"{'cin':>5}{'cout':>5} k s {'H':>4} cnt | {'fwd us':>8} {'TF/s':>6} {'GB/s':>6} | {'dgrad us':>8} {'TF/s':>6} | {'wgrad us':>8} {'TF/s':>6}")
    for (cin, cout, k, s, H), (mod, cnt) in seen.items():
        conv = mod.conv
        geo = F_._Geometry((batch, cin, H, H), conv.weight.shape, (s, s), F_._pair_(conv.padding), dt)
        xk = torch.randn((batch, geo.Cin_k, geo.H, geo.W), device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w, wt = F_._WeightCache().get(conv.weight, dt, cout, geo.cin_pad)
        y = ops.new_act(batch, cout, geo.Ho, geo.Wo, dt, dev)
        y.normal_()
        stats = torch.zeros((ops.STAT_REPS, 2 * cout), dtype=torch.float64, device=dev)
        d = geo.desc(dt, geo.Cin_k, cout)
        flop = 2.0 * batch * geo.Ho * geo.Wo * cout * cin * k * k
        byts = 2.0 * (xk.numel() + y.numel())
        t_f = timeit(lambda: ops.conv_fwd(d, xk, w, y, 0, stats=stats))
        if geo.needs_pack:
            t_d = 0.0
        else:
            dx = ops.new_act(batch, cin, geo.H, geo.W, dt, dev)
            t_d = timeit(lambda: ops.conv_dgrad(geo.desc(dt, cin, cout), y, wt, dx))
        K = geo.kdims[0] * geo.kdims[1] * geo.Cin_k
        dw = torch.zeros((cout, K), dtype=torch.float32, device=dev)
        t_w = timeit(lambda: ops.conv_wgrad(d, xk, y, dw))
        print(f"{cin:5d}{cout:5d} {k} {s} {H:4d} {cnt:3d} | {t_f:8.1f} {flop / t_f / 1e6:6.1f} {byts / t_f / 1e3:6.0f} | {t_d:8.1f} "
              f"{(flop / t_d / 1e6 if t_d else 0):6.1f} | {t_w:8.1f} {flop / t_w / 1e6:6.1f}")
        tot["fwd"] += t_f * cnt
        tot["dgrad"] += t_d * cnt
        tot["wgrad"] += t_w * cnt
        totflop += flop * cnt
    for k_, v in tot.items():
        print(f"total {k_}: {v / 1e3:.2f} ms  ({totflop / v / 1e6:.1f} TF/s)")


if __name__ == "__main__":
    main()
