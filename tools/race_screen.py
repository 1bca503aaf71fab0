#!/usr/bin/env python3
"""Race screen of the fp16 conv kernels: forward and dgrad have no atomics in their data path, so repeated launches on the
same operands must be BIT-identical; a mismatch means an LDS-DMA / barrier race (rare wrong tiles).
usage (GPU box): python tools/race_screen.py [model] [batch] [size] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel, ops, functional as F_  # noqa: E402
from ayolov2_amd.modules import Conv  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov5s"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    dev = torch.device("cuda")
    model = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"{name}.yaml")).to(dev)
    shapes = []
    hs = [m.register_forward_hook(lambda mod, i, o: shapes.append((mod, tuple(i[0].shape)))) for m in model.modules() if isinstance(m, Conv)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.eval()
        model.use_plan = False
        model(torch.rand(1, 3, size, size, device=dev))
    for h in hs:
        h.remove()
    seen = {}
    for mod, xs in shapes:
        conv = mod.conv
        seen.setdefault((conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], xs[2]), mod)
    dt = torch.float16
    bad = 0
    # a second stream keeps the chip busy with copies while the screen runs (races show under load)
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for (cin, cout, k, s, H), mod in seen.items():
        conv = mod.conv
        geo = F_._Geometry((batch, cin, H, H), conv.weight.shape, (s, s), F_._pair_(conv.padding), dt)
        xk = torch.randn((batch, geo.Cin_k, geo.H, geo.W), device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w, wt = F_._WeightCache().get(conv.weight, dt, cout, geo.cin_pad)
        d = geo.desc(dt, geo.Cin_k, cout)
        ys, dxs = [], []
        dy = ops.new_act(batch, cout, geo.Ho, geo.Wo, dt, dev)
        dy.normal_()
        for r in range(reps):
            with torch.cuda.stream(side):
                junk[:128 << 20].copy_(junk[128 << 20:])
            y = ops.new_act(batch, cout, geo.Ho, geo.Wo, dt, dev)
            stats = torch.zeros((ops.STAT_REPS, 2 * cout), dtype=torch.float64, device=dev)
            ops.conv_fwd(d, xk, w, y, 0, stats=stats)
            ys.append(y)
            if not geo.needs_pack:
                dx = ops.new_act(batch, cin, geo.H, geo.W, dt, dev)
                ops.conv_dgrad(geo.desc(dt, cin, cout), dy, wt, dx)
                dxs.append(dx)
        torch.cuda.synchronize()
        nf = sum(int((ys[0] != t).sum()) for t in ys[1:])
        nd = sum(int((dxs[0] != t).sum()) for t in dxs[1:]) if dxs else 0
        bad += nf + nd
        print(f"{cin:5d}->{cout:5d} k{k} s{s} {H:4d}: fwd mismatching elements {nf}, dgrad {nd}", flush=True)
    print("RACE SCREEN:", "clean" if bad == 0 else f"{bad} mismatching elements")


if __name__ == "__main__":
    main()
