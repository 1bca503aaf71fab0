#!/usr/bin/env python3
"""Transform on load, per layer (VERDICT r3 item 1 stage A: "measure it, do not cost it"): for every 1x1 / stride-1 conv of the
model, batch 64 at 640^2, the two-launch route  [BatchNorm + SiLU pass writes the activation] + [conv reads it]  against the
one-launch route  [conv reads the pre-activation z and forms the activation on the way to the MFMAs]  (ayolo_conv_fwd_xf), each
timed alone on the chip with HIP events.  usage: python tools/xf_sweep.py [model] [batch] [size]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import YOLOModel, ops, functional as F_  # noqa: E402
from ayolov2_amd.modules import Conv  # noqa: E402
from tools.conv_sweep import timeit  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "yolov5s"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 640
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
        c = mod.conv
        if c.kernel_size != (1, 1) or c.stride != (1, 1):
            continue
        seen.setdefault((c.in_channels, c.out_channels, xs[2]), [mod, 0])[1] += 1
    dt = torch.float16
    print(f"{'cin':>5}{'cout':>5}{'H':>5} cnt | {'bn+act us':>9} {'conv us':>8} {'sum':>8} | {'on-load us':>10} {'+store-back':>11} | {'saved us':>8} {'(store-back)':>12}  x cnt")
    tot2 = tot1 = tots = 0.0
    for (cin, cout, H), (mod, cnt) in sorted(seen.items(), key=lambda kv: -kv[0][2]):
        z = torch.randn((batch, cin, H, H), device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        a = ops.new_act(batch, cin, H, H, dt, dev)
        scale = (torch.rand(cin, device=dev) + 0.5).float()
        shift = torch.randn(cin, device=dev).float()
        w, _ = F_._WeightCache().get(mod.conv.weight, dt, cout, cin)
        y = ops.new_act(batch, cout, H, H, dt, dev)
        stats = torch.zeros((ops.STAT_REPS, 2 * cout), dtype=torch.float64, device=dev)
        d = ops.make_desc(dt, batch, H, H, cin, cin, cout, cout, (1, 1), (1, 1), (0, 0), H, H)
        t_a = timeit(lambda: ops.affine_act(z, a, scale, shift, 1), 10)
        t_c = timeit(lambda: ops.conv_fwd(d, a, w, y, 0, stats=stats), 10)
        t_x = timeit(lambda: ops.conv_fwd_xf(d, z, scale, shift, 1, w, y, 0, stats=stats), 10)
        t_s = timeit(lambda: ops.conv_fwd_xf(d, z, scale, shift, 1, w, y, 0, stats=stats, store=a), 10)
        print(f"{cin:5d}{cout:5d}{H:5d} {cnt:3d} | {t_a:9.1f} {t_c:8.1f} {t_a + t_c:8.1f} | {t_x:10.1f} {t_s:11.1f} | {t_a + t_c - t_x:8.1f} {t_a + t_c - t_s:12.1f}")
        tot2 += (t_a + t_c) * cnt
        tot1 += t_x * cnt
        tots += t_s * cnt
    print(f"all 1x1 layers of the model: two launches {tot2 / 1e3:.3f} ms, transform on load {tot1 / 1e3:.3f} ms, with store-back {tots / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
