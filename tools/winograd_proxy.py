#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) verdict by proxy: in the transformed domain the layer is sixteen independent GEMMs
[pixels / 4 x Cin] x [Cin x Cout], i.e. ONE 1x1 conv over 4x the pixels (2.25x fewer MACs, 4x the activation rows on both sides) --
before any input / output transform.  Times the direct 3x3 kernel of a layer (inference epilogue: folded BatchNorm + SiLU) against
the 1x1 kernel at exactly that GEMM shape, alone on the chip.  If the GEMM part alone is not well below the direct kernel, no
Winograd kernel can win.  usage: python tools/winograd_proxy.py B Cin Cout H W [B Cin Cout H W ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import ops, functional as F_  # noqa: E402
from ayolov2_amd._lib import EPI_AFFINE_SILU  # noqa: E402
from tools.conv_sweep import timeit  # noqa: E402


def one(B, Cin, Cout, H, W):
    dt, dev = torch.float16, torch.device("cuda")
    res = []
    for k, hh in ((3, H), (1, 2 * H)):                    # 1x1 over 4x the pixels: (2H) x (2W) map
        ww = W if k == 3 else 2 * W
        x = torch.randn(B, Cin, hh, ww, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w32 = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
        w, _ = F_._WeightCache().get(w32, dt, Cout, Cin)
        y = ops.new_act(B, Cout, hh, ww, dt, dev)
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        d = ops.make_desc(dt, B, hh, ww, Cin, Cin, Cout, Cout, (k, k), (1, 1), (k // 2, k // 2), hh, ww)
        res.append(timeit(lambda: ops.conv_fwd(d, x, w, y, EPI_AFFINE_SILU, scale=sc, shift=sh), 10))
    fl = 2.0 * B * H * W * Cin * Cout * 9
    print(f"{B:3d} x {Cin:4d} -> {Cout:4d} @ {H:4d} x {W:4d}: direct 3x3 {res[0]:8.1f} us ({fl / res[0] / 1e6:6.0f} TF/s)   sixteen GEMMs as a 1x1 on "
          f"{2 * H} x {2 * W} {res[1]:8.1f} us   ratio {res[1] / res[0]:.2f}")


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    for i in range(0, len(a), 5):
        one(*a[i:i + 5])
