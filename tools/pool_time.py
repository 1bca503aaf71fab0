#!/usr/bin/env python3
"""SPPF max-pool forward / backward alone on the chip (YOLOv5s at batch 64: 256 channels on 20 x 20, channel slices of the
1024-wide concat buffer), HIP events.  k = 5 takes the strip kernels, k = 7 the per-pixel ones (the round-3 path, for reference)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import ops  # noqa: E402
from tools.conv_sweep import timeit  # noqa: E402

B, C, H = 64, 256, 20
cat = torch.randn(B, 4 * C, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last)
g = torch.randn_like(cat)
for k in (5, 7):
    x, y = cat[:, :C], cat[:, C:2 * C]
    _, arg = ops.maxpool_fwd(x, k, y)
    t_f = timeit(lambda: ops.maxpool_fwd(x, k, y), 20)
    t_b = timeit(lambda: ops.maxpool_bwd(arg, g[:, C:2 * C], k, g[:, :C], accumulate=True), 20)
    print(f"k={k}: forward {t_f:6.1f} us   backward {t_b:6.1f} us")
# the whole cascade in one launch per direction (ayolo_sppf_pool_fwd / _bwd; four channel groups per workgroup on this shape)
arg3 = ops.sppf_pool_fwd(cat, C)
t_f = timeit(lambda: ops.sppf_pool_fwd(cat, C), 20)
t_b = timeit(lambda: ops.sppf_pool_bwd(arg3, g, C), 20)
print(f"cascade (three pools): forward {t_f:6.1f} us   backward {t_b:6.1f} us")
