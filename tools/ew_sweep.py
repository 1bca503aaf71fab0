#!/usr/bin/env python3
"""Achieved HBM bandwidth of the elementwise kernels on the YOLOv5s activation shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ayolov2_amd import ops
from tools.conv_sweep import timeit
dev = torch.device("cuda"); dt = torch.float16; B = 64
shapes = [(32, 320), (64, 160), (32, 160), (128, 80), (64, 80), (256, 40), (128, 40), (512, 20), (256, 20)]
print(f"{'C':>5}{'H':>5} | {'affine us':>9} {'GB/s':>6} | {'bwd_red us':>10} {'GB/s':>6} | {'bwd_app us':>10} {'GB/s':>6} | copy2d GB/s")
for C, H in shapes:
    z = torch.randn(B, C, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    a = torch.empty_like(z); da = torch.randn_like(z); dz = torch.empty_like(z)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
    mean = torch.randn(C, device=dev) * 0.1; inv = torch.rand(C, device=dev) + 0.5
    n = z.numel() * 2
    t1 = timeit(lambda: ops.affine_act(z, a, sc, sh, 1))
    from ayolov2_amd._lib import call
    sums = torch.zeros(ops.STAT_REPS, 2 * C, dtype=torch.float64, device=dev)
    code = ops.dtype_code(dt); npix = B * H * H
    t2 = timeit(lambda: call("ayolo_bn_act_bwd_reduce", code, z.data_ptr(), C, da.data_ptr(), C, npix, C, mean.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, sums.data_ptr(), ops.STAT_REPS, torch.cuda.current_stream().cuda_stream))
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    t3 = timeit(lambda: call("ayolo_bn_act_bwd_apply", code, z.data_ptr(), C, da.data_ptr(), C, dz.data_ptr(), C, npix, C, mean.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, sums.data_ptr(), ops.STAT_REPS, dg.data_ptr(), db.data_ptr(), 1.0, torch.cuda.current_stream().cuda_stream))
    t4 = timeit(lambda: ops.copy2d(z, a))
    print(f"{C:5d}{H:5d} | {t1:9.1f} {2*n/t1/1e3:6.0f} | {t2:10.1f} {2*n/t2/1e3:6.0f} | {t3:10.1f} {3*n/t3/1e3:6.0f} | {2*n/t4/1e3:6.0f}")
