#!/usr/bin/env python3
"""How much of a conv forward / BN-backward-reduce launch is the replicated-atomics tail?  (stats on/off, replica count)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ayolov2_amd import ops, functional as F_
from ayolov2_amd._lib import call
from tools.conv_sweep import timeit
dev = torch.device("cuda"); dt = torch.float16; B = 64
for (cin, cout, k, s, H) in [(64, 64, 1, 1, 160), (128, 128, 1, 1, 80), (128, 128, 3, 1, 40), (512, 256, 1, 1, 20), (256, 256, 1, 1, 20)]:
    geo = F_._Geometry((B, cin, H, H), (cout, cin, k, k), (s, s), (k // 2, k // 2), dt)
    x = torch.randn(B, cin, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w32 = torch.randn(cout, cin, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    w, wt = F_._WeightCache().get(w32, dt, cout, geo.cin_pad)
    y = ops.new_act(B, cout, geo.Ho, geo.Wo, dt, dev)
    d = geo.desc(dt, cin, cout)
    out = []
    for reps in (0, 8, 64):
        st = torch.zeros((max(reps, 1), 2 * cout), dtype=torch.float32, device=dev) if reps else None
        s_bak = ops.STAT_REPS
        ops.STAT_REPS = max(reps, 1)
        t = timeit(lambda: ops.conv_fwd(d, x, w, y, 0, stats=st), reps=10)
        ops.STAT_REPS = s_bak
        out.append(t)
    z = y; da = torch.randn_like(z)
    mean = torch.zeros(cout, device=dev); inv = torch.ones(cout, device=dev); sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    code = ops.dtype_code(dt); npix = B * geo.Ho * geo.Wo
    red = []
    for reps in (8, 64):
        sums = torch.zeros(reps, 2 * cout, device=dev)
        red.append(timeit(lambda: call("ayolo_bn_act_bwd_reduce", code, z.data_ptr(), cout, da.data_ptr(), cout, npix, cout, mean.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, sums.data_ptr(), reps, torch.cuda.current_stream().cuda_stream), reps=10))
    print(f"{cin:4d}->{cout:4d} k{k} H{H:4d} | conv fwd us: no-stats {out[0]:7.1f}  reps8 {out[1]:7.1f}  reps64 {out[2]:7.1f} | bn reduce us: reps8 {red[0]:7.1f} reps64 {red[1]:7.1f}")
