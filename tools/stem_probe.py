#!/usr/bin/env python3
"""Phase times of k_stem_wgrad (probe build): realtime marks per workgroup -- start, first tile staged, tile loop done, LDS reduction
done, atomics done.  usage (GPU box): AYOLO_LIB=$PWD/ab/libayolo_probe.so python tools/stem_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ayolov2_amd import _lib, functional as F_, ops
B, H = 64, 640
geo = F_._Geometry((B, 3, H, H), (32, 3, 6, 6), (2, 2), (2, 2), torch.float16)
xk = torch.randn((B, geo.Cin_k, geo.H, geo.W), device="cuda").half().contiguous(memory_format=torch.channels_last)
dy = torch.randn((B, 32, geo.Ho, geo.Wo), device="cuda").half().contiguous(memory_format=torch.channels_last)
dw = torch.zeros((32, 144), dtype=torch.float32, device="cuda")
d = geo.desc(torch.float16, geo.Cin_k, 32)
for _ in range(3):
    ops.conv_wgrad(d, xk, dy, dw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv_wgrad(d, xk, dy, dw); e1.record(); torch.cuda.synchronize()
print("launch us", e0.elapsed_time(e1) * 1e3)
N = 96
buf = np.zeros(512 * N, dtype=np.uint64)
lib = _lib.lib()
lib.ayolo_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
assert lib.ayolo_probe_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(512, N)[:, :5].astype(np.int64)
o = t[:, 0].min()
names = ["start", "first tile staged", "loop done", "LDS reduce done", "end"]
for k in range(5):
    v = (t[:, k] - o) / 100.0
    print(f"{names[k]:20s} min {v.min():8.2f} us  median {np.median(v):8.2f}  max {v.max():8.2f}")
