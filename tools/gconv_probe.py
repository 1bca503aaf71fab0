#!/usr/bin/env python3
"""Where does a k_gconv step go?  Runs one conv forward through the probe build (ab/libayolo_probe.so: conv.hip compiled with
-DAYOLO_PROBE: `make -C ayolov2_amd/csrc probe`) and prints, per workgroup class, the s_memtime marks of the
first tile's step loop: [top | DMA wait done | barrier passed | MFMAs issued] per step, in shader cycles.
usage (GPU box): AYOLO_LIB=$PWD/ab/libayolo_probe.so python tools/gconv_probe.py B Cin Cout k s p H W [dgrad]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import _lib, functional as F_  # noqa: E402

N = 96


def main():
    B, Cin, Cout, k, s, p, H, W = (int(a) for a in sys.argv[1:9])
    x = torch.randn(B, Cin, H, W, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    cache = F_._WeightCache()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            y = F_.ConvFn.apply(x, w, (s, s), (p, p), cache)
        torch.cuda.synchronize()
        ev0.record()
        y = F_.ConvFn.apply(x, w, (s, s), (p, p), cache)
        ev1.record()
    torch.cuda.synchronize()
    if len(sys.argv) > 9 and sys.argv[9] == "wgrad":         # the weight-gradient kernel of the same layer (probe marks of k_wgrad)
        from ayolov2_amd import ops
        geo = F_._Geometry((B, Cin, H, W), w.shape, (s, s), (p, p), torch.float16)
        d = geo.desc(torch.float16, geo.Cin_k, Cout)
        dw = torch.zeros((Cout, geo.kdims[0] * geo.kdims[1] * geo.Cin_k), dtype=torch.float32, device="cuda")
        dy = torch.randn_like(y)
        for _ in range(3):
            ops.conv_wgrad(d, x, dy, dw)
        torch.cuda.synchronize()
        ev0.record()
        ops.conv_wgrad(d, x, dy, dw)
        ev1.record()
        torch.cuda.synchronize()
    print(f"launch {ev0.elapsed_time(ev1) * 1e3:.1f} us (with probe overhead)")
    lib = _lib.lib()
    buf = np.zeros(512 * N, dtype=np.uint64)
    lib.ayolo_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
    assert lib.ayolo_probe_read(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(512, N).astype(np.int64)
    live = t[:, 0] > 0
    print("workgroups recorded:", int(live.sum()))
    t0 = t[live, 0].min()
    for b in (0, 1, 8, int(live.sum()) // 2, int(live.sum()) - 1):
        r = t[b]
        if r[0] == 0:
            continue
        marks = r[2:N - 6]
        nstep = int((marks > 0).sum()) // 4
        print(f"--- workgroup {b}: prologue marks (k_gconv: tap table / rows / barrier; k_wgrad: N-4 = loop drained): +{r[N - 4] - r[0]}, +{r[N - 5] - r[0]}, +{r[N - 6] - r[0]}, "
              f"loop entered +{r[1] - r[0]}; end +{r[N - 1] - r[0]} cycles, {nstep} steps recorded; after last step -> end {r[N - 1] - marks[4 * nstep - 1]}")
        prev = r[1]
        for i in range(nstep):
            a, bq, c, d = marks[4 * i:4 * i + 4]
            print(f"  step {i:2d}: gap {a - prev:6d} | vmcnt wait {bq - a:6d} | barrier {c - bq:6d} | fetch+issue+mfma {d - c:6d}")
            prev = d
    # distribution over workgroups
    steps = []
    for b in np.nonzero(live)[0]:
        marks = t[b, 2:N - 6]
        n = int((marks > 0).sum()) // 4
        for i in range(n):
            a, bq, c, d = marks[4 * i:4 * i + 4]
            steps.append((i, bq - a, c - bq, d - c))
    steps = np.array(steps)
    print("mean per step index over all workgroups: wait / barrier / body")
    for i in range(int(steps[:, 0].max()) + 1):
        m = steps[steps[:, 0] == i]
        print(f"  step {i:2d}: {m[:, 1].mean():8.0f} {m[:, 2].mean():8.0f} {m[:, 3].mean():8.0f}   (n={len(m)})")
    # s_memtime is not comparable across workgroups; s_memrealtime (100 MHz, chip-wide) is
    rs, re_ = t[live, N - 3], t[live, N - 2]
    o = rs.min()
    print(f"realtime (10 ns ticks): starts +0..+{rs.max() - o}, ends +{re_.min() - o}..+{re_.max() - o}")
    print("  start offsets sorted (every 16th):", sorted(int(v - o) for v in rs)[::16])
    print("  end offsets sorted (every 16th):  ", sorted(int(v - o) for v in re_)[::16])
    print("  duration ticks: mean %.0f min %d max %d" % ((re_ - rs).mean(), (re_ - rs).min(), (re_ - rs).max()))
    dur = (t[live, N - 1] - t[live, 0])
    print(f"workgroup duration cycles: mean {dur.mean():.0f} min {dur.min()} max {dur.max()}; span of starts {t[live, 0].max() - t0}, last end +{t[live, N - 1].max() - t0}")


if __name__ == "__main__":
    main()
