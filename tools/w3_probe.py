#!/usr/bin/env python3
"""Where does a k_wgrad3 step go?  Runs one 3x3 weight gradient through the probe build (`make -C ayolov2_amd/csrc probe` ->
ab/libayolo_probe.so) and prints, per workgroup, the s_memtime marks of its step loop: [DMA wait | barrier | issue of the next
step | sub-steps] per step, in shader cycles.
usage (GPU box): AYOLO_LIB=$PWD/ab/libayolo_probe.so python tools/w3_probe.py B Cin Cout s H W"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ayolov2_amd import _lib, ops, functional as F_  # noqa: E402

N = 128


def main():
    B, Cin, Cout, s, H, W = (int(a) for a in sys.argv[1:7])
    dt = torch.float16
    geo = F_._Geometry((B, Cin, H, W), (Cout, Cin, 3, 3), (s, s), (1, 1), dt)
    x = torch.randn(B, Cin, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, geo.Ho, geo.Wo, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    d = geo.desc(dt, Cin, Cout)
    dw = torch.zeros((Cout, 9 * Cin), dtype=torch.float32, device="cuda")
    g = (ctypes.c_int64 * 24)()
    _lib.check(_lib.lib().ayolo_wgrad3_geometry(d, g, 24), "geometry")
    names = "TC RPS PX nsub strips NB CB NP SL tn tc nrows ppr rowpitch plo ple xstage stage UP XP NU x_bytes y_bytes lds".split()
    print({k: int(v) for k, v in zip(names, g)})
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.conv_wgrad(d, x, dy, dw)
    torch.cuda.synchronize()
    ev0.record()
    ops.conv_wgrad(d, x, dy, dw)
    ev1.record()
    torch.cuda.synchronize()
    print(f"launch + reduction {ev0.elapsed_time(ev1) * 1e3:.1f} us (with probe overhead)")
    lib = _lib.lib()
    buf = np.zeros(512 * N, dtype=np.uint64)
    lib.ayolo_probe3_read.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
    assert lib.ayolo_probe3_read(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(512, N).astype(np.int64)
    live = t[:, 0] > 0
    print("workgroups recorded:", int(live.sum()))
    rows = []
    for b in np.nonzero(live)[0]:
        r = t[b]
        nm = int(r[N - 2])
        nst = (min(nm, N - 2) - 3) // 4
        steps = [(r[3 + 4 * i] - (r[2 + 4 * i] if i else r[2]), r[4 + 4 * i] - r[3 + 4 * i], r[5 + 4 * i] - r[4 + 4 * i], r[6 + 4 * i] - r[5 + 4 * i])
                 for i in range(nst)]
        rows.append((b, r[1] - r[0], r[2] - r[1], steps, r[N - 1] - r[3 + 4 * nst - 1] if nst else 0, r[N - 1] - r[0], nm))
    for b, pro, iss0, steps, epi, life, nm in rows[:2] + rows[len(rows) // 2:len(rows) // 2 + 1]:
        print(f"--- workgroup {b}: prologue {pro}, first issue {iss0}, {len(steps)} steps recorded ({nm} marks), epilogue (reduce + store) {epi}, life {life} cycles")
        for i, (w, ba, iss, cmp_) in enumerate(steps[:12]):
            print(f"  step {i:2d}: dma wait {w:6d} | barrier {ba:6d} | issue next {iss:6d} | sub-steps {cmp_:6d}")
    allsteps = np.array([st for _, _, _, steps, _, _, _ in rows for st in steps[1:-1]])
    if len(allsteps):
        m = allsteps.mean(0)
        print(f"mean over {len(allsteps)} inner steps: dma wait {m[0]:.0f} | barrier {m[1]:.0f} | issue next {m[2]:.0f} | sub-steps {m[3]:.0f}   (sum {m.sum():.0f})")
    ep = np.array([(t[b][N - 6] - t[b][3 + 4 * ((min(int(t[b][N - 2]), N - 2) - 3) // 4) - 1], t[b][N - 5] - t[b][N - 6], t[b][N - 4] - t[b][N - 5],
                    t[b][N - 1] - t[b][N - 4]) for b in np.nonzero(live)[0] if t[b][N - 6] > 0])
    if len(ep):
        m = ep.mean(0)
        print(f"epilogue parts (mean): last step -> final barrier {m[0]:.0f} | slice reduction through LDS {m[1]:.0f} | stores issued {m[2]:.0f} | stores landed {m[3]:.0f}")
    print("mean prologue %.0f, first issue %.0f, epilogue %.0f, life %.0f" % (np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows]),
                                                                           np.mean([r[4] for r in rows]), np.mean([r[5] for r in rows])))


if __name__ == "__main__":
    main()
