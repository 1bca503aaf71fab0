#!/usr/bin/env python3
"""Index algebra of two kernels planned for the next round, checked on the CPU against torch (run: python
tools/experiments/next_round_math.py).  Not product code: the point is to have the sharing identities verified before any HIP
is written ("patch-tile loader").

1. forward 3x3 / stride 2 / pad 1 (even W): per input row offset dh the taps dw = -1 / +1 read the ODD input columns
   2*ow - 1 / 2*ow + 1 -- one run of odd columns (row j <-> column 2*(ow0 + j - 1) + 1) serves both by a shift of one row, the
   lane with ow == 0 zeroes the dw = -1 fragment; the tap dw = 0 reads the even columns.  6 row-group loads per 32-channel
   chunk instead of 9 tap loads.
2. stem 6x6 / stride 2 / pad 2 on the image packed as pixel PAIRS of 4 channels (8 values = 16 bytes per pair): in pair space
   the kernel is 6 x 3 with stride (2, 1); for one dh the three horizontal taps of an output pixel are three CONSECUTIVE
   16-byte pair rows, i.e. one contiguous 48-byte window of a [pair][8] LDS image -- no im2col duplication, x is DMA'd once
   per dh (6 x (TP + 2) x 16 bytes per tile instead of 4.5 x 16 KiB).
"""
import numpy as np
import torch
import torch.nn.functional as F


def check_fwd_s2(B=2, Cin=3, Cout=4, H=6, W=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, None, 2, 1).permute(0, 2, 3, 1).reshape(-1, Cout).numpy()
    Ho, Wo = H // 2, W // 2
    P = B * Ho * Wo
    m = np.arange(P)
    ow, oh, n = m % Wo, (m // Wo) % Ho, m // (Wo * Ho)
    xn = x.numpy()
    out = np.zeros((P, Cout))
    loads = 0
    for dh in (-1, 0, 1):
        # odd-column run: row j <-> aligned output index mu = j - 1, input pixel (n, 2*oh + dh, 2*ow + 1)
        mu = np.arange(-1, P)
        ok = mu >= 0
        owr, ohr, nr = np.where(ok, mu % Wo, 0), np.where(ok, (mu // Wo) % Ho, 0), np.where(ok, mu // (Wo * Ho), 0)
        ih = 2 * ohr + dh
        ok &= (ih >= 0) & (ih < H)
        odd = np.zeros((P + 1, Cin))
        odd[ok] = xn[nr[ok], :, ih[ok], 2 * owr[ok] + 1]
        loads += 1
        f_p1 = odd[m + 1]                                      # dw = +1: own row
        f_m1 = odd[m].copy()                                   # dw = -1: the row before
        f_m1[ow == 0] = 0.0                                    # column -1 is padding; the row before is the previous image row
        out += f_p1 @ w[:, :, dh + 1, 2].numpy().T + f_m1 @ w[:, :, dh + 1, 0].numpy().T
        ih0 = 2 * oh + dh
        ok0 = (ih0 >= 0) & (ih0 < H)
        even = np.zeros((P, Cin))
        even[ok0] = xn[n[ok0], :, ih0[ok0], 2 * ow[ok0]]
        loads += 1
        out += even @ w[:, :, dh + 1, 1].numpy().T
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)
    return loads


def check_stem_pairs(B=2, Cout=5, H=12, W=16, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, 3, 6, 6, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, None, 2, 2).permute(0, 2, 3, 1).reshape(-1, Cout).numpy()
    Ho, Wo = H // 2, W // 2
    # image as pixel pairs of 4 channels: pair q = columns (2q, 2q + 1), 8 values [c0..c3 of col 2q | c0..c3 of col 2q + 1]
    xp = np.zeros((B, H, Wo, 8))
    xp[..., 0:3] = x.permute(0, 2, 3, 1).numpy()[:, :, 0::2]
    xp[..., 4:7] = x.permute(0, 2, 3, 1).numpy()[:, :, 1::2]
    # weights in the same packing: for dh and pair offset dq in {-1, 0, +1}: kernel columns j = 2*(dq + 1) + {0, 1}
    wp = np.zeros((Cout, 6, 3, 8))
    for dq in range(3):
        wp[:, :, dq, 0:3] = w[:, :, :, 2 * dq].permute(0, 2, 1).numpy()
        wp[:, :, dq, 4:7] = w[:, :, :, 2 * dq + 1].permute(0, 2, 1).numpy()
    out = np.zeros((B, Ho, Wo, Cout))
    for i in range(6):                                         # dh = i - 2, vertical stride 2
        rows = np.zeros((B, Ho, Wo + 2, 8))                    # one zero pair of halo on each side
        for oh in range(Ho):
            ih = 2 * oh + i - 2
            if 0 <= ih < H:
                rows[:, oh, 1:Wo + 1] = xp[:, ih]
        # the three horizontal taps of output pixel ow = pairs ow - 1, ow, ow + 1 = ONE contiguous 24-value window
        win = np.concatenate([rows[:, :, 0:Wo], rows[:, :, 1:Wo + 1], rows[:, :, 2:Wo + 2]], axis=-1)      # (B, Ho, Wo, 24)
        out += win @ wp[:, i].reshape(Cout, 24).T
    np.testing.assert_allclose(out.reshape(-1, Cout), ref, rtol=1e-12, atol=1e-12)


if __name__ == "__main__":
    print("forward 3x3 s2: row-group loads per chunk =", check_fwd_s2(), "(per-tap path: 9)")
    check_fwd_s2(1, 2, 3, 2, 2, 3)
    check_stem_pairs()
    check_stem_pairs(1, 3, 4, 4, 2)
    print("stem in pair space: one contiguous 3-pair window per (output pixel, dh) -- ok")
