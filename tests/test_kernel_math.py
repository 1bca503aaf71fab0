"""CPU restatements of the index algebra behind two conv kernels (csrc/conv.hip), checked against torch on small cases --
executable documentation of WHY the kernels may share LDS rows between taps; the kernels themselves are tested on the GPU
(tests/test_gpu_conv.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _flat_rows(x):
    """(B, C, H, W) -> [B*H*W][C] in the kernels' flattened (n, h, w) pixel order."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).numpy()


@pytest.mark.parametrize("shape", [(2, 5, 7, 6, 9), (1, 3, 4, 1, 5), (3, 4, 8, 5, 1), (2, 6, 6, 2, 2)])
def test_row_sharing_identity_3x3_stride1(shape):
    """k_gconv3: in flattened pixel order the input pixel of output pixel m under tap (dh, dw) is m + dh*W + dw, so one run
    of input rows per dh serves the three dw taps by a shift of one row; what the shift cannot express is the zero padding
    at the left / right border (the kernel zeroes those lanes' fragments) and the top / bottom border (the loader's
    out-of-range offset: the row index leaves the image)."""
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    ref = _flat_rows(F.conv2d(x, w, None, 1, 1))
    xr = _flat_rows(x)
    P = B * H * W
    m = np.arange(P)
    ow = m % W
    oh = (m // W) % H
    out = np.zeros((P, Cout))
    for dh in (-1, 0, 1):
        # the loader: row j of the run <-> flattened pixel m0 + j - 1 shifted by dh image rows; zero when oh + dh leaves the image
        run = np.zeros((P + 2, Cin))
        mu = np.arange(-1, P + 1)
        ok = (mu >= 0) & (mu < P)
        ohr = np.where(ok, (mu // W) % H, 0)
        ok &= (ohr + dh >= 0) & (ohr + dh < H)
        src = np.clip(mu + dh * W, 0, P - 1)
        run[ok] = xr[src[ok]]
        for dw in (-1, 0, 1):
            frag = run[m + 1 + dw].copy()                       # LDS rows p, p + 1, p + 2
            if dw == -1:
                frag[ow == 0] = 0.0                              # m - 1 is the previous image row's last pixel
            if dw == 1:
                frag[ow == W - 1] = 0.0
            out += frag @ w[:, :, dh + 1, dw + 1].numpy().T
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("shape", [(2, 3, 5, 8, 6), (1, 4, 2, 2, 2), (2, 2, 3, 6, 10)])
def test_four_shift_decomposition_of_stride2_dgrad(shape):
    """k_dgrad_s2: for a 3x3 / stride 2 / pad 1 conv on an even-sized map, dx[2*oh + a, 2*ow + b] of residue class (a, b)
    sums the taps (i, j) with a + 1 - i and b + 1 - j even, each reading dy at (oh + dh', ow + dw') with dh' = (a + 1 - i) / 2,
    dw' = (b + 1 - j) / 2 in {0, 1}: nine (class, tap) products over FOUR shifts of the dy tile.  The kernel's product order
    [A: shift (0,0) x classes 0,1,2,3 | B: (0,1) x 1,3 | C: (1,0) x 2,3 | D: (1,1) x 3] with weight taps
    {4,5,7,8 | 3,6 | 1,2 | 0} is exactly that enumeration."""
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, None, 2, 1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    Ho, Wo = H // 2, W // 2
    # enumeration from the definition
    products = []
    for a in (0, 1):
        for b in (0, 1):
            for i in range(3):
                for j in range(3):
                    if (a + 1 - i) % 2 == 0 and (b + 1 - j) % 2 == 0:
                        products.append(((a + 1 - i) // 2, (b + 1 - j) // 2, a * 2 + b, i * 3 + j))
    table = [(0, 0, 0, 4), (0, 0, 1, 5), (0, 0, 2, 7), (0, 0, 3, 8), (0, 1, 1, 3), (0, 1, 3, 6), (1, 0, 2, 1), (1, 0, 3, 2), (1, 1, 3, 0)]
    assert sorted(products) == sorted(table)
    assert all(s in ((0, 0), (0, 1), (1, 0), (1, 1)) for s in {(p[0], p[1]) for p in products})
    # and the sum it stands for
    dy = gy.numpy()
    dx = np.zeros((B, Cin, H, W))
    for dh, dw, cls, tap in table:
        a, b = cls >> 1, cls & 1
        shifted = np.zeros((B, Cout, Ho, Wo))
        shifted[:, :, :Ho - dh, :Wo - dw] = dy[:, :, dh:, dw:]            # rows / columns beyond the dy map are zero
        dx[:, :, a::2, b::2] += np.einsum("bohw,oc->bchw", shifted, w[:, :, tap // 3, tap % 3].numpy())
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=1e-12, atol=1e-12)
