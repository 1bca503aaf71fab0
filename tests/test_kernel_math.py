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


# ---------------------------------------------------------------------------------------------------
# k_wgrad3 (csrc/wgrad3.hip): the kernel's data movement restated lane by lane.  The step geometry comes from the library
# (ayolo_wgrad3_geometry: host code, no GPU); everything the GPU does with it -- which 16 bytes every DMA lane fetches and where
# they land in LDS, which LDS addresses every lane hands to the transposing fragment read, which accumulator element ends up
# in which dw element, how the slices of a block are summed -- is followed here with the kernel's own expressions, on an LDS
# image that starts as NaN (a fragment read that touches bytes no DMA wrote poisons the result, as it could on the GPU).
# ---------------------------------------------------------------------------------------------------
def _w3_geometry(B, C, N, H, W, s, ldx, ldy):
    import ctypes
    from ayolov2_amd import _lib, ops
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    d = ops.make_desc(torch.float16, B, H, W, C, ldx, N, ldy, (3, 3), (s, s), (1, 1), Ho, Wo)
    out = (ctypes.c_int64 * 24)()
    _lib.check(_lib.lib().ayolo_wgrad3_geometry(d, out, 24), "ayolo_wgrad3_geometry")
    names = "TC RPS PX nsub strips NB CB NP SL tn tc nrows ppr rowpitch plo ple xstage stage UP XP NU x_bytes y_bytes lds".split()
    return dict(zip(names, [int(v) for v in out])), Ho, Wo


def _w3_emulate(x, dy, g, s, steps_per_item):
    """x: (B, XH, XW, ldx) float16 NHWC, dy: (B, OH, OW, ldy) float16 -> dw (N, 9 * C) float64, plus the number of workspace slots."""
    B, XH, XW, ldx = x.shape
    _, OH, OW, ldy = dy.shape
    C, N = g["C"], g["N"]
    TC, RPS, PX, nsub, strips = g["TC"], g["RPS"], g["PX"], g["nsub"], g["strips"]
    NB, CB, NP, SL, tn, tc = g["NB"], g["CB"], g["NP"], g["SL"], g["tn"], g["tc"]
    nrows, ppr, rowpitch, plo, ple, xstage, stage = g["nrows"], g["ppr"], g["rowpitch"], g["plo"], g["ple"], g["xstage"], g["stage"]
    UP, XP, NU = g["UP"], g["XP"], g["NU"]
    K = 9 * C
    G_OOB, XOOB, M32 = 0x80000000, 0x40000000, 0xFFFFFFFF
    xf, yf = x.reshape(-1), dy.reshape(-1)
    lane = np.arange(64)
    uch = RPS * steps_per_item
    uranges = -(-NU // uch)
    slotf = tn * tc * NB * CB * 9 * 1024       # floats of a TILE-MAJOR slot: [tile][n-block][c-block][tap][32 rows][32 channels]
    slots = np.zeros((uranges, slotf))
    written = np.zeros((uranges, slotf), dtype=np.int32)
    xrowb, yrowb = XW * ldx * 2, OW * ldy * 2

    def dma(lds, dst, buf, nbytes, off):
        """one LDS-DMA piece: lane l moves 16 bytes from byte offset off[l] (zeros beyond the descriptor) to LDS dst + 16 l"""
        off = off & M32
        ok = off < nbytes
        idx = np.where(ok, off, 0)[:, None] // 2 + np.arange(8)
        vals = np.where(ok[:, None], buf[np.minimum(idx, buf.size - 1)], 0).astype(np.float16)
        lds[(dst // 2 + lane * 8)[:, None] + np.arange(8)] = vals

    def frag(lds, lo, hi):
        """ds_read_b64_tr_b16 x 2 -> [16 pixels][32 channels]: lane supplies the address of pixel (q >> 2) (+ 4), 4 channels"""
        F = np.zeros((16, 32))
        q = lane & 15
        for h, a in ((0, lo), (1, hi)):
            px = (lane >> 5) * 8 + h * 4 + (q >> 2)
            ch = ((lane >> 4) & 1) * 16 + (q & 3) * 4
            F[px[:, None], ch[:, None] + np.arange(4)] = lds[(a // 2)[:, None] + np.arange(4)]
        return F

    q16 = lane & 15
    rowl = q16 >> 2
    chanb = ((q16 & 3) * 4 + ((lane >> 4) & 1) * 16) * 2

    def xoff(pp):
        pp = np.minimum(pp, PX - 1)
        return (pp // TC) * (s * rowpitch) + (pp % TC) * 64 + chanb

    NW = 8                                  # wavefronts per workgroup; SL = NW / NP of them share a block
    for tile in range(tn * tc):
        for zz in range(uranges):
            tni, tci = tile // tc, tile % tc
            nb0, cb0 = tni * NB, tci * CB
            u0 = zz * uch
            u1 = min(u0 + uch, NU)
            nsteps = -(-(u1 - u0) // RPS)
            lds = np.full(max(g["lds"], 2 * stage) // 2, np.nan, dtype=np.float16)
            acc = np.zeros((NW, 9, 32, 32))
            wc = []
            for wave in range(NW):
                pair = wave & (NP - 1)
                slice_ = wave if NP == 1 else (wave >> 1 if NP == 2 else wave >> 2)
                nb, cb = (pair >> 1, pair & 1) if CB == 2 else (pair, 0)
                tapo = []
                for t in range(9):
                    dh, dw = t // 3, t % 3
                    colo = dw * 64 if s == 1 else (plo if dw == 1 else (64 if dw == 2 else 0))
                    tapo.append(dh * rowpitch + cb * (plo + ple) + colo)
                DYL = ((lane >> 5) * 8 + rowl) * 64 + chanb + nb * nsub * 1024
                wc.append(dict(pair=pair, slice=slice_, nb=nb, cb=cb, DYL=DYL, tapo=tapo))
            ndy = NB * nsub
            tabs = {}

            def setup_strip(c0):
                """the loader tables of a strip, shared by the wavefronts: XT[j][lane], DT[e][lane]"""
                cbsz = (plo + ple) >> 4
                XT, DT = [], []
                for j in range(ppr):
                    ci = j * 64 + lane
                    cbi = (ci >= cbsz).astype(int)
                    rem = ci - cbi * cbsz
                    even = rem >= (plo >> 4)
                    rem2 = np.where(even, rem - (plo >> 4), rem)
                    q, ch = rem2 >> 2, rem2 & 3
                    ic = c0 - 1 + q if s == 1 else np.where(even, 2 * (c0 + q), 2 * (c0 + q) - 1)
                    chan = (cb0 + cbi) * 32 + ch * 8
                    ok = (ci < CB * cbsz) & (ic >= 0) & (ic < XW) & (chan < C)
                    XT.append(np.where(ok, (ic * ldx + chan) * 2, XOOB))
                for e in range(ndy):
                    nbk, sub = e // nsub, e % nsub
                    pp = sub * 16 + (lane >> 2)
                    row, col = pp // TC, pp % TC
                    chan = (nb0 + nbk) * 32 + (lane & 3) * 8
                    ok = (pp < PX) & (c0 + col < OW) & (chan < N)
                    DT.append(np.where(ok, (((c0 + col) * ldy + chan) * 2) | row, XOOB))
                tabs["XT"], tabs["DT"] = XT, DT

            cur = {}

            def cursor_reset():
                V0 = s * u0
                cur.update(xn=V0 // XP, xvi=V0 % XP, yn=u0 // UP, yoh=u0 % UP, yu=u0)

            def cursor_step():
                cur["xvi"] += s * RPS
                if cur["xvi"] >= XP:
                    cur["xvi"] -= XP
                    cur["xn"] += 1
                cur["yoh"] += RPS
                cur["yu"] += RPS
                if cur["yoh"] >= UP:
                    cur["yoh"] -= UP
                    cur["yn"] += 1

            def issue(sb):
                # row bases of the whole window / of the step's dy rows, one row per lane
                vi, n = cur["xvi"] + lane, np.full(64, cur["xn"])
                n = np.where(vi >= XP, n + 1, n)
                vi = np.where(vi >= XP, vi - XP, vi)
                ih = (vi - 1) & M32
                rbv = np.where((ih < XH) & (n < B) & (lane < nrows), (n * XH + ih) * xrowb, G_OOB)
                oh, n2 = cur["yoh"] + lane, np.full(64, cur["yn"])
                n2 = np.where(oh >= UP, n2 + 1, n2)
                oh = np.where(oh >= UP, oh - UP, oh)
                dyv = np.where((lane < RPS) & (oh < OH) & (cur["yu"] + lane < u1), (n2 * OH + oh) * yrowb, G_OOB)
                for wave in range(NW):
                    for q in range(wave, nrows * ppr, 8):
                        r, j = q // ppr, q % ppr
                        dma(lds, sb + r * rowpitch + j * 1024, xf, g["x_bytes"], int(rbv[r]) + tabs["XT"][j])
                    for e in range(7 - wave, ndy, 8):
                        dc = tabs["DT"][e]
                        dma(lds, sb + xstage + e * 1024, yf, g["y_bytes"], dyv[dc & 3] + (dc & ~3))

            def compute(sb):
                pp_lane = 8 * (lane >> 5) + rowl
                for wave in range(NW):
                    w = wc[wave]
                    sub = w["slice"]
                    while sub < nsub:
                        ppc = pp_lane + 16 * sub
                        xo0, xo1 = xoff(ppc), xoff(ppc + 4)
                        ya = sb + xstage + w["DYL"] + sub * 1024
                        A = frag(lds, ya, ya + 256)
                        for t in range(9):
                            acc[wave, t] += A.T @ frag(lds, sb + xo0 + w["tapo"][t], sb + xo1 + w["tapo"][t])
                        sub += SL

            assert SL * NP == NW
            G = strips * nsteps
            strip_ld, st_ld = 0, 0
            setup_strip(0)
            cursor_reset()
            issue(0)
            for gi in range(G):
                sb = stage if gi & 1 else 0
                if gi + 1 < G:
                    st_ld += 1
                    if st_ld == nsteps:
                        st_ld, strip_ld = 0, strip_ld + 1
                        setup_strip(strip_ld * TC)
                        cursor_reset()
                    else:
                        cursor_step()
                    issue(stage - sb)
                compute(sb)
            # slices of a block summed in slice order by slice 0, which stores the block
            for wave in range(NW):
                w = wc[wave]
                if w["slice"] != 0:
                    continue
                tot = acc[wave].copy()
                for sl in range(1, SL):
                    other = next(v for v in range(NW) if wc[v]["pair"] == w["pair"] and wc[v]["slice"] == sl)
                    tot += acc[other]
                base = (tile * NB * CB + w["nb"] * CB + w["cb"]) * 9 * 1024      # the kernel's store: whole blocks, edge tiles padded
                for t in range(9):
                    blk = slots[zz, base + t * 1024:base + (t + 1) * 1024].reshape(32, 32)
                    blk[:] = tot[t]
                    written[zz, base + t * 1024:base + (t + 1) * 1024] += 1
    assert (written == 1).all(), "every element of every workspace slot is stored exactly once"
    # k_wgrad_reduce's view of a slot (w3_perm in csrc/conv.hip): dense element (n, t * C + c) -> tile-major offset
    n, col = np.divmod(np.arange(N * K), K)
    t, c = np.divmod(col, C)
    off = (((n // (NB * 32) * tc + c // (CB * 32)) * NB * CB + ((n >> 5) % NB) * CB + ((c >> 5) % CB)) * 9 + t) * 1024 + (n & 31) * 32 + (c & 31)
    return slots.sum(0)[off].reshape(N, K), uranges


@pytest.mark.parametrize("case", [
    (2, 32, 32, 8, 16, 1, 0, 3),       # one block: four wavefronts split the sub-steps
    (2, 32, 64, 12, 24, 2, 0, 2),      # stride 2, two n-blocks x two slices
    (1, 64, 64, 8, 20, 1, 8, 4),       # 2 x 2 blocks, x a channel slice of a wider buffer
    (2, 40, 72, 10, 12, 2, 0, 1),      # ragged channel blocks, two tiles along N
    (1, 96, 32, 6, 8, 1, 0, 2),        # one n-block x two c-blocks, two tiles along C
    (2, 32, 128, 9, 13, 2, 0, 3),      # odd map (bottom padding row, strip wider than the map), four n-blocks
    (3, 64, 64, 5, 4, 1, 0, 2),        # tiny map: steps straddle images
])
def test_wgrad3_patch_index_math(case):
    B, C, N, H, W, s, xpad, spi = case
    ldx, ldy = C + xpad, N
    geo, Ho, Wo = _w3_geometry(B, C, N, H, W, s, ldx, ldy)
    geo.update(C=C, N=N)
    gen = torch.Generator().manual_seed(sum(case))
    xt = torch.randn(B, C, H, W, generator=gen).half()
    dyt = torch.randn(B, N, Ho, Wo, generator=gen).half()
    x = np.full((B, H, W, ldx), 7.0, dtype=np.float16)               # the other channels of the wider buffer must never be read
    x[..., :C] = xt.permute(0, 2, 3, 1).numpy()
    dy = dyt.permute(0, 2, 3, 1).contiguous().numpy()
    dw, nslots = _w3_emulate(x, dy, geo, s, spi)
    xr = xt.double().requires_grad_(False)
    w = torch.zeros(N, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w, None, s, 1).backward(dyt.double())
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * C).numpy()      # [n][(dh, dw)][c]
    assert nslots >= 1
    np.testing.assert_allclose(dw, ref, rtol=1e-9, atol=1e-9)


def _dpp(v, ctrl):
    """What a lane reads under a DPP control (all lanes active, bound_ctrl): the source lane's value, per 16-lane row."""
    out = np.empty_like(v)
    for lane in range(64):
        row, i = lane & ~15, lane & 15
        if ctrl == 0x128:                      # row_ror:8
            src = (i + 8) % 16
        elif ctrl == 0x141:                    # row_half_mirror: i <-> 7 - i inside each half row
            src = (i & 8) | (7 - (i & 7))
        elif ctrl == 0x4E:                     # quad_perm [2,3,0,1]
            src = (i & ~3) | [2, 3, 0, 1][i & 3]
        elif ctrl == 0xB1:                     # quad_perm [1,0,3,2]
            src = (i & ~3) | [1, 0, 3, 2][i & 3]
        else:
            raise AssertionError(ctrl)
        out[lane] = v[row + src]
    return out


def _permlane16_swap(a, b):
    """v_permlane16_swap: the odd rows of the first operand <-> the even rows of the second (rows of 16 lanes)."""
    a2, b2 = a.copy(), b.copy()
    for row in (0, 2):
        a2[(row + 1) * 16:(row + 2) * 16] = b[row * 16:(row + 1) * 16]
        b2[row * 16:(row + 1) * 16] = a[(row + 1) * 16:(row + 2) * 16]
    return a2, b2


@pytest.mark.parametrize("nv", [16, 32])
def test_half_wavefront_reduce_scatter_lane_map(nv):
    """csrc/conv.hip half_reduce_scatter / rs_index / rs_reports, lane by lane: every lane holds nv partial sums (value r = one
    channel of its 32-lane half); after the butterfly lane l must hold the total of value rs_index(l) over the 32 lanes of ITS half
    (nv = 16: both lanes of a pair hold it, the even one reports), every value of a half reported exactly once -- the contract the
    conv epilogues' single fp64 LDS atomic per sum relies on.  Integers, so the check is exact."""
    rng = np.random.default_rng(nv)
    v = [rng.integers(-1000, 1000, size=64).astype(np.int64) for _ in range(nv)]          # v[r][lane]
    want = np.stack([[x[:32].sum(), x[32:].sum()] for x in v])                             # [r][half]
    lanes = np.arange(64)
    cur = [x.copy() for x in v]
    nxt = []
    for j in range(nv // 2):                                                               # level 16
        a2, b2 = _permlane16_swap(cur[2 * j], cur[2 * j + 1])
        nxt.append(a2 + b2)
    cur = nxt
    for ctrl, bit in ((0x128, 8), (0x141, 4), (0x4E, 2)):
        hi = (lanes & bit) != 0
        nxt = []
        for j in range(len(cur) // 2):
            s0, s1 = cur[2 * j], cur[2 * j + 1]
            keep, oth = np.where(hi, s1, s0), np.where(hi, s0, s1)
            nxt.append(keep + _dpp(oth, ctrl))
        cur = nxt
    if nv == 32:
        hi = (lanes & 1) != 0
        res = np.where(hi, cur[1], cur[0]) + _dpp(np.where(hi, cur[0], cur[1]), 0xB1)
    else:
        assert len(cur) == 1
        res = cur[0] + _dpp(cur[0], 0xB1)
    rs_index = ((lanes >> 4) & 1) + 2 * ((lanes >> 3) & 1) + 4 * ((lanes >> 2) & 1) + 8 * ((lanes >> 1) & 1) + (16 * (lanes & 1) if nv == 32 else 0)
    reports = np.ones(64, bool) if nv == 32 else (lanes & 1) == 0
    for lane in range(64):
        assert res[lane] == want[rs_index[lane], lane >> 5], (lane, rs_index[lane])
    for half in (0, 1):
        got = sorted(rs_index[(lanes >> 5 == half) & reports])
        assert got == list(range(nv)), got
