"""HIP conv / BN / pool kernels (fp32 exact-MFMA mode and fp16 mode) against plain PyTorch fp32 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, Cin, Cout, k, s, p, H, W)
SHAPES = [
    (2, 32, 64, 3, 2, 1, 40, 40),
    (2, 64, 32, 1, 1, 0, 20, 24),
    (3, 32, 32, 3, 1, 1, 13, 17),      # odd sizes: ragged pixel tiles
    (1, 128, 256, 3, 2, 1, 16, 16),
    (2, 256, 128, 1, 1, 0, 8, 8),
    (1, 64, 64, 3, 1, 1, 33, 9),
    (2, 16, 48, 3, 1, 1, 12, 12),      # Cout not a multiple of 32
    (2, 32, 64, 3, 1, 1, 24, 32),      # pixel count an exact multiple of the 128 / 256-pixel tiles
    (1, 64, 160, 3, 1, 1, 20, 40),     # ragged pixel tiles, Cout > 128 (two output-channel tiles, the second ragged)
    (2, 128, 32, 3, 1, 1, 8, 16),      # 4 channel chunks, single tile per image
    (2, 32, 64, 3, 2, 1, 13, 17),      # stride 2 on odd sizes: dgrad residue classes differ in size -> one launch per class
    (1, 64, 32, 1, 2, 0, 16, 16),      # 1x1 stride 2: three tap-less residue classes (zero gradient rows)
    (2, 48, 96, 3, 1, 1, 12, 12),      # yolov5m widths: 6 / 12 sixteen-byte channel chunks per pixel (not a power of two)
    (1, 96, 48, 3, 2, 1, 16, 16),
    (2, 80, 160, 1, 1, 0, 10, 12),     # yolov5x widths
    (2, 1280, 640, 1, 1, 0, 4, 4),     # deepest 1x1 of yolov5x: K = 1280, ragged 640-wide output
    (1, 320, 320, 3, 1, 1, 8, 8),
    # k_gconv3 (3x3 stride 1: one x row run shared by the three taps of a kernel row): maps narrower than the halo, a
    # single column / single row (every pixel is at the left AND right border), pixel counts around the tile sizes
    (2, 32, 32, 3, 1, 1, 7, 1),
    (1, 64, 32, 3, 1, 1, 1, 50),
    (3, 32, 32, 3, 1, 1, 2, 2),
    (2, 32, 64, 3, 1, 1, 16, 16),      # 512 pixels: exactly two 256-pixel tiles, image boundary on the tile boundary
    (1, 96, 96, 3, 1, 1, 15, 17),      # 255 pixels (one short of a tile), three 32-channel chunks
    (1, 32, 32, 3, 1, 1, 257, 1),      # 257 pixels in one column: the halo row of tile 0 is the first pixel of tile 1
    (1, 80, 160, 3, 1, 1, 20, 24),     # yolov5x: 80 channels = 2.5 chunks of 32, the last one half beyond C (zero-filled x and W)
    (2, 72, 40, 3, 1, 1, 9, 9),
    # k_pw (round 6: streaming 1x1 kernel, weights in registers): K = 128 / 256 with >= 128 output channels, ragged last pixel tile,
    # one / two / three channel tiles, fewer tiles than workgroup slots and more
    (2, 128, 128, 1, 1, 0, 7, 9),
    (3, 256, 256, 1, 1, 0, 13, 11),
    (2, 256, 128, 1, 1, 0, 16, 16),
    (2, 128, 384, 1, 1, 0, 5, 5),
    (5, 128, 136, 1, 1, 0, 64, 65),
    # ... K = 512 as two half-K sub-tiles per pixel tile (forward: Cin = 512; dgrad: Cout = 512): ragged tile, one / four channel tiles,
    # several pixel tiles per workgroup
    (2, 512, 256, 1, 1, 0, 9, 11),
    (1, 512, 512, 1, 1, 0, 20, 20),
    (3, 256, 512, 1, 1, 0, 7, 5),
    (16, 512, 128, 1, 1, 0, 40, 40),
]


# YOLOv5l / YOLOv5s layers at the geometry of the train step (batch 16 / 64 at 640x640): every tile variant the dispatcher
# picks at full size (128- / 256-pixel tiles, 32..128 output channels per tile, K up to 4608, two-half fragment fetch)
FULL_SIZE = [
    (16, 64, 128, 3, 2, 1, 320, 320),
    (16, 64, 64, 1, 1, 0, 160, 160),
    (16, 64, 64, 3, 1, 1, 160, 160),
    (16, 128, 256, 3, 2, 1, 160, 160),
    (16, 256, 256, 3, 1, 1, 40, 40),
    (16, 512, 256, 1, 1, 0, 80, 80),
    (16, 512, 1024, 3, 2, 1, 40, 40),
    (16, 512, 512, 3, 1, 1, 20, 20),
    (16, 1024, 1024, 1, 1, 0, 20, 20),
    (16, 2048, 512, 1, 1, 0, 20, 20),
    (64, 32, 64, 3, 2, 1, 320, 320),
    (64, 128, 128, 3, 1, 1, 40, 40),
    (64, 512, 256, 1, 1, 0, 20, 20),
    (64, 128, 128, 1, 1, 0, 40, 40),
    (64, 256, 256, 1, 1, 0, 40, 40),
    (64, 128, 128, 1, 1, 0, 80, 80),
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", FULL_SIZE)
def test_full_size_layers_fp16_vs_fp32_mode(shape):
    """Forward, data gradient and weight gradient of one conv at full-size launch geometry: the fp16 kernels against the
    exact-fp32 mode of the same kernels (itself held to 1e-4 of the CPU above) on operands pre-rounded to fp16, so only the
    fp16 output rounding (2^-11 relative) and the summation order differ."""
    from ayolov2_amd import functional as F_
    B, Cin, Cout, k, s, p, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g).half().float().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5).half().float()
    res = []
    gy = None
    for amp in (False, True):
        xg = x.clone().requires_grad_(True)
        wg = w.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            y = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
        if gy is None:
            gy = torch.randn(y.shape, device="cuda", generator=g).half().float()
        y.backward(gy.to(y.dtype))
        res.append((y.detach().float(), xg.grad.float(), wg.grad.float()))
    for a32, a16, what, tol in zip(res[0], res[1], ("y", "dx", "dw"), (1e-3, 1e-3, 1e-4)):
        scale = float(a32.abs().max())
        err = float((a16 - a32).abs().max())
        assert err <= tol * scale, (what, err, scale)


def _tol(dt):
    return (1e-4, 1e-4) if dt == torch.float32 else (3e-3, 3e-3)   # fp16: same rounded operands, fp32 accumulate -> only the output rounding


def _rel_err(got, want):
    got, want = got.detach(), want.detach()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_fwd_dgrad_wgrad(shape, dt):
    from ayolov2_amd import functional as F_
    B, Cin, Cout, k, s, p, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    if dt == torch.float16:
        x, w = x.half().float(), w.half().float()       # same rounded operands on both sides
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn(yr.shape, generator=g)
    if dt == torch.float16:
        gy = gy.half().float()
    yr.backward(gy)

    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    if dt == torch.float16:
        with torch.autocast("cuda", dtype=torch.float16):
            yg = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
    else:
        yg = F_.ConvFn.apply(xg, wg, (s, s), (p, p), F_._WeightCache())
    assert yg.dtype == dt
    yg.backward(gy.cuda().to(dt))
    rt = _tol(dt)[0]
    e = _rel_err(yg.float().cpu(), yr.detach())
    assert e < rt, f"fwd rel err {e}"
    e = _rel_err(xg.grad.float().cpu(), xr.grad)
    assert e < rt, f"dgrad rel err {e}"
    e = _rel_err(wg.grad.float().cpu(), wr.grad)
    assert e < rt, f"wgrad rel err {e}"


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 16, 20), (2, 128, 8, 10), (2, 256, 4, 5), (2, 128, 32, 40), (1, 32, 8, 8), (2, 64, 1, 5)])
def test_head_conv(shape, dt):
    """YOLOHead 1x1 conv + bias into the (B, na, ny, nx, no) fp32 layout (the EPI_HEAD epilogue), every element checked
    (a stale last accumulator register once survived the whole-network tolerance)."""
    from ayolov2_amd import functional as F_
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(B + Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(255, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(255, generator=g)
    if dt == torch.float16:
        x, w = x.half().float(), w.half().float()
    ref = F.conv2d(x, w, b).view(B, 3, 85, H, W).permute(0, 1, 3, 4, 2)
    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last)
    if dt == torch.float16:
        with torch.autocast("cuda", dtype=torch.float16):
            raw = F_.HeadConvFn.apply(xg, w.cuda(), b.cuda(), 3, 85, F_._WeightCache())
    else:
        raw = F_.HeadConvFn.apply(xg, w.cuda(), b.cuda(), 3, 85, F_._WeightCache())
    assert raw.dtype == torch.float32 and raw.shape == ref.shape
    assert _rel_err(raw.cpu(), ref) < (1e-5 if dt == torch.float32 else 2e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_stem_conv(dt):
    """6x6/s2/p2 on a 3-channel NCHW image (fp16: pixel-pair packed 6x3 conv; fp32: channel-padded)."""
    from ayolov2_amd.modules import Conv
    from oracle.model_ref import RConv
    torch.manual_seed(0)
    ref = RConv(3, 32, 6, 2, 2)
    m = Conv(3, 32, 6, 2, 2)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    x = torch.rand(2, 3, 64, 96)
    ref.train(); m.train()
    yr = ref(x)
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    if dt == torch.float16:
        with torch.autocast("cuda", dtype=torch.float16):
            yg = m(x.cuda())
    else:
        yg = m(x.cuda())
    yg.backward(gy.cuda().to(yg.dtype))
    rt = 2e-4 if dt == torch.float32 else 3e-2
    assert _rel_err(yg.float().cpu(), yr.detach()) < rt
    assert _rel_err(m.conv.weight.grad.float().cpu(), ref.conv.weight.grad) < rt * 2
    assert _rel_err(m.batch_norm.weight.grad.cpu(), ref.batch_norm.weight.grad) < rt * 2
    assert _rel_err(m.batch_norm.bias.grad.cpu(), ref.batch_norm.bias.grad) < rt * 2
    if dt == torch.float32:
        np.testing.assert_allclose(m.batch_norm.running_mean.cpu(), ref.batch_norm.running_mean, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.batch_norm.running_var.cpu(), ref.batch_norm.running_var, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [(64, 64, 3, 1), (64, 128, 3, 2), (128, 64, 1, 1)])
def test_conv_bn_silu_block_train(cfg, dt):
    """Conv -> BatchNorm(batch stats) -> SiLU forward + backward vs torch.nn on the CPU."""
    from ayolov2_amd.modules import Conv
    from oracle.model_ref import RConv
    cin, cout, k, s = cfg
    torch.manual_seed(1)
    ref = RConv(cin, cout, k, s)
    with torch.no_grad():
        ref.batch_norm.weight.uniform_(0.5, 1.5)
        ref.batch_norm.bias.uniform_(-0.5, 0.5)
    m = Conv(cin, cout, k, s)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    x = torch.randn(4, cin, 20, 20)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    xg = x.cuda().requires_grad_(True)
    if dt == torch.float16:
        with torch.autocast("cuda", dtype=torch.float16):
            yg = m(xg)
    else:
        yg = m(xg)
    yg.backward(gy.cuda().to(yg.dtype))
    rt = 2e-4 if dt == torch.float32 else 3e-2
    assert _rel_err(yg.float().cpu(), yr.detach()) < rt
    assert _rel_err(xg.grad.float().cpu(), xr.grad) < rt * 2
    assert _rel_err(m.conv.weight.grad.float().cpu(), ref.conv.weight.grad) < rt * 2
    assert _rel_err(m.batch_norm.weight.grad.cpu(), ref.batch_norm.weight.grad) < rt * 2
    assert _rel_err(m.batch_norm.bias.grad.cpu(), ref.batch_norm.bias.grad) < rt * 2


@pytest.mark.parametrize("shape", [(4, 32, 32, 3, 1, 1, 40, 40), (4, 64, 64, 3, 1, 1, 40, 40), (2, 128, 128, 3, 1, 1, 40, 40),
                                   (8, 128, 256, 3, 1, 1, 80, 80), (2, 80, 160, 3, 1, 1, 24, 20), (4, 128, 128, 1, 1, 0, 40, 40),
                                   (2, 64, 128, 3, 2, 1, 40, 40)])
def test_conv_epilogue_statistics_fp16(shape):
    """The BatchNorm sums a conv leaves behind (sum z, sum z^2 over the rounded fp16 outputs, per channel) for every channel-tile
    width and both conv kernels -- k_gconv keeps them in registers across tiles, k_gconv3 reduces them per tile with DPP row
    sums + LDS atomics -- against the sums of the stored output itself (fp32 summation order is the only difference)."""
    from ayolov2_amd import functional as F_, ops
    B, Cin, Cout, k, s_, p_, H, W = shape
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    geo = F_._Geometry((B, Cin, H, W), wt.shape, (s_, s_), (p_, p_), dt)
    w, _ = F_._WeightCache().get(wt.contiguous(memory_format=torch.channels_last), dt, Cout, geo.cin_pad)
    y = ops.new_act(B, Cout, geo.Ho, geo.Wo, dt, x.device)
    stats = torch.zeros((ops.STAT_REPS, 2 * Cout), dtype=torch.float64, device="cuda")
    ops.conv_fwd(geo.desc(dt, geo.Cin_k, Cout), x, w, y, 0, stats=stats)
    torch.cuda.synchronize()
    tot = stats.double().sum(0).cpu()
    yf = y.double()
    P = B * geo.Ho * geo.Wo                                   # fp32 accumulation: ~1e-7 of the summed magnitudes
    np.testing.assert_allclose(tot[:Cout].numpy(), yf.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-6 * P)
    np.testing.assert_allclose(tot[Cout:].numpy(), (yf * yf).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-6 * P)
    ref = F.conv2d(x.float(), wt.half().float(), None, s_, p_)
    assert float((y.float() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(2, 32, 64, 40, 40), (1, 64, 32, 8, 6), (2, 32, 80, 10, 12), (1, 48, 64, 34, 2), (3, 16, 32, 6, 6),
                                   (2, 64, 128, 64, 48), (1, 8, 96, 18, 66), (4, 32, 64, 2, 2), (2, 128, 64, 24, 16), (1, 256, 128, 12, 20),
                                   (2, 192, 96, 10, 10)])
def test_dgrad_stride2_all_classes(shape):
    """k_dgrad_s2 (data gradient of 3x3 / stride 2 / pad 1 convs with <= 64 input channels: the four residue classes share
    the dy rows and their accumulators live in one workgroup) against torch's conv_transpose-style reference, fp16 with
    pre-rounded operands; also into a pre-filled buffer (accumulate)."""
    from ayolov2_amd import functional as F_, ops
    B, Cin, Cout, H, W = shape
    dt = torch.float16
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g).half().float().requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().float()
    y = F.conv2d(x, w, None, 2, 1)
    gy = torch.randn(y.shape, generator=g).half().float()
    y.backward(gy)
    geo = F_._Geometry((B, Cin, H, W), w.shape, (2, 2), (1, 1), dt)
    _, wt = F_._WeightCache().get(w.cuda().contiguous(memory_format=torch.channels_last), dt, Cout, geo.cin_pad)
    dy = gy.cuda().to(dt).contiguous(memory_format=torch.channels_last)
    dx = ops.new_act(B, geo.cin_pad, H, W, dt, dy.device)
    ops.conv_dgrad(geo.desc(dt, geo.cin_pad, Cout), dy, wt, dx)
    ref = x.grad
    got = dx[:, :Cin].float().cpu()
    assert float((got - ref).abs().max()) <= 3e-3 * float(ref.abs().max()), float((got - ref).abs().max())
    base = torch.randn(dx.shape, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    dx2 = base.clone()
    ops.conv_dgrad(geo.desc(dt, geo.cin_pad, Cout), dy, wt, dx2, accumulate=True)
    exp = base[:, :Cin].float().cpu() + ref
    assert float((dx2[:, :Cin].float().cpu() - exp).abs().max()) <= 4e-3 * float(exp.abs().max())


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_conv_eval_and_fuse(dt):
    from ayolov2_amd.modules import Conv
    from oracle.model_ref import RConv
    torch.manual_seed(2)
    ref = RConv(32, 64, 3, 1)
    with torch.no_grad():
        ref.batch_norm.running_mean.uniform_(-0.3, 0.3)
        ref.batch_norm.running_var.uniform_(0.5, 2.0)
        ref.batch_norm.weight.uniform_(0.5, 1.5)
        ref.batch_norm.bias.uniform_(-0.5, 0.5)
    ref.eval()
    m = Conv(32, 64, 3, 1)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    x = torch.randn(2, 32, 24, 24)
    with torch.no_grad():
        yr = ref(x)
        xg = x.cuda()
        if dt == torch.float16:
            m = m.half()
            xg = xg.half()
        y1 = m(xg)
        y2 = m.fuse()(xg)
    rt = 2e-4 if dt == torch.float32 else 3e-2
    assert _rel_err(y1.float().cpu(), yr) < rt
    assert _rel_err(y2.float().cpu(), yr) < rt


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_maxpool_upsample(dt):
    from ayolov2_amd import functional as F_
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 20, 20, generator=g)
    if dt == torch.float16:
        x = x.half().float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 5, 1, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = F_.MaxPoolFn.apply(xg, 5)
    yg.backward(gy.cuda().to(dt))
    np.testing.assert_array_equal(yg.detach().float().cpu().numpy(), yr.detach().numpy())
    rt = 1e-6 if dt == torch.float32 else 5e-3
    assert _rel_err(xg.grad.float().cpu(), xr.grad) <= rt
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="nearest")
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = F_.Upsample2xFn.apply(xg)
    yg.backward(gy.cuda().to(dt))
    np.testing.assert_array_equal(yg.detach().float().cpu().numpy(), yr.detach().numpy())
    assert _rel_err(xg.grad.float().cpu(), xr.grad) <= rt


@pytest.mark.parametrize("shape,k", [((1, 8, 13, 17), 5), ((2, 8, 4, 3), 5), ((1, 8, 9, 11), 3), ((1, 16, 23, 7), 9), ((3, 8, 20, 20), 5)])
def test_maxpool_ties_and_strips(shape, k):
    """Max-pool forward works in strips of 5 output rows with a horizontal-then-vertical reduction (k = 5 specialised): map
    heights that are not multiples of the strip, maps smaller than the window, other window sizes -- on inputs quantised to
    four levels so that nearly every window has ties: the gradient then lands where torch's row-major first-maximum rule
    puts it."""
    from ayolov2_amd import functional as F_
    g = torch.Generator().manual_seed(sum(shape) + k)
    x = torch.randint(0, 4, shape, generator=g).float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, 1, k // 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = F_.MaxPoolFn.apply(xg, k)
    yg.backward(gy.cuda())
    np.testing.assert_array_equal(yg.detach().cpu().numpy(), yr.detach().numpy())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)


SPPF_SHAPES = [(2, 16, 20, 20), (1, 8, 7, 9), (3, 256, 20, 20), (1, 32, 40, 40), (2, 24, 4, 3), (1, 16, 10, 10)]


def _sppf_inputs(shape, mode, g):
    B, C, H, W = shape
    if mode == "normal":
        x = torch.randn(B, C, H, W, generator=g)
    elif mode == "ties":                                     # four levels: nearly every window has ties, zeros of both signs
        x = torch.randint(-1, 3, (B, C, H, W), generator=g).float()
        x = torch.where((x == 0) & (torch.rand(x.shape, generator=g) < 0.5), torch.full_like(x, -0.0), x)
    else:                                                    # NaN / inf sprinkled in
        x = torch.randn(B, C, H, W, generator=g)
        r = torch.rand(x.shape, generator=g)
        x = torch.where(r < 0.02, torch.full_like(x, float("nan")), x)
        x = torch.where((r >= 0.02) & (r < 0.04), torch.full_like(x, float("inf")), x)
        x = torch.where((r >= 0.04) & (r < 0.08), torch.full_like(x, float("-inf")), x)
    return x.half()


@pytest.mark.parametrize("shape", SPPF_SHAPES)
@pytest.mark.parametrize("mode", ["normal", "ties", "special"])
def test_sppf_pool_cascade_equals_three_pool_launches(shape, mode):
    """ayolo_sppf_pool_fwd / _bwd (kindle SPPF's three chained MaxPool2d(5, 1, 2), res/configs/model/yolov5s.yaml:33, as one
    launch per direction over the LDS-resident map) against three ayolo_maxpool_fwd / _bwd launches on the same concat buffer:
    values and recorded window positions bit for bit -- on ties (first maximum in row-major order), signed zeros, infinities and
    NaN (torch's rule: the last NaN of the scan) -- and the input gradient: the cascade's sums are exact (fp64 atomics), the
    launches' sequential fp32 sums almost always are, so at most a stray last-place difference is tolerated.  Both NCG = 2 and
    NCG = 1 geometries (C not a multiple of 16; 40 x 40 maps)."""
    from ayolov2_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + len(mode))
    x = _sppf_inputs(shape, mode, g)
    cat_a = torch.zeros(B, 4 * C + 8, H, W, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)[:, :4 * C]
    cat_a[:, :C] = x.cuda()
    cat_b = cat_a.clone(memory_format=torch.preserve_format)[:, :4 * C] if False else torch.zeros_like(cat_a)
    cat_b[:, :C] = x.cuda()
    args_b = []
    for j in range(3):
        _, a = ops.maxpool_fwd(cat_b[:, j * C:(j + 1) * C], 5, y=cat_b[:, (j + 1) * C:(j + 2) * C])
        args_b.append(a)
    args_a = ops.sppf_pool_fwd(cat_a, C)
    torch.cuda.synchronize()
    va, vb = cat_a.float(), cat_b.float()
    assert torch.equal(torch.isnan(va), torch.isnan(vb))
    assert torch.equal(torch.nan_to_num(va, nan=0.0), torch.nan_to_num(vb, nan=0.0))
    if mode != "special":                                    # (an all -inf window records its first in-image tap; the scan kernels tap 0)
        for j in range(3):
            assert torch.equal(args_a[j], args_b[j]), j
    else:
        for j in range(3):
            ok = torch.isfinite(cat_b[:, (j + 1) * C:(j + 2) * C].permute(0, 2, 3, 1)) | torch.isnan(cat_b[:, (j + 1) * C:(j + 2) * C].permute(0, 2, 3, 1)) \
                | (cat_b[:, (j + 1) * C:(j + 2) * C].permute(0, 2, 3, 1) == float("inf"))
            assert torch.equal(args_a[j][ok], args_b[j][ok]), j
    if mode == "special":
        return
    # backward: same gradient of the concat buffer into both routes
    d = torch.randn(B, 4 * C, H, W, generator=g).half()
    da = torch.zeros_like(cat_a)
    da.copy_(d.cuda())
    db = da.clone(memory_format=torch.preserve_format)
    for j in (2, 1, 0):
        ops.maxpool_bwd(args_b[j], db[:, (j + 1) * C:(j + 2) * C], 5, dx=db[:, j * C:(j + 1) * C], accumulate=True)
    dxa = ops.sppf_pool_bwd(args_a, da, C)
    torch.cuda.synchronize()
    assert torch.equal(da[:, C:], d.cuda()[:, C:])           # slices 1..3 untouched
    a, b = dxa.float(), db[:, :C].float()
    diff = (a != b)
    assert float(diff.float().mean()) <= 1e-4, float(diff.float().mean())
    assert float((a - b).abs().max()) <= 2.0 ** -10 * float(b.abs().max())


def test_sppf_pool_cascade_vs_torch_autograd():
    """The same cascade against torch's own max_pool2d chain with autograd on the CPU (the oracle's SPPF body)."""
    from ayolov2_amd import ops
    B, C, H, W = 2, 16, 20, 20
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 6, (B, C, H, W), generator=g).float()
    xr = x.clone().requires_grad_(True)
    y1 = F.max_pool2d(xr, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    cr = torch.cat((xr, y1, y2, y3), 1)
    d = torch.randint(-3, 4, cr.shape, generator=g).float()
    cr.backward(d)
    cat = torch.zeros(B, 4 * C, H, W, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    cat[:, :C] = x.cuda().half()
    arg = ops.sppf_pool_fwd(cat, C)
    np.testing.assert_array_equal(cat.float().cpu().numpy(), cr.detach().numpy())
    dc = torch.zeros_like(cat)
    dc.copy_(d.cuda().half())
    dx = ops.sppf_pool_bwd(arg, dc, C)
    np.testing.assert_array_equal(dx.float().cpu().numpy(), xr.grad.numpy())      # small integers: every sum exact in fp16


def test_wide_pixel_tile_variants():
    """k_gconv's 256-pixel-tile instantiations are chosen only for large maps (>= 131072 output pixels); force them
    (AYOLO_GCONV_TP=256, read once per process) on the small shapes so that every element is checked, fp32 and fp16."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AYOLO_GCONV_TP="256")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_check.py")], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("ok ", "BAD"))]
    assert out.returncode == 0 and len(lines) >= 30, out.stderr[-2000:]
    bad = [l for l in lines if l.startswith("BAD")]
    assert not bad, "\n".join(bad)


# (B, Cin, Cout, k, s, p, H, W, segments as (c0, C) of dx's channels): every dgrad kernel that carries the BN-backward sums
BNR_CASES = [
    ((2, 64, 32, 1, 1, 0, 20, 24), [(0, 64)]),                 # k_gconv, 64-channel tile
    ((2, 64, 64, 1, 1, 0, 40, 40), [(0, 32), (32, 32)]),       # two segments (C3 concat buffer), one per wave
    ((2, 128, 64, 1, 1, 0, 24, 24), [(0, 64), (64, 64)]),      # 128-channel tile: per-tile reduction, segment per (wave, mi)
    ((2, 256, 128, 1, 1, 0, 16, 16), [(0, 128), (128, 128)]),  # two channel tiles, one segment each
    ((2, 32, 64, 1, 1, 0, 33, 17), [(0, 32)]),                 # 32-channel tile, ragged pixel tiles
    ((3, 32, 32, 3, 1, 1, 13, 17), [(0, 32)]),                 # k_gconv3, registers across tiles
    ((2, 128, 128, 3, 1, 1, 20, 20), [(0, 128)]),              # k_gconv3, 128-channel tile (per-tile reduction)
    ((2, 32, 64, 3, 2, 1, 40, 40), [(0, 32)]),                 # k_dgrad_s2, 32-channel tile (the stem's output gradient)
    ((2, 64, 128, 3, 2, 1, 24, 24), [(0, 64)]),                # k_dgrad_s2, 64-channel tile
    ((1, 128, 256, 3, 2, 1, 16, 16), [(0, 128)]),              # k_gconv walking the four residue classes
    ((2, 64, 64, 3, 2, 1, 13, 17), [(0, 64)]),                 # odd map: one launch per residue class
    ((2, 96, 48, 1, 1, 0, 12, 12), [(8, 80)]),                 # a segment that is a strict sub-range of dx's channels
    ((2, 128, 128, 1, 1, 0, 13, 17), [(0, 128)]),              # k_pw (K = Cout = 128), ragged last tile
    ((3, 256, 256, 1, 1, 0, 9, 11), [(0, 128), (128, 128)]),   # k_pw (K = 256), two channel tiles, a segment each
    ((2, 256, 128, 1, 1, 0, 40, 40), [(0, 96), (96, 160)]),    # k_pw, segment boundary inside a channel tile (32-channel blocks)
    ((2, 256, 512, 1, 1, 0, 10, 10), [(0, 256)]),              # k_pw with K = Cout = 512 (two sub-tiles per pixel tile), a wave without pixels
    ((3, 512, 512, 1, 1, 0, 9, 11), [(0, 256), (256, 256)]),   # ... four channel tiles, a segment per pair
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BNR_CASES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_dgrad_with_bn_backward_sums(case, accumulate):
    """ayolo_conv_dgrad_bn: dx is bit-identical to the plain dgrad, and the sums it leaves in the accumulators equal what
    ayolo_bn_act_bwd_reduce computes from that dx (same rounded values; only the fp32 summation order differs)."""
    from ayolov2_amd import functional as F_, ops
    from ayolov2_amd._lib import call
    shape, segs = case
    B, Cin, Cout, k, s, p, H, W = shape
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + len(segs))
    geo = F_._Geometry((B, Cin, H, W), (Cout, Cin, k, k), (s, s), (p, p), dt)
    w32 = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    _, wt = F_._WeightCache().get(w32, dt, Cout, geo.cin_pad)
    dy = ops.new_act(B, Cout, geo.Ho, geo.Wo, dt, "cuda")
    dy.copy_(torch.randn(dy.shape, device="cuda", generator=g))
    base = ops.new_act(B, Cin, H, W, dt, "cuda")
    base.copy_(torch.randn(base.shape, device="cuda", generator=g))
    d = geo.desc(dt, Cin, Cout)
    # reference: plain dgrad, then the separate reduce pass per segment
    dx_ref = base.clone(memory_format=torch.preserve_format) if accumulate else ops.new_act(B, Cin, H, W, dt, "cuda")
    ops.conv_dgrad(d, dy, wt, dx_ref, accumulate=accumulate)
    dx = base.clone(memory_format=torch.preserve_format) if accumulate else ops.new_act(B, Cin, H, W, dt, "cuda")
    seg_args, refs = [], []
    for c0, C in segs:
        z = ops.new_act(B, C, H, W, dt, "cuda")
        z.copy_(torch.randn(z.shape, device="cuda", generator=g) * 1.5 + 0.3)
        mean = torch.randn(C, device="cuda", generator=g) * 0.3
        invstd = torch.rand(C, device="cuda", generator=g) + 0.5
        gamma = torch.randn(C, device="cuda", generator=g)
        beta = torch.randn(C, device="cuda", generator=g) * 0.5
        mi = torch.cat((mean, invstd)).contiguous()
        sums = torch.zeros((ops.STAT_REPS, 2 * C), dtype=torch.float64, device="cuda")
        seg_args.append((z, mi, gamma, beta, sums, c0))
        ref = torch.zeros((ops.STAT_REPS, 2 * C), dtype=torch.float64, device="cuda")
        da = dx_ref[:, c0:c0 + C]
        call("ayolo_bn_act_bwd_reduce", ops.dtype_code(dt), z.data_ptr(), C, da.data_ptr(), Cin, B * H * W, C, mean.data_ptr(),
             invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, ref.data_ptr(), ops.STAT_REPS, torch.cuda.current_stream().cuda_stream)
        refs.append(ref)
    ops.conv_dgrad_bn(d, dy, wt, dx, seg_args, act=1, accumulate=accumulate)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    for (z, mi, gamma, beta, sums, c0), ref in zip(seg_args, refs):
        got, want = sums.sum(0).double().cpu(), ref.sum(0).double().cpu()
        C = z.shape[1]
        # scale: the L1 mass of the summands is ~ npix * |du|; compare against the largest channel total and that mass
        mass = float(dx_ref[:, c0:c0 + C].float().abs().sum() / C)
        err = float((got - want).abs().max())
        assert err <= 2e-5 * mass + 1e-6, (err, mass, float(want.abs().max()))


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
@pytest.mark.parametrize("accumulate", [False, True])
def test_bn_backward_apply_forwards_the_shortcut_gradient(dt, accumulate):
    """ayolo_bn_act_bwd_apply_res: the BatchNorm + SiLU backward pass of a block whose output also fed a shortcut add writes
    d(shortcut) (+)= da from the da values it reads anyway (kindle Bottleneck: x + cv2(cv1(x))); dz / dgamma / dbeta must be
    those of the plain pass bit for bit, the shortcut gradient exact (a copy, or one fp addition in the storage dtype),
    also into a channel slice of a wider buffer."""
    from ayolov2_amd import ops
    torch.manual_seed(7)
    B, C, H, W = 3, 64, 9, 11                         # 297 pixels: the four-pixel main loop and the tail
    z = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    da = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    mean = z.float().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.float().var((0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    ref = ops.bn_act_bwd(z, da, mean, invstd, gamma, beta, 1)
    wide = torch.randn(B, 2 * C, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    old = wide.clone()
    dres = wide[:, C:]                                 # channel slice: row stride 2C
    out = ops.bn_act_bwd(z, da, mean, invstd, gamma, beta, 1, dres=dres, res_accumulate=accumulate)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    want = (old[:, C:].float() + da.float()).to(dt) if accumulate else da
    assert torch.equal(wide[:, C:], want)
    assert torch.equal(wide[:, :C], old[:, :C])        # the other half of the wide buffer is untouched


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(3, 32, 32, 9, 11, 1), (2, 64, 64, 16, 16, 1), (1, 128, 128, 5, 7, 0), (2, 32, 96, 12, 12, 1), (1, 256, 256, 4, 4, 1)])
def test_bn_backward_apply_of_a_merged_pair_equals_two_passes(dt, shape):
    """ayolo_bn_act_bwd_apply2: the apply passes of the two blocks of a merged cv1 | cv2 conv (kindle C3,
    res/configs/model/yolov5s.yaml:23-52) as ONE launch over whole rows of the shared z / dz buffers must equal two
    ayolo_bn_act_bwd_apply launches over the half rows bit for bit: dz, dgamma and dbeta of both blocks.  Each block has its own
    output-gradient buffer (one contiguous, one a channel slice of a wider buffer, as in the plan), statistics and sums."""
    from ayolov2_amd import ops
    from ayolov2_amd._lib import BnApplySeg, call
    torch.manual_seed(31)
    B, C0, C1, H, W, act = shape
    Ct, npix = C0 + C1, B * H * W
    z = torch.randn(B, Ct, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    da0 = torch.randn(B, C0, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    wide = torch.randn(B, C1 + 32, H, W, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    da1 = wide[:, 32:]                                                              # row stride C1 + 32
    das, c0s, Cs = (da0, da1), (0, C0), (C0, C1)
    code, st = ops.dtype_code(dt), torch.cuda.current_stream().cuda_stream
    mean = z.float().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.float().var((0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = torch.rand(Ct, device="cuda") + 0.5, torch.randn(Ct, device="cuda")
    sums, want_dz = [], torch.zeros_like(z)
    want_g, want_b = torch.empty(Ct, device="cuda"), torch.empty(Ct, device="cuda")
    for k in range(2):
        c0, C = c0s[k], Cs[k]
        zk, sl = z[:, c0:c0 + C], slice(c0, c0 + C)
        su = torch.zeros(ops.STAT_REPS, 2 * C, dtype=torch.float64, device="cuda")
        call("ayolo_bn_act_bwd_reduce", code, zk.data_ptr(), Ct, das[k].data_ptr(), ops.nhwc_info(das[k])[4], npix, C, mean[sl].data_ptr(),
             invstd[sl].data_ptr(), gamma[sl].data_ptr(), beta[sl].data_ptr(), act, su.data_ptr(), ops.STAT_REPS, st)
        sums.append(su)
        call("ayolo_bn_act_bwd_apply", code, zk.data_ptr(), Ct, das[k].data_ptr(), ops.nhwc_info(das[k])[4], want_dz[:, sl].data_ptr(), Ct, npix, C,
             mean[sl].data_ptr(), invstd[sl].data_ptr(), gamma[sl].data_ptr(), beta[sl].data_ptr(), act, su.data_ptr(), ops.STAT_REPS,
             want_g[sl].data_ptr(), want_b[sl].data_ptr(), 1.0, st)
    dz = torch.zeros_like(z)
    got_g, got_b = torch.empty(Ct, device="cuda"), torch.empty(Ct, device="cuda")
    segs = []
    for k in range(2):
        sl = slice(c0s[k], c0s[k] + Cs[k])
        g = BnApplySeg()
        g.da, g.save_mean, g.save_invstd = das[k].data_ptr(), mean[sl].data_ptr(), invstd[sl].data_ptr()
        g.gamma, g.beta, g.sums = gamma[sl].data_ptr(), beta[sl].data_ptr(), sums[k].data_ptr()
        g.dgamma, g.dbeta, g.ldda, g.C = got_g[sl].data_ptr(), got_b[sl].data_ptr(), ops.nhwc_info(das[k])[4], Cs[k]
        segs.append(g)
    import ctypes
    call("ayolo_bn_act_bwd_apply2", code, z.data_ptr(), Ct, dz.data_ptr(), Ct, npix, ctypes.byref(segs[0]), ctypes.byref(segs[1]), act,
         ops.STAT_REPS, 1.0, st)
    torch.cuda.synchronize()
    assert torch.equal(dz, want_dz)
    assert torch.equal(got_g, want_g) and torch.equal(got_b, want_b)


def test_bn_backward_apply_two_gib_tensors_take_the_pointer_path():
    """k_bn_bwd_apply addresses tensors below 2 GiB through 32-bit buffer offsets (B32) and larger ones through 64-bit pointers:
    the same pass over a 2 GiB activation (64 x 64 x 512 x 512 fp16 = 2^31 bytes: not below the limit) and over its two halves
    (B32) must agree bit for bit.  With zero sums the per-channel constants do not depend on the pixel count, so the halves are
    exactly the rows of the whole."""
    from ayolov2_amd import ops
    from ayolov2_amd._lib import call
    B, C, H, W = 64, 64, 512, 512
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(11)
    z = torch.empty(B, H, W, C, device="cuda", dtype=dt).normal_(generator=g)            # NHWC storage, rows of C
    da = torch.empty(B, H, W, C, device="cuda", dtype=dt).normal_(generator=g)
    assert z.numel() * 2 == 1 << 31
    mean = torch.randn(C, device="cuda") * 0.1
    invstd = torch.rand(C, device="cuda") + 0.5
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    sums = torch.zeros(ops.STAT_REPS, 2 * C, dtype=torch.float64, device="cuda")
    dgm, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    code = ops.dtype_code(dt)
    st = torch.cuda.current_stream().cuda_stream

    def run(zz, dd, out, npix):
        call("ayolo_bn_act_bwd_apply", code, zz.data_ptr(), C, dd.data_ptr(), C, out.data_ptr(), C, npix, C, mean.data_ptr(), invstd.data_ptr(),
             gamma.data_ptr(), beta.data_ptr(), 1, sums.data_ptr(), ops.STAT_REPS, dgm.data_ptr(), dbt.data_ptr(), 1.0, st)

    npix = B * H * W
    whole = torch.empty_like(z)
    run(z, da, whole, npix)
    halves = torch.empty_like(z)
    h = B // 2
    run(z[:h], da[:h], halves[:h], npix // 2)
    run(z[h:], da[h:], halves[h:], npix // 2)
    torch.cuda.synchronize()
    assert torch.equal(whole, halves)
    assert bool(torch.isfinite(whole.float()).all()) and float(whole.float().abs().max()) > 0


@pytest.mark.parametrize("shape", [(2, 32, 64, 96), (3, 16, 70, 100), (1, 48, 38, 132), (2, 64, 24, 40), (2, 80, 64, 96)])
def test_stem_weight_gradient_kernel(shape):
    """k_stem_wgrad (the packed 6x6 / stride 2 / pad 2 stem, fp16, Cout <= 64: input patch staged once per 4 x 64 output tile,
    fragments by transposing LDS reads at a 16-byte pixel stride) against torch's fp32 weight gradient on fp16-rounded
    operands: 1e-4 of the largest element, for every YOLOv5 stem width up to l (16 / 32 / 48 / 64 output channels) and maps
    that end inside a tile in both directions; 80 channels (YOLOv5x) stay on the generic k_wgrad path, same bar."""
    from ayolov2_amd import functional as F_
    B, Cout, H, W = shape
    g = torch.Generator().manual_seed(B + Cout + H)
    x = torch.rand(B, 3, H, W, generator=g).half().float()
    w = (torch.randn(Cout, 3, 6, 6, generator=g) / 108 ** 0.5).half().float()
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, 2, 2)
    gy = torch.randn(yr.shape, generator=g).half().float()
    yr.backward(gy)
    wg = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        yg = F_.ConvFn.apply(x.cuda(), wg, (2, 2), (2, 2), F_._WeightCache())
    yg.backward(gy.cuda().half())
    assert _rel_err(yg.float().cpu(), yr.detach()) < 2e-3
    assert _rel_err(wg.grad.float().cpu(), wr.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 32, 64, 96), (3, 16, 70, 100), (1, 48, 38, 132), (2, 64, 24, 40), (64, 32, 128, 128)])
def test_stem_forward_kernel_and_statistics(shape):
    """k_stem_fwd (training forward of the packed stem: patch staged once per 4 x 64 tile, every B fragment one 16-byte LDS
    read, weight fragments register-resident): the stored fp16 output against torch's fp32 conv on fp16-rounded operands
    (3e-3 of the largest element = the output rounding) and the BatchNorm sums against the sums of the stored output."""
    from ayolov2_amd import functional as F_, ops
    B, Cout, H, W = shape
    dt = torch.float16
    g = torch.Generator().manual_seed(B + Cout + H)
    x = torch.rand(B, 3, H, W, generator=g).half().float()
    wt = (torch.randn(Cout, 3, 6, 6, generator=g) / 108 ** 0.5).half().float()
    geo = F_._Geometry((B, 3, H, W), wt.shape, (2, 2), (2, 2), dt)
    assert geo.packed_stem
    xk = F_._prepare_input(x.cuda(), geo, dt)
    w, _ = F_._WeightCache().get(wt.cuda().contiguous(memory_format=torch.channels_last), dt, Cout, geo.cin_pad)
    y = ops.new_act(B, Cout, geo.Ho, geo.Wo, dt, xk.device)
    y.fill_(float("nan"))                                       # every valid element must be written
    stats = torch.zeros((ops.STAT_REPS, 2 * Cout), dtype=torch.float64, device="cuda")
    ops.conv_fwd(geo.desc(dt, geo.Cin_k, Cout), xk, w, y, 0, stats=stats)
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, None, 2, 2)
    assert float((y.float().cpu() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())
    tot = stats.sum(0).cpu()
    yf = y.double()
    P = B * geo.Ho * geo.Wo
    np.testing.assert_allclose(tot[:Cout].numpy(), yf.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-6 * P)
    np.testing.assert_allclose(tot[Cout:].numpy(), (yf * yf).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-6 * P)


@pytest.mark.parametrize("shape", [(2, 32, 64, 96), (3, 16, 70, 100), (2, 64, 24, 40), (1, 48, 38, 132)])
@pytest.mark.parametrize("act", [1, 0])
def test_stem_backward_in_one_launch(shape, act):
    """ayolo_stem_bn_wgrad: BatchNorm + SiLU backward of the stem's output gradient and the weight gradient of its conv in one
    kernel (dz formed from (da, z) on the way to LDS) against the two-kernel sequence ayolo_bn_act_bwd_apply -> ayolo_conv_wgrad
    with the same sums: dgamma / dbeta identical, dw within 2e-3 of its largest element (the fused kernel folds two of the
    per-channel constants, so a dz element may round to the neighbouring fp16 value)."""
    from ayolov2_amd import functional as F_, ops
    B, Cout, H, W = shape
    dt = torch.float16
    torch.manual_seed(B + Cout + H + act)
    x = torch.rand(B, 3, H, W)
    geo = F_._Geometry((B, 3, H, W), (Cout, 3, 6, 6), (2, 2), (2, 2), dt)
    xk = F_._prepare_input(x.cuda(), geo, dt)
    z = torch.randn(B, Cout, geo.Ho, geo.Wo, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    da = torch.randn(B, Cout, geo.Ho, geo.Wo, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    mean = z.float().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.float().var((0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    npix = B * geo.Ho * geo.Wo
    code = ops.dtype_code(dt)
    sums = ops.zero_stats(Cout, z.device)
    from ayolov2_amd._lib import call
    call("ayolo_bn_act_bwd_reduce", code, z.data_ptr(), Cout, da.data_ptr(), Cout, npix, Cout, mean.data_ptr(), invstd.data_ptr(),
         gamma.data_ptr(), beta.data_ptr(), act, sums.data_ptr(), ops.STAT_REPS, torch.cuda.current_stream().cuda_stream)
    dz = torch.empty_like(z)
    dg_ref, db_ref = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    call("ayolo_bn_act_bwd_apply", code, z.data_ptr(), Cout, da.data_ptr(), Cout, dz.data_ptr(), Cout, npix, Cout, mean.data_ptr(),
         invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, sums.data_ptr(), ops.STAT_REPS, dg_ref.data_ptr(), db_ref.data_ptr(),
         1.0, torch.cuda.current_stream().cuda_stream)
    d = geo.desc(dt, geo.Cin_k, Cout)
    dw_ref = torch.zeros((Cout, 144), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(d, xk, dz, dw_ref)
    dw = torch.zeros_like(dw_ref)
    dg, db = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    ops.stem_bn_wgrad(d, xk, z, da, mean, invstd, gamma, beta, act, sums, dw, dg, db)
    torch.cuda.synchronize()
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    assert float((dw - dw_ref).abs().max()) <= 2e-3 * float(dw_ref.abs().max())


@pytest.mark.parametrize("shape", [(2, 32, 64, 96), (3, 16, 70, 100), (2, 64, 24, 40)])
@pytest.mark.parametrize("act", [True, False])
def test_stem_forward_kernel_inference_epilogue(shape, act):
    """k_stem_fwd with the inference executor's epilogue (per-channel scale / shift = folded BatchNorm, optional SiLU) against
    torch on fp16-rounded operands: 3e-3 of the largest element (the fp16 output rounding)."""
    from ayolov2_amd import functional as F_, ops, _lib
    B, Cout, H, W = shape
    dt = torch.float16
    g = torch.Generator().manual_seed(B + Cout + H + int(act))
    x = torch.rand(B, 3, H, W, generator=g).half().float()
    wt = (torch.randn(Cout, 3, 6, 6, generator=g) / 108 ** 0.5).half().float()
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    geo = F_._Geometry((B, 3, H, W), wt.shape, (2, 2), (2, 2), dt)
    xk = F_._prepare_input(x.cuda(), geo, dt)
    w, _ = F_._WeightCache().get(wt.cuda().contiguous(memory_format=torch.channels_last), dt, Cout, geo.cin_pad)
    y = ops.new_act(B, Cout, geo.Ho, geo.Wo, dt, xk.device)
    y.fill_(float("nan"))
    ops.conv_fwd(geo.desc(dt, geo.Cin_k, Cout), xk, w, y, _lib.EPI_AFFINE_SILU if act else _lib.EPI_AFFINE, scale.cuda(), shift.cuda())
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, None, 2, 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act:
        ref = F.silu(ref)
    assert float((y.float().cpu() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
def test_wgrad_grouped_launch_vs_single_layer_and_bit_reproducible(dt):
    """ayolo_wgrad_group_run (one launch per tile class over the item list of several layers, split-K
    partials stored with plain stores, fixed-order reduction) against the single-layer entry on the same operands -- all three
    tile classes, 3x3 / stride 2 / 1x1 layers, one layer whose dy arrives through an override slot, overwrite and accumulate
    reductions -- and against torch; two runs of either route must agree BIT FOR BIT (no atomics anywhere)."""
    import ctypes
    from ayolov2_amd import _lib, ops
    from ayolov2_amd._lib import WgradJob
    g = torch.Generator(device="cuda").manual_seed(77)
    # (B, Cin, Cout, k, s, p, H, W)
    layers = [(4, 64, 32, 3, 1, 1, 24, 20), (4, 32, 64, 3, 2, 1, 40, 40), (3, 128, 160, 1, 1, 0, 20, 24), (2, 64, 256, 3, 1, 1, 10, 12),
              (4, 96, 48, 1, 1, 0, 16, 16)]
    ops_, wants = [], []
    for (B, Ci, Co, k, s, p, H, W) in layers:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(B, Ci, H, W, device="cuda", generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, Co, Ho, Wo, device="cuda", generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        want = torch.nn.grad.conv2d_weight(x.float().cpu(), (Co, Ci, k, k), dy.float().cpu(), stride=s, padding=p)
        d = ops.make_desc(dt, B, H, W, Ci, Ci, Co, Co, (k, k), (s, s), (p, p), Ho, Wo)
        ops_.append((d, x, dy))
        wants.append(want.permute(0, 2, 3, 1).contiguous())            # KRSC, the layout of dw
    n = len(layers)

    def single():
        outs = []
        for d, x, dy in ops_:
            dw = torch.zeros((d.Cout, d.kh, d.kw, d.Cin), dtype=torch.float32, device="cuda")
            ops.conv_wgrad(d, x, dy, dw, alpha=0.5)
            outs.append(dw)
        return outs

    lib = _lib.lib()
    base = [torch.full((d.Cout, d.kh, d.kw, d.Cin), 3.0, dtype=torch.float32, device="cuda") for d, _, _ in ops_]

    def grouped():
        dws = [b.clone() for b in base]
        arr = (WgradJob * n)()
        for k, ((d, x, dy), dw) in enumerate(zip(ops_, dws)):
            arr[k].conv = d
            arr[k].x, arr[k].dw = x.data_ptr(), dw.data_ptr()
            arr[k].dy = 0 if k == 2 else dy.data_ptr()                 # layer 2: dy through override slot 1
            arr[k].alpha, arr[k].dy_slot, arr[k].overwrite = 0.5, (1 if k == 2 else -1), (0 if k == 4 else 1)
        tb, wb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _lib.check(lib.ayolo_wgrad_group_size(arr, n, ctypes.byref(tb), ctypes.byref(wb)), "size")
        host = ctypes.create_string_buffer(tb.value)
        _lib.check(lib.ayolo_wgrad_group_build(arr, n, host, tb.value), "build")
        dev = torch.frombuffer(host, dtype=torch.uint8).clone().cuda()
        ws = torch.empty(max(wb.value, 16), dtype=torch.uint8, device="cuda")
        ovr = (ctypes.c_void_p * 2)(None, ops_[2][2].data_ptr())
        _lib.check(lib.ayolo_wgrad_group_run(ctypes.addressof(host), dev.data_ptr(), ws.data_ptr(), ws.numel(), ovr, 2,
                                            torch.cuda.current_stream().cuda_stream), "run")
        torch.cuda.synchronize()
        return dws

    s1, s2, g1, g2 = single(), single(), grouped(), grouped()
    tol = 1e-4 if dt == torch.float32 else 2e-3
    for k in range(n):
        assert torch.equal(s1[k], s2[k]) and torch.equal(g1[k], g2[k]), f"layer {k}: not bit-reproducible"
        want = 0.5 * wants[k]
        scale = float(want.abs().max())
        assert float((s1[k].cpu() - want).abs().max()) <= tol * scale, k
        got = g1[k].cpu() - (3.0 if k == 4 else 0.0)                     # layer 4 accumulates onto its old contents
        assert float((got - want).abs().max()) <= tol * scale + (1e-5 if k == 4 else 0.0), k


@pytest.mark.parametrize("shape", [(4, 64, 64, 40, 40, 1), (2, 128, 32, 24, 20, 1), (3, 512, 256, 10, 12, 1), (2, 256, 255, 16, 16, 0),
                                   (1, 48, 160, 33, 7, 1), (8, 32, 64, 64, 64, 1)])
def test_conv_transform_on_load_equals_materialised_route(shape):
    """ayolo_conv_fwd_xf (stage A): the 1x1 consumer reads the producer's pre-activation z and forms
    act(z * scale + shift) on the way to the MFMAs.  Against the two-launch route -- ayolo_affine_act writes the activation,
    ayolo_conv_fwd reads it -- output AND BatchNorm statistics must agree BIT FOR BIT (same operand bits, same MFMA order);
    the last case is the YOLOHead epilogue (fp32 logits + bias, Cout 255), one case has a channel-slice input (ld > C)."""
    from ayolov2_amd import ops, _lib
    B, Ci, Co, H, W, act = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    head = Co == 255
    ld = Ci + 32 if Ci == 128 else Ci                                   # channel slice of a wider z buffer
    zfull = torch.randn(B, ld, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    z = zfull[:, :Ci]
    scale = (torch.rand(Ci, device="cuda", generator=g) + 0.5).float()
    shift = torch.randn(Ci, device="cuda", generator=g).float()
    cop = (Co + 7) // 8 * 8
    w = torch.zeros(cop, 1, 1, Ci, device="cuda", dtype=torch.float16)
    w[:Co] = (torch.randn(Co, 1, 1, Ci, device="cuda", generator=g) / Ci ** 0.5).half()
    bias = torch.randn(cop, device="cuda", generator=g).float() if head else None
    cd = Co if head else cop                                            # YOLOHead: Cout = na * no exactly, rows padded to 8
    d = ops.make_desc(torch.float16, B, H, W, Ci, ld, cd, cop, (1, 1), (1, 1), (0, 0), H, W)
    a = ops.new_act(B, Ci, H, W, torch.float16, "cuda")
    ops.affine_act(z, a, scale, shift, act)
    da = ops.make_desc(torch.float16, B, H, W, Ci, Ci, cd, cop, (1, 1), (1, 1), (0, 0), H, W)
    outs = []
    for route in ("materialised", "on_load"):
        if head:
            y = torch.zeros((B, H, W, cop), dtype=torch.float32, device="cuda")
            stats = None
        else:
            y = ops.new_act(B, cop, H, W, torch.float16, "cuda")
            stats = torch.zeros((ops.STAT_REPS, 2 * cop), dtype=torch.float64, device="cuda")
        epi = _lib.EPI_HEAD if head else _lib.EPI_NONE
        if route == "materialised":
            ops.conv_fwd(da, a, w, y, epi, shift=bias, stats=stats, head_no=85 if head else 0)
        else:
            # store-back: the first channel tile also writes the activation it forms -- it must BE the pass's output
            back = torch.full_like(a, float("nan"))
            ops.conv_fwd_xf(d, z, scale, shift, act, w, y, epi, shift=bias, stats=stats, head_no=85 if head else 0, store=back)
            torch.cuda.synchronize()
            assert torch.equal(back, a), "store-back differs from the materialised activation"
        torch.cuda.synchronize()
        outs.append((y.clone(), None if stats is None else stats.sum(0).clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if not head:
        assert torch.equal(outs[0][1], outs[1][1])
    # and the route is right at all: against torch fp32 on the rounded activation
    u = z.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref_a = (u * torch.sigmoid(u) if act else u).half().float()
    ref = F.conv2d(ref_a.cpu(), w[:Co].permute(0, 3, 1, 2).float().cpu(), bias[:Co].cpu() if head else None)
    got = outs[1][0].permute(0, 3, 1, 2)[:, :Co].float().cpu() if head else outs[1][0][:, :Co].float().cpu()
    assert _rel_err(got, ref) <= 4e-3


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 40, 40, (1, 1), (True, True)), (2, 32, 96, 64, 24, 20, (1, 0), (True, False)),
                                  (3, 128, 128, 256, 10, 12, (0, 1), (False, True)), (2, 64, 32, 48, 33, 7, (1, 1), (True, True)),
                                  (3, 256, 256, 256, 10, 12, (1, 1), (True, True)), (2, 256, 256, 512, 9, 7, (1, 0), (True, False))])
def test_conv_transform_on_load_two_segments(case):
    """C3's cv3 reads [last Bottleneck output | cv2 half]: two input segments from two buffers with their own channel strides,
    each virtual (the producer's z, transformed on load) or a plain activation.  Forward vs the conv over the materialised concat
    -- bit for bit, statistics included -- and the weight gradient as two grouped jobs writing the column blocks of one dw
    against the single-layer weight gradient over the materialised concat (bit for bit: same operand bits, same split-K order
    per column block is NOT guaranteed, so 1e-6 relative)."""
    import ctypes
    from ayolov2_amd import ops, _lib
    from ayolov2_amd._lib import WgradJob
    B, C0, C1, Co, H, W, acts, virt = case
    g = torch.Generator(device="cuda").manual_seed(sum(case[:6]) + 5)
    Ci = C0 + C1
    z0full = torch.randn(B, C0 + 16, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    z1full = torch.randn(B, 2 * C1, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    z0, z1 = z0full[:, :C0], z1full[:, C1:]                              # channel slices of wider buffers
    scale = (torch.rand(Ci, device="cuda", generator=g) + 0.5).float()
    shift = torch.randn(Ci, device="cuda", generator=g).float()
    cat = ops.new_act(B, Ci, H, W, torch.float16, "cuda")
    for zz, c0, C, act, v in ((z0, 0, C0, acts[0], virt[0]), (z1, C0, C1, acts[1], virt[1])):
        if v:
            ops.affine_act(zz, cat[:, c0:c0 + C], scale[c0:c0 + C].contiguous(), shift[c0:c0 + C].contiguous(), act)
        else:
            cat[:, c0:c0 + C] = zz                                       # a plain segment IS the activation
    w = (torch.randn(Co, 1, 1, Ci, device="cuda", generator=g) / Ci ** 0.5).half()
    d = ops.make_desc(torch.float16, B, H, W, Ci, Ci, Co, Co, (1, 1), (1, 1), (0, 0), H, W)
    outs = []
    for route in ("materialised", "on_load"):
        y = ops.new_act(B, Co, H, W, torch.float16, "cuda")
        stats = torch.zeros((ops.STAT_REPS, 2 * Co), dtype=torch.float64, device="cuda")
        if route == "materialised":
            ops.conv_fwd(d, cat, w, y, _lib.EPI_NONE, stats=stats)
        else:
            back = cat.clone()
            for c0, C, v in ((0, C0, virt[0]), (C0, C1, virt[1])):
                if v:
                    back[:, c0:c0 + C] = float("nan")                     # virtual segments are written back, plain ones are left alone
            ops.conv_fwd_xf(d, [(z0, acts[0], virt[0]), (z1, acts[1], virt[1])], scale, shift, 0, w, y, _lib.EPI_NONE, stats=stats, store=back)
            torch.cuda.synchronize()
            assert torch.equal(back, cat), "store-back differs from the materialised concat"
        torch.cuda.synchronize()
        outs.append((y.clone(), stats.sum(0).clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # weight gradient: two jobs, one per segment, into the column blocks of one [Co][Ci] matrix
    dy = torch.randn(B, Co, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    dw_ref = torch.zeros((Co, 1, 1, Ci), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(d, cat, dy, dw_ref)
    dw = torch.full((Co, 1, 1, Ci), 7.0, dtype=torch.float32, device="cuda")
    lib = _lib.lib()
    arr = (WgradJob * 2)()
    for k, (zz, c0, C, act, v) in enumerate(((z0, 0, C0, acts[0], virt[0]), (z1, C0, C1, acts[1], virt[1]))):
        arr[k].conv = ops.make_desc(torch.float16, B, H, W, C, ops.nhwc_info(zz)[4], Co, Co, (1, 1), (1, 1), (0, 0), H, W)
        arr[k].x, arr[k].dy, arr[k].dw = zz.data_ptr(), dy.data_ptr(), dw.data_ptr() + 4 * c0
        arr[k].alpha, arr[k].dy_slot, arr[k].overwrite, arr[k].dw_ld = 1.0, -1, 1, Ci
        if v:
            sc, sh = scale[c0:c0 + C], shift[c0:c0 + C]
            arr[k].xscale, arr[k].xshift, arr[k].xact = sc.data_ptr(), sh.data_ptr(), act
    tb, wb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(lib.ayolo_wgrad_group_size(arr, 2, ctypes.byref(tb), ctypes.byref(wb)), "size")
    host = ctypes.create_string_buffer(tb.value)
    _lib.check(lib.ayolo_wgrad_group_build(arr, 2, host, tb.value), "build")
    dev = torch.frombuffer(host, dtype=torch.uint8).clone().cuda()
    ws = torch.empty(max(wb.value, 16), dtype=torch.uint8, device="cuda")
    _lib.check(lib.ayolo_wgrad_group_run(ctypes.addressof(host), dev.data_ptr(), ws.data_ptr(), ws.numel(), None, 0,
                                        torch.cuda.current_stream().cuda_stream), "run")
    torch.cuda.synchronize()
    assert float((dw - dw_ref).abs().max()) <= 1e-5 * float(dw_ref.abs().max())


def test_batched_weight_cast_equals_per_layer_cast():
    """ayolo_cast_weights (all layers of a plan in one launch, 64 x 64 tiles with an LDS transpose, 16-byte rows on interior
    tiles) against ayolo_cast_weight (one thread per element): both copies bit-identical, for layers that take the vector
    path (multiples of 64), ragged ones (3 input channels, 255 outputs, padded rows), a 3x3 layer and a transposed copy that
    is a column slice of a wider matrix."""
    from ayolov2_amd import ops
    from ayolov2_amd._lib import call
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    cases = [(64, 1, 64, 64, 64, 0), (128, 9, 192, 128, 192, 0), (255, 1, 128, 256, 128, 0), (32, 18, 8, 32, 8, 0), (80, 9, 3, 96, 32, 0),
             (64, 1, 128, 64, 128, 256), (48, 1, 100, 64, 104, 0)]
    job_t = np.dtype([("w32", "<u8"), ("w", "<u8"), ("wt", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"),
                      ("Cout_pad", "<i4"), ("Cin_pad", "<i4"), ("wt_ld", "<i4")])
    jobs = np.zeros(len(cases), dtype=job_t)
    keep, want = [], []
    for k, (co, taps, ci, cop, cip, ld) in enumerate(cases):
        w32 = torch.randn(co, taps, ci, generator=g).to(dev)
        w = torch.full((cop, taps, cip), 7.0, dtype=torch.float16, device=dev)
        wide = torch.full((cip, taps, ld or cop), 7.0, dtype=torch.float16, device=dev)
        wt = wide[:, :, 64:64 + cop] if ld else wide
        rw = torch.empty(cop, taps, cip, dtype=torch.float16, device=dev)
        rt = torch.empty(cip, taps, cop, dtype=torch.float16, device=dev)
        call("ayolo_cast_weight", w32.data_ptr(), co, taps, 1, ci, cop, cip, ops.dtype_code(torch.float16), rw.data_ptr(), rt.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        jobs[k] = (w32.data_ptr(), w.data_ptr(), wt.data_ptr(), co, taps, ci, cop, cip, ld)
        keep.append((w32, w, wide, wt))
        want.append((rw, rt))
    tab = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
    call("ayolo_cast_weights", tab.data_ptr(), len(cases), ops.dtype_code(torch.float16), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for (w32, w, wide, wt), (rw, rt), case in zip(keep, want, cases):
        assert torch.equal(w, rw), case
        assert torch.equal(wt, rt), case
        if case[5]:
            assert bool((wide[:, :, :64] == 7.0).all()) and bool((wide[:, :, 64 + case[3]:] == 7.0).all()), case


W3_SHAPES = [  # (B, Cin, Cout, stride, H, W)
    (2, 32, 32, 1, 24, 40),      # one block: eight wavefronts split the sub-steps
    (3, 64, 64, 1, 20, 20),      # 2 x 2 blocks, four-row steps that straddle images
    (2, 64, 64, 1, 80, 80),      # the 80 x 80 Bottleneck layer of YOLOv5s: column strips
    (2, 128, 128, 1, 40, 40),    # 2 x 2 tiles of 2 x 2 blocks
    (2, 40, 72, 1, 12, 16),      # ragged channel blocks
    (4, 256, 256, 1, 20, 20),
    (2, 32, 64, 2, 64, 96),      # stride 2: odd / even column planes
    (2, 64, 128, 2, 40, 40),
    (2, 128, 32, 2, 9, 13),      # odd map: bottom padding row, strip wider than the map
]


@pytest.mark.parametrize("shape", W3_SHAPES)
def test_wgrad3_patch_kernel_vs_torch_and_generic(shape, monkeypatch):
    """k_wgrad3 (csrc/wgrad3.hip: the 3x3 weight gradient on a once-staged input patch -- retired from the default route in round 6,
    AYOLO_WGRAD3 = 1 / 2 routes the stride-1 / all 3x3 layers to it) through
    the C ABI's ayolo_conv_wgrad: against plain PyTorch fp32 on the CPU (same fp16-rounded operands, fp32 accumulation on both
    sides), against the generic k_wgrad on the same device buffers, bit-reproducible from run to run, and accumulating into dw
    (alpha) like the generic entry."""
    import ctypes
    from ayolov2_amd import _lib, ops, functional as F_
    B, Cin, Cout, s, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g).half().float()
    dt = torch.float16
    geo = F_._Geometry((B, Cin, H, W), (Cout, Cin, 3, 3), (s, s), (1, 1), dt)
    dy = torch.randn(B, Cout, geo.Ho, geo.Wo, generator=g).half().float()
    wr = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, wr, None, s, 1).backward(dy)
    ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)                       # [n][(dh, dw)][c]
    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last)
    dyg = dy.cuda().to(dt).contiguous(memory_format=torch.channels_last)
    d = geo.desc(dt, Cin, Cout)
    out = {}
    for name, env in (("patch", "2"), ("generic", "0")):
        monkeypatch.setenv("AYOLO_WGRAD3", env)
        if name == "patch":
            gq = (ctypes.c_int64 * 24)()
            _lib.check(_lib.lib().ayolo_wgrad3_geometry(d, gq, 24), "ayolo_wgrad3_geometry")
        dw = torch.zeros((Cout, 9 * Cin), dtype=torch.float32, device="cuda")
        ops.conv_wgrad(d, xg, dyg, dw)
        dw2 = torch.zeros_like(dw)
        ops.conv_wgrad(d, xg, dyg, dw2)
        assert torch.equal(dw, dw2), name + ": not bit-reproducible"
        ops.conv_wgrad(d, xg, dyg, dw2, alpha=0.5)                               # accumulates
        torch.cuda.synchronize()
        out[name] = dw.cpu()
        assert _rel_err(dw2.cpu(), 1.5 * out[name]) < 1e-6, name
    assert _rel_err(out["patch"], ref) < 2e-4, _rel_err(out["patch"], ref)
    assert _rel_err(out["generic"], ref) < 2e-4
    assert _rel_err(out["patch"], out["generic"]) < 2e-5
