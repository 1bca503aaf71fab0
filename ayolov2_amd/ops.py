"""Tensor-level wrappers over the C ABI.  PyTorch is used here only for device memory and streams.

Activation convention: a logical (B, C, H, W) torch tensor whose memory is NHWC (``channels_last`` strides),
possibly a channel slice of a wider buffer (then ``ld`` = channel count of the parent).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import F16, F32, ConvDesc, call


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return F16
    if dt == torch.float32:
        return F32
    raise _lib.AyoloError(f"unsupported dtype {dt}")


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.AyoloError(f"{what}: the HIP path needs a tensor on the GPU (got {t.device}); there is no CPU fallback")


def nhwc_info(t: torch.Tensor) -> Tuple[int, int, int, int, int]:
    """(B, C, H, W, ld) of an NHWC-in-memory tensor (possibly a channel slice of a wider buffer: ld > C)."""
    if t.dim() != 4:
        raise _lib.AyoloError(f"expected a 4-D activation, got shape {tuple(t.shape)}")
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if W > 1:
        ld = sw
    elif H > 1:
        ld = sh
    elif B > 1:
        ld = sb
    else:
        ld = C
    ok = (C == 1 or sc == 1) and ld >= C
    if W > 1 and H > 1:
        ok = ok and sh == W * ld
    if B > 1:
        ok = ok and sb == H * W * ld
    if not ok:
        raise _lib.AyoloError(f"tensor is not NHWC in memory: shape {tuple(t.shape)} strides {t.stride()}")
    return B, C, H, W, ld


def is_nhwc(t: torch.Tensor) -> bool:
    try:
        nhwc_info(t)
        return True
    except _lib.AyoloError:
        return False


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return t itself if it already is NHWC in memory (16-byte aligned), else a channels_last copy."""
    if is_nhwc(t) and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous(memory_format=torch.channels_last)


def new_act(B: int, C: int, H: int, W: int, dtype: torch.dtype, device) -> torch.Tensor:
    return torch.empty((B, C, H, W), dtype=dtype, device=device, memory_format=torch.channels_last)


def make_desc(dt: torch.dtype, B, H, W, Cin, ldx, Cout, ldy, k, s, p, Ho, Wo) -> ConvDesc:
    return ConvDesc(dtype_code(dt), B, H, W, Cin, ldx, Cout, ldy, k[0], k[1], s[0], s[1], p[0], p[1], Ho, Wo)


# --------------------------------------------------------------------------------------------------
STAT_REPS = 8    # replicas of the per-channel BN accumulators (see ayolo.h)


class _ZeroArena:
    """Bump allocator of zero-initialised fp32 scratch (BN accumulators).  One fill kernel per training step
    (``reset`` at the start of the model forward) replaces one ``torch.zeros`` launch per layer per pass.  A slice
    is only valid inside the call that took it (all work is stream-ordered on the current stream)."""

    def __init__(self):
        self.buf = None
        self.off = 0

    def reset(self):
        if self.buf is not None and self.off > 0:
            self.buf[:self.off].zero_()
        self.off = 0

    def take(self, n: int, device) -> torch.Tensor:
        n = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            self.buf = torch.zeros(max(1 << 21, 2 * n), dtype=torch.float32, device=device)
            self.off = 0
        out = self.buf[self.off:self.off + n]
        self.off += n
        return out


ARENA = _ZeroArena()


def zero_stats(C: int, device) -> torch.Tensor:
    """Zeroed BatchNorm accumulators [STAT_REPS][2 * C], fp64 (ayolo.h: every statistics accumulator is double)."""
    n = STAT_REPS * 2 * C
    return ARENA.take(2 * n, device)[:2 * n].view(torch.float64).view(STAT_REPS, 2 * C)


def conv_fwd(desc: ConvDesc, x, w, y, epilogue=_lib.EPI_NONE, scale=None, shift=None, stats=None, head_no=0):
    reps = stats.shape[0] if (stats is not None and stats.dim() == 2) else 1
    call("ayolo_conv_fwd", desc, _ptr(x), _ptr(w), _ptr(y), epilogue, _ptr(scale), _ptr(shift), _ptr(stats), reps, head_no,
         _stream())


def conv_fwd_xf(desc: ConvDesc, z, xscale, xshift, xact, w, y, epilogue=_lib.EPI_NONE, shift=None, stats=None, head_no=0, store=None):
    """1x1 conv over a (partly) VIRTUAL input (ayolo_conv_fwd_xf: transform on load).  One segment: `z` is the producer's
    pre-activation, the operand is act(z * xscale + xshift).  Two segments: z = [(tensor, act, virt), (tensor, act, virt)] --
    channel slices side by side in the conv's input channels; virt = False marks a plain (materialised) activation."""
    from ._lib import XfSeg
    segs = z if isinstance(z, (list, tuple)) else [(z, xact, True)]
    arr = (XfSeg * len(segs))()
    for k, (t, act, virt) in enumerate(segs):
        _, C, _, _, ld = nhwc_info(t)
        arr[k].x, arr[k].ld, arr[k].C, arr[k].act, arr[k].virt = t.data_ptr(), ld, C, int(act), int(bool(virt))
    reps = stats.shape[0] if (stats is not None and stats.dim() == 2) else 1
    lds = nhwc_info(store)[4] if store is not None else 0      # store: the materialised activation the first channel tile writes back
    call("ayolo_conv_fwd_xf", desc, arr, len(segs), _ptr(xscale), _ptr(xshift), None, 0, _ptr(store), lds, _ptr(w), _ptr(y), epilogue,
         _ptr(shift), _ptr(stats), reps, head_no, _stream())


def conv_dgrad(desc: ConvDesc, dy, wt, dx, accumulate=False):
    call("ayolo_conv_dgrad", desc, _ptr(dy), _ptr(wt), _ptr(dx), int(accumulate), _stream())


def conv_dgrad_bn(desc: ConvDesc, dy, wt, dx, segs, act: int, accumulate=False):
    """dgrad with the BatchNorm-backward sums of the block(s) that produced x in its epilogue (ayolo_conv_dgrad_bn).
    segs: [(z, save_mean|save_invstd [2C], gamma, beta, sums [reps][2C], c0)] -- channels [c0, c0 + C) of dx."""
    from ._lib import BnSeg
    arr = (BnSeg * len(segs))()
    reps = None
    for k, (z, mi, gamma, beta, sums, c0) in enumerate(segs):
        _, C, _, _, ldz = nhwc_info(z)
        arr[k].z, arr[k].mean_invstd, arr[k].gamma, arr[k].beta, arr[k].sums = _ptr(z), _ptr(mi), _ptr(gamma), _ptr(beta), _ptr(sums)
        arr[k].ldz, arr[k].c0, arr[k].C = ldz, int(c0), C
        r = sums.shape[0] if sums.dim() == 2 else 1
        assert reps in (None, r)
        reps = r
    call("ayolo_conv_dgrad_bn", desc, _ptr(dy), _ptr(wt), _ptr(dx), int(accumulate), arr, len(segs), int(act), reps, _stream())


_WGRAD_WS: dict = {}         # (device, stream) -> split-K workspace of the single-layer weight gradient (grown on demand)


def wgrad_workspace(nbytes: int, device) -> "torch.Tensor":
    """Workspace for ayolo_conv_wgrad's split-K partial sums: one buffer per (device, stream) -- launches on a stream are
    ordered, so consecutive layers can share it -- grown to the largest request seen."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WGRAD_WS[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        while len(_WGRAD_WS) > 16:
            _WGRAD_WS.pop(next(iter(_WGRAD_WS)))
    return ws


def conv_wgrad(desc: ConvDesc, x, dy, dw, alpha=1.0):
    """dw += alpha * sum_pixels dy (x) x (ayolo_conv_wgrad: split-K partials in a workspace, fixed-order reduction)."""
    need = int(_lib.lib().ayolo_conv_wgrad_workspace(desc))
    ws = wgrad_workspace(need, dw.device) if need else None
    call("ayolo_conv_wgrad", desc, _ptr(x), _ptr(dy), _ptr(dw), float(alpha), _ptr(ws), ws.numel() if ws is not None else 0, _stream())


def cast_weight(w32_krsc: torch.Tensor, Cout, kh, kw, Cin, Cout_pad, Cin_pad, dtype: torch.dtype, want_w=True,
                want_wt=True):
    """fp32 [Cout][kh][kw][Cin] -> (w [Cout_pad][kh][kw][Cin_pad], wt [Cin_pad][kh][kw][Cout_pad]) of `dtype`."""
    dev = w32_krsc.device
    w = torch.empty((Cout_pad, kh, kw, Cin_pad), dtype=dtype, device=dev) if want_w else None
    wt = torch.empty((Cin_pad, kh, kw, Cout_pad), dtype=dtype, device=dev) if want_wt else None
    call("ayolo_cast_weight", _ptr(w32_krsc), Cout, kh, kw, Cin, Cout_pad, Cin_pad, dtype_code(dtype), _ptr(w), _ptr(wt),
         _stream())
    return w, wt


def bn_finalize(stats, C, count, gamma, beta, eps, momentum, running_mean, running_var):
    dev = stats.device
    save_mean = torch.empty(C, dtype=torch.float32, device=dev)
    save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty(C, dtype=torch.float32, device=dev)
    reps = stats.shape[0] if stats.dim() == 2 else 1
    call("ayolo_bn_finalize", _ptr(stats), reps, C, float(count), _ptr(gamma), _ptr(beta), float(eps), float(momentum),
         _ptr(running_mean), _ptr(running_var), _ptr(save_mean), _ptr(save_invstd), _ptr(scale), _ptr(shift), _stream())
    from ._lib import bump_versions
    bump_versions((running_mean, running_var))
    return save_mean, save_invstd, scale, shift


def affine_act(z, a, scale, shift, act: int):
    B, C, H, W, ldz = nhwc_info(z)
    _, _, _, _, lda = nhwc_info(a)
    call("ayolo_affine_act", dtype_code(z.dtype), _ptr(z), ldz, _ptr(a), lda, B * H * W, C, _ptr(scale), _ptr(shift), act,
         _stream())


def bn_act_bwd(z, da, save_mean, save_invstd, gamma, beta, act: int, want_param_grads=True, dres=None, res_accumulate=False):
    """Returns (dz, dgamma, dbeta) for a = act(bn(z)) with batch statistics.  `dres` (NHWC tensor of da's shape, any channel
    stride): the gradient buffer of a shortcut fed by the same output, dres (+)= da written by the apply pass."""
    B, C, H, W, ldz = nhwc_info(z)
    _, _, _, _, ldda = nhwc_info(da)
    npix = B * H * W
    dev = z.device
    sums = zero_stats(C, dev)
    dz = new_act(B, C, H, W, z.dtype, dev)
    dt = dtype_code(z.dtype)
    call("ayolo_bn_act_bwd_reduce", dt, _ptr(z), ldz, _ptr(da), ldda, npix, C, _ptr(save_mean), _ptr(save_invstd),
         _ptr(gamma), _ptr(beta), act, _ptr(sums), STAT_REPS, _stream())
    dgamma = torch.empty(C, dtype=torch.float32, device=dev) if want_param_grads else None
    dbeta = torch.empty(C, dtype=torch.float32, device=dev) if want_param_grads else None
    if dres is None:
        call("ayolo_bn_act_bwd_apply", dt, _ptr(z), ldz, _ptr(da), ldda, _ptr(dz), C, npix, C, _ptr(save_mean),
             _ptr(save_invstd), _ptr(gamma), _ptr(beta), act, _ptr(sums), STAT_REPS, _ptr(dgamma), _ptr(dbeta), 1.0, _stream())
    else:
        call("ayolo_bn_act_bwd_apply_res", dt, _ptr(z), ldz, _ptr(da), ldda, _ptr(dz), C, npix, C, _ptr(save_mean),
             _ptr(save_invstd), _ptr(gamma), _ptr(beta), act, _ptr(sums), STAT_REPS, _ptr(dgamma), _ptr(dbeta), 1.0,
             _ptr(dres), nhwc_info(dres)[4], int(bool(res_accumulate)), _stream())
    return dz, dgamma, dbeta


def stem_bn_wgrad(desc, x, z, da, save_mean, save_invstd, gamma, beta, act: int, sums, dw, dgamma=None, dbeta=None,
                  alpha: float = 1.0):
    """Backward of the stem block in one launch (ayolo_stem_bn_wgrad): dw += alpha * weight gradient with
    dz = bn_act_backward(da, z; sums) formed on the fly; dgamma / dbeta from the sums.  `desc`: packed-stem descriptor whose
    ldy is the row stride of da."""
    call("ayolo_stem_bn_wgrad", desc, _ptr(x), _ptr(z), nhwc_info(z)[4], _ptr(da), _ptr(save_mean), _ptr(save_invstd),
         _ptr(gamma), _ptr(beta), act, _ptr(sums), sums.shape[0], _ptr(dw), _ptr(dgamma), _ptr(dbeta), alpha, 1.0, _stream())


def maxpool_fwd(x, k: int, y=None, want_argmax=True):
    B, C, H, W, ldx = nhwc_info(x)
    if y is None:
        y = new_act(B, C, H, W, x.dtype, x.device)
    _, _, _, _, ldy = nhwc_info(y)
    arg = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device) if want_argmax else None
    call("ayolo_maxpool_fwd", dtype_code(x.dtype), _ptr(x), ldx, _ptr(y), ldy, _ptr(arg), B, H, W, C, k, _stream())
    return y, arg


def maxpool_bwd(argmax, dy, k: int, dx=None, accumulate=False):
    B, C, H, W, lddy = nhwc_info(dy)
    if dx is None:
        dx = new_act(B, C, H, W, dy.dtype, dy.device)
    _, _, _, _, lddx = nhwc_info(dx)
    call("ayolo_maxpool_bwd", dtype_code(dy.dtype), _ptr(argmax), _ptr(dy), lddy, _ptr(dx), lddx, B, H, W, C, k,
         int(accumulate), _stream())
    return dx


def sppf_pool_fwd(cat, C: int, want_argmax=True):
    """kindle SPPF's three chained 5 x 5 max-pools in one launch on the concat buffer `cat` (4 C channels: x | y1 | y2 | y3;
    slice 0 is read, slices 1..3 written).  Returns the three window-position planes uint8[3][B][H][W][C] (or None)."""
    B, C4, H, W, ld = nhwc_info(cat)
    assert C4 == 4 * C and cat.dtype == torch.float16
    arg = torch.empty((3, B, H, W, C), dtype=torch.uint8, device=cat.device) if want_argmax else None
    call("ayolo_sppf_pool_fwd", dtype_code(cat.dtype), _ptr(cat), ld, _ptr(arg), B, H, W, C, _stream())
    return arg


def sppf_pool_bwd(argmax, dcat, C: int):
    """Backward of the cascade: slice 0 of `dcat` (the concat buffer's gradient) becomes d(x)."""
    B, C4, H, W, ld = nhwc_info(dcat)
    assert C4 == 4 * C and dcat.dtype == torch.float16
    call("ayolo_sppf_pool_bwd", dtype_code(dcat.dtype), _ptr(argmax), _ptr(dcat), ld, B, H, W, C, _stream())
    return dcat[:, :C]


def upsample2x_fwd(x, y=None):
    B, C, H, W, ldx = nhwc_info(x)
    if y is None:
        y = new_act(B, C, 2 * H, 2 * W, x.dtype, x.device)
    _, _, _, _, ldy = nhwc_info(y)
    call("ayolo_upsample2x_fwd", dtype_code(x.dtype), _ptr(x), ldx, _ptr(y), ldy, B, H, W, C, _stream())
    return y


def upsample2x_bwd(dy, dx=None, accumulate=False):
    B, C, H2, W2, lddy = nhwc_info(dy)
    H, W = H2 // 2, W2 // 2
    if dx is None:
        dx = new_act(B, C, H, W, dy.dtype, dy.device)
    _, _, _, _, lddx = nhwc_info(dx)
    call("ayolo_upsample2x_bwd", dtype_code(dy.dtype), _ptr(dy), lddy, _ptr(dx), lddx, B, H, W, C, int(accumulate),
         _stream())
    return dx


def pack_input(img: torch.Tensor, dtype: torch.dtype, Cpad: int) -> torch.Tensor:
    """(B, C, H, W) fp32 NCHW -> (B, Cpad, H, W) logical, NHWC in memory, channels >= C zero."""
    B, C, H, W = img.shape
    img = img.contiguous()
    if img.dtype != torch.float32:
        img = img.float()
    y = new_act(B, Cpad, H, W, dtype, img.device)
    call("ayolo_pack_input", _ptr(img), B, C, H, W, dtype_code(dtype), _ptr(y), Cpad, _stream())
    return y


def copy2d(x, y, accumulate=False):
    B, C, H, W, ldx = nhwc_info(x)
    _, _, _, _, ldy = nhwc_info(y)
    call("ayolo_copy2d", dtype_code(x.dtype), _ptr(x), ldx, _ptr(y), ldy, B * H * W, C, int(accumulate), _stream())
    return y


def head_decode(raw, anchors_px, stride: float, out, row_off: int):
    B, na, ny, nx, no = raw.shape
    if raw.stride(4) != 1:
        raw = raw.contiguous()
    strides = (_lib.c_int64 * 4)(*raw.stride()[:4])
    call("ayolo_head_decode", _ptr(raw), strides, B, na, ny, nx, no, _ptr(anchors_px), float(stride), _ptr(out),
         out.shape[1], row_off, _stream())


def head_grad_pack(draw, dtype: torch.dtype, ldz: int, want_bias=True):
    B, na, ny, nx, no = draw.shape
    draw = draw.contiguous().float()
    dz = new_act(B, ldz, ny, nx, dtype, draw.device)
    dbias = torch.zeros(na * no, dtype=torch.float32, device=draw.device) if want_bias else None
    call("ayolo_head_grad_pack", _ptr(draw), B, na, ny, nx, no, dtype_code(dtype), _ptr(dz), ldz, _ptr(dbias), _stream())
    return dz, dbias
