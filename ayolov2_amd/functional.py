"""Autograd bindings: each Function's forward/backward is a handful of calls into libayolo_hip.so.

The math mirrors what torch/cuDNN did under kindle's modules in the reference:
Conv2d(bias=False) -> BatchNorm2d (batch statistics in training) -> SiLU, MaxPool2d(5,1,2), nearest 2x upsample,
and the YOLOHead 1x1 conv with bias.
"""
from __future__ import annotations

import copy
from typing import Optional

import torch
from torch import nn

from . import _lib, ops
from ._lib import EPI_AFFINE, EPI_AFFINE_SILU, EPI_HEAD, EPI_NONE


def compute_dtype(weight: torch.Tensor) -> torch.dtype:
    """fp16 under autocast (reference trains with torch.cuda.amp, yolo_trainer.py:322) or for .half() models."""
    if torch.is_autocast_enabled():
        return torch.float16
    return torch.float16 if weight.dtype == torch.float16 else torch.float32


def _ce(dt: torch.dtype) -> int:
    return 8 if dt == torch.float16 else 4


def _pair_(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class _WeightCache:
    """compute-dtype copies of a conv weight ([Cout_pad][kh][kw][Cin_pad] and its transpose), rebuilt only
    when the master weight changes (once per optimiser step)."""

    def __init__(self):
        self.key = None
        self.w = None
        self.wt = None

    def get(self, weight: torch.Tensor, dt: torch.dtype, cout_pad: int, cin_pad: int):
        key = (weight.data_ptr(), weight._version, dt, cout_pad, cin_pad, weight.device)
        if key != self.key:
            Cout, Cin, kh, kw = weight.shape
            w32 = weight.detach().permute(0, 2, 3, 1)      # KRSC view of an OIHW tensor
            if w32.dtype != torch.float32 or not w32.is_contiguous():
                w32 = w32.float().contiguous()
            self.w, self.wt = ops.cast_weight(w32, Cout, kh, kw, Cin, cout_pad, cin_pad, dt)
            self.key = key
        return self.w, self.wt


class _Geometry:
    """How a conv is presented to the kernels (incl. the stem's pixel-pair packing)."""

    def __init__(self, x_shape, weight_shape, stride, padding, dt, allow_packed_stem: bool = True):
        B, Cx, H, W = x_shape
        Cout, Cin, kh, kw = weight_shape
        sh, sw = stride
        ph, pw = padding
        ce = _ce(dt)
        self.packed_stem = False
        self.cin_pad = _round_up(Cin, ce)
        self.B, self.Cout = B, Cout
        self.Ho = (H + 2 * ph - kh) // sh + 1
        self.Wo = (W + 2 * pw - kw) // sw + 1
        if allow_packed_stem and Cin % ce != 0 and Cin <= 4 and dt == torch.float16 and kw % 2 == 0 and sw == 2 and pw % 2 == 0 and W % 2 == 0:
            # 4-channel NHWC image viewed as (B, H, W/2, 8): k6/s2/p2 along W becomes k3/s1/p1 over pixel pairs
            self.packed_stem = True
            self.cin_pad = 4
            self.kdims = (kh, kw // 2)
            self.sdims = (sh, 1)
            self.pdims = (ph, pw // 2)
            self.H, self.W, self.Cin_k = H, W // 2, 8
        else:
            self.kdims, self.sdims, self.pdims = (kh, kw), (sh, sw), (ph, pw)
            self.H, self.W, self.Cin_k = H, W, self.cin_pad
        self.needs_pack = Cin % ce != 0

    def desc(self, dt, ldx, ldy, cout=None):
        return ops.make_desc(dt, self.B, self.H, self.W, self.Cin_k, ldx, cout if cout is not None else self.Cout, ldy,
                             self.kdims, self.sdims, self.pdims, self.Ho, self.Wo)


def _prepare_input(x: torch.Tensor, geo: _Geometry, dt: torch.dtype) -> torch.Tensor:
    """Returns the NHWC tensor the kernels read (logical (B, Cin_k, H, W_k))."""
    if geo.needs_pack and (x.shape[1] > 4 or x.requires_grad):
        # an activation whose channel count is off the vector width (the rank of a Tucker factor): zero-padded copy
        B, C, H, W = x.shape
        xp = torch.zeros((B, geo.cin_pad, H, W), dtype=dt, device=x.device).contiguous(memory_format=torch.channels_last)
        xp[:, :C] = x.to(dt)
        return xp
    if geo.needs_pack:
        xp = ops.pack_input(x, dt, geo.cin_pad)          # (B, cin_pad, H, W) NHWC
        if geo.packed_stem:
            B, _, H, W = xp.shape
            xp = xp.as_strided((B, 8, H, W // 2), (H * W * 4, 1, W * 4, 8))
        return xp
    if x.dtype != dt:
        x = x.to(dt)
    return ops.to_nhwc(x)


class ConvBnActFn(torch.autograd.Function):
    """a = act(bn_train(conv(x, w)))  -- training-mode forward with saved statistics, and its backward."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, stride, padding, eps, momentum, act, cache):
        dt = compute_dtype(weight)
        # pixel-pair packing is for the image only: an input that wants a gradient keeps the plain (zero-padded) geometry
        geo = _Geometry(x.shape, weight.shape, stride, padding, dt, allow_packed_stem=not x.requires_grad)
        xk = _prepare_input(x, geo, dt)
        _, _, _, _, ldx = ops.nhwc_info(xk)
        Cout = weight.shape[0]
        w, wt = cache.get(weight, dt, Cout, geo.cin_pad)
        dev = xk.device
        z = ops.new_act(geo.B, Cout, geo.Ho, geo.Wo, dt, dev)
        stats = ops.zero_stats(Cout, dev)
        d = geo.desc(dt, ldx, Cout)
        ops.conv_fwd(d, xk, w, z, EPI_NONE, stats=stats)
        count = geo.B * geo.Ho * geo.Wo
        g32 = gamma.float() if gamma is not None else None
        b32 = beta.float() if beta is not None else None
        save_mean, save_invstd, scale, shift = ops.bn_finalize(stats, Cout, count, g32, b32, eps, momentum,
                                                               running_mean, running_var)
        a = ops.new_act(geo.B, Cout, geo.Ho, geo.Wo, dt, dev)
        ops.affine_act(z, a, scale, shift, act)
        ctx.save_for_backward(xk, z, save_mean, save_invstd, g32, b32, wt)
        ctx.geo, ctx.dt, ctx.act, ctx.ldx = geo, dt, act, ldx
        ctx.weight_shape = tuple(weight.shape)
        return a

    @staticmethod
    def backward(ctx, da):
        xk, z, save_mean, save_invstd, g32, b32, wt = ctx.saved_tensors
        geo, dt, act = ctx.geo, ctx.dt, ctx.act
        Cout, Cin, kh, kw = ctx.weight_shape
        if da.dtype != dt:
            da = da.to(dt)
        da = ops.to_nhwc(da)
        dz, dgamma, dbeta = ops.bn_act_bwd(z, da, save_mean, save_invstd, g32, b32, act)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.new_act(geo.B, geo.cin_pad, geo.H, geo.W, dt, dz.device)
            ops.conv_dgrad(geo.desc(dt, geo.cin_pad, Cout), dz, wt, dx)
            if geo.cin_pad != Cin:
                dx = dx[:, :Cin]
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(geo, dt, xk, ctx.ldx, dz, Cout, Cout, Cin, kh, kw)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None


def _wgrad(geo: _Geometry, dt, xk, ldx, dz, ldy, Cout, Cin, kh, kw, cout_pad=None) -> torch.Tensor:
    """fp32 weight gradient as a logical OIHW tensor with channels_last (KRSC) strides."""
    cout_pad = cout_pad or Cout
    K = geo.kdims[0] * geo.kdims[1] * geo.Cin_k
    dwk = torch.zeros((cout_pad, K), dtype=torch.float32, device=dz.device)
    ops.conv_wgrad(geo.desc(dt, ldx, ldy, cout=cout_pad), xk, dz, dwk)
    dwk = dwk[:Cout].view(Cout, kh, kw, geo.cin_pad)[..., :Cin]
    return dwk.permute(0, 3, 1, 2)


class ConvFn(torch.autograd.Function):
    """y = conv(x, w) with no normalisation / activation (inner convs of a Tucker-decomposed block)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, cache):
        dt = compute_dtype(weight)
        # pixel-pair packing is for the image only: an input that wants a gradient keeps the plain (zero-padded) geometry
        geo = _Geometry(x.shape, weight.shape, stride, padding, dt, allow_packed_stem=not x.requires_grad)
        xk = _prepare_input(x, geo, dt)
        _, _, _, _, ldx = ops.nhwc_info(xk)
        Cout = weight.shape[0]
        cout_pad = _round_up(Cout, _ce(dt))
        w, wt = cache.get(weight, dt, cout_pad, geo.cin_pad)
        y = ops.new_act(geo.B, cout_pad, geo.Ho, geo.Wo, dt, xk.device)
        ops.conv_fwd(geo.desc(dt, ldx, cout_pad, cout=cout_pad), xk, w, y, EPI_NONE)
        ctx.save_for_backward(xk, wt)
        ctx.geo, ctx.dt, ctx.ldx, ctx.cout_pad = geo, dt, ldx, cout_pad
        ctx.weight_shape = tuple(weight.shape)
        return y if cout_pad == Cout else y[:, :Cout]

    @staticmethod
    def backward(ctx, dy):
        xk, wt = ctx.saved_tensors
        geo, dt, cout_pad = ctx.geo, ctx.dt, ctx.cout_pad
        Cout, Cin, kh, kw = ctx.weight_shape
        if dy.dtype != dt:
            dy = dy.to(dt)
        if cout_pad != Cout:
            full = torch.zeros((geo.B, cout_pad, geo.Ho, geo.Wo), dtype=dt, device=dy.device).contiguous(
                memory_format=torch.channels_last)
            full[:, :Cout] = dy
            dy = full
        dy = ops.to_nhwc(dy)
        _, _, _, _, ldy = ops.nhwc_info(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.new_act(geo.B, geo.cin_pad, geo.H, geo.W, dt, dy.device)
            ops.conv_dgrad(geo.desc(dt, geo.cin_pad, ldy, cout=cout_pad), dy, wt, dx)
            if geo.cin_pad != Cin:
                dx = dx[:, :Cin]
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(geo, dt, xk, ctx.ldx, dy, ldy, Cout, Cin, kh, kw, cout_pad=cout_pad)
        return dx, dw, None, None, None


def conv_affine_act_eval(x, weight, scale, shift, stride, padding, act: int, cache) -> torch.Tensor:
    """Inference: y = act(conv(x,w)*scale + shift) in ONE kernel (BN folded into the epilogue)."""
    dt = compute_dtype(weight)
    geo = _Geometry(x.shape, weight.shape, stride, padding, dt)
    xk = _prepare_input(x, geo, dt)
    _, _, _, _, ldx = ops.nhwc_info(xk)
    Cout = weight.shape[0]
    cout_pad = _round_up(Cout, _ce(dt))
    w, _ = cache.get(weight, dt, cout_pad, geo.cin_pad)
    y = ops.new_act(geo.B, cout_pad, geo.Ho, geo.Wo, dt, xk.device)
    epi = EPI_AFFINE_SILU if act else EPI_AFFINE
    if scale is None and shift is None and not act:
        epi = EPI_NONE
    ops.conv_fwd(geo.desc(dt, ldx, cout_pad), xk, w, y, epi,
                 scale=None if scale is None else scale.float().contiguous(),
                 shift=None if shift is None else shift.float().contiguous())
    return y if cout_pad == Cout else y[:, :Cout]


class HeadConvFn(torch.autograd.Function):
    """raw[b, a, y, x, o] = conv1x1(x, w)[b, a*no + o, y, x] + bias  (fp32 logits), and its backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, na, no, cache):
        dt = compute_dtype(weight)
        geo = _Geometry(x.shape, weight.shape, (1, 1), (0, 0), dt)
        xk = _prepare_input(x, geo, dt)
        B, Cin, H, W, ldx = ops.nhwc_info(xk)
        Cout = weight.shape[0]
        cout_pad = _round_up(Cout, 8)
        w, wt = cache.get(weight, dt, cout_pad, geo.cin_pad)
        buf = torch.empty((B, H, W, cout_pad), dtype=torch.float32, device=xk.device)       # NHWC logits
        ops.conv_fwd(geo.desc(dt, ldx, cout_pad), xk, w, buf, EPI_HEAD,
                     shift=bias.float().contiguous() if bias is not None else None, head_no=no)
        raw = buf.as_strided((B, na, H, W, no), (H * W * cout_pad, no, W * cout_pad, cout_pad, 1))
        raw._ayolo_head = (cout_pad, dt)          # lets the fused loss hand its gradient over in dz layout
        ctx.raw_ptr = buf.data_ptr()
        ctx.save_for_backward(xk, wt)
        ctx.geo, ctx.dt, ctx.ldx, ctx.cout_pad, ctx.na, ctx.no = geo, dt, ldx, cout_pad, na, no
        ctx.weight_shape = tuple(weight.shape)
        ctx.has_bias = bias is not None
        return raw

    @staticmethod
    def backward(ctx, draw):
        xk, wt = ctx.saved_tensors
        geo, dt, cout_pad = ctx.geo, ctx.dt, ctx.cout_pad
        Cout, Cin, kh, kw = ctx.weight_shape
        from .losses import take_packed_head_grad
        pk = take_packed_head_grad(draw, cout_pad, dt, ctx.raw_ptr)
        if pk is not None:
            dz = pk[0].view(geo.B, geo.H, geo.W, cout_pad).permute(0, 3, 1, 2)     # NHWC memory, NCHW view
            dbias = pk[1]
        else:
            dz, dbias = ops.head_grad_pack(draw, dt, cout_pad, want_bias=ctx.has_bias)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.new_act(geo.B, Cin, geo.H, geo.W, dt, dz.device)
            ops.conv_dgrad(geo.desc(dt, Cin, cout_pad, cout=cout_pad), dz, wt, dx)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(geo, dt, xk, ctx.ldx, dz, cout_pad, Cout, Cin, kh, kw, cout_pad=cout_pad)
        return dx, dw, (dbias if ctx.has_bias else None), None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        x = ops.to_nhwc(x)
        y, arg = ops.maxpool_fwd(x, k, want_argmax=x.requires_grad or torch.is_grad_enabled())
        ctx.save_for_backward(arg)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        return ops.maxpool_bwd(arg, ops.to_nhwc(dy), ctx.k), None


class Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_fwd(ops.to_nhwc(x))

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2x_bwd(ops.to_nhwc(dy))
