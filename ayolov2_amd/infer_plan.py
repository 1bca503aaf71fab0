"""Inference executor: the eval-mode forward of a YOLOModel (val.py:331 ``model.fuse().eval()`` path, TTA, the
Tucker-decomposed model of decompose_model.py) compiled ONCE per (input shape, dtype) into one straight-line list of
kernel launches over static buffers, enqueued by a single ``ayolo_run_ops`` call -- the forward-only sibling of the
training plan (plan.py).

What it removes compared with the per-module path: every torch glue kernel (concat, shortcut adds, slicing of padded
channels, per-call scale/shift arithmetic, weight casts) and ~300 Python-level launches per forward.

* Conv-BN-SiLU is one launch: BatchNorm (running statistics) and a conv bias are folded into the conv epilogue's
  per-channel scale/shift; the folded vectors and the compute-dtype weight copies are refreshed only when a parameter
  or buffer changed (version counters), not per forward.
* C3: ``cv1 | cv2`` run as one conv writing both halves of the concat buffer; the Bottleneck chain then updates the
  first half IN PLACE (``x + cv2(cv1(x))`` is the conv epilogue AYOLO_EPI_AFFINE_SILU_RES added onto x), so the concat,
  the shortcut adds and the copies disappear.  SPPF pools write their concat slices, UpSample writes the neck's.
* A Tucker-decomposed block (``.conv`` = Sequential 1x1 -> kxk -> 1x1, decomposition.py:363-424) is three launches over
  rank-padded buffers (ranks rounded up to the 8-channel vector width with zero weights), BN + activation in the third.
* YOLOHead: logits in the (B, na, ny, nx, no) layout + the decode kernel writing the (B, sum na*ny*nx, no) prediction.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from . import functional as F_
from ._lib import EPI_AFFINE, EPI_AFFINE_RES, EPI_AFFINE_SILU, EPI_AFFINE_SILU_RES, EPI_HEAD, EPI_NONE
from .modules import C3, SPPF, Bottleneck, Concat, Conv, UpSample, YOLOHead, _act_code, _pair
from .plan import (OP_CAST_WEIGHT, OP_CAST_WEIGHTS, OP_CONV_FWD, OP_MAXPOOL_FWD, OP_PACK_INPUT, OP_SPPF_FWD, OP_UPSAMPLE_FWD, Act, Op,
                   PlanUnsupported, _op)

OP_HEAD_DECODE = 20
MAX_INFER_PLANS = 4


TUCKER_FORMS = os.environ.get("AYOLO_TUCKER_FORM", "auto")      # auto | factors | first | last | dense (A/B switch of _tucker_form)


class _MergedConv:
    """Consecutive LINEAR convs of a Tucker block (1x1 factor, k x k core, 1x1 factor: decomposition.py:363-424 puts no
    activation between them) presented as ONE conv: the composed weight is computed in fp32 when a source changes.  Only the
    last part may carry a bias (a bias in front of a zero-padded k x k conv does not commute with the padding)."""

    def __init__(self, parts: Sequence[nn.Conv2d]):
        self.parts = list(parts)
        core = max(self.parts, key=lambda c: c.kernel_size[0] * c.kernel_size[1])
        if sum(1 for c in self.parts if c.kernel_size != (1, 1)) > 1 or any(c.bias is not None for c in self.parts[:-1]):
            raise PlanUnsupported("Tucker merge: more than one k x k part or an inner bias")
        for c in self.parts:
            if c is not core and (_pair(c.stride) != (1, 1) or _pair(c.padding) != (0, 0)):
                raise PlanUnsupported("Tucker merge: strided / padded factor conv")
        self.kernel_size, self.stride, self.padding = core.kernel_size, core.stride, core.padding
        self.dilation, self.groups = (1, 1), 1
        self.in_channels, self.out_channels = self.parts[0].in_channels, self.parts[-1].out_channels
        self.bias = self.parts[-1].bias
        self.weight = None
        self.recompute()

    def sources(self) -> List[torch.Tensor]:
        return [t for c in self.parts for t in (c.weight, c.bias) if t is not None]

    @torch.no_grad()
    def recompute(self) -> None:
        w = None
        for c in self.parts:
            cw = c.weight.detach().float()
            if w is None:
                w = cw
            elif cw.shape[2:] == (1, 1):                      # 1x1 after what has been composed so far
                w = torch.einsum("oa,aihw->oihw", cw[:, :, 0, 0], w)
            elif w.shape[2:] == (1, 1):                       # k x k after a 1x1
                w = torch.einsum("oahw,ai->oihw", cw, w[:, :, 0, 0])
            else:
                raise PlanUnsupported("Tucker merge: two spatial kernels")
        self.weight = w.contiguous()


class _Fold:
    """One conv launch's host-side state: the fp32 KRSC weight source, the folded scale / shift vectors."""

    def __init__(self, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], cout_pad: int, device):
        self.conv, self.bn = conv, bn
        self.scale = torch.ones(cout_pad, dtype=torch.float32, device=device) if bn is not None else None
        self.shift = torch.zeros(cout_pad, dtype=torch.float32, device=device) if (bn is not None or conv.bias is not None) else None
        self.w32: Optional[torch.Tensor] = None

    def tensors(self) -> List[torch.Tensor]:
        if isinstance(self.conv, _MergedConv):
            ts = self.conv.sources()
        else:
            ts = [self.conv.weight]
            if self.conv.bias is not None:
                ts.append(self.conv.bias)
        if self.bn is not None:
            ts += [t for t in (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var) if t is not None]
        return ts

    @torch.no_grad()
    def refresh(self, w32_buf: torch.Tensor) -> None:
        """fp32 KRSC copy of the weight for the cast kernel + BatchNorm / bias folded into scale / shift."""
        if isinstance(self.conv, _MergedConv):
            self.conv.recompute()
        w = self.conv.weight.detach()
        w32_buf.copy_(w.permute(0, 2, 3, 1))                       # (Cout, kh, kw, Cin) fp32, from any dtype / layout
        co = w.shape[0]
        if self.bn is not None:
            bn = self.bn
            rv, rm = bn.running_var.float(), bn.running_mean.float()
            g = bn.weight.float() if bn.weight is not None else torch.ones_like(rv)
            b = bn.bias.float() if bn.bias is not None else torch.zeros_like(rv)
            sc = g / torch.sqrt(rv + bn.eps)
            sh = b - rm * sc
            if self.conv.bias is not None:
                sh = sh + self.conv.bias.float() * sc
            self.scale[:co].copy_(sc)
            self.shift[:co].copy_(sh)
        elif self.conv.bias is not None:
            self.shift[:co].copy_(self.conv.bias.float())


class InferPlan:
    def __init__(self, model, x_shape: Sequence[int], dt: torch.dtype, device):
        self.model, self.dt, self.device = model, dt, device
        self.B, self.Cimg, self.H, self.W = x_shape
        self.keep: List[torch.Tensor] = []
        self.fwd: List[Op] = []
        self.casts: List[Op] = []
        self.folds: List[Tuple[_Fold, torch.Tensor]] = []
        self.pack_op: Optional[Op] = None
        self.raw_specs: list = []
        self.tucker_forms: List[str] = []                      # launch form chosen for every Tucker block (_tucker_form)
        self.out: Optional[torch.Tensor] = None
        self._versions = None
        self._compile()

    # ------------------------------------------------------------------ helpers
    def _new_act(self, C, H, W) -> Act:
        t = ops.new_act(self.B, C, H, W, self.dt, self.device)
        self.keep.append(t)
        return Act(t)

    def _ce(self) -> int:
        return 8 if self.dt == torch.float16 else 4

    def _conv(self, conv: nn.Conv2d, bn, act: int, x_t: torch.Tensor, dst: Optional[Act], residual_in_place: bool = False,
              image: bool = False) -> Act:
        """One conv launch: y = act(bn(conv(x))) [+ bias], written into `dst` (a channel slice of a wider buffer is fine) or
        a new buffer whose channel count is rounded up to the vector width (extra channels are exact zeros)."""
        if not isinstance(conv, (nn.Conv2d, _MergedConv)) or conv.groups != 1 or _pair(conv.dilation) != (1, 1):
            raise PlanUnsupported("non-standard conv")
        dt, dev, ce = self.dt, self.device, self._ce()
        Cout, Cin, kh, kw = conv.weight.shape
        xshape = (self.B, self.Cimg, self.H, self.W) if image else (x_t.shape[0], Cin, x_t.shape[2], x_t.shape[3])
        # (the pixel-pair packing of the stem only applies to the packed IMAGE, not to a 2..4-channel Tucker factor output)
        geo = F_._Geometry(xshape, conv.weight.shape, _pair(conv.stride), _pair(conv.padding), dt, allow_packed_stem=image)
        if image:
            packed = ops.new_act(self.B, geo.cin_pad, self.H, self.W, dt, dev)
            self.keep.append(packed)
            self.pack_op = _op(OP_PACK_INPUT, i=(self.B, self.Cimg, self.H, self.W, ops.dtype_code(dt), geo.cin_pad), p=(None, packed))
            self.fwd.append(self.pack_op)
            xk = packed.as_strided((self.B, 8, self.H, self.W // 2), (self.H * self.W * 4, 1, self.W * 4, 8)) if geo.packed_stem else packed
        else:
            # the producer's buffer holds Cin channels rounded up to the vector width (extra channels are zeros)
            if x_t.shape[1] < geo.cin_pad:
                raise PlanUnsupported("input buffer narrower than the padded channel count")
            xk = x_t if x_t.shape[1] == geo.cin_pad else x_t[:, :geo.cin_pad]
        ldx = ops.nhwc_info(xk)[4]
        cout_pad = F_._round_up(Cout, ce)
        if dst is not None:
            if dst.C != Cout or Cout % ce:
                raise PlanUnsupported("concat slice with a channel count off the vector width")
            y = dst
        else:
            y = self._new_act(cout_pad, geo.Ho, geo.Wo)
        ldy = ops.nhwc_info(y.t)[4]
        wc = torch.empty((cout_pad, kh, kw, geo.cin_pad), dtype=dt, device=dev)
        w32 = torch.empty((Cout, kh, kw, Cin), dtype=torch.float32, device=dev)
        self.keep += [wc, w32]
        self.casts.append(_op(OP_CAST_WEIGHT, i=(Cout, kh, kw, Cin, cout_pad, geo.cin_pad, ops.dtype_code(dt), 0), p=(w32, wc, None)))
        fold = _Fold(conv, bn, cout_pad, dev)
        self.folds.append((fold, w32))
        if fold.scale is None and fold.shift is None and not act and not residual_in_place:
            epi = EPI_NONE
        elif residual_in_place:
            epi = EPI_AFFINE_SILU_RES if act else EPI_AFFINE_RES
        else:
            epi = EPI_AFFINE_SILU if act else EPI_AFFINE
        self.fwd.append(_op(OP_CONV_FWD, i=(epi, 1, 0), p=(xk, wc, y.t, fold.scale, fold.shift, None),
                            conv=geo.desc(dt, ldx, ldy, cout=cout_pad if dst is None else Cout)))
        return y

    def _tucker_form(self, convs: List[nn.Conv2d], xt: Optional[torch.Tensor], image: bool) -> str:
        """How a Tucker block 1x1 (Cin -> r1) -> k x k (r1 -> r2) -> 1x1 (r2 -> Cout) is launched.  The three convs are linear
        with nothing in between, so any adjacent pair -- or all three -- may be multiplied out when the plan is compiled:
          factors: three launches (fewest FLOP, two rank-wide intermediates through HBM, the first at INPUT resolution);
          first  : k x k (Cin -> r2) then 1x1: no r1 intermediate (at stride 2 it alone is as large as the block's output);
          last   : 1x1 then k x k (r1 -> Cout);
          dense  : the block's own k x k conv again (what an HBM-bound layer with ranks near C/2 is cheapest as).
        Chosen per block by a two-term roofline per launch, max(bytes / 3 TB/s, FLOP / 500 TF/s) + 6 us, with the rates this
        executor's conv kernels sustain on such layers (profiles/r03_conv_layer_sweep.txt).  AYOLO_TUCKER_FORM forces one
        form for A/B runs.  Reference: decomposition.py:363-424 (the Sequential), decompose_model.py:63-74."""
        if len(convs) != 3 or convs[0].kernel_size != (1, 1) or convs[2].kernel_size != (1, 1) or convs[0].bias is not None \
                or convs[1].bias is not None or _pair(convs[1].dilation) != (1, 1):
            return "factors"
        if TUCKER_FORMS in ("factors", "first", "last", "dense"):
            return TUCKER_FORMS
        ce = self._ce()
        es = 2 if self.dt == torch.float16 else 4
        core = convs[1]
        cin, r1, r2, cout = convs[0].in_channels, convs[0].out_channels, core.out_channels, convs[2].out_channels
        kh, kw = core.kernel_size
        H, W = (self.H, self.W) if image else (xt.shape[2], xt.shape[3])
        sh, sw = _pair(core.stride)
        ph, pw = _pair(core.padding)
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        pin, pout = self.B * H * W, self.B * Ho * Wo
        up = lambda c: F_._round_up(c, ce)
        cin_e, r1p, r2p = (4 if image else up(cin)), up(r1), up(r2)

        def launch(ci, co, taps, p_in, p_out):
            byts = es * (p_in * ci + p_out * co)
            flop = 2.0 * p_out * taps * ci * co
            return max(byts / 3.0e12, flop / 5.0e14) + 6e-6

        k2 = kh * kw
        cost = {"factors": launch(cin_e, r1p, 1, pin, pin) + launch(r1p, r2p, k2, pin, pout) + launch(r2p, cout, 1, pout, pout),
                "first": launch(cin_e, r2p, k2, pin, pout) + launch(r2p, cout, 1, pout, pout),
                "last": launch(cin_e, r1p, 1, pin, pin) + launch(r1p, cout, k2, pin, pout),
                "dense": launch(cin_e, cout, k2, pin, pout)}
        return min(cost, key=cost.get)

    def _block(self, mod: Conv, x: Optional[Act], dst: Optional[Act], residual_in_place: bool = False, image: bool = False) -> Act:
        """A kindle Conv block in eval mode: conv (plain, fused-with-bias, or the 3-conv Tucker Sequential) -> BN -> act."""
        act = _act_code(mod.activation)
        bn = getattr(mod, "batch_norm", None)
        conv = mod.conv
        xt = None if image else x.t
        if isinstance(conv, nn.Sequential):
            convs = list(conv)
            if len(convs) < 2 or not all(isinstance(c, nn.Conv2d) for c in convs):
                raise PlanUnsupported("unexpected members in a decomposed block")
            form = self._tucker_form(convs, xt, image)
            self.tucker_forms.append(form)
            if form == "dense":
                groups = [convs]
            elif form == "first":
                groups = [convs[:2], convs[2:]]
            elif form == "last":
                groups = [convs[:1], convs[1:]]
            else:
                groups = [[c] for c in convs]
            t = xt
            for j, g in enumerate(groups):
                c = g[0] if len(g) == 1 else _MergedConv(g)
                if j + 1 < len(groups):                        # factor / core launches: plain, no bias
                    t = self._conv(c, None, 0, t, None, image=(image and j == 0)).t
                else:
                    return self._conv(c, bn, act, t, dst, residual_in_place, image=(image and j == 0))
        return self._conv(conv, bn, act, xt, dst, residual_in_place, image)

    # ------------------------------------------------------------------ composite blocks
    def _c3(self, m: C3, x: Act, dst: Optional[Act]) -> Act:
        h = m.cv1.conv.out_channels if isinstance(m.cv1.conv, nn.Conv2d) else m.cv1.conv[-1].out_channels
        _, _, H, W = x.t.shape
        cat = self._new_act(2 * h, H, W)
        first = cat.slice(0, h)
        merged = self._merged_pair(m.cv1, m.cv2, x, cat)
        if not merged:
            self._block(m.cv1, x, first)
            self._block(m.cv2, x, cat.slice(h, 2 * h))
        for b in m.m:
            y1 = self._block(b.cv1, first, None)
            # x + cv2(cv1(x)) over x itself (shortcut), or plain overwrite of the (now dead) chain input
            self._block(b.cv2, y1, first, residual_in_place=bool(b.add))
        return self._block(m.cv3, cat, dst)

    def _merged_pair(self, a: Conv, b: Conv, x: Act, cat: Act) -> bool:
        """C3's cv1 | cv2 as ONE conv over the shared input, writing both halves of the concat buffer."""
        ca, cb = a.conv, b.conv
        ok = (isinstance(ca, nn.Conv2d) and isinstance(cb, nn.Conv2d) and ca.kernel_size == cb.kernel_size and ca.stride == cb.stride
              and ca.padding == cb.padding and ca.in_channels == cb.in_channels and ca.out_channels == cb.out_channels
              and ca.out_channels % self._ce() == 0 and _act_code(a.activation) == _act_code(b.activation)
              and (ca.bias is None) == (cb.bias is None)
              and (getattr(a, "batch_norm", None) is None) == (getattr(b, "batch_norm", None) is None))
        if not ok:
            return False
        dt, dev = self.dt, self.device
        h, Cin, kh, kw = ca.weight.shape
        geo = F_._Geometry(tuple(x.t.shape[:1]) + (Cin,) + tuple(x.t.shape[2:]), (2 * h, Cin, kh, kw), _pair(ca.stride), _pair(ca.padding), dt)
        if geo.needs_pack or x.t.shape[1] != geo.cin_pad:
            return False
        ldx, ldy = ops.nhwc_info(x.t)[4], ops.nhwc_info(cat.t)[4]
        wc = torch.empty((2 * h, kh, kw, geo.cin_pad), dtype=dt, device=dev)
        self.keep.append(wc)
        scale = shift = None
        for j, (mod, c) in enumerate(((a, ca), (b, cb))):
            w32 = torch.empty((h, kh, kw, Cin), dtype=torch.float32, device=dev)
            self.keep.append(w32)
            self.casts.append(_op(OP_CAST_WEIGHT, i=(h, kh, kw, Cin, h, geo.cin_pad, ops.dtype_code(dt), 0), p=(w32, wc[j * h:(j + 1) * h], None)))
            fold = _Fold(c, getattr(mod, "batch_norm", None), h, dev)
            if j == 0:
                scale = torch.ones(2 * h, dtype=torch.float32, device=dev) if fold.scale is not None else None
                shift = torch.zeros(2 * h, dtype=torch.float32, device=dev) if fold.shift is not None else None
            # the two folds write into the halves of the shared scale / shift vectors
            fold.scale = scale[j * h:(j + 1) * h] if scale is not None else None
            fold.shift = shift[j * h:(j + 1) * h] if shift is not None else None
            self.folds.append((fold, w32))
        act = _act_code(a.activation)
        epi = EPI_NONE if (scale is None and shift is None and not act) else (EPI_AFFINE_SILU if act else EPI_AFFINE)
        self.keep += [t for t in (scale, shift) if t is not None]
        self.fwd.append(_op(OP_CONV_FWD, i=(epi, 1, 0), p=(x.t, wc, cat.t, scale, shift, None), conv=geo.desc(dt, ldx, ldy, cout=2 * h)))
        return True

    def _sppf(self, m: SPPF, x: Act, dst: Optional[Act]) -> Act:
        h = m.cv1.conv.out_channels if isinstance(m.cv1.conv, nn.Conv2d) else m.cv1.conv[-1].out_channels
        _, _, H, W = x.t.shape
        if h % self._ce():
            raise PlanUnsupported("SPPF width off the vector width")
        cat = self._new_act(4 * h, H, W)
        self._block(m.cv1, x, cat.slice(0, h))
        k = m.pool.kernel_size
        code = ops.dtype_code(self.dt)
        # the three chained pools in one launch on the LDS-resident map where a workgroup gets runs of >= 32 bytes (the 20 x 20 maps of
        # 640 x 640 inputs; at 40 x 40 only one channel group fits and three pool launches are faster)
        if k == 5 and self.dt == torch.float16 and _lib.lib().ayolo_sppf_pool_supported(code, H, W, h) >= 2:
            self.fwd.append(_op(OP_SPPF_FWD, i=(code, ops.nhwc_info(cat.t)[4], self.B, H, W, h), p=(cat.t, None)))
        else:
            for j in range(3):
                src, d = cat.slice(j * h, (j + 1) * h), cat.slice((j + 1) * h, (j + 2) * h)
                self.fwd.append(_op(OP_MAXPOOL_FWD, i=(code, ops.nhwc_info(src.t)[4], ops.nhwc_info(d.t)[4], self.B, H, W, h, k), p=(src.t, d.t, None)))
        return self._block(m.cv2, cat, dst)

    def _upsample(self, x: Act, dst: Optional[Act]) -> Act:
        B, C, H, W = x.t.shape
        out = dst if dst is not None else self._new_act(C, 2 * H, 2 * W)
        self.fwd.append(_op(OP_UPSAMPLE_FWD, i=(ops.dtype_code(self.dt), ops.nhwc_info(x.t)[4], ops.nhwc_info(out.t)[4], B, H, W, C), p=(x.t, out.t)))
        return out

    def _head(self, head: YOLOHead, xs: List[Act]) -> None:
        dt, dev = self.dt, self.device
        total = 0
        for lvl, x in enumerate(xs):
            conv = head.conv[lvl]
            if not isinstance(conv, nn.Conv2d):
                raise PlanUnsupported("decomposed head conv")
            Cout, Cin = conv.weight.shape[:2]
            cp = F_._round_up(Cout, 8)
            B, _, H, W = x.t.shape
            geo = F_._Geometry((B, Cin, H, W), conv.weight.shape, (1, 1), (0, 0), dt)
            if x.t.shape[1] != geo.cin_pad:
                raise PlanUnsupported("head input width")
            wc = torch.empty((cp, 1, 1, geo.cin_pad), dtype=dt, device=dev)
            w32 = torch.empty((Cout, 1, 1, Cin), dtype=torch.float32, device=dev)
            buf = torch.empty((B, H, W, cp), dtype=torch.float32, device=dev)
            self.keep += [wc, w32, buf]
            self.casts.append(_op(OP_CAST_WEIGHT, i=(Cout, 1, 1, Cin, cp, geo.cin_pad, ops.dtype_code(dt), 0), p=(w32, wc, None)))
            fold = _Fold(conv, None, cp, dev)
            self.folds.append((fold, w32))
            self.fwd.append(_op(OP_CONV_FWD, i=(EPI_HEAD, 1, head.no), p=(x.t, wc, buf, None, fold.shift, None),
                                conv=geo.desc(dt, ops.nhwc_info(x.t)[4], cp)))
            self.raw_specs.append((buf, (B, head.na, H, W, head.no), (H * W * cp, head.no, W * cp, cp, 1)))
            total += head.na * H * W
        self.out = torch.empty((self.B, total, head.no), dtype=torch.float32, device=dev)
        off = 0
        for lvl, (buf, shape, strides) in enumerate(self.raw_specs):
            B, na, H, W, no = shape
            apx = head.anchor_grid[lvl].reshape(-1, 2).float().contiguous().to(dev)
            self.keep.append(apx)
            self.fwd.append(_op(OP_HEAD_DECODE, i=(B, na, H, W, no, strides[0], strides[1], strides[2], strides[3], off),
                                f=(head._strides_py[lvl],), l=(total,), p=(buf, apx, self.out)))
            off += na * H * W

    # ------------------------------------------------------------------ whole model
    def _compile(self) -> None:
        model = self.model
        layers, routes = list(model.model), model.routes
        # concat destinations (as in the training plan): producer layer -> (concat layer, channel offset)
        ch: List[int] = []
        hw: List[Tuple[int, int]] = []
        for i, m in enumerate(layers):
            frm = routes[i]
            fl = frm if isinstance(frm, list) else [frm]
            srcs = [(i + f) if f < 0 else f for f in fl]
            cin = [self.Cimg if s < 0 else ch[s] for s in srcs]
            sin = [(self.H, self.W) if s < 0 else hw[s] for s in srcs]
            if isinstance(m, Conv):
                c = m.conv if isinstance(m.conv, nn.Conv2d) else m.conv[1]
                cout = m.conv.out_channels if isinstance(m.conv, nn.Conv2d) else m.conv[-1].out_channels
                s, k, p = _pair(c.stride), _pair(c.kernel_size), _pair(c.padding)
                ch.append(cout)
                hw.append(((sin[0][0] + 2 * p[0] - k[0]) // s[0] + 1, (sin[0][1] + 2 * p[1] - k[1]) // s[1] + 1))
            elif isinstance(m, (C3, SPPF)):
                last = m.cv3 if isinstance(m, C3) else m.cv2
                ch.append(last.conv.out_channels if isinstance(last.conv, nn.Conv2d) else last.conv[-1].out_channels)
                hw.append(sin[0])
            elif isinstance(m, UpSample):
                ch.append(cin[0]); hw.append((sin[0][0] * 2, sin[0][1] * 2))
            elif isinstance(m, Concat):
                if m.dimension != 1:
                    raise PlanUnsupported("concat on a non-channel dim")
                ch.append(sum(cin)); hw.append(sin[0])
            elif isinstance(m, YOLOHead):
                ch.append(0); hw.append((0, 0))
            else:
                raise PlanUnsupported(type(m).__name__)
        dest: Dict[int, Tuple[int, int]] = {}
        for j, m in enumerate(layers):
            if isinstance(m, Concat):
                off = 0
                for f in routes[j]:
                    s = (j + f) if f < 0 else f
                    if s in dest or s < 0 or ch[s] % self._ce():
                        raise PlanUnsupported("concat input cannot be written in place")
                    dest[s] = (j, off)
                    off += ch[s]
        cat_bufs: Dict[int, Act] = {}
        outs: List[Optional[Act]] = []
        for i, m in enumerate(layers):
            frm = routes[i]
            fl = frm if isinstance(frm, list) else [frm]
            srcs = [(i + f) if f < 0 else f for f in fl]
            xin = [None if s < 0 else outs[s] for s in srcs]
            dst = None
            if i in dest:
                j, off = dest[i]
                if j not in cat_bufs:
                    cat_bufs[j] = self._new_act(ch[j], hw[j][0], hw[j][1])
                dst = cat_bufs[j].slice(off, off + ch[i])
            if isinstance(m, Conv):
                out = self._block(m, xin[0], dst, image=(srcs[0] < 0))
            elif isinstance(m, C3):
                out = self._c3(m, xin[0], dst)
            elif isinstance(m, SPPF):
                out = self._sppf(m, xin[0], dst)
            elif isinstance(m, UpSample):
                out = self._upsample(xin[0], dst)
            elif isinstance(m, Concat):
                out = cat_bufs[i]
            elif isinstance(m, YOLOHead):
                self._head(m, xin)
                out = None
            outs.append(out)
        if self.pack_op is None or self.out is None:
            raise PlanUnsupported("model without an image stem / YOLOHead")
        # the cast jobs run as their own one-op list, only when a weight changed
        job_t = np.dtype([("w32", "<u8"), ("w", "<u8"), ("wt", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"),
                          ("Cout_pad", "<i4"), ("Cin_pad", "<i4"), ("wt_ld", "<i4")])
        jobs = np.zeros(len(self.casts), dtype=job_t)
        for k, o in enumerate(self.casts):
            jobs[k] = (o.p[0] or 0, o.p[1] or 0, 0, o.i[0], o.i[1] * o.i[2], o.i[3], o.i[4], o.i[5], 0)
        tab = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.device)
        self.keep.append(tab)
        self.cast_arr = (Op * 1)(_op(OP_CAST_WEIGHTS, i=(len(self.casts), ops.dtype_code(self.dt)), p=(tab,)))
        self.fwd_arr = (Op * len(self.fwd))(*self.fwd)
        self.pack_idx = next(k for k, o in enumerate(self.fwd) if o is self.pack_op)
        self.decode_idx = [k for k, o in enumerate(self.fwd) if (o.kind & 0xff) == OP_HEAD_DECODE]
        self._tracked = [t for f, _ in self.folds for t in f.tensors()]
        self._ptrs = tuple(t.data_ptr() for t in self._tracked)

    # ------------------------------------------------------------------ execution
    def valid(self) -> bool:
        return tuple(t.data_ptr() for t in self._tracked) == self._ptrs

    def _trunk(self, x: torch.Tensor, st, n_ops: int) -> None:
        """refresh derived weights if a parameter changed, point the pack op at `x`, run the first `n_ops` ops"""
        vers = tuple(t._version for t in self._tracked)
        if vers != self._versions:                            # a parameter / buffer changed: refold + recast everything
            for fold, w32 in self.folds:
                fold.refresh(w32)
            _lib.check(_lib.lib().ayolo_run_ops(self.cast_arr, 1, st), "ayolo_run_ops(cast)")
            self._versions = vers
        self._x_keep = x
        self.fwd_arr[self.pack_idx].p[0] = x.data_ptr()
        _lib.check(_lib.lib().ayolo_run_ops(self.fwd_arr, n_ops, st), "ayolo_run_ops(inference)")

    @staticmethod
    def _as_input(x: torch.Tensor) -> torch.Tensor:
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        return x

    def run(self, x: torch.Tensor):
        x = self._as_input(x)
        st = torch.cuda.current_stream().cuda_stream
        # The decoded prediction -- what val.py keeps, post-processes and may hold across batches (`out, train_out =
        # model(img)`, train_utils.py:441-444) -- is written into a tensor that belongs to THIS call: the decode ops are
        # pointed at a fresh allocation (caching allocator: no device malloc, no extra copy).  model.static_outputs = True
        # keeps the round-2 behaviour (one static buffer, overwritten by the next forward of the same shape).
        out = self.out if getattr(self.model, "static_outputs", False) else torch.empty_like(self.out)
        for k in self.decode_idx:
            self.fwd_arr[k].p[2] = out.data_ptr()
        self._trunk(x, st, len(self.fwd))
        # the raw per-level logits stay views of the executor's static buffers (valid until the next forward of this shape)
        raws = [buf.as_strided(shape, strides) for buf, shape, strides in self.raw_specs]
        return out, raws

    @property
    def rows(self) -> int:
        return int(self.out.shape[1])

    def run_augmented(self, x: torch.Tensor, merged: torch.Tensor, row_off: int, win: Tuple[int, int], scale: float, flip: int,
                      extent: float) -> None:
        """One forward of test-time augmentation (tta.inference_with_tta): everything up to the head convs as `run`, then the
        decode of every level writes its rows -- already de-scaled and de-flipped -- at `row_off` of the merged prediction
        `merged` (B, rows of all augmentations, no); rows outside the window `win` (the clipped tails) are not stored."""
        assert self.decode_idx == list(range(len(self.fwd) - len(self.decode_idx), len(self.fwd))), "decode ops close the list"
        assert merged.is_contiguous() and merged.dtype == torch.float32 and merged.shape[0] == self.B and merged.shape[2] == self.out.shape[2]
        x = self._as_input(x)
        st = torch.cuda.current_stream().cuda_stream
        self._trunk(x, st, len(self.fwd) - len(self.decode_idx))
        for k in self.decode_idx:
            o = self.fwd_arr[k]
            strides = (_lib.c_int64 * 4)(o.i[5], o.i[6], o.i[7], o.i[8])
            _lib.call("ayolo_head_decode_aug", o.p[0], strides, o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.p[1], o.f[0],
                      merged.data_ptr(), merged.shape[1], row_off + o.i[9], float(scale), int(flip), float(extent),
                      int(win[0]), int(win[1]), st)


def eval_plan_for(model, x: torch.Tensor) -> Optional["InferPlan"]:
    """The cached inference plan of (input shape, dtype, device), compiled on first use; None if the structure is unsupported."""
    w = next((p for p in model.parameters()), None)
    dt = torch.float16 if (torch.is_autocast_enabled() or (w is not None and w.dtype == torch.float16)) else torch.float32
    key = ("eval", tuple(x.shape), dt, x.device)
    cache = model.__dict__.setdefault("_plans", {})
    plan = cache.get(key)
    if plan is False:
        return None
    if plan is not None and not plan.valid():
        plan = None
    if plan is None:
        try:
            plan = InferPlan(model, tuple(x.shape), dt, x.device)
        except PlanUnsupported:
            cache[key] = False
            return None
        live = [k for k, v in cache.items() if v is not False and k and k[0] == "eval"]
        while len(live) >= MAX_INFER_PLANS:
            cache.pop(live.pop(0))
        cache[key] = plan
    return plan


def plan_forward_eval(model, x: torch.Tensor):
    """Eval forward through the cached inference plan; returns (decoded, raws) or None if the structure is unsupported.
    `decoded` is a tensor of its own (the decode kernels write into a fresh allocation per call); the raw per-level logits
    are views of plan-owned static buffers that the next forward of the same shape overwrites."""
    plan = eval_plan_for(model, x)
    if plan is None:
        return None
    out, raws = plan.run(x)
    head = model.model[-1]
    if getattr(head, "out_xyxy", False):
        out = out.clone()
        xy, wh = out[..., :2].clone(), out[..., 2:4].clone()
        out[..., :2], out[..., 2:4] = xy - wh / 2, xy + wh / 2
    return out, raws
