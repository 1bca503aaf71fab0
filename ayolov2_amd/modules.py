"""nn.Module blocks of the YOLOv5 backbone-neck-head, keeping the attribute surface the reference's callers
touch on kindle's modules:

* every conv block exposes ``.conv`` (an ``nn.Conv2d``, or the 3-conv ``nn.Sequential`` that
  scripts/tensor_decomposition/decomposition.py:325-335 swaps in) as a direct child;
* ``YOLOHead`` exposes ``nl, na, nc, no, anchors, anchor_grid, stride, conv (ModuleList), out_xyxy``
  (scripts/loss/losses.py:201-221, scripts/utils/anchors.py:185-224, export.py:171).

Forward/backward arithmetic runs in libayolo_hip.so (see functional.py); these classes hold parameters only.
The kindle source is not available (un-vendored PyPI dependency), so BatchNorm uses the torch defaults
(eps 1e-5, momentum 0.1) and conv weights the torch default init -- documented in DESIGN.md.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Union

import torch
from torch import nn

from . import functional as F_
from . import ops


def autopad(k, p=None):
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _act_code(activation: Optional[str]) -> int:
    if activation is None or activation in ("Identity", "Linear", "None"):
        return 0
    if activation in ("SiLU", "Swish"):
        return 1
    raise NotImplementedError(f"activation {activation!r}: only SiLU / identity have HIP epilogues")


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv(nn.Module):
    """Conv2d(bias=False) -> BatchNorm2d -> activation (yolov5s.yaml:21-50 `Conv` rows)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, stride: int = 1,
                 padding: Optional[int] = None, groups: int = 1, activation: Optional[str] = "SiLU") -> None:
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped convolution is not on the YOLOv5 hot path")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, autopad(kernel_size, padding), bias=False)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.batch_norm = nn.BatchNorm2d(out_channels)
        self.activation = activation
        self._caches = {}

    # weight-copy caches are runtime state, never part of a checkpoint / deepcopy
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_caches"] = {}
        return st

    def _cache(self, key) -> F_._WeightCache:
        c = self._caches.get(key)
        if c is None:
            c = self._caches[key] = F_._WeightCache()
        return c

    def _bn_affine(self):
        bn = getattr(self, "batch_norm", None)
        if bn is None:
            return None, None
        rv, rm = bn.running_var.float(), bn.running_mean.float()
        g = bn.weight.float() if bn.weight is not None else torch.ones_like(rv)
        b = bn.bias.float() if bn.bias is not None else torch.zeros_like(rv)
        scale = g / torch.sqrt(rv + bn.eps)
        return scale, b - rm * scale

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops.require_cuda(x, "Conv.forward")
        act = _act_code(self.activation)
        bn = getattr(self, "batch_norm", None)
        conv = self.conv
        if isinstance(conv, nn.Sequential):            # Tucker-decomposed: 1x1 -> kxk -> 1x1, BN/act after the last
            convs = list(conv)
            for i, c in enumerate(convs[:-1]):
                x = self._plain(x, c, i)
            last = convs[-1]
            scale, shift = self._bn_affine()
            if last.bias is not None:
                shift = last.bias.float() * (scale if scale is not None else 1.0) + (shift if shift is not None else 0.0)
            if self.training and bn is not None:
                # fine-tuning a decomposed block: the factor / core convs are plain autograd convs, the last 1x1 carries the
                # batch-statistics BatchNorm + activation (a bias in front of a training-mode BN only shifts the mean)
                if bn.running_mean is not None and bn.running_mean.dtype != torch.float32:
                    raise NotImplementedError("training needs fp32 BatchNorm buffers (use autocast, not model.half())")
                if last.bias is not None or bn.momentum is None:
                    raise NotImplementedError("decomposed block with a biased last conv / cumulative-average BatchNorm in training")
                y = F_.ConvBnActFn.apply(x, last.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                         _pair(last.stride), _pair(last.padding), bn.eps, bn.momentum, act,
                                         self._cache(len(convs) - 1))
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
                return y
            return F_.conv_affine_act_eval(x, last.weight, scale, shift, _pair(last.stride), _pair(last.padding), act,
                                           self._cache(len(convs) - 1))
        if self.training and bn is not None:
            if bn.running_mean is not None and bn.running_mean.dtype != torch.float32:
                raise NotImplementedError("training needs fp32 BatchNorm buffers (use autocast, not model.half())")
            if bn.momentum is None:
                raise NotImplementedError("cumulative-average BatchNorm")
            y = F_.ConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     _pair(conv.stride), _pair(conv.padding), bn.eps, bn.momentum, act, self._cache(0))
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            return y
        scale, shift = self._bn_affine()
        if conv.bias is not None:                       # fused (BN folded by fuse()) or a plain biased conv
            shift = conv.bias.float() * (scale if scale is not None else 1.0) + (shift if shift is not None else 0.0)
        return F_.conv_affine_act_eval(x, conv.weight, scale, shift, _pair(conv.stride), _pair(conv.padding), act,
                                       self._cache(0))

    def _plain(self, x, c: nn.Conv2d, idx: int):
        if c.bias is not None:
            return F_.conv_affine_act_eval(x, c.weight, None, c.bias, _pair(c.stride), _pair(c.padding), 0, self._cache(idx))
        return F_.ConvFn.apply(x, c.weight, _pair(c.stride), _pair(c.padding), self._cache(idx))

    def fuse(self) -> "Conv":
        """Fold BatchNorm into the conv (kindle `model.fuse()`, val.py:331)."""
        bn = getattr(self, "batch_norm", None)
        if bn is None or isinstance(self.conv, nn.Sequential):
            return self
        conv = self.conv
        with torch.no_grad():
            scale, shift = self._bn_affine()
            w = (conv.weight.float() * scale.view(-1, 1, 1, 1)).to(conv.weight.dtype)
            fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
            fused = fused.to(conv.weight.device)
            fused.weight.data = w.contiguous(memory_format=torch.channels_last)
            cb = conv.bias.float() * scale if conv.bias is not None else 0.0
            fused.bias.data = (shift + cb).to(conv.weight.dtype)
            fused.requires_grad_(False)
        self.conv = fused
        del self.batch_norm
        self._caches = {}
        return self


class Bottleneck(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, shortcut: bool = True, groups: int = 1,
                 expansion: float = 0.5, activation: Optional[str] = "SiLU") -> None:
        super().__init__()
        hidden = int(out_channels * expansion)
        self.cv1 = Conv(in_channels, hidden, 1, 1, activation=activation)
        self.cv2 = Conv(hidden, out_channels, 3, 1, groups=groups, activation=activation)
        self.add = shortcut and in_channels == out_channels

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3(nn.Module):
    """CSP bottleneck with 3 convolutions (yolov5s.yaml:23-52 `C3` rows)."""

    def __init__(self, in_channels: int, out_channels: int, n_repeat: int = 1, shortcut: bool = True, groups: int = 1,
                 expansion: float = 0.5, activation: Optional[str] = "SiLU") -> None:
        super().__init__()
        hidden = int(out_channels * expansion)
        self.cv1 = Conv(in_channels, hidden, 1, 1, activation=activation)
        self.cv2 = Conv(in_channels, hidden, 1, 1, activation=activation)
        self.cv3 = Conv(2 * hidden, out_channels, 1, activation=activation)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, groups, expansion=1.0, activation=activation)
                                 for _ in range(n_repeat)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


class MaxPool5(nn.Module):
    def __init__(self, kernel_size: int = 5):
        super().__init__()
        self.kernel_size = kernel_size

    def forward(self, x):
        ops.require_cuda(x, "MaxPool.forward")
        return F_.MaxPoolFn.apply(x, self.kernel_size)


class SPPF(nn.Module):
    """Spatial pyramid pooling - fast (yolov5s.yaml:33)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 5, activation: Optional[str] = "SiLU") -> None:
        super().__init__()
        hidden = in_channels // 2
        self.cv1 = Conv(in_channels, hidden, 1, 1, activation=activation)
        self.cv2 = Conv(hidden * 4, out_channels, 1, 1, activation=activation)
        self.pool = MaxPool5(kernel_size)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.pool(x)
        y2 = self.pool(y1)
        return self.cv2(torch.cat((x, y1, y2, self.pool(y2)), dim=1))


class UpSample(nn.Module):
    def __init__(self, size=None, scale_factor=2, mode: str = "nearest") -> None:
        super().__init__()
        if size is not None or int(scale_factor) != 2 or mode != "nearest":
            raise NotImplementedError("only nearest x2 upsampling is on the YOLOv5 hot path")
        self.scale_factor = 2

    def forward(self, x):
        ops.require_cuda(x, "UpSample.forward")
        return F_.Upsample2xFn.apply(x)


class Concat(nn.Module):
    def __init__(self, dimension: int = 1) -> None:
        super().__init__()
        self.dimension = dimension

    def forward(self, xs: Sequence[torch.Tensor]):
        return torch.cat(list(xs), dim=self.dimension)


class YOLOHead(nn.Module):
    """Detection head.  train: list of nl raw tensors (B, na, ny, nx, nc+5);
    eval: (decoded (B, sum na*ny*nx, nc+5), raw list)  -- train_utils.py:441-444, losses.py:245-256."""

    def __init__(self, n_classes: int, anchors: Sequence[Sequence[float]], in_channels: Sequence[int],
                 strides: Sequence[float]) -> None:
        super().__init__()
        self.nc = int(n_classes)
        self.no = self.nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.out_xyxy = False
        a = torch.tensor(anchors, dtype=torch.float32).view(self.nl, -1, 2)
        self._strides_py = [float(s) for s in strides]      # host copy: no device sync in forward
        self.register_buffer("stride", torch.tensor(list(strides), dtype=torch.float32))
        self.register_buffer("anchors", a / self.stride.view(-1, 1, 1))                 # stride units (loss)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))    # pixels (inference)
        self.conv = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in in_channels)
        self._caches = {}
        self._initialize_biases()

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_caches"] = {}
        return st

    def _cache(self, key):
        c = self._caches.get(key)
        if c is None:
            c = self._caches[key] = F_._WeightCache()
        return c

    def _initialize_biases(self) -> None:
        for conv, s in zip(self.conv, self.stride.tolist()):
            b = conv.bias.view(self.na, -1).detach().clone()
            b[:, 4] += math.log(8 / (640 / s) ** 2)
            b[:, 5:] += math.log(0.6 / (self.nc - 0.99))
            conv.bias = nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, xs: Sequence[torch.Tensor]):
        raws: List[torch.Tensor] = []
        for i, x in enumerate(xs):
            ops.require_cuda(x, "YOLOHead.forward")
            conv = self.conv[i]
            if isinstance(conv, nn.Sequential):      # decomposed head conv (decomposition.py:263-264)
                raise NotImplementedError("Tucker-decomposed head convs")
            raws.append(F_.HeadConvFn.apply(x, conv.weight, conv.bias, self.na, self.no, self._cache(i)))
        if self.training:
            return raws
        B = raws[0].shape[0]
        total = sum(r.shape[1] * r.shape[2] * r.shape[3] for r in raws)
        out = torch.empty((B, total, self.no), dtype=torch.float32, device=raws[0].device)
        off = 0
        for i, r in enumerate(raws):
            apx = self.anchor_grid[i].reshape(-1, 2).float().contiguous()
            ops.head_decode(r, apx, self._strides_py[i], out, off)
            off += r.shape[1] * r.shape[2] * r.shape[3]
        if self.out_xyxy:
            xy, wh = out[..., :2].clone(), out[..., 2:4].clone()
            out[..., :2], out[..., 2:4] = xy - wh / 2, xy + wh / 2
        return out, raws
