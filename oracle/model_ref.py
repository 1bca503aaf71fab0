"""CPU oracle for the model side (TEST INFRASTRUCTURE ONLY -- see oracle/ops_ref.py header).

Pure-PyTorch (torch.nn, CPU, fp32) restatement of the network the reference builds through the un-vendored
``kindle`` package (environment.yml:42; NOT in /root/reference, not installable here): the architecture is
fully determined by the model yaml (res/configs/model/yolov5s.yaml:1-58) plus the module semantics the
reference's call sites rely on:

  Conv      = Conv2d(bias=False, pad=k//2 unless given) -> BatchNorm2d -> SiLU      (yolov5s.yaml:21-50)
  Bottleneck= Conv1x1 -> Conv3x3 (+x)                                              (inside C3, expansion 1.0)
  C3        = cv3(cat(m(cv1(x)), cv2(x)))                                           (yolov5s.yaml:23-52)
  SPPF      = cv2(cat(x, p(x), p(p(x)), p(p(p(x))))), p = MaxPool2d(5,1,2)          (yolov5s.yaml:33)
  UpSample  = nearest x2, Concat = cat(dim=1)                                       (yolov5s.yaml:37-51)
  YOLOHead  = 1x1 conv with bias per level; train: (B,na,ny,nx,no) raw logits (losses.py:245-256);
              eval: xy = (sig*2-0.5+grid)*stride, wh = (sig*2)^2*anchor_px (losses.py:254-255), cat over levels
              P3->P5 (tta_utils.py:52-58).

Pinned by the README parameter counts (README.md:206-211) -- "architecture pinned, numerics unpinned": kindle's
BN eps/momentum and init are unverifiable, torch defaults are used on both sides of every parity test.
Module / parameter NAMES equal the product's so a state_dict moves between the two.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Union

import torch
import yaml
from torch import nn


def _div8(x: float) -> int:
    return int(math.ceil(x / 8) * 8)


class RConv(nn.Module):
    def __init__(self, cin, cout, k=1, s=1, p=None, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, k // 2 if p is None else p, bias=False)
        self.batch_norm = nn.BatchNorm2d(cout)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.batch_norm(self.conv(x)))


class RBottleneck(nn.Module):
    def __init__(self, cin, cout, shortcut=True, e=0.5):
        super().__init__()
        h = int(cout * e)
        self.cv1 = RConv(cin, h, 1, 1)
        self.cv2 = RConv(h, cout, 3, 1)
        self.add = shortcut and cin == cout

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class RC3(nn.Module):
    def __init__(self, cin, cout, n=1, shortcut=True, e=0.5):
        super().__init__()
        h = int(cout * e)
        self.cv1 = RConv(cin, h, 1, 1)
        self.cv2 = RConv(cin, h, 1, 1)
        self.cv3 = RConv(2 * h, cout, 1)
        self.m = nn.Sequential(*[RBottleneck(h, h, shortcut, 1.0) for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class RSPPF(nn.Module):
    def __init__(self, cin, cout, k=5):
        super().__init__()
        h = cin // 2
        self.cv1 = RConv(cin, h, 1, 1)
        self.cv2 = RConv(h * 4, cout, 1, 1)
        self.pool = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.pool(x)
        y2 = self.pool(y1)
        return self.cv2(torch.cat((x, y1, y2, self.pool(y2)), 1))


class RHead(nn.Module):
    def __init__(self, nc, anchors, chans, strides):
        super().__init__()
        self.nc, self.no, self.nl, self.na = nc, nc + 5, len(anchors), len(anchors[0]) // 2
        a = torch.tensor(anchors, dtype=torch.float32).view(self.nl, -1, 2)
        self.register_buffer("stride", torch.tensor(strides, dtype=torch.float32))
        self.register_buffer("anchors", a / self.stride.view(-1, 1, 1))
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.conv = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in chans)

    def forward(self, xs):
        raws, z = [], []
        for i, x in enumerate(xs):
            y = self.conv[i](x)
            B, _, ny, nx = y.shape
            y = y.view(B, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            raws.append(y)
            if not self.training:
                gy, gx = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
                grid = torch.stack((gx, gy), 2).view(1, 1, ny, nx, 2).float()
                s = y.sigmoid()
                xy = (s[..., 0:2] * 2.0 - 0.5 + grid) * self.stride[i]
                wh = (s[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
                z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(B, -1, self.no))
        return raws if self.training else (torch.cat(z, 1), raws)


class RefYOLO(nn.Module):
    def __init__(self, cfg: Union[str, Dict[str, Any]]):
        super().__init__()
        if isinstance(cfg, str):
            with open(cfg) as f:
                cfg = yaml.safe_load(f)
        gd, gw = float(cfg["depth_multiple"]), float(cfg["width_multiple"])
        rows = list(cfg["backbone"]) + list(cfg["head"])
        layers, ch, red = [], [], []
        self.routes = []
        cprev, rprev = int(cfg.get("input_channel", 3)), 1
        for i, row in enumerate(rows):
            frm, rep, name, args = row[0], row[1], row[2], list(row[3])
            fl = frm if isinstance(frm, list) else [frm]
            fa = [i + f if f < 0 else f for f in fl]
            cin = [cprev if (j == i - 1 or i == 0) else ch[j] for j in fa]
            rin = [rprev if (j == i - 1 or i == 0) else red[j] for j in fa]
            n = max(round(rep * gd), 1) if rep > 1 else rep
            if name == "Conv":
                co = _div8(args[0] * gw)
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else None
                m, cout, r = RConv(cin[0], co, k, s, p), co, rin[0] * s
            elif name == "C3":
                co = _div8(args[0] * gw)
                m, cout, r = RC3(cin[0], co, n, args[1] if len(args) > 1 else True), co, rin[0]
            elif name == "SPPF":
                co = _div8(args[0] * gw)
                m, cout, r = RSPPF(cin[0], co, args[1] if len(args) > 1 else 5), co, rin[0]
            elif name == "UpSample":
                m, cout, r = nn.Upsample(scale_factor=2, mode="nearest"), cin[0], rin[0] // 2
            elif name == "Concat":
                m, cout, r = None, sum(cin), rin[0]
            elif name == "YOLOHead":
                m, cout, r = RHead(args[0], args[1], cin, [float(v) for v in rin]), 0, 0
            else:
                raise NotImplementedError(name)
            layers.append(m if m is not None else nn.Identity())
            self.routes.append((frm, name))
            ch.append(cout)
            red.append(r)
            cprev, rprev = cout, r
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        outs = []
        for i, m in enumerate(self.model):
            frm, name = self.routes[i]
            if isinstance(frm, list):
                xin = [x if f == -1 else outs[i + f if f < 0 else f] for f in frm]
                x = torch.cat(xin, 1) if name == "Concat" else m(xin)
            else:
                x = m(x if frm == -1 else outs[i + frm if frm < 0 else frm])
            outs.append(x)
        return x
