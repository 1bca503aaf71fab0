"""YOLOModel: builds the backbone-neck-head ``nn.Module`` from the reference's model yaml
(res/configs/model/yolov5{n,s,m,l,x}.yaml) and keeps the surface the reference's train.py / val.py /
decompose_model.py / losses.py require of ``kindle.YOLOModel``:

    YOLOModel(cfg: dict | str, verbose=False)
    .model            indexable container, parameter names ``model.{i}.``; ``model[-1]`` is the YOLOHead
    .stride           tensor([8., 16., 32.])
    .model_parser.cfg the parsed yaml dict
    .fuse() / .export() / .profile()
    forward(x)        train: list of nl raw tensors; eval: (decoded, raw list)
"""
from __future__ import annotations

import math
import time
from types import SimpleNamespace
from typing import Any, Dict, List, Sequence, Union

import torch
import yaml
from torch import nn

from .modules import C3, SPPF, Bottleneck, Concat, Conv, UpSample, YOLOHead


def make_divisible(x: float, divisor: int = 8) -> int:
    return int(math.ceil(x / divisor) * divisor)


class _Layer(SimpleNamespace):
    pass


class ModelParser:
    def __init__(self, cfg: Union[str, Dict[str, Any]]):
        if isinstance(cfg, str):
            with open(cfg, "r") as f:
                cfg = yaml.safe_load(f)
        self.cfg: Dict[str, Any] = cfg


class YOLOModel(nn.Module):
    def __init__(self, cfg: Union[str, Dict[str, Any]] = "yolov5s.yaml", verbose: bool = False) -> None:
        super().__init__()
        self.model_parser = ModelParser(cfg)
        c = self.model_parser.cfg
        self.input_channel = int(c.get("input_channel", 3))
        depth, width = float(c["depth_multiple"]), float(c["width_multiple"])
        rows = list(c["backbone"]) + list(c["head"])
        layers: List[nn.Module] = []
        self.routes: List[Any] = []
        ch: List[int] = []            # output channels per layer
        red: List[int] = []           # spatial reduction (stride) per layer
        c_prev, r_prev = self.input_channel, 1
        for i, row in enumerate(rows):
            frm, rep, name, args = row[0], row[1], row[2], list(row[3])
            kw = dict(row[4]) if len(row) > 4 and row[4] else {}
            act = kw.get("activation", "SiLU")
            f_list = frm if isinstance(frm, list) else [frm]
            f_abs = [(i + f) if f < 0 else f for f in f_list]
            cin = [c_prev if j == i - 1 or i == 0 and j == -1 else ch[j] for j in f_abs] if i > 0 else [c_prev]
            rin = [r_prev if j == i - 1 else red[j] for j in f_abs] if i > 0 else [1]
            n = max(round(rep * depth), 1) if rep > 1 else rep
            if name == "Conv":
                out = make_divisible(args[0] * width)
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else None
                m: nn.Module = Conv(cin[0], out, k, s, p, activation=act)
                cout, r = out, rin[0] * s
            elif name == "C3":
                out = make_divisible(args[0] * width)
                shortcut = args[1] if len(args) > 1 else True
                m = C3(cin[0], out, n, shortcut, activation=act)
                cout, r = out, rin[0]
            elif name == "Bottleneck":
                out = make_divisible(args[0] * width)
                m = Bottleneck(cin[0], out, *(args[1:]), activation=act)
                cout, r = out, rin[0]
            elif name == "SPPF":
                out = make_divisible(args[0] * width)
                m = SPPF(cin[0], out, args[1] if len(args) > 1 else 5, activation=act)
                cout, r = out, rin[0]
            elif name == "UpSample":
                m = UpSample(args[0] if len(args) > 0 else None, args[1] if len(args) > 1 else 2)
                cout, r = cin[0], rin[0] // 2 if rin[0] > 1 else rin[0]
            elif name == "Concat":
                m = Concat(args[0] if args else 1)
                cout, r = sum(cin), rin[0]
            elif name == "YOLOHead":
                m = YOLOHead(args[0], args[1], cin, [float(x) for x in rin])
                cout, r = 0, 0
            else:
                raise NotImplementedError(f"module {name!r} is outside the YOLOv5 hot path (SURVEY.md section 2)")
            layers.append(m)
            self.routes.append(frm)
            ch.append(cout)
            red.append(r)
            c_prev, r_prev = cout, r
        self.model = nn.Sequential(*layers)
        self.save = sorted({(i + f) if f < 0 else f for i, frm in enumerate(self.routes)
                            for f in (frm if isinstance(frm, list) else [frm]) if f != -1})
        head = self.model[-1]
        self.stride = head.stride.clone() if isinstance(head, YOLOHead) else torch.tensor([32.0])
        self.names = [str(i) for i in range(getattr(head, "nc", 0))]
        if verbose:
            n_param = sum(p.numel() for p in self.parameters())
            print(f"YOLOModel: {len(layers)} layers, {n_param:,d} parameters")

    # ------------------------------------------------------------------------------------------
    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save of the module (yolo_trainer.py:379-386 deep-copies the EMA model into every
        checkpoint): the cached executor plans are derived data -- GBs of static activations and ctypes op arrays that
        cannot be pickled -- and are rebuilt on the copy's first forward."""
        state = self.__dict__.copy()
        state.pop("_plans", None)
        return state

    def forward(self, x: torch.Tensor):
        if self.training:
            from . import ops
            if x.is_cuda and x.dim() == 4 and getattr(self, "use_plan", True):
                from .plan import plan_forward_train
                raws = plan_forward_train(self, x)     # static-plan fast path (None: unsupported structure)
                if raws is not None:
                    return raws
            if getattr(self, "_ayolo_grad_sync", None) is not None:
                raise RuntimeError("FlatGradDDP needs the plan executor (model not plannable / use_plan off): "
                                   "wrap with torch DistributedDataParallel instead")
            ops.ARENA.reset()          # one fill for all BN accumulators of this step
        elif x.is_cuda and x.dim() == 4 and getattr(self, "use_plan", True) and not torch.is_grad_enabled():
            from .infer_plan import plan_forward_eval
            res = plan_forward_eval(self, x)           # static inference executor (None: unsupported structure)
            if res is not None:
                return res
        outs: List[Any] = []
        for i, m in enumerate(self.model):
            frm = self.routes[i]
            if isinstance(frm, list):
                xin = [x if f == -1 else outs[(i + f) if f < 0 else f] for f in frm]
            elif frm != -1:
                xin = outs[(i + frm) if frm < 0 else frm]
            else:
                xin = x
            x = m(xin)
            outs.append(x if i in self.save else None)
        return x

    def fuse(self) -> "YOLOModel":
        for m in self.modules():
            if isinstance(m, Conv):
                m.fuse()
        return self

    def export(self, verbose: bool = False) -> "YOLOModel":
        """kindle's export() switches the head to a deployment output; here it is fuse() + eval()."""
        self.fuse().eval()
        return self

    def profile(self, input_size=(640, 640), batch_size: int = 1, n_run: int = 10, verbose: bool = False):
        dev = next(self.parameters()).device
        x = torch.rand(batch_size, self.input_channel, *input_size, device=dev)
        with torch.no_grad():
            self(x)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_run):
                self(x)
            if dev.type == "cuda":
                torch.cuda.synchronize()
        return SimpleNamespace(total_run_time=time.perf_counter() - t0, n_run=n_run)
