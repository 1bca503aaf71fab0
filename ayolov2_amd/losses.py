"""YOLO loss with the reference's interface (scripts/loss/losses.py:168-391 ``ComputeLoss``).

It sits between the forward and backward of a train step and is < 1 % of its FLOPs, so
it is written with device-agnostic torch ops; the tensors it consumes/produces stay on the GPU and its backward
feeds ``HeadConvFn.backward`` (the HIP path).  Semantics follow the reference:

* ``build_targets``: anchor matching ``max(r, 1/r) < anchor_t`` and the 3-neighbour 0.5-cell offsets
  (losses.py:303-391);
* box loss ``1 - CIoU`` on ``(sigmoid*2-0.5, (sigmoid*2)^2*anchor)`` (losses.py:254-260), objectness BCE against
  the detached IoU with ``balance = [4, 1, 0.4]`` (losses.py:204-206, 285-286), class BCE with label smoothing
  (losses.py:277-279); returns ``(loss * batch_size, cat(lbox, lobj, lcls, loss).detach())`` (losses.py:297-300).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from .metrics import bbox_iou


def smooth_bce(eps: float = 0.1) -> Tuple[float, float]:
    return 1.0 - 0.5 * eps, 0.5 * eps


def _unwrap(model: nn.Module) -> nn.Module:
    if isinstance(model, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)) or type(model).__name__ == "FlatGradDDP":
        return model.module
    return model


class FocalLoss(nn.Module):
    def __init__(self, loss_fcn: nn.BCEWithLogitsLoss, gamma: float = 1.5, alpha: float = 0.25) -> None:
        super().__init__()
        self.loss_fcn, self.gamma, self.alpha = loss_fcn, gamma, alpha
        self.reduction = loss_fcn.reduction
        self.loss_fcn.reduction = "none"

    def forward(self, pred, true):
        loss = self.loss_fcn(pred, true)
        p = torch.sigmoid(pred)
        p_t = true * p + (1 - true) * (1 - p)
        loss = loss * (true * self.alpha + (1 - true) * (1 - self.alpha)) * (1.0 - p_t) ** self.gamma
        return loss.mean() if self.reduction == "mean" else loss.sum() if self.reduction == "sum" else loss


class _FusedLossFn(torch.autograd.Function):
    """Value and analytic gradient of the whole loss in six HIP launches (csrc/loss.hip) instead of ~600 torch kernels
    and their autograd graph.  Same arithmetic as the torch-op path below (which stays as the reference semantics for
    focal loss / autobalance / sort_obj_iou and for CPU tensors)."""

    @staticmethod
    def forward(ctx, cfg, tcls, tbox, indices, anchors, *preds):
        from . import _lib
        from .ops import _stream
        dev = preds[0].device
        nl = len(preds)
        cells = [p.shape[0] * p.shape[1] * p.shape[2] * p.shape[3] for p in preds]
        ns = [int(indices[i][0].shape[0]) for i in range(nl)]
        own = torch.zeros(2 * sum(cells), dtype=torch.int32, device=dev)          # [own | head], one fill
        score = torch.empty(max(sum(ns), 1) * 6, dtype=torch.float32, device=dev)  # [score | next (int32) | rowbox x4]
        nrow = max(sum(ns), 1)
        acc = torch.empty(3 * nl * 64, dtype=torch.float64, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        arr = (_lib.LossLevel * nl)()
        keep = []
        co = no_ = 0
        for i, p in enumerate(preds):
            L = arr[i]
            L.pred = p.data_ptr()
            L.sb, L.sa, L.sy, L.sx = p.stride(0), p.stride(1), p.stride(2), p.stride(3)
            L.B, L.na, L.ny, L.nx, L.no = p.shape
            L.n = ns[i]
            if ns[i]:
                cols = [indices[i][0], indices[i][1], indices[i][2], indices[i][3], tcls[i]]
                cols = [c.contiguous() if c.dtype == torch.int64 else c.long().contiguous() for c in cols]
                tb, an = tbox[i].float().contiguous(), anchors[i].float().contiguous()
                keep += cols + [tb, an]
                L.b, L.a, L.gj, L.gi, L.tcls = (c.data_ptr() for c in cols)
                L.tbox, L.anch = tb.data_ptr(), an.data_ptr()
            L.own = own.data_ptr() + 4 * co
            L.head = own.data_ptr() + 4 * (sum(cells) + co)
            L.score = score.data_ptr() + 4 * no_
            L.next = score.data_ptr() + 4 * (nrow + no_)
            L.rowbox = score.data_ptr() + 4 * (2 * nrow + 4 * no_)
            L.balance = float(cfg["balance"][i])
            L.grad = None
            co += cells[i]
            no_ += ns[i]
        consts = (cfg["cp"], cfg["cn"], cfg["cls_pw"], cfg["obj_pw"], cfg["gr"], cfg["box"], cfg["obj"], cfg["cls"])
        _lib.call("ayolo_yolo_loss_fwd", arr, nl, *consts, acc.data_ptr(), out.data_ptr(), _stream())
        ctx.arr, ctx.consts, ctx.keep = arr, consts, keep + [own, score]
        ctx.packed_ok = bool(cfg.get("packed", True))
        ctx.save_for_backward(*preds)
        items = out[1:5]
        ctx.mark_non_differentiable(items)
        return out[0:1], items

    @staticmethod
    def backward(ctx, g_loss, _g_items):
        from . import _lib
        from .ops import _stream, dtype_code
        g = g_loss.detach().reshape(-1)[:1].float().contiguous()
        preds = ctx.saved_tensors
        # Packed mode: every logit tensor is a YOLOHead output of this package (tagged by HeadConvFn / the plan), whose
        # backward understands the side channel: the gradient is written directly as that conv's backward operand
        # (NHWC, compute dtype) + bias gradient, and autograd only carries a zero-stride placeholder that points to it.
        tags = [getattr(p, "_ayolo_head", None) for p in preds]
        if ctx.packed_ok and all(t is not None for t in tags) and len({t[1] for t in tags}) == 1:
            grads = []
            for i, (p, tag) in enumerate(zip(preds, tags)):
                ldz, dt = tag[0], tag[1]
                B, na, ny, nx, no = p.shape
                dz = torch.empty((B * ny * nx, ldz), dtype=dt, device=p.device)
                # the plan executor hands over the (zeroed) slot of the head bias in its gradient arena: the bias gradient is
                # accumulated in place, ready for the first all-reduce bucket; otherwise a fresh accumulator
                dbias = tag[2] if len(tag) > 2 and tag[2] is not None else torch.zeros(na * no, dtype=torch.float32, device=p.device)
                L = ctx.arr[i]
                L.dz, L.ldz, L.dz_dtype, L.dbias = dz.data_ptr(), ldz, dtype_code(dt), dbias.data_ptr()
                ph = torch.zeros((), dtype=torch.float32, device=p.device).expand(p.shape)
                ph._ayolo_packed = (dz, dbias)
                if len(_PENDING_PACKED) > 64:             # backward passes that never reached their head
                    _PENDING_PACKED.clear()
                _PENDING_PACKED[p.data_ptr()] = (dz, dbias)
                grads.append(ph)
            _lib.call("ayolo_yolo_loss_bwd_packed", ctx.arr, len(grads), *ctx.consts, g.data_ptr(), _stream())
            return (None, None, None, None, None, *grads)
        grads = []
        for i, p in enumerate(preds):
            gr = torch.empty(p.shape, dtype=torch.float32, device=p.device)       # contiguous (B,na,ny,nx,no)
            ctx.arr[i].grad = gr.data_ptr()
            grads.append(gr)
        _lib.call("ayolo_yolo_loss_bwd", ctx.arr, len(grads), *ctx.consts, g.data_ptr(), _stream())
        return (None, None, None, None, None, *grads)


# Packed loss gradients that have been produced but not yet consumed by their YOLOHead backward, keyed by the address
# of the head's raw-logit buffer.  The placeholder autograd carries is all zeros, so if the raw tensor has a SECOND
# differentiable consumer (distillation / auxiliary loss) autograd hands the head `0 + other gradients` as a new dense
# tensor that no longer carries the payload attribute: the head backward then finds the payload here and adds the two.
_PENDING_PACKED: dict = {}


def take_packed_head_grad(draw, ldz: int, dtype, raw_ptr: Optional[int] = None):
    """For YOLOHead backward implementations: returns (dz, dbias) -- the head conv's backward operand (NHWC rows of
    `ldz` channels, compute dtype) and bias gradient -- when the fused loss produced its gradient in that layout for
    the head whose raw-logit buffer starts at `raw_ptr`, else None (`draw` is then an ordinary dense gradient).
    Gradients of other consumers of the same logits that reached the head next to the packed one are added in."""
    pend = _PENDING_PACKED.pop(raw_ptr, None) if raw_ptr is not None else None
    if draw is None and pend is None:
        return None
    pk = getattr(draw, "_ayolo_packed", None) if draw is not None else None
    if pk is not None:
        dz, dbias = pk
        if dz.shape[1] == ldz and dz.dtype == dtype:
            return dz, dbias
        raise RuntimeError("packed head gradient has the wrong layout for this head")
    if pend is not None:
        dz, dbias = pend
        if dz.shape[1] != ldz or dz.dtype != dtype:
            raise RuntimeError("packed head gradient has the wrong layout for this head")
        if draw is not None:
            # the loss gradient (packed) plus what the other consumers of these logits sent (dense): rare, plain torch
            from . import ops
            dz2, dbias2 = ops.head_grad_pack(draw, dtype, ldz, want_bias=True)
            B, _, H, W = dz2.shape
            dz = dz + dz2.permute(0, 2, 3, 1).reshape(B * H * W, ldz)
            dbias.add_(dbias2)              # in place: `dbias` may be the head bias' slot in the plan's gradient arena
        return dz, dbias
    if draw.dim() == 5 and all(s == 0 for s in draw.stride()) and draw.numel() > 1:
        raise RuntimeError("packed head-gradient placeholder without payload reached a YOLOHead backward; "
                           "set ComputeLoss.packed_head_grad = False")
    return None


class ComputeLoss:
    def __init__(self, model: nn.Module, autobalance: bool = False) -> None:
        self.sort_obj_iou = False
        m = _unwrap(model)
        device = next(m.parameters()).device
        hyp: Dict[str, Any] = m.hyp
        bce_cls: nn.Module = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([hyp["cls_pw"]], device=device))
        bce_obj: nn.Module = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([hyp["obj_pw"]], device=device))
        self.cp, self.cn = smooth_bce(eps=hyp.get("label_smoothing", 0.0))
        if hyp["fl_gamma"] > 0:
            bce_cls, bce_obj = FocalLoss(bce_cls, hyp["fl_gamma"]), FocalLoss(bce_obj, hyp["fl_gamma"])
        head = m.model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(head.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.ssi = list(head.stride).index(16) if autobalance else 0
        self.BCEcls, self.BCEobj, self.gr, self.hyp, self.autobalance = bce_cls, bce_obj, 1.0, hyp, autobalance
        self.na, self.nc, self.nl, self.anchors = head.na, head.nc, head.nl, head.anchors
        self._anchors_cpu = None
        self.fused = True          # use csrc/loss.hip when the configuration allows (see _fusable)
        self.packed_head_grad = True   # hand the gradient to this package's YOLOHead backward in its operand layout

    def _fusable(self, preds) -> bool:
        return (self.fused and self.hyp["fl_gamma"] <= 0 and not self.autobalance and not self.sort_obj_iou
                and all(p.is_cuda and p.dtype == torch.float32 and p.dim() == 5 and p.stride(4) == 1 for p in preds)
                and len(preds) <= 8)

    def prepare(self, targets: torch.Tensor, pred_shapes, device=None):
        """Target assignment done on the HOST (labels come from the CPU data loader) before / while the forward
        runs, so the loss itself contains no device->host synchronisation: the boolean-mask gathers of
        ``build_targets`` would otherwise stall the stream between forward and backward.  Returns what
        ``build_targets`` returns, moved to ``device``; pass it to ``__call__(..., prepared=...)``."""
        device = device if device is not None else targets.device
        if self._anchors_cpu is None:
            self._anchors_cpu = self.anchors.detach().float().cpu()
        t_cpu = targets.detach().float().cpu()
        # numpy, not torch-CPU: torch's intra-op thread pool (one spinning OpenMP thread per host core) starves the
        # HIP runtime's helper threads and stalls the GPU for tens of ms every other step
        tcls, tbox, indices, anch = self._build_targets_np(t_cpu.numpy(), pred_shapes, self._anchors_cpu.numpy())
        if torch.device(device).type != "cuda":
            return tcls, tbox, indices, anch
        # two pinned staging buffers -> two asynchronous H2D copies (pageable copies would block on the stream)
        ints = [c for c in tcls] + [i for idx in indices for i in idx]
        flts = [b.reshape(-1) for b in tbox] + [a.reshape(-1) for a in anch]
        ni, nf = sum(t.numel() for t in ints), sum(t.numel() for t in flts)
        # The copies are asynchronous and the host runs about one step ahead of the GPU: a staging buffer may only be
        # rewritten after the copy that read it has completed.  Ring of three buffer pairs, each with the event recorded
        # behind its copies (the wait is a no-op unless the host is three steps ahead).
        ring = getattr(self, "_pin_ring", None)
        if ring is None:
            ring = self._pin_ring = [{"i": None, "f": None, "ev": None} for _ in range(3)]
            self._pin_next = 0
        slot = ring[self._pin_next % len(ring)]
        self._pin_next += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["i"] is None or slot["i"].numel() < ni:
            slot["i"] = torch.empty(max(ni, 1) * 2, dtype=torch.int64).pin_memory()
        if slot["f"] is None or slot["f"].numel() < nf:
            slot["f"] = torch.empty(max(nf, 1) * 2, dtype=torch.float32).pin_memory()
        pin_i, pin_f = slot["i"], slot["f"]
        mode = getattr(self, "h2d_mode", "same")
        if mode == "pageable":
            di = torch.cat(ints).to(device) if ni else torch.empty(0, dtype=torch.int64, device=device)
            df = torch.cat(flts).to(device) if nf else torch.empty(0, device=device)
        else:
            torch.cat(ints, out=pin_i[:ni]) if ni else None
            torch.cat(flts, out=pin_f[:nf]) if nf else None
            if mode == "side":     # copy engine on its own stream; the compute stream only waits on the event
                if getattr(self, "_copy_stream", None) is None:
                    self._copy_stream = torch.cuda.Stream(device=device)
                cur = torch.cuda.current_stream(device)
                with torch.cuda.stream(self._copy_stream):
                    di = pin_i[:ni].to(device, non_blocking=True)
                    df = pin_f[:nf].to(device, non_blocking=True)
                    slot["ev"] = torch.cuda.Event()
                    slot["ev"].record()
                cur.wait_stream(self._copy_stream)
                di.record_stream(cur)
                df.record_stream(cur)
            else:
                di = pin_i[:ni].to(device, non_blocking=True)
                df = pin_f[:nf].to(device, non_blocking=True)
                slot["ev"] = torch.cuda.Event()
                slot["ev"].record()
        oi, of = 0, 0

        def take_i(t):
            nonlocal oi
            v = di[oi:oi + t.numel()]
            oi += t.numel()
            return v

        def take_f(t):
            nonlocal of
            v = df[of:of + t.numel()].view(t.shape)
            of += t.numel()
            return v

        d_tcls = [take_i(c) for c in tcls]
        d_idx = [tuple(take_i(i) for i in idx) for idx in indices]
        d_tbox = [take_f(b) for b in tbox]
        d_anch = [take_f(a) for a in anch]
        return d_tcls, d_tbox, d_idx, d_anch

    def __call__(self, preds: List[torch.Tensor], targets: torch.Tensor, prepared=None) -> Tuple[torch.Tensor, torch.Tensor]:
        device = preds[0].device
        lcls = torch.zeros(1, device=device)
        lbox = torch.zeros(1, device=device)
        lobj = torch.zeros(1, device=device)
        tcls, tbox, indices, anchors = prepared if prepared is not None else self.build_targets(preds, targets)
        if self._fusable(preds):
            cfg = {"balance": self.balance, "cp": self.cp, "cn": self.cn, "cls_pw": float(self.hyp["cls_pw"]),
                   "obj_pw": float(self.hyp["obj_pw"]), "gr": float(self.gr), "box": float(self.hyp["box"]),
                   "obj": float(self.hyp["obj"]), "cls": float(self.hyp["cls"]), "packed": self.packed_head_grad}
            loss_bs, items = _FusedLossFn.apply(cfg, tcls, tbox, indices, anchors, *preds)
            return loss_bs, items
        for i, pi in enumerate(preds):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros_like(pi[..., 0], device=device)
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, :2].sigmoid() * 2.0 - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                iou = bbox_iou(torch.cat((pxy, pwh), 1).T, tbox[i], x1y1x2y2=False, c_iou=True)
                lbox = lbox + (1.0 - iou).mean()
                score = iou.detach().clamp(0).type(tobj.dtype)
                if self.sort_obj_iou:
                    order = torch.argsort(score)
                    b, a, gj, gi, score = b[order], a[order], gj[order], gi[order], score[order]
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * score
                if self.nc > 1:
                    # one-hot class targets; scatter_ with a Python scalar launches no host->device copy (an
                    # index_put of a CPU scalar would block on the stream)
                    t = torch.full_like(ps[:, 5:], self.cn, device=device)
                    t.scatter_(1, tcls[i].view(-1, 1), self.cp)
                    lcls = lcls + self.BCEcls(ps[:, 5:], t)
            obji = self.BCEobj(pi[..., 4], tobj)
            lobj = lobj + obji * self.balance[i]
            if self.autobalance:
                self.balance[i] = self.balance[i] * 0.9999 + 0.0001 / obji.detach().item()
        if self.autobalance:
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        lbox = lbox * self.hyp["box"]
        lobj = lobj * self.hyp["obj"]
        lcls = lcls * self.hyp["cls"]
        bs = preds[0].shape[0]
        loss = lbox + lobj + lcls
        return loss * bs, torch.cat((lbox, lobj, lcls, loss)).detach()

    def _build_targets_np(self, targets, pred_shapes, anchors_all):
        """float32 numpy twin of ``build_targets`` (same operations in the same order)."""
        import numpy as np
        f32 = np.float32
        na, nt = self.na, targets.shape[0]
        tcls, tbox, indices, anch = [], [], [], []
        gain = np.ones(7, f32)
        ai = np.repeat(np.arange(na, dtype=f32).reshape(na, 1), nt, 1)
        tg = np.concatenate((np.broadcast_to(targets.astype(f32), (na, nt, 6)), ai[:, :, None]), 2)
        g = f32(0.5)
        off = np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], f32) * g
        for i in range(self.nl):
            anchors = anchors_all[i].astype(f32)
            shape = pred_shapes[i]
            gain[2:6] = np.array([shape[3], shape[2], shape[3], shape[2]], f32)
            t = tg * gain
            if nt:
                r = t[:, :, 4:6] / anchors[:, None]
                keep = np.maximum(r, f32(1.0) / r).max(2) < f32(self.hyp["anchor_t"])
                t = t[keep]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                jk = (np.mod(gxy, f32(1.0)) < g) & (gxy > f32(1.0))
                lm = (np.mod(gxi, f32(1.0)) < g) & (gxi > f32(1.0))
                sel = np.stack((np.ones_like(jk[:, 0]), jk[:, 0], jk[:, 1], lm[:, 0], lm[:, 1]))
                t = np.broadcast_to(t, (5,) + t.shape)[sel]
                offsets = (np.zeros_like(gxy)[None] + off[:, None])[sel]
            else:
                t = tg[0]
                offsets = f32(0)
            b, c = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
            gxy, gwh = t[:, 2:4], t[:, 4:6]
            gij = (gxy - offsets).astype(np.int64)
            gi, gj = gij[:, 0], gij[:, 1]
            a = t[:, 6].astype(np.int64)
            indices.append((torch.from_numpy(b), torch.from_numpy(a), torch.from_numpy(np.clip(gj, 0, int(shape[2]) - 1)),
                            torch.from_numpy(np.clip(gi, 0, int(shape[3]) - 1))))
            tbox.append(torch.from_numpy(np.concatenate((gxy - gij.astype(f32), gwh), 1).astype(f32)))
            anch.append(torch.from_numpy(anchors[a]))
            tcls.append(torch.from_numpy(c))
        return tcls, tbox, indices, anch

    def build_targets(self, preds: List[torch.Tensor], targets: torch.Tensor, anchors: Optional[torch.Tensor] = None):
        """targets: (nt, 6) [image, class, x, y, w, h] normalised -> per level (classes, boxes, indices, anchors)."""
        all_anchors = self.anchors if anchors is None else anchors
        na, nt = self.na, targets.shape[0]
        dev = targets.device
        tcls, tbox, indices, anch = [], [], [], []
        gain = torch.ones(7, device=dev)
        ai = torch.arange(na, device=dev).float().view(na, 1).repeat(1, nt)
        targets = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)      # (na, nt, 7)
        g = 0.5
        off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], device=dev).float() * g
        for i in range(self.nl):
            anchors = all_anchors[i].to(dev)
            shape = preds[i].shape
            gain[2:6] = torch.tensor([shape[3], shape[2], shape[3], shape[2]], device=dev, dtype=gain.dtype)
            t = targets * gain
            if nt:
                r = t[:, :, 4:6] / anchors[:, None]
                keep = torch.max(r, 1.0 / r).max(2)[0] < self.hyp["anchor_t"]
                t = t[keep]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1.0 < g) & (gxy > 1.0)).T
                l, m = ((gxi % 1.0 < g) & (gxi > 1.0)).T
                sel = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[sel]
                offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
            else:
                t = targets[0]
                offsets = 0
            b, c = t[:, :2].long().T
            gxy, gwh = t[:, 2:4], t[:, 4:6]
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            a = t[:, 6].long()
            indices.append((b, a, gj.clamp_(0, int(shape[2]) - 1), gi.clamp_(0, int(shape[3]) - 1)))
            tbox.append(torch.cat((gxy - gij, gwh), 1))
            anch.append(anchors[a])
            tcls.append(c)
        return tcls, tbox, indices, anch
