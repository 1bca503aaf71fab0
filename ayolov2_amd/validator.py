"""Validation statistics with the reference's interface (scripts/utils/train_utils.py:217-520 ``YoloValidator``,
scripts/utils/metrics.py:446-548 ``compute_ap`` / ``ap_per_class``).

The per-image matching of detections to labels (``process_batch``: IoU matrix, IoU-sorted ``np.unique`` by detection,
then by label, train_utils.py:294-333) runs on the MI355X for ALL images of a batch in one go
(``ayolo_match_detections``), so a validation step has no per-image device->host copy; only the (N, 10) correctness
matrix and the conf / class columns travel at the end, once per batch.  ``ap_per_class`` is the reference's numpy
procedure (it runs once per epoch on a few 10^5 rows on the host).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import call
from .general import scale_coords, xywh2xyxy
from .metrics import non_max_suppression
from .ops import _stream


# --------------------------------------------------------------------------------------------------
# matching
# --------------------------------------------------------------------------------------------------
def match_batch(detections: Sequence[torch.Tensor], labels: Sequence[torch.Tensor], iouv: torch.Tensor) -> List[torch.Tensor]:
    """detections[i]: (n_i, 6) x1,y1,x2,y2,conf,cls in NMS output order; labels[i]: (m_i, 5) cls,x1,y1,x2,y2 (same
    coordinate space) -> per image a (n_i, len(iouv)) bool tensor, train_utils.py:294-333 semantics."""
    assert len(detections) == len(labels)
    B = len(detections)
    dev = iouv.device
    ops.require_cuda(iouv, "match_batch")
    ns = [int(d.shape[0]) for d in detections]
    ms = [int(l.shape[0]) for l in labels]
    N, M = sum(ns), sum(ms)
    niou = int(iouv.shape[0])
    correct = torch.zeros((N, niou), dtype=torch.uint8, device=dev)
    if N and M:
        det = torch.cat([d.reshape(-1, 6) for d in detections]).float().contiguous()
        lab = torch.cat([l.reshape(-1, 5) for l in labels]).float().contiguous()
        det_img = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(ns)).to(dev, non_blocking=True)
        lab_off = torch.tensor(np.concatenate(([0], np.cumsum(ms))), dtype=torch.int32).to(dev, non_blocking=True)
        iouv_f = iouv.float().contiguous()
        best_l = torch.empty(N, dtype=torch.int32, device=dev)
        best_iou = torch.empty(N, dtype=torch.float32, device=dev)
        owner = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        call("ayolo_match_detections", det.data_ptr(), det_img.data_ptr(), N, lab.data_ptr(), lab_off.data_ptr(), M,
             iouv_f.data_ptr(), niou, best_l.data_ptr(), best_iou.data_ptr(), owner.data_ptr(), correct.data_ptr(), _stream())
    return list(correct.bool().split(ns))


def process_batch(detections: torch.Tensor, labels: torch.Tensor, iouv: torch.Tensor) -> torch.Tensor:
    """Single-image form with the reference's argument meaning (train_utils.py:294-333, minus ``self``)."""
    return match_batch([detections], [labels], iouv)[0]


# --------------------------------------------------------------------------------------------------
# AP (host, once per epoch) -- metrics.py:446-548
# --------------------------------------------------------------------------------------------------
def compute_ap(recall, precision) -> Tuple[float, np.ndarray, np.ndarray]:
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)                                     # 101-point interpolation (COCO)
    integrate = getattr(np, "trapezoid", None) or np.trapz
    return integrate(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray, plot: bool = False,
                 save_dir: str = ".", names: Optional[list] = None):
    """(p, r, ap, f1, unique_classes) at the max-mean-F1 confidence; ``plot`` is accepted and ignored (the reference's
    matplotlib curves are out of scope)."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l = (target_cls == c).sum()
        n_p = i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    return p[:, i], r[:, i], ap, f1[:, i], unique_classes.astype("int32")


# --------------------------------------------------------------------------------------------------
# validator
# --------------------------------------------------------------------------------------------------
class YoloValidator:
    """The statistics half of the reference's validator (model call -> NMS -> matching -> AP); data loading, logging,
    plotting, TensorRT and the COCO-json writer stay in the reference.

    cfg_hyp needs ``conf_t`` / ``iou_t``; ``single_cls`` as in cfg_train (train_utils.py:461-469)."""

    def __init__(self, model: torch.nn.Module, device: torch.device, cfg_hyp: Dict[str, Any], single_cls: bool = False,
                 half: bool = False, hybrid_label: bool = False, nms_type: str = "nms", loss_fn=None, tta: bool = False,
                 tta_scales: Sequence[float] = (1, 0.83, 0.67), tta_flips: Sequence[Optional[int]] = (None, 3, None)) -> None:
        self.model, self.device, self.cfg_hyp = model, device, cfg_hyp
        self.tta, self.tta_scales, self.tta_flips = tta, list(tta_scales), list(tta_flips)
        self.single_cls, self.half, self.hybrid_label, self.nms_type, self.loss_fn = single_cls, half, hybrid_label, nms_type, loss_fn
        self.iouv = torch.linspace(0.5, 0.95, 10).to(device)        # mAP@0.5:0.95 (train_utils.py:236-237)
        self.niou = self.iouv.numel()
        self.nc = int(getattr(model, "nc", 0)) or int(model.model[-1].nc)
        self.init_statistics()

    def init_statistics(self) -> None:
        self.seen = 0
        self.loss = torch.zeros(3, device=self.device)
        # dt: seconds spent in pre-process / inference / NMS over the run (train_utils.py:420-470).  On the GPU the stages are
        # asynchronous, so they are bracketed by events on the stream and summed when the statistics are read -- the
        # reference's host clocks around un-synchronised launches would time the enqueue, and a synchronise per stage would
        # serialise the loop
        self.statistics: Dict[str, Any] = {"stats": [], "dt": [0.0, 0.0, 0.0]}
        self._dt_events: list = []

    def _mark(self):
        if self.device.type != "cuda":
            import time
            return time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stage_times(self) -> List[float]:
        """[pre-process, inference, NMS] seconds so far; resolves the pending events (one synchronise)."""
        for marks in self._dt_events:
            for k, (a, b) in enumerate(marks):
                if isinstance(a, float):
                    self.statistics["dt"][k] += b - a
                else:
                    b.synchronize()
                    self.statistics["dt"][k] += a.elapsed_time(b) * 1e-3
        self._dt_events = []
        return self.statistics["dt"]

    @staticmethod
    def convert_target(targets: torch.Tensor, width: int, height: int, n_batch: int = 0) -> torch.Tensor:
        """normalised xywh labels -> pixels (train_utils.py:106-123; works on a copy)."""
        targets = targets.clone()
        targets[:, 2:] *= torch.tensor([width, height, width, height], dtype=targets.dtype).to(targets.device, non_blocking=True)
        return targets

    def process_batch(self, detections: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        return process_batch(detections, labels, self.iouv)

    def convert_trt_out(self, out: torch.Tensor, n_objs: torch.Tensor) -> List[torch.Tensor]:
        """Fixed-shape NMS output ``(B, keepTopK, 6)`` + per-image counts -> the validator's ragged rows
        (train_utils.py:262-283); the counts are the one host read of this path."""
        return [out[i, :int(n)] for i, n in enumerate(n_objs.tolist())]

    @torch.no_grad()
    def validation_step(self, val_batch, batch_idx: int = 0) -> None:
        from ._lib import roctx_range
        imgs, targets, paths, shapes = val_batch
        targets_cpu = targets.detach().float().cpu()                 # labels come from the host loader: no sync needed later
        t0 = self._mark()
        with roctx_range("val.pre_process"):
            imgs = imgs.to(self.device, non_blocking=True)
            if imgs.dtype == torch.uint8:                            # prepare_img (train_utils.py:256-261)
                imgs = imgs.float() / 255.0
            imgs = imgs.half() if self.half else imgs.float()
            targets = targets.to(self.device, non_blocking=True)
        _, _, height, width = imgs.shape
        t1 = self._mark()
        with roctx_range("val.inference"):
            if self.tta:                                             # train_utils.py:425-433
                from .tta import inference_with_tta
                outs = inference_with_tta(self.model, imgs, self.tta_scales, self.tta_flips)
            else:
                outs = self.model(imgs)
        t2 = self._mark()
        out, train_out = (outs[0], outs[1]) if isinstance(outs, (tuple, list)) and len(outs) == 2 else (outs, None)
        trt_case = isinstance(train_out, torch.Tensor)               # engine with the NMS plugin appended (train_utils.py:456-457)
        if self.loss_fn is not None and train_out is not None and not trt_case:
            self.loss += self.loss_fn([x.float() for x in train_out], targets)[1][:3]
        targets = self.convert_target(targets, width, height)
        targets_cpu = self.convert_target(targets_cpu, width, height)
        lb = [targets[targets[:, 0] == i, 1:] for i in range(imgs.shape[0])] if self.hybrid_label else None
        t3 = self._mark()
        with roctx_range("val.nms"):
            if trt_case:
                out = self.convert_trt_out(out, train_out)
            else:
                out = non_max_suppression(out, self.cfg_hyp["conf_t"], self.cfg_hyp["iou_t"], multi_label=True, labels=lb or (),
                                          agnostic=self.single_cls, nms_type=self.nms_type)
        t4 = self._mark()
        self._dt_events.append(((t0, t1), (t1, t2), (t3, t4)))
        self.statistics_per_image(imgs, out, targets, shapes, paths, targets_cpu=targets_cpu)

    def statistics_per_image(self, img: torch.Tensor, out: List[torch.Tensor], targets: torch.Tensor, shapes, paths=None,
                             targets_cpu: Optional[torch.Tensor] = None) -> None:
        """train_utils.py:335-401 for a whole batch: native-space boxes, one matching call, one D2H of the results."""
        n_img = min(len(out), len(shapes))
        tc = targets_cpu if targets_cpu is not None else targets.detach().cpu()
        tcls_all = tc[:, 1].tolist()
        timg = tc[:, 0].long().tolist()
        dets, labs, keep = [], [], []
        for si in range(n_img):
            pred = out[si]
            sel = [k for k, v in enumerate(timg) if v == si]
            labels = targets[sel, 1:] if sel else targets[:0, 1:]
            nl = len(sel)
            tcls = [tcls_all[k] for k in sel]
            self.seen += 1
            if len(pred) == 0:
                if nl:
                    self.statistics["stats"].append((np.zeros((0, self.niou), bool), np.zeros(0, np.float32), np.zeros(0, np.float32), tcls))
                continue
            if self.single_cls:
                pred[:, 5] = 0
            predn = pred.clone()
            shape, ratio_pad = shapes[si][0], shapes[si][1]
            scale_coords(img[si].shape[1:], predn[:, :4], shape, ratio_pad)
            if nl:
                tbox = xywh2xyxy(labels[:, 1:5])
                scale_coords(img[si].shape[1:], tbox, shape, ratio_pad)
                labelsn = torch.cat((labels[:, 0:1], tbox), 1)
            else:
                labelsn = labels[:0, :5]
            dets.append(predn)
            labs.append(labelsn)
            keep.append((pred, tcls))
        if not dets:
            return
        corrects = match_batch(dets, labs, self.iouv)
        # one device->host transfer for the whole batch
        ns = [int(d.shape[0]) for d in dets]
        packed = torch.cat([torch.cat((c.float(), p[:, 4:6].float()), 1) for c, (p, _) in zip(corrects, keep)]).cpu().numpy()
        o = 0
        for n, (_, tcls) in zip(ns, keep):
            blk = packed[o:o + n]
            o += n
            self.statistics["stats"].append((blk[:, :self.niou] > 0.5, blk[:, self.niou], blk[:, self.niou + 1], tcls))

    def compute_statistics(self) -> Dict[str, Any]:
        """train_utils.py:474-520: (mp, mr, map50, map) + per-class arrays."""
        stats = [np.concatenate(x, 0) for x in zip(*self.statistics["stats"])] if self.statistics["stats"] else []
        res: Dict[str, Any] = {"mp": 0.0, "mr": 0.0, "map50": 0.0, "map": 0.0, "seen": self.seen, "dt": list(self.stage_times())}
        if len(stats) and stats[0].any():
            p, r, ap, f1, ap_class = ap_per_class(*stats)
            ap50, ap_m = ap[:, 0], ap.mean(1)
            res.update(mp=float(p.mean()), mr=float(r.mean()), map50=float(ap50.mean()), map=float(ap_m.mean()), p=p, r=r,
                       ap50=ap50, ap=ap_m, ap_class=ap_class)
            res["nt"] = np.bincount(stats[3].astype(np.int64), minlength=max(self.nc, 1))
        else:
            res["nt"] = np.zeros(1)
        return res
