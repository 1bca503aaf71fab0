"""CPU oracle for the detection-side task math (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package ``ayolov2_amd`` never does (a product path that reached this file would
void every parity claim).

numpy/float32 restatement of the reference's L3 functions.  Every function cites the reference
``file:line`` (relative to /root/reference) whose arithmetic it follows, operation for operation, so
that float32 results are the same IEEE sequence the reference's torch-CPU path produces:

* ``xywh2xyxy``            scripts/utils/general.py:297-321
* ``clip_coords``          scripts/utils/general.py:203-230
* ``scale_coords``         scripts/utils/general.py:324-358
* ``box_iou``              scripts/utils/metrics.py:138-164
* ``bbox_iou``             scripts/utils/metrics.py:60-135
* ``non_max_suppression``  scripts/utils/metrics.py:285-443
* ``batched_nms``          scripts/utils/nms.py:15-116
* ``batched_nms_trt``      TensorRT OSS ``batchedNMSPlugin`` ("BatchedNMS_TRT" v1) with the fields the reference sets
  (scripts/model_converter/model_converter.py:268-388); third-party binary, NOT in /root/reference: restated from the
  published plugin source (sortScoresPerClass -> allClassNMS -> sortScoresPerImage -> gatherNMSOutputs).
  "parity unpinned" at that leaf.
* ``tv_nms`` / ``tv_batched_nms``  torchvision==0.10.1 (environment.yml:28; NOT vendored, restated from
  the published source -- see oracle/nms_oracle.c header).  "parity unpinned" at that leaf.

Pinned by ``tests/golden/*.npz`` which ``tools/make_golden.py`` produced by importing the reference's
own Python (with this file's ``tv_nms`` bound as the ``torchvision.ops.nms`` stand-in).

Deliberate, documented deviations:
* the 10 s wall-clock ``time_limit`` break (metrics.py:328,439-441) is not reproduced ("never reached");
* sort tie order is defined as STABLE (equal keys keep ascending input order).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-C", _HERE, "-s"])
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int64
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_double, ctypes.c_void_p]
        lib.oracle_box_iou.restype = None
        lib.oracle_box_iou.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p]
        lib.oracle_argsort_desc.restype = None
        lib.oracle_argsort_desc.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def argsort_desc(key) -> np.ndarray:
    """Stable descending argsort (ties keep ascending input order)."""
    key = _f32(key).reshape(-1)
    out = np.empty(key.shape[0], dtype=np.int64)
    _lib().oracle_argsort_desc(key.ctypes.data, key.shape[0], out.ctypes.data)
    return out


# ----------------------------------------------------------------------------------------------
# torchvision leaves
# ----------------------------------------------------------------------------------------------
def tv_nms(boxes, scores, iou_threshold: float, cls=None) -> np.ndarray:
    """torchvision.ops.nms (0.10.1 CPU kernel semantics): kept indices, descending score."""
    boxes = _f32(boxes).reshape(-1, 4)
    scores = _f32(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    c = None
    if cls is not None:
        c = _f32(cls).reshape(-1)
    nk = _lib().oracle_nms(boxes.ctypes.data, scores.ctypes.data, c.ctypes.data if c is not None else None,
                           n, float(iou_threshold), keep.ctypes.data)
    return keep[:nk].copy()


def tv_batched_nms(boxes, scores, idxs, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.boxes.batched_nms, 0.10.1: `_batched_nms_vanilla` when boxes.numel() > 4000
    (per-class nms, result re-sorted by score), else the coordinate trick
    (offset = idx * (boxes.max() + 1))."""
    boxes = _f32(boxes).reshape(-1, 4)
    scores = _f32(scores).reshape(-1)
    idxs = _f32(idxs).reshape(-1)
    if boxes.size == 0:
        return np.empty((0,), dtype=np.int64)
    if boxes.size > 4000:
        # per-class greedy; kept set re-sorted by score (stable) == one greedy pass over the
        # score-sorted list in which only same-class pairs interact.
        return tv_nms(boxes, scores, iou_threshold, cls=idxs)
    max_coordinate = boxes.max()
    offsets = idxs * (max_coordinate + np.float32(1))
    return tv_nms(boxes + offsets[:, None], scores, iou_threshold)


# ----------------------------------------------------------------------------------------------
# general.py
# ----------------------------------------------------------------------------------------------
def xywh2xyxy(x, ratio=(1.0, 1.0), wh=(1.0, 1.0), pad=(0.0, 0.0)) -> np.ndarray:
    """general.py:297-321: ratio*wh*(c -/+ size/2) + pad."""
    x = _f32(x)
    y = x.copy()
    rx = np.float32(ratio[0] * wh[0])
    ry = np.float32(ratio[1] * wh[1])
    px, py = np.float32(pad[0]), np.float32(pad[1])
    hw = x[:, 2] / np.float32(2)
    hh = x[:, 3] / np.float32(2)
    y[:, 0] = rx * (x[:, 0] - hw) + px
    y[:, 1] = ry * (x[:, 1] - hh) + py
    y[:, 2] = rx * (x[:, 0] + hw) + px
    y[:, 3] = ry * (x[:, 1] + hh) + py
    return y


def clip_coords(boxes, wh, inplace: bool = True) -> np.ndarray:
    """general.py:203-230 (xyxy clipped to (w, h))."""
    b = boxes if inplace else np.array(boxes, copy=True)
    w, h = np.float32(wh[0]), np.float32(wh[1])
    b[:, 0] = np.clip(b[:, 0], 0, w)
    b[:, 1] = np.clip(b[:, 1], 0, h)
    b[:, 2] = np.clip(b[:, 2], 0, w)
    b[:, 3] = np.clip(b[:, 3], 0, h)
    return b


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None) -> np.ndarray:
    """general.py:324-358 (in place on coords; shapes are (h, w))."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = ((img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2)
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= np.float32(pad[0])
    coords[:, [1, 3]] -= np.float32(pad[1])
    coords[:, :4] /= np.float32(gain)
    clip_coords(coords, (img0_shape[1], img0_shape[0]))
    return coords


# ----------------------------------------------------------------------------------------------
# metrics.py
# ----------------------------------------------------------------------------------------------
def box_iou(box1, box2) -> np.ndarray:
    """metrics.py:138-164 dense (N, M) IoU."""
    a = _f32(box1).reshape(-1, 4)
    b = _f32(box2).reshape(-1, 4)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    if out.size:
        _lib().oracle_box_iou(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out


def bbox_iou(box1, box2, x1y1x2y2=True, g_iou=False, d_iou=False, c_iou=False, eps=1e-7) -> np.ndarray:
    """metrics.py:60-135.  box1 is (4, n), box2 is (n, 4); float32 throughout (eps added as f32)."""
    b1 = _f32(box1)
    b2 = _f32(box2).T
    e = np.float32(eps)
    two = np.float32(2)
    if x1y1x2y2:
        b1_x1, b1_y1, b1_x2, b1_y2 = b1[0], b1[1], b1[2], b1[3]
        b2_x1, b2_y1, b2_x2, b2_y2 = b2[0], b2[1], b2[2], b2[3]
    else:
        b1_x1, b1_x2 = b1[0] - b1[2] / two, b1[0] + b1[2] / two
        b1_y1, b1_y2 = b1[1] - b1[3] / two, b1[1] + b1[3] / two
        b2_x1, b2_x2 = b2[0] - b2[2] / two, b2[0] + b2[2] / two
        b2_y1, b2_y2 = b2[1] - b2[3] / two, b2[1] + b2[3] / two
    zero = np.float32(0)
    inter = np.maximum(np.minimum(b1_x2, b2_x2) - np.maximum(b1_x1, b2_x1), zero) * np.maximum(
        np.minimum(b1_y2, b2_y2) - np.maximum(b1_y1, b2_y1), zero)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + e
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + e
    union = w1 * h1 + w2 * h2 - inter + e
    iou = inter / union
    if g_iou or d_iou or c_iou:
        cw = np.maximum(b1_x2, b2_x2) - np.minimum(b1_x1, b2_x1)
        ch = np.maximum(b1_y2, b2_y2) - np.minimum(b1_y1, b2_y1)
        if c_iou or d_iou:
            c2 = cw ** 2 + ch ** 2 + e
            rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / np.float32(4)
            if d_iou:
                return iou - rho2 / c2
            v = np.float32(4 / math.pi ** 2) * (np.arctan(w2 / h2) - np.arctan(w1 / h1)) ** 2
            alpha = v / (v - iou + np.float32(1 + eps))
            return iou - (rho2 / c2 + v * alpha)
        c_area = cw * ch + e
        return iou - (c_area - union) / c_area
    return iou


_MAX_WH = 4096
_MAX_NMS = 30000


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300, nms_type="nms") -> List[np.ndarray]:
    """metrics.py:285-443.  Returns one (n_i, 6) float32 array [x1,y1,x2,y2,conf,cls] per image."""
    pred = _f32(prediction)
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    B, _, no = pred.shape
    nc = no - 5
    ct = np.float32(conf_thres)            # torch compares a f32 tensor with the python scalar in f32
    it32 = np.float32(iou_thres)
    multi_label = bool(multi_label) and nc > 1
    out: List[np.ndarray] = [np.zeros((0, 6), np.float32) for _ in range(B)]
    for xi in range(B):
        x = pred[xi]
        x = x[x[:, 4] > ct].copy()                                   # :337
        if labels and len(labels[xi]):                               # :340-346
            lab = _f32(labels[xi])
            v = np.zeros((lab.shape[0], nc + 5), np.float32)
            v[:, :4] = lab[:, 1:5]
            v[:, 4] = 1.0
            v[np.arange(lab.shape[0]), lab[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]                                        # :353
        box = xywh2xyxy(x[:, :4])                                    # :356
        if multi_label:                                              # :359-361
            i, j = np.nonzero(x[:, 5:] > ct)
            x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(np.float32)), 1)
        else:                                                        # :363-364
            j = np.argmax(x[:, 5:], axis=1)                          # first max index, as torch
            conf = x[np.arange(x.shape[0]), j + 5]
            x = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)[conf > ct]
        if classes is not None:                                      # :367-368
            x = x[np.isin(x[:, 5], np.asarray(classes, np.float32))]
        n = x.shape[0]
        if not n:
            continue
        if n > _MAX_NMS:                                             # :378-379
            x = x[argsort_desc(x[:, 4])[:_MAX_NMS]]
        if nms_type == "nms":                                        # :382-388
            c = x[:, 5:6] * np.float32(0 if agnostic else _MAX_WH)
            i = tv_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
            out[xi] = x[i]
        elif nms_type == "batched_nms":                              # :391-397
            c = x[:, 5] * np.float32(0) if agnostic else x[:, 5]
            i = tv_batched_nms(x[:, :4].copy(), x[:, 4], c, iou_thres)[:max_det]
            out[xi] = x[i]
        elif nms_type == "fast_nms":                                 # :400-405 (unsorted)
            c = x[:, 5] * np.float32(0) if agnostic else x[:, 5]
            boxes = x[:, :4] + c.reshape(-1, 1) * np.float32(_MAX_WH)
            iou = np.triu(box_iou(boxes, boxes), k=1)
            keep = iou.max(0) < it32
            out[xi] = x[keep][:max_det]
        elif nms_type == "matrix_nms":                               # :408-417 (keeps all, decays conf)
            iou = np.triu(box_iou(x[:, :4], x[:, :4]), k=1)
            m = iou.max(0).reshape(-1, 1)
            decay = np.exp(-(iou ** 2 - m ** 2) / np.float32(0.5)).min(0)
            x[:, 4] *= decay
            out[xi] = x[:max_det]
        elif nms_type == "merge_nms":                                # :418-435
            c = x[:, 5:6] * np.float32(0 if agnostic else _MAX_WH)
            boxes, scores = x[:, :4] + c, x[:, 4]
            i = tv_nms(boxes, scores, iou_thres)[:max_det]
            if 1 < n < 3e3:
                iou = box_iou(boxes[i], boxes) > it32
                weights = iou * scores[None]
                x[i, :4] = (weights @ x[:, :4]).astype(np.float32) / weights.sum(1, keepdims=True)
                i = i[iou.sum(1) > 1]
            out[xi] = x[i]
        else:
            raise ValueError(f"unknown nms_type {nms_type!r}")
    return out


def batched_nms(prediction, conf_thres=0.001, iou_thres=0.65, nms_box=500, agnostic=False,
                nms_type="nms") -> List[np.ndarray]:
    """nms.py:15-116 (`agnostic=True` ADDS the class offset there -- SURVEY 0.7)."""
    pred = _f32(prediction)
    B = pred.shape[0]
    ct = np.float32(conf_thres)
    it32 = np.float32(iou_thres)
    rows = []
    for b in range(B):
        idx = argsort_desc(pred[b, :, 4])[:nms_box]                  # :41
        o = pred[b, idx]                                             # :42
        confs = o[:, 5:] * o[:, 4:5]                                 # :45
        j, k = np.nonzero(confs > ct)                                # :46
        x = np.concatenate((o[j, :4], confs[j, k, None], k[:, None].astype(np.float32)), 1)  # :47
        xywh = x[:, :4].copy()                                       # :50-54
        two = np.float32(2.0)
        x[:, 0] = xywh[:, 0] - xywh[:, 2] / two
        x[:, 1] = xywh[:, 1] - xywh[:, 3] / two
        x[:, 2] = xywh[:, 0] + xywh[:, 2] / two
        x[:, 3] = xywh[:, 1] + xywh[:, 3] / two
        rows.append(x)
    outputs: List[np.ndarray] = []
    for b in range(B):
        o = rows[b]
        bboxes = o[:, :4] + o[:, 5:6] * np.float32(4096) if agnostic else o[:, :4]   # :58-62
        if nms_type == "nms":
            keep = tv_nms(bboxes, o[:, 4], iou_thres)
        elif nms_type == "batched_nms":
            keep = tv_batched_nms(o[:, :4].copy(), o[:, 4], o[:, 5], iou_thres)
        elif nms_type == "fast_nms":
            bb = o[:, :4] + o[:, 5].reshape(-1, 1) * np.float32(4096)
            if bb.shape[0] == 0:
                outputs.append(o)
                continue
            iou = np.triu(box_iou(bb, bb), k=1)
            keep = iou.max(0) < it32
        elif nms_type == "matrix_nms":
            bb = o[:, :4] + o[:, 5].reshape(-1, 1) * np.float32(4096)
            if bb.shape[0] == 0:
                outputs.append(o)
                continue
            iou = np.triu(box_iou(bb, bb), k=1)
            m = iou.max(0).reshape(-1, 1)
            decay = np.exp(-(iou ** 2 - m ** 2) / np.float32(0.5)).min(0)
            o[:, 4] *= decay
            keep = np.ones(bb.shape[0], dtype=bool)
        elif nms_type == "merge_nms":
            keep = tv_nms(bboxes, o[:, 4], iou_thres)
            iou = box_iou(bboxes[keep], bboxes) > it32
            weights = iou * o[:, 4][None]
            o[keep, :4] = (weights @ o[:, :4]).astype(np.float32) / weights.sum(1, keepdims=True)
            keep = keep[iou.sum(1) > 1]
        else:
            raise ValueError(f"unknown nms_type {nms_type!r}")
        outputs.append(o[keep])
    return outputs


# ----------------------------------------------------------------------------------------------
# head decode (kindle YOLOHead eval path; layout proven by losses.py:245-256,350 and
# tta_utils.py:52-58; YOLOv5 v6 parametrisation mirrors losses.py:254-255)
# ----------------------------------------------------------------------------------------------
def head_decode(raw: Sequence[np.ndarray], anchors_px: np.ndarray, strides: Sequence[float]) -> np.ndarray:
    """raw[i]: (B, na, ny, nx, no) logits -> (B, sum(na*ny*nx), no) [cx,cy,w,h (pixels), obj, cls...]."""
    outs = []
    for i, r in enumerate(raw):
        r = np.asarray(r, np.float64)
        B, na, ny, nx, no = r.shape
        s = 1.0 / (1.0 + np.exp(-r))
        gy, gx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        grid = np.stack((gx, gy), -1)[None, None].astype(np.float64)
        y = s.copy()
        y[..., 0:2] = (s[..., 0:2] * 2.0 - 0.5 + grid) * float(strides[i])
        y[..., 2:4] = (s[..., 2:4] * 2.0) ** 2 * np.asarray(anchors_px[i], np.float64).reshape(1, na, 1, 1, 2)
        outs.append(y.reshape(B, -1, no))
    return np.concatenate(outs, 1).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# validator matching + AP (scripts/utils/train_utils.py:294-333, scripts/utils/metrics.py:446-548)
# --------------------------------------------------------------------------------------------------
def process_batch(detections: np.ndarray, labels: np.ndarray, iouv: np.ndarray) -> np.ndarray:
    """detections (N,6) x1,y1,x2,y2,conf,cls; labels (M,5) cls,x1,y1,x2,y2 -> correct (N, len(iouv)) bool.
    Follows train_utils.py:311-332 step by step (float32 arithmetic like the torch original)."""
    detections = np.asarray(detections, np.float32)
    labels = np.asarray(labels, np.float32)
    iouv = np.asarray(iouv, np.float32)
    correct = np.zeros((detections.shape[0], iouv.shape[0]), bool)
    if detections.shape[0] == 0 or labels.shape[0] == 0:
        return correct
    iou = box_iou(labels[:, 1:], detections[:, :4])
    x = np.where((iou >= iouv[0]) & (labels[:, 0:1] == detections[:, 5]))
    if x[0].shape[0]:
        matches = np.concatenate((np.stack(x, 1).astype(np.float32), iou[x[0], x[1]][:, None].astype(np.float32)), 1)
        if x[0].shape[0] > 1:
            matches = matches[matches[:, 2].argsort()[::-1]]
            matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
            matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
        correct[matches[:, 1].astype(np.int64)] = matches[:, 2:3] >= iouv
    return correct


def compute_ap(recall, precision):
    """metrics.py:446-473: 101-point interpolated AP of the precision envelope."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    integrate = getattr(np, "trapezoid", None) or np.trapz          # numpy 2 renamed trapz
    ap = integrate(np.interp(x, mrec, mpre), x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls):
    """metrics.py:476-548 without the plotting branch: (p, r, ap, f1, unique_classes)."""
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


def _trt_bbox_size(b: np.ndarray) -> np.float32:
    """allClassNMS.cu bboxSize with normalized = false: 0 for an inverted box, else (w + 1) * (h + 1)."""
    if b[2] < b[0] or b[3] < b[1]:
        return np.float32(0.0)
    return np.float32((np.float32(b[2] - b[0]) + np.float32(1.0)) * (np.float32(b[3] - b[1]) + np.float32(1.0)))


def _trt_jaccard(a: np.ndarray, b: np.ndarray) -> np.float32:
    """allClassNMS.cu jaccardOverlap with normalized = false (offset 1): disjoint boxes intersect in (0, 0, 0, 0),
    whose extents become 0 + 1."""
    if b[0] > a[2] or b[2] < a[0] or b[1] > a[3] or b[3] < a[1]:
        ix1 = iy1 = ix2 = iy2 = np.float32(0.0)
    else:
        ix1, iy1 = max(a[0], b[0]), max(a[1], b[1])
        ix2, iy2 = min(a[2], b[2]), min(a[3], b[3])
    w = np.float32(np.float32(ix2 - ix1) + np.float32(1.0))
    h = np.float32(np.float32(iy2 - iy1) + np.float32(1.0))
    if w > 0 and h > 0:
        inter = np.float32(w * h)
        return np.float32(inter / np.float32(np.float32(_trt_bbox_size(a) + _trt_bbox_size(b)) - inter))
    return np.float32(0.0)


def batched_nms_trt(boxes, scores, top_k: int = 512, keep_top_k: int = 100, score_threshold: float = 0.001,
                    iou_threshold: float = 0.65):
    """BatchedNMS_TRT (shareLocation = 1, backgroundLabelId = -1, isNormalized = 0, clipBoxes = 0), the layer
    scripts/model_converter/model_converter.py:268-388 appends to the engine.  boxes (B, N, 4) xyxy, scores (B, N, nc).
    Returns (num_detections (B, 1) int32, nmsed_boxes (B, keep, 4), nmsed_scores (B, keep), nmsed_classes (B, keep));
    padding rows are box 0, score 0, class -1 (gatherNMSOutputs)."""
    boxes, scores = _f32(boxes), _f32(scores)
    B, N, nc = scores.shape
    thr, it = np.float32(score_threshold), np.float32(iou_threshold)
    num = np.zeros((B, 1), np.int32)
    ob = np.zeros((B, keep_top_k, 4), np.float32)
    osc = np.zeros((B, keep_top_k), np.float32)
    ocl = np.full((B, keep_top_k), -1.0, np.float32)
    for b in range(B):
        kept = []                                        # (score, class, position in the class's sorted list, box index)
        for c in range(nc):
            sc = scores[b, :, c]
            idx = np.nonzero(sc > thr)[0]                # sortScoresPerClass: others get score 0 / index -1
            idx = idx[argsort_desc(sc[idx])][:top_k]     # stable descending, top K fed to the NMS step
            alive = np.ones(len(idx), bool)
            for i in range(len(idx)):
                if not alive[i]:
                    continue
                kept.append((sc[idx[i]], c, i, idx[i]))
                for j in range(i + 1, len(idx)):
                    if alive[j] and _trt_jaccard(boxes[b, idx[i]], boxes[b, idx[j]]) > it:
                        alive[j] = False
        # sortScoresPerImage: stable descending over the (class, position) array
        kept.sort(key=lambda t: (-float(t[0]), t[1], t[2]))
        kept = kept[:keep_top_k]
        num[b, 0] = len(kept)
        for j, (s_, c, _, i) in enumerate(kept):
            ob[b, j], osc[b, j], ocl[b, j] = boxes[b, i], s_, c
    return num, ob, osc, ocl


# YOLO class index -> COCO category id (scripts/utils/multi_queue.py:78-159 `label_fixer`)
COCO80_TO_91 = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34, 35, 36,
                37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 67,
                70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]


def coco_rows(names, outputs, img_size, shapes=None):
    """scripts/utils/multi_queue.py:204-305 `_add_outputs` + `add_predicted_box` for one batch: per image
    scale_coords(img_size, bbox, original shape) (ResultWriterTorch.scale_coords, :316-339), xyxy -> [x, y, w, h]
    (:262-266), the json objects with the category table.  outputs: per image (n, 6) float32 or None."""
    from pathlib import Path
    objs = []
    for i, name in enumerate(names):
        if outputs[i] is None or len(outputs[i]) == 0:
            continue
        o = _f32(outputs[i])
        bbox, conf = o[:, :4].copy(), o[:, 4:]
        if shapes is not None:
            bbox = scale_coords(img_size, bbox, shapes[i][0])
            bbox[:, 2] = bbox[:, 2] - bbox[:, 0]
            bbox[:, 3] = bbox[:, 3] - bbox[:, 1]
        for row, c in zip(bbox, conf):
            objs.append({"image_id": int(Path(name).stem), "category_id": COCO80_TO_91[int(c[1])],
                         "bbox": [float(p) for p in row], "score": float(c[0])})
    return objs
