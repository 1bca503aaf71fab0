"""Detection task math with the reference's Python signatures (scripts/utils/metrics.py), running on the GPU
through libayolo_hip.so:

* ``box_iou``               metrics.py:138-164
* ``bbox_iou``              metrics.py:60-135  (differentiable, plain torch ops: it is < 1 % of a train step)
* ``non_max_suppression``   metrics.py:285-443 (all five ``nms_type`` branches)

The HIP pipeline (filter -> key sort -> wave64 IoU bit-matrix -> greedy scan) returns kept *indices* bit-identical
to the CPU path; the two host synchronisations are the candidate counts and the kept counts (the reference's
output -- a list of variable-length tensors -- needs the latter in any case).
The reference's 10 s wall-clock ``time_limit`` break (metrics.py:328,439-441) is never reached and not reproduced.
"""
from __future__ import annotations

import os

import math
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import call
from .general import xywh2xyxy  # noqa: F401  (re-exported like the reference module)

MAX_WH = 4096
MAX_NMS = 30000
NMS_BY_CLASS = os.environ.get("AYOLO_NMS_BY_CLASS", "1") != "0"      # per-(image, class) segments in the `nms` branch


def _stream():
    return torch.cuda.current_stream().cuda_stream


def thr_as_float_for_double_compare(thr: float) -> float:
    """Largest float32 t with t <= thr:  (float ovr > t)  ==  ((double)ovr > thr)  -- torchvision's CPU kernel
    compares the float IoU against the *double* threshold."""
    t = np.float32(thr)
    if float(t) > thr:
        t = np.nextafter(t, np.float32(-np.inf))
    return float(t)


def box_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """(N, 4) x (M, 4) xyxy -> (N, M) IoU, inter / (area1 + area2 - inter)."""
    ops.require_cuda(box1, "box_iou")
    a = box1.float().contiguous()
    b = box2.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    call("ayolo_box_iou", a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream())
    return out


def bbox_iou(box1, box2, x1y1x2y2=True, g_iou=False, d_iou=False, c_iou=False, eps=1e-7):
    """IoU / GIoU / DIoU / CIoU of box1 (4, n) against box2 (n, 4); differentiable."""
    box2 = box2.T
    if x1y1x2y2:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    else:
        b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
        b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
        b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
        b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    iw = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0)
    ih = (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    inter = iw * ih
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not (g_iou or d_iou or c_iou):
        return iou
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    if c_iou or d_iou:
        c2 = cw ** 2 + ch ** 2 + eps
        rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
        if d_iou:
            return iou - rho2 / c2
        v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
        with torch.no_grad():
            alpha = v / (v - iou + (1 + eps))
        return iou - (rho2 / c2 + v * alpha)
    c_area = cw * ch + eps
    return iou - (c_area - union) / c_area


# --------------------------------------------------------------------------------------------------
# shared GPU pipeline
# --------------------------------------------------------------------------------------------------
class _Candidates:
    """Result of stage A+B: per-image segments of candidate rows [x1,y1,x2,y2,conf,cls] in processing order."""

    def __init__(self, sdet: torch.Tensor, counts: np.ndarray, device):
        self.sdet = sdet
        self.counts = counts                                   # per image, before any truncation
        self.offsets = np.concatenate(([0], np.cumsum(counts)))[:-1].astype(np.int64)
        self.device = device


def _collect_candidates(pred: torch.Tensor, conf_thres: float, multi_label: bool, require_obj: bool,
                        classes: Optional[Sequence[int]], rows: Optional[torch.Tensor], order_by_seq: bool) -> _Candidates:
    B, N, no = pred.shape
    nc = no - 5
    dev = pred.device
    rows_per_img = N if rows is None else rows.shape[1]
    class_mask = None
    if classes is not None:
        words = np.zeros((nc + 63) // 64, dtype=np.uint64)
        for c in classes:
            c = int(c)
            if 0 <= c < nc:
                words[c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        class_mask = torch.from_numpy(words.view(np.int64)).to(dev)
    seq_bits, total_bits = _lib.c_int(0), _lib.c_int(0)
    _lib.check(_lib.lib().ayolo_nms_key_bits(B, rows_per_img, nc if multi_label else 1, int(order_by_seq), seq_bits, total_bits),
               "ayolo_nms_key_bits (B*N*nc too large for a 64-bit sort key)")
    worst = B * rows_per_img * (nc if multi_label else 1)
    capacity = min(worst, max(1 << 16, B * rows_per_img * 2))
    counters = torch.zeros(1 + B, dtype=torch.int32, device=dev)
    while True:
        det = torch.empty((capacity, 6), dtype=torch.float32, device=dev)
        keys = torch.empty(capacity, dtype=torch.int64, device=dev)
        counters.zero_()
        call("ayolo_nms_candidates", pred.data_ptr(), B, N, no, float(np.float32(conf_thres)), int(multi_label),
             int(require_obj), ops._ptr(class_mask), ops._ptr(rows), rows_per_img, det.data_ptr(), keys.data_ptr(),
             counters.data_ptr(), capacity, int(order_by_seq), _stream())
        host = counters.cpu().numpy().astype(np.int64)          # sync 1: candidate counts
        total = int(host[0])
        if total <= capacity:
            break
        capacity = total
    counts = host[1:1 + B]
    if total == 0:
        return _Candidates(torch.empty((0, 6), dtype=torch.float32, device=dev), counts, dev)
    keys_out = torch.empty(total, dtype=torch.int64, device=dev)
    vals_in = torch.empty(total, dtype=torch.int32, device=dev)
    vals_out = torch.empty(total, dtype=torch.int32, device=dev)
    call("ayolo_iota_u32", vals_in.data_ptr(), total, _stream())
    ws_bytes = _lib.c_size_t(0)
    call("ayolo_sort_pairs_u64", keys.data_ptr(), keys_out.data_ptr(), vals_in.data_ptr(), vals_out.data_ptr(), total, 0,
         int(total_bits.value), None, ws_bytes, _stream())
    ws = torch.empty(max(int(ws_bytes.value), 16), dtype=torch.uint8, device=dev)
    ws_bytes2 = _lib.c_size_t(ws.numel())
    call("ayolo_sort_pairs_u64", keys.data_ptr(), keys_out.data_ptr(), vals_in.data_ptr(), vals_out.data_ptr(), total, 0,
         int(total_bits.value), ws.data_ptr(), ws_bytes2, _stream())
    sdet = torch.empty((total, 6), dtype=torch.float32, device=dev)
    call("ayolo_gather_rows", det.data_ptr(), vals_out.data_ptr(), sdet.data_ptr(), total, 6, _stream())
    return _Candidates(sdet, counts, dev)


def _greedy_nms(cand: _Candidates, seg_n: np.ndarray, iou_thres: float, scales: Union[float, torch.Tensor],
                modes: np.ndarray, max_out: int) -> Tuple[torch.Tensor, torch.Tensor, np.ndarray]:
    """Stage C+D on segments [offsets[b], offsets[b]+seg_n[b]).  Returns (out (B,max_out,6), out_idx, kept counts)."""
    dev = cand.device
    B = len(seg_n)
    max_n = int(seg_n.max()) if B else 0
    max_out = max(1, min(max_out, max(max_n, 1)))
    out = torch.empty((B, max_out, 6), dtype=torch.float32, device=dev)
    out_idx = torch.empty((B, max_out), dtype=torch.int32, device=dev)
    out_count = torch.zeros(B, dtype=torch.int32, device=dev)
    if max_n == 0:
        return out, out_idx, np.zeros(B, dtype=np.int64)
    words = (seg_n + 63) // 64
    mask_sizes = seg_n * words
    mask_off = np.concatenate(([0], np.cumsum(mask_sizes)))[:-1].astype(np.int64)
    mask = torch.empty(max(int(mask_sizes.sum()), 1), dtype=torch.int64, device=dev)
    seg_off_d = torch.from_numpy(cand.offsets.astype(np.int32)).to(dev)
    seg_n_d = torch.from_numpy(seg_n.astype(np.int32)).to(dev)
    mask_off_d = torch.from_numpy(mask_off).to(dev)
    if isinstance(scales, torch.Tensor):
        scales_d = scales.float().contiguous()
    else:
        scales_d = torch.full((B,), float(scales), dtype=torch.float32, device=dev)
    class_aware = int(modes.any())
    if class_aware:    # class-aware (per-class) images must not see a coordinate offset
        scales_d = torch.where(torch.from_numpy(modes.astype(bool)).to(dev), torch.zeros_like(scales_d), scales_d)
    mixed = bool(class_aware and not modes.all())
    thr_f = thr_as_float_for_double_compare(iou_thres)
    if not mixed:
        call("ayolo_nms_mask", cand.sdet.data_ptr(), seg_off_d.data_ptr(), seg_n_d.data_ptr(), mask_off_d.data_ptr(), B,
             max_n, thr_f, 0.0, scales_d.data_ptr(), class_aware, mask.data_ptr(), _stream())
    else:
        # two launches over disjoint image subsets (segments of the other subset get n = 0)
        for flag in (0, 1):
            sel = torch.from_numpy((modes == flag)).to(dev)
            n_sub = torch.where(sel, seg_n_d, torch.zeros_like(seg_n_d))
            call("ayolo_nms_mask", cand.sdet.data_ptr(), seg_off_d.data_ptr(), n_sub.data_ptr(), mask_off_d.data_ptr(), B,
                 max_n, thr_f, 0.0, scales_d.data_ptr(), flag, mask.data_ptr(), _stream())
    call("ayolo_nms_reduce", cand.sdet.data_ptr(), seg_off_d.data_ptr(), seg_n_d.data_ptr(), mask_off_d.data_ptr(),
         mask.data_ptr(), B, max_out, out.data_ptr(), out_idx.data_ptr(), out_count.data_ptr(), max_n, _stream())
    kept = out_count.cpu().numpy().astype(np.int64)             # sync 2: kept counts
    return out, out_idx, kept


def _greedy_nms_by_class(cand: _Candidates, seg_n: np.ndarray, iou_thres: float, nc: int, max_det: int,
                         output: List[torch.Tensor]) -> bool:
    """Class-aware greedy NMS of the `nms` branch (metrics.py:383-388: boxes offset by cls * 4096) evaluated per (image,
    class) segment instead of over all candidate pairs of an image.  The offset makes boxes of different classes
    disjoint whenever the candidates' coordinates span less than 4096, so the global greedy scan in confidence order
    equals independent per-class scans; with ~80 classes that is ~80x fewer box pairs.  Same kernels, same arithmetic on
    the offset boxes (both boxes of a pair carry the same offset); the kept boxes of an image are merged back in
    confidence order and cut to `max_det`.  Returns False (nothing done) when the span condition does not hold."""
    dev = cand.device
    B = len(seg_n)
    tot = int(seg_n.sum())
    if tot == 0:
        return True
    nseg = B * nc
    st = _stream()
    sel_off = np.concatenate(([0], np.cumsum(seg_n))).astype(np.int64)         # (B+1,) compact start of every image
    meta = torch.from_numpy(np.concatenate((cand.offsets[:B], sel_off)).astype(np.int32)).to(dev)
    seg_off_d, sel_off_d = meta[:B], meta[B:]
    # zeroed scratch, one fill: [summary (2 x int64) | span (2) | out_count (nseg) | kept (B) | flags (tot)]
    zi = torch.zeros(6 + nseg + B + tot, dtype=torch.int32, device=dev)
    summary, span_d = zi[0:4], zi[4:6]
    out_count, kept_d, flags = zi[6:6 + nseg], zi[6 + nseg:6 + nseg + B], zi[6 + nseg + B:]
    rows1 = torch.empty((tot, 6), dtype=torch.float32, device=dev)             # per image, in confidence order
    keys = torch.empty((2, tot), dtype=torch.int64, device=dev)
    vals = torch.empty((3, tot), dtype=torch.int32, device=dev)                # [iota | perm | scan scratch]
    call("ayolo_nms_class_keys", cand.sdet.data_ptr(), seg_off_d.data_ptr(), sel_off_d.data_ptr(), B, nc, tot,
         rows1.data_ptr(), keys[0].data_ptr(), vals[0].data_ptr(), span_d.data_ptr(), st)
    # stable radix sort on (image, class): groups in key order, confidence order inside
    key_bits = max(1, int(nseg - 1).bit_length())
    ws_bytes = _lib.c_size_t(0)
    call("ayolo_sort_pairs_u64", keys[0].data_ptr(), keys[1].data_ptr(), vals[0].data_ptr(), vals[1].data_ptr(), tot, 0,
         key_bits, None, ws_bytes, st)
    ws = torch.empty(max(int(ws_bytes.value), 16), dtype=torch.uint8, device=dev)
    ws_bytes = _lib.c_size_t(ws.numel())
    call("ayolo_sort_pairs_u64", keys[0].data_ptr(), keys[1].data_ptr(), vals[0].data_ptr(), vals[1].data_ptr(), tot, 0,
         key_bits, ws.data_ptr(), ws_bytes, st)
    perm = vals[1]
    lay32 = torch.empty(2 * nseg, dtype=torch.int32, device=dev)
    mask_off_d = torch.empty(nseg, dtype=torch.int64, device=dev)
    seg_off2, seg_n2 = lay32[:nseg], lay32[nseg:]
    call("ayolo_nms_class_layout", keys[1].data_ptr(), tot, nseg, seg_off2.data_ptr(), seg_n2.data_ptr(),
         mask_off_d.data_ptr(), summary.data_ptr(), st)
    host = zi[:6].cpu().numpy()                                               # one sync: layout summary + coordinate span
    max_n, mask_words = (int(v) for v in host[:4].view(np.int64))
    codes = host[4:6].view(np.uint32)
    bits = np.where(codes & np.uint32(0x80000000), codes & np.uint32(0x7FFFFFFF), ~codes).astype(np.uint32)
    hi, lo = bits.view(np.float32)                                            # max(coord), max(-coord) = -min(coord)
    if not (hi + lo < np.float32(MAX_WH)):
        return False
    rows2 = torch.empty((tot, 6), dtype=torch.float32, device=dev)
    call("ayolo_gather_rows", rows1.data_ptr(), perm.data_ptr(), rows2.data_ptr(), tot, 6, st)
    max_out = max(1, min(max_det, max_n))
    mask = torch.empty(max(mask_words, 1), dtype=torch.int64, device=dev)
    scales_d = torch.full((nseg,), float(MAX_WH), dtype=torch.float32, device=dev)
    out = torch.empty((nseg, max_out, 6), dtype=torch.float32, device=dev)
    out_idx = torch.empty((nseg, max_out), dtype=torch.int32, device=dev)
    thr_f = thr_as_float_for_double_compare(iou_thres)
    call("ayolo_nms_mask", rows2.data_ptr(), seg_off2.data_ptr(), seg_n2.data_ptr(), mask_off_d.data_ptr(), nseg, max_n, thr_f,
         0.0, scales_d.data_ptr(), 0, mask.data_ptr(), st)
    call("ayolo_nms_reduce", rows2.data_ptr(), seg_off2.data_ptr(), seg_n2.data_ptr(), mask_off_d.data_ptr(), mask.data_ptr(),
         nseg, max_out, out.data_ptr(), out_idx.data_ptr(), out_count.data_ptr(), max_n, st)
    # merge: kept rows back into their image's confidence order, first max_det per image
    res = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
    margs = (rows1.data_ptr(), out_idx.data_ptr(), out_count.data_ptr(), seg_off2.data_ptr(), perm.data_ptr(), nseg, max_out,
             sel_off_d.data_ptr(), B, tot, max_det, flags.data_ptr(), vals[2].data_ptr(), res.data_ptr(), kept_d.data_ptr())
    ws_bytes = _lib.c_size_t(0)
    call("ayolo_nms_class_merge", *margs, None, ws_bytes, st)
    if int(ws_bytes.value) > ws.numel():
        ws = torch.empty(int(ws_bytes.value), dtype=torch.uint8, device=dev)
    ws_bytes = _lib.c_size_t(ws.numel())
    call("ayolo_nms_class_merge", *margs, ws.data_ptr(), ws_bytes, st)
    kept = kept_d.cpu().numpy()                                               # sync: final counts
    for b in range(B):
        k = int(kept[b])
        if k:
            output[b] = res[b, :k]
    return True


import threading

_FAST_TLS = threading.local()   # per host thread: {(device, B, N, no, multi_label) -> [capacity, workspace, pinned status copy]}
                                # (the reference's builder validates with one thread per device: the pinned status buffer of a
                                # key must not be shared between two threads' copy -> synchronize -> read sequences)
NMS_FAST = os.environ.get("AYOLO_NMS_FAST", "1") != "0"           # the one-call LDS-resident route of the `nms` branch


def _nms_class_fast(pred: torch.Tensor, *args) -> bool:
    with torch.cuda.device(pred.device):                   # kernels, stream and allocations on the device the prediction lives on
        return _nms_class_fast_on_device(pred, *args)


def _nms_class_fast_on_device(pred: torch.Tensor, conf_thres: float, iou_thres: float, multi_label: bool,
                              classes: Optional[Sequence[int]], max_det: int, output: List[torch.Tensor]) -> bool:
    """The class-aware `nms` branch in ONE library call (ayolo_nms_class_fast: no library sorts, no intermediate host
    read).  The work buffers are sized from the previous call of the same shape; the single host read at the end returns
    the per-image counts together with the status flags.  Returns False when a limit of that path did not hold (the
    general path then recomputes everything); a candidate-buffer overflow is retried once with the reported size."""
    B, N, no = pred.shape
    nc = no - 5
    dev = pred.device
    class_mask = None
    if classes is not None:
        words = np.zeros((nc + 63) // 64, dtype=np.uint64)
        for c in classes:
            c = int(c)
            if 0 <= c < nc:
                words[c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        class_mask = torch.from_numpy(words.view(np.int64)).to(dev)
    key = (dev, B, N, no, multi_label)
    states = _FAST_TLS.__dict__.setdefault("states", {})
    st = states.get(key)
    worst = B * N * (nc if multi_label else 1)
    if st is None:
        st = states[key] = [min(worst, max(1 << 16, B * N * 2)), None, torch.empty(2 + 2 * B, dtype=torch.int32).pin_memory()]
        while len(states) > 8:
            states.pop(next(iter(states)))
    thr_f = thr_as_float_for_double_compare(iou_thres)
    stream = torch.cuda.current_stream(dev)                                    # the stream of pred's device, not the current device's
    for _ in range(2):
        capacity = int(st[0])
        need = _lib.c_size_t(0)
        args = (pred.data_ptr(), B, N, no, float(np.float32(conf_thres)), int(multi_label), ops._ptr(class_mask), thr_f, int(max_det),
                MAX_NMS, capacity)
        call("ayolo_nms_class_fast", *args, None, need, None, None, stream.cuda_stream)
        if st[1] is None or st[1].numel() < int(need.value):
            st[1] = torch.empty(int(need.value), dtype=torch.uint8, device=dev)
        need = _lib.c_size_t(st[1].numel())
        out = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
        status = torch.empty(2 + 2 * B, dtype=torch.int32, device=dev)
        call("ayolo_nms_class_fast", *args, st[1].data_ptr(), need, out.data_ptr(), status.data_ptr(), stream.cuda_stream)
        st[2].copy_(status, non_blocking=True)
        stream.synchronize()                                                   # the one host read of this path
        host = st[2].numpy()
        flags, total = int(host[0]), int(np.uint32(host[1]))
        if flags & 1:                                                          # candidate buffer too small: resize, once
            st[0] = min(worst, total + total // 4 + 1024)
            continue
        if flags:
            return False
        st[0] = max(min(worst, total + total // 2 + 1024), 1 << 14)            # next call: 1.5x what this one needed
        for b in range(B):
            k = int(host[2 + b])
            if k:
                output[b] = out[b, :k]
        return True
    return False


def _tv_batched_strategy(cand: _Candidates, seg_n: np.ndarray, class_agnostic: bool):
    """torchvision 0.10.1 ops.boxes.batched_nms: per-class NMS when boxes.numel() > 4000, else the coordinate
    trick offset = idx * (boxes.max() + 1).  Returns (scales tensor, modes array)."""
    dev = cand.device
    B = len(seg_n)
    if class_agnostic:     # idxs are all zero: one class, no offset in either strategy
        return torch.zeros(B, dtype=torch.float32, device=dev), np.zeros(B, dtype=np.int64)
    modes = (seg_n * 4 > 4000).astype(np.int64)
    maxc = torch.zeros(B, dtype=torch.float32, device=dev)
    seg_off_d = torch.from_numpy(cand.offsets.astype(np.int32)).to(dev)
    seg_n_d = torch.from_numpy(seg_n.astype(np.int32)).to(dev)
    call("ayolo_seg_max_coord", cand.sdet.data_ptr(), seg_off_d.data_ptr(), seg_n_d.data_ptr(), B, maxc.data_ptr(), _stream())
    return maxc, modes


def _merge(cand: _Candidates, b: int, n: int, kept_idx: torch.Tensor, nk: int, scale: float, iou_thres: float):
    """merge_nms tail (metrics.py:424-433): weighted-mean boxes + redundancy filter for image b."""
    dev = cand.device
    off = int(cand.offsets[b])
    seg = cand.sdet[off:off + n]
    merged = torch.empty((nk, 4), dtype=torch.float32, device=dev)
    red = torch.empty(nk, dtype=torch.int32, device=dev)
    kidx = kept_idx[:nk].contiguous()
    call("ayolo_merge_boxes", seg.data_ptr(), n, float(scale), kidx.data_ptr(), nk, float(np.float32(iou_thres)),
         merged.data_ptr(), red.data_ptr(), _stream())
    return merged, red.bool()


def non_max_suppression(prediction: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        classes: Optional[list] = None, agnostic: bool = False, multi_label: bool = False,
                        labels: Union[tuple, list] = (), max_det: int = 300, nms_type: str = "nms") -> List[torch.Tensor]:
    """Same contract as the reference: list (per image) of (n, 6) tensors [x1, y1, x2, y2, conf, cls]."""
    ops.require_cuda(prediction, "non_max_suppression")
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if nms_type not in ("nms", "batched_nms", "fast_nms", "matrix_nms", "merge_nms"):
        raise ValueError(f"Wrong NMS type {nms_type!r}")
    pred = prediction.detach().float().contiguous()
    B, N, no = pred.shape
    nc = no - 5
    dev = pred.device
    multi_label = bool(multi_label) and nc > 1
    if labels and any(len(l) for l in labels):               # a-priori labels: appended rows (obj = cls = 1)
        lmax = max(len(l) for l in labels)
        extra = torch.zeros((B, lmax, no), dtype=torch.float32, device=dev)
        for i, l in enumerate(labels):
            if len(l):
                l = torch.as_tensor(l, dtype=torch.float32, device=dev)
                extra[i, :len(l), :4] = l[:, 1:5]
                extra[i, :len(l), 4] = 1.0
                extra[i, torch.arange(len(l), device=dev), l[:, 0].long() + 5] = 1.0
        pred = torch.cat((pred, extra), 1).contiguous()
        N = pred.shape[1]
    if (nms_type == "nms" and not agnostic and nc > 1 and NMS_BY_CLASS and NMS_FAST and B <= 1024 and B * nc <= 12288
            and max_det <= 1024 and nc * max_det <= 32768):
        fast_out: List[torch.Tensor] = [torch.zeros((0, 6), dtype=torch.float32, device=dev)] * B
        if _nms_class_fast(pred, conf_thres, iou_thres, multi_label, classes, max_det, fast_out):
            return fast_out
    by_seq = nms_type in ("fast_nms", "matrix_nms")           # those branches run on UNSORTED candidates
    cand = _collect_candidates(pred, conf_thres, multi_label, True, classes, None, by_seq)
    seg_n = np.minimum(cand.counts, MAX_NMS)
    # metrics.py:378-379: an image with more than max_nms candidates is sorted by confidence and cut to max_nms BEFORE its
    # branch runs, so fast_nms / matrix_nms see THAT image in confidence order (stable here; torch's argsort leaves the
    # order of equal confidences open) and every other image in its original order
    cand_sorted = None
    if by_seq and (cand.counts > MAX_NMS).any():
        cand_sorted = _collect_candidates(pred, conf_thres, multi_label, True, classes, None, False)
    empty = torch.zeros((0, 6), dtype=torch.float32, device=dev)
    output: List[torch.Tensor] = [empty] * B

    if nms_type in ("nms", "merge_nms", "batched_nms"):
        if nms_type == "batched_nms":
            scales, modes = _tv_batched_strategy(cand, seg_n, agnostic)
        else:
            scales, modes = (0.0 if agnostic else float(MAX_WH)), np.zeros(B, dtype=np.int64)
        if nms_type == "nms" and not agnostic and nc > 1 and NMS_BY_CLASS and int(seg_n.max()) > 512:   # (general path)
            if _greedy_nms_by_class(cand, seg_n, iou_thres, nc, max_det, output):
                return output
        out, out_idx, kept = _greedy_nms(cand, seg_n, iou_thres, scales, modes, max_det)
        for b in range(B):
            k = int(kept[b])
            if k == 0:
                continue
            res = out[b, :k]
            n = int(seg_n[b])
            if nms_type == "merge_nms" and 1 < n < 3e3:
                merged, red = _merge(cand, b, n, out_idx[b], k, 0.0 if agnostic else float(MAX_WH), iou_thres)
                res = torch.cat((merged, res[:, 4:]), 1)[red]
            output[b] = res
        return output

    # fast_nms / matrix_nms: column reductions of the upper-triangular IoU matrix, candidates in original order
    for b in range(B):
        n = int(seg_n[b])
        if n == 0:
            continue
        src = cand_sorted if (cand_sorted is not None and cand.counts[b] > MAX_NMS) else cand
        off = int(src.offsets[b])
        x = src.sdet[off:off + n]
        boxes = x[:, :4].contiguous()
        cls = x[:, 5].contiguous()
        colmax = torch.empty(n, dtype=torch.float32, device=dev)
        if nms_type == "fast_nms":
            scale = 0.0 if agnostic else float(MAX_WH)
            call("ayolo_iou_colmax", boxes.data_ptr(), cls.data_ptr(), scale, n, colmax.data_ptr(), _stream())
            keep = colmax < float(np.float32(iou_thres))
            output[b] = x[keep][:max_det]
        else:
            call("ayolo_iou_colmax", boxes.data_ptr(), None, 0.0, n, colmax.data_ptr(), _stream())
            decay = torch.empty(n, dtype=torch.float32, device=dev)
            call("ayolo_matrix_nms_decay", boxes.data_ptr(), None, 0.0, n, colmax.data_ptr(), decay.data_ptr(), _stream())
            res = x[:max_det].clone()
            res[:, 4] = res[:, 4] * decay[:max_det]
            output[b] = res
    return output
