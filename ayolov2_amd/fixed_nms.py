"""Fixed-shape batched NMS with the TensorRT ``BatchedNMS_TRT`` contract.

The reference appends that plugin to its TensorRT engines (scripts/model_converter/model_converter.py:268-388:
``shareLocation=1, backgroundLabelId=-1, numClasses, topK, keepTopK, scoreThreshold, iouThreshold, isNormalized=0,
clipBoxes=0``; inputs boxes ``(B, N, 1, 4)`` and scores ``(B, N, nc) = obj * cls``) and reads its four outputs
``num_detections / nmsed_boxes / nmsed_scores / nmsed_classes`` back through ``YoloValidator.convert_trt_out``
(scripts/utils/train_utils.py:262-283).  The plugin itself is a third-party binary (TensorRT OSS ``batchedNMSPlugin``);
its published algorithm is restated in ``oracle/ops_ref.py::batched_nms_trt`` -- parity with the binary is unpinned.

MI355X design: every buffer has a size fixed by ``(B, N, nc, topK, keepTopK, capacity)`` and the host never reads a
count back, so the whole sequence (11 launches + two radix sorts) is asynchronous on the caller's stream and can be
captured in a hipGraph -- unlike ``metrics.non_max_suppression``, whose ragged outputs need three host syncs.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib
from ._lib import call


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class BatchedNMS:
    """The plugin's fields (model_converter.py:316-376) and its call contract.

    ``capacity`` bounds the number of (box, class) pairs with ``score > score_threshold`` that can be held before the
    per-class top-K cut; the default is one pair per proposal (``B * N``), never less than ``B * nc * top_k``.  Pairs
    beyond the capacity are dropped in an unspecified order; ``self.overflow`` (a device tensor, read it lazily) tells.
    """

    def __init__(self, num_classes: int, top_k: int = 512, keep_top_k: int = 100, score_threshold: float = 0.001,
                 iou_threshold: float = 0.65, share_location: bool = True, background_label_id: int = -1,
                 is_normalized: bool = False, clip_boxes: bool = False, capacity: Optional[int] = None) -> None:
        if not share_location or background_label_id != -1 or is_normalized or clip_boxes:
            raise NotImplementedError("BatchedNMS: only the field values the reference sets are implemented "
                                      "(shareLocation=1, backgroundLabelId=-1, isNormalized=0, clipBoxes=0)")
        if keep_top_k > top_k:
            raise ValueError("BatchedNMS: keepTopK must not exceed topK")
        self.num_classes, self.top_k, self.keep_top_k = int(num_classes), int(top_k), int(keep_top_k)
        self.score_threshold, self.iou_threshold = float(score_threshold), float(iou_threshold)
        self.capacity = capacity
        self.overflow: Optional[torch.Tensor] = None
        self._buf = {}

    # ------------------------------------------------------------------------------------------------------------
    def _buffers(self, B: int, N: int, dev: torch.device):
        key = (B, N, dev)
        if key in self._buf:
            return self._buf[key]
        nc, top_k = self.num_classes, self.top_k
        nseg = B * nc
        cap = self.capacity if self.capacity is not None else max(B * N, nseg * top_k)
        cap = int(min(cap, B * N * nc))
        max_out = min(top_k, self.keep_top_k)
        E = nseg * max_out
        rb, cb, ib, fb = _lib.c_int(0), _lib.c_int(0), _lib.c_int(0), _lib.c_int(0)
        _lib.check(_lib.lib().ayolo_trt_nms_key_bits(B, N, nc, rb, cb, ib), "ayolo_trt_nms_key_bits (B*N*nc too large for a 64-bit key)")
        call("ayolo_trt_nms_final_keys", None, None, B, nc, max_out, None, None, fb, None)
        i64 = lambda *s: torch.empty(s, dtype=torch.int64, device=dev)      # noqa: E731
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)      # noqa: E731
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)    # noqa: E731
        b = dict(cap=cap, max_out=max_out, E=E, bits1=rb.value + 32 + cb.value + ib.value, bits2=fb.value,
                 det=f32(cap, 6), sdet=f32(cap, 6), keys=i64(2, cap), vals=i32(2, cap), counters=i32(1 + B),
                 lay=i32(2, nseg), mask_off=i64(nseg), mask=i64(nseg * top_k * ((top_k + 63) // 64)),
                 out=f32(nseg, max_out, 6), out_idx=i32(nseg, max_out), out_count=i32(nseg),
                 fkeys=i64(2, E), fvals=i32(2, E))
        call("ayolo_iota_u32", b["vals"][0].data_ptr(), cap, _stream())
        ws = _lib.c_size_t(0)
        call("ayolo_sort_pairs_u64", b["keys"][0].data_ptr(), b["keys"][1].data_ptr(), b["vals"][0].data_ptr(),
             b["vals"][1].data_ptr(), cap, 0, b["bits1"], None, ws, _stream())
        ws2 = _lib.c_size_t(0)
        call("ayolo_sort_pairs_u64", b["fkeys"][0].data_ptr(), b["fkeys"][1].data_ptr(), b["fvals"][0].data_ptr(),
             b["fvals"][1].data_ptr(), E, 0, b["bits2"], None, ws2, _stream())
        b["ws"] = torch.empty(max(int(ws.value), int(ws2.value), 16), dtype=torch.uint8, device=dev)
        self._buf[key] = b
        return b

    def from_prediction(self, pred: torch.Tensor, box_xyxy: bool = True
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """NMS of a decoded head output ``(B, N, 5 + nc)`` -- what the engine's slice / gather / multiply layers in front
        of the plugin compute (model_converter.py:271-306), fused into the candidate filter: boxes = ``pred[..., :4]``
        (xyxy when the head was exported with ``out_xyxy``, export.py:171; else xywh, converted like metrics.py:356),
        scores = ``pred[..., 4:5] * pred[..., 5:]``.  Returns ``(num_detections (B, 1) int32, nmsed_boxes (B, keep, 4),
        nmsed_scores (B, keep), nmsed_classes (B, keep))``."""
        if not pred.is_cuda:
            raise _lib.AyoloError("BatchedNMS runs on the HIP device only (there is no CPU fallback)")
        if pred.dim() != 3 or pred.shape[2] != 5 + self.num_classes:
            raise ValueError(f"BatchedNMS: expected (B, N, {5 + self.num_classes}), got {tuple(pred.shape)}")
        pred = pred.float().contiguous()
        B, N, no = pred.shape
        dev, nc, st = pred.device, self.num_classes, _stream()
        b = self._buffers(B, N, dev)
        cap, max_out, E, nseg = b["cap"], b["max_out"], b["E"], B * nc
        keys, vals, fkeys, fvals = b["keys"], b["vals"], b["fkeys"], b["fvals"]
        seg_off2, seg_n2 = b["lay"][0], b["lay"][1]
        call("ayolo_trt_nms_candidates", pred.data_ptr(), B, N, no, self.score_threshold, int(box_xyxy), b["det"].data_ptr(),
             keys[0].data_ptr(), b["counters"].data_ptr(), cap, st)
        wsb = _lib.c_size_t(b["ws"].numel())
        call("ayolo_sort_pairs_u64", keys[0].data_ptr(), keys[1].data_ptr(), vals[0].data_ptr(), vals[1].data_ptr(), cap, 0,
             b["bits1"], b["ws"].data_ptr(), wsb, st)
        call("ayolo_gather_rows", b["det"].data_ptr(), vals[1].data_ptr(), b["sdet"].data_ptr(), cap, 6, st)
        call("ayolo_trt_nms_layout", keys[1].data_ptr(), cap, B, N, nc, self.top_k, seg_off2.data_ptr(), seg_n2.data_ptr(),
             b["mask_off"].data_ptr(), st)
        call("ayolo_trt_nms_mask", b["sdet"].data_ptr(), seg_off2.data_ptr(), seg_n2.data_ptr(), b["mask_off"].data_ptr(), nseg,
             self.top_k, self.iou_threshold, b["mask"].data_ptr(), st)
        call("ayolo_nms_reduce", b["sdet"].data_ptr(), seg_off2.data_ptr(), seg_n2.data_ptr(), b["mask_off"].data_ptr(),
             b["mask"].data_ptr(), nseg, max_out, b["out"].data_ptr(), b["out_idx"].data_ptr(), b["out_count"].data_ptr(),
             self.top_k, st)
        call("ayolo_trt_nms_final_keys", b["out"].data_ptr(), b["out_count"].data_ptr(), B, nc, max_out, fkeys[0].data_ptr(),
             fvals[0].data_ptr(), None, st)
        wsb = _lib.c_size_t(b["ws"].numel())
        call("ayolo_sort_pairs_u64", fkeys[0].data_ptr(), fkeys[1].data_ptr(), fvals[0].data_ptr(), fvals[1].data_ptr(), E, 0,
             b["bits2"], b["ws"].data_ptr(), wsb, st)
        keep = self.keep_top_k
        num = torch.empty((B, 1), dtype=torch.int32, device=dev)
        boxes = torch.empty((B, keep, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((B, keep), dtype=torch.float32, device=dev)
        classes = torch.empty((B, keep), dtype=torch.float32, device=dev)
        call("ayolo_trt_nms_emit", fkeys[1].data_ptr(), fvals[1].data_ptr(), b["out"].data_ptr(), B, nc, max_out, keep,
             num.data_ptr(), boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(), st)
        self.overflow = b["counters"][0] > cap           # stays on the device: read it only when you can afford the sync
        return num, boxes, scores, classes

    def __call__(self, boxes: torch.Tensor, scores: torch.Tensor):
        """The plugin's own two inputs: boxes ``(B, N, 1, 4)`` xyxy (shareLocation) and scores ``(B, N, nc)``."""
        B, N = scores.shape[:2]
        ones = torch.ones((B, N, 1), dtype=torch.float32, device=scores.device)      # score * 1.0f is exact
        pred = torch.cat((boxes.reshape(B, N, 4).float(), ones, scores.float()), 2)
        return self.from_prediction(pred, box_xyxy=True)


class NMSEngine(torch.nn.Module):
    """A model with the plugin appended, i.e. what the reference's ``TrtWrapper`` hands its validator: forward returns
    ``(out (B, keepTopK, 6) [x1, y1, x2, y2, score, class], n_objs (B,))`` -- the second element is a TENSOR, which is how
    ``YoloValidator.validation_step`` recognises the TensorRT case (train_utils.py:456-457)."""

    def __init__(self, model: torch.nn.Module, nms: BatchedNMS, box_xyxy: bool = False) -> None:
        super().__init__()
        self.model, self.nms, self.box_xyxy = model, nms, box_xyxy
        self.nc = nms.num_classes

    @torch.no_grad()
    def forward(self, imgs: torch.Tensor):
        outs = self.model(imgs)
        pred = outs[0] if isinstance(outs, (tuple, list)) else outs
        num, boxes, scores, classes = self.nms.from_prediction(pred, box_xyxy=self.box_xyxy)
        return torch.cat((boxes, scores[..., None], classes[..., None]), 2), num.reshape(-1)


def convert_trt_out(num_detections: torch.Tensor, nmsed_boxes: torch.Tensor, nmsed_scores: torch.Tensor,
                    nmsed_classes: torch.Tensor) -> List[torch.Tensor]:
    """The four plugin outputs -> the validator's per-image ``(n, 6)`` ``[x1, y1, x2, y2, conf, cls]`` rows
    (train_utils.py:262-283 applied to ``cat(boxes, scores, classes)``); one host sync for the counts."""
    out = torch.cat((nmsed_boxes, nmsed_scores[..., None], nmsed_classes[..., None]), 2)
    n = num_detections.reshape(-1).tolist()
    return [out[i, :k] for i, k in enumerate(n)]
