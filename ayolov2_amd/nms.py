"""``batched_nms`` with the reference's signature (scripts/utils/nms.py:15-116), on the HIP pipeline.

Differences from ``metrics.non_max_suppression`` that are reproduced, not fixed:
top-``nms_box`` proposals by OBJECTNESS first (no objectness threshold), always multi-label, no ``max_det`` cap,
and ``agnostic=True`` ADDS the ``cls * 4096`` offset.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from . import ops
from ._lib import call
from .metrics import (MAX_WH, _collect_candidates, _greedy_nms, _merge, _stream, _tv_batched_strategy)


def _topk_rows_by_objectness(pred: torch.Tensor, nms_box: int) -> torch.Tensor:
    """Per image, proposal indices of the nms_box largest objectness values (stable: ties keep index order)."""
    B, N, no = pred.shape
    dev = pred.device
    keys = torch.empty(B * N, dtype=torch.int64, device=dev)
    vals = torch.empty(B * N, dtype=torch.int32, device=dev)
    call("ayolo_nms_obj_keys", pred.data_ptr(), B, N, no, keys.data_ptr(), vals.data_ptr(), _stream())
    keys_out = torch.empty_like(keys)
    vals_out = torch.empty_like(vals)
    from . import _lib
    ws_bytes = _lib.c_size_t(0)
    call("ayolo_sort_pairs_u64", keys.data_ptr(), keys_out.data_ptr(), vals.data_ptr(), vals_out.data_ptr(), B * N, 0, 64,
         None, ws_bytes, _stream())
    ws = torch.empty(max(int(ws_bytes.value), 16), dtype=torch.uint8, device=dev)
    ws_bytes2 = _lib.c_size_t(ws.numel())
    call("ayolo_sort_pairs_u64", keys.data_ptr(), keys_out.data_ptr(), vals.data_ptr(), vals_out.data_ptr(), B * N, 0, 64,
         ws.data_ptr(), ws_bytes2, _stream())
    k = min(nms_box, N)
    return vals_out.view(B, N)[:, :k].contiguous()


def batched_nms(prediction: torch.Tensor, conf_thres: float = 0.001, iou_thres: float = 0.65, nms_box: int = 500,
                agnostic: bool = False, nms_type: str = "nms") -> List[torch.Tensor]:
    ops.require_cuda(prediction, "batched_nms")
    if nms_type not in ("nms", "batched_nms", "fast_nms", "matrix_nms", "merge_nms"):
        raise ValueError(f"Wrong NMS type {nms_type!r}")
    pred = prediction.detach().float().contiguous()
    B, N, no = pred.shape
    dev = pred.device
    rows = _topk_rows_by_objectness(pred, nms_box)
    by_seq = nms_type in ("fast_nms", "matrix_nms")
    cand = _collect_candidates(pred, conf_thres, True, False, None, rows, by_seq)
    seg_n = cand.counts.copy()
    empty = torch.zeros((0, 6), dtype=torch.float32, device=dev)
    outputs: List[torch.Tensor] = [empty] * B
    off_scale = float(MAX_WH) if agnostic else 0.0

    if nms_type in ("nms", "merge_nms", "batched_nms"):
        if nms_type == "batched_nms":
            scales, modes = _tv_batched_strategy(cand, seg_n, False)
        else:
            scales, modes = off_scale, np.zeros(B, dtype=np.int64)
        max_n = int(seg_n.max()) if B else 0
        out, out_idx, kept = _greedy_nms(cand, seg_n, iou_thres, scales, modes, max(max_n, 1))
        for b in range(B):
            k = int(kept[b])
            if k == 0:
                continue
            res = out[b, :k]
            if nms_type == "merge_nms":
                merged, red = _merge(cand, b, int(seg_n[b]), out_idx[b], k, off_scale, iou_thres)
                res = torch.cat((merged, res[:, 4:]), 1)[red]
            outputs[b] = res
        return outputs

    for b in range(B):
        n = int(seg_n[b])
        if n == 0:
            continue
        off = int(cand.offsets[b])
        x = cand.sdet[off:off + n]
        boxes = x[:, :4].contiguous()
        cls = x[:, 5].contiguous()
        colmax = torch.empty(n, dtype=torch.float32, device=dev)
        call("ayolo_iou_colmax", boxes.data_ptr(), cls.data_ptr(), float(MAX_WH), n, colmax.data_ptr(), _stream())
        if nms_type == "fast_nms":
            outputs[b] = x[colmax < float(np.float32(iou_thres))]
        else:
            decay = torch.empty(n, dtype=torch.float32, device=dev)
            call("ayolo_matrix_nms_decay", boxes.data_ptr(), cls.data_ptr(), float(MAX_WH), n, colmax.data_ptr(),
                 decay.data_ptr(), _stream())
            res = x.clone()
            res[:, 4] = res[:, 4] * decay
            outputs[b] = res
    return outputs
