"""Box-format helpers with the reference's signatures (scripts/utils/general.py:203-358).  Tiny elementwise
glue -- plain torch/numpy ops on whatever device the boxes live on."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch


def clip_coords(boxes, wh: Tuple[float, float], inplace: bool = True):
    """Clip xyxy boxes to (width, height)."""
    if isinstance(boxes, torch.Tensor):
        if not inplace:
            boxes = boxes.clone()
        boxes[:, 0].clamp_(0, wh[0])
        boxes[:, 1].clamp_(0, wh[1])
        boxes[:, 2].clamp_(0, wh[0])
        boxes[:, 3].clamp_(0, wh[1])
    else:
        if not inplace:
            boxes = np.copy(boxes)
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, wh[0])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, wh[1])
    return boxes


def xywh2xyxy(x, ratio=(1.0, 1.0), wh=(1.0, 1.0), pad=(0.0, 0.0)):
    """[cx, cy, w, h] -> ratio*wh*[x1, y1, x2, y2] + pad."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    hw, hh = x[:, 2] / 2, x[:, 3] / 2
    y[:, 0] = ratio[0] * wh[0] * (x[:, 0] - hw) + pad[0]
    y[:, 1] = ratio[1] * wh[1] * (x[:, 1] - hh) + pad[1]
    y[:, 2] = ratio[0] * wh[0] * (x[:, 0] + hw) + pad[0]
    y[:, 3] = ratio[1] * wh[1] * (x[:, 1] + hh) + pad[1]
    return y


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Rescale xyxy coords from img1_shape (h, w) to img0_shape (h, w), undoing the letterbox."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = ((img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2)
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape[::-1])
    return coords
