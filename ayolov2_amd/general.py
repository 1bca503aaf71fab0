"""Box-format helpers with the reference's signatures (scripts/utils/general.py:203-358).  Tiny elementwise
glue -- plain torch/numpy ops on whatever device the boxes live on."""
from __future__ import annotations

from typing import Optional, Tuple

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


def xyxy2xywh(x, wh: Tuple[float, float] = (1.0, 1.0), clip_eps: Optional[float] = None, check_validity: bool = True):
    """(n, 4) corner boxes -> centre / size boxes divided by the image size `wh` (interface of general.py:252-294, pinned by
    golden G3b).  Both axes are handled at once: centre = ((lo + hi) / 2) / wh, size = (hi - lo) / wh; with
    `check_validity` a box that overhangs [0, 1] is shrunk on that axis by twice the overhang (low side first, then the
    high side from the already shrunk size) and everything is clipped to [1e-12, 1].  `clip_eps` is accepted for interface
    parity: the reference clips a COPY and then derives every output column from the unclipped input, so it changes nothing
    (G3b `sized_clip` == `sized`)."""
    tensor = isinstance(x, torch.Tensor)
    size = x.new_tensor(wh[:2]) if tensor else np.asarray(wh[:2], dtype=x.dtype)
    lo, hi = x[:, 0:2], x[:, 2:4]
    centre = ((lo + hi) / 2) / size
    extent = (hi - lo) / size
    if check_validity:
        below = centre - extent / 2                       # < 0: the box starts before the image
        extent = extent + (below.clamp(max=0) if tensor else np.minimum(below, 0)) * 2
        above = centre + extent / 2                       # > 1: it ends beyond it
        extent = extent - ((above.clamp(min=1) if tensor else np.maximum(above, 1)) - 1) * 2
    out = torch.cat((centre, extent), 1) if tensor else np.concatenate((centre, extent), 1)
    return out.clip(1e-12, 1) if check_validity else out


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
