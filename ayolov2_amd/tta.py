"""Test-time augmentation with the reference's interface (scripts/utils/tta_utils.py:15-86, ``scale_img`` from
scripts/utils/torch_utils.py:305-331).  Pure tensor glue around the model's eval forward (resize / flip / pad before,
de-scale / de-flip / tail clipping after); the forwards themselves run on the HIP path."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn


def scale_img(img: torch.Tensor, ratio: float = 1.0, same_shape: bool = False, gs: int = 32) -> torch.Tensor:
    """(bs,3,y,x) scaled by `ratio`, padded with the ImageNet mean to a multiple of `gs` (torch_utils.py:305-331)."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = (math.ceil(x * ratio / gs) * gs for x in (h, w))
    return F.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def descale_pred(p: torch.Tensor, flips: Optional[int], scale: float, img_size: Sequence[int]) -> torch.Tensor:
    """Inverse of the augmentation on decoded predictions, in place (tta_utils.py:15-37): 2 = up-down, 3 = left-right."""
    p[..., :4] /= scale
    if flips == 2:
        p[..., 1] = img_size[0] - p[..., 1]
    elif flips == 3:
        p[..., 0] = img_size[1] - p[..., 0]
    return p


def clip_augmented(model: nn.Module, y: List[torch.Tensor]) -> List[torch.Tensor]:
    """Drop the largest-stride rows of the first and the smallest-stride rows of the last augmentation
    (tta_utils.py:40-59)."""
    nl = model.model[-1].nl
    g = sum(4 ** x for x in range(nl))
    e = 1
    i = (y[0].shape[1] // g) * sum(4 ** x for x in range(e))
    y[0] = y[0][:, :-i]
    i = (y[-1].shape[1] // g) * sum(4 ** (nl - 1 - x) for x in range(e))
    y[-1] = y[-1][:, i:]
    return y


def inference_with_tta(model: nn.Module, x: torch.Tensor, s: Sequence[float], f: Sequence[Optional[int]]) -> Tuple[torch.Tensor, None]:
    """One eval forward per (scale, flip) pair, predictions mapped back and concatenated (tta_utils.py:62-86)."""
    img_size = x.shape[-2:]
    y = []
    for si, fi in zip(s, f):
        xi = scale_img(x.flip(fi) if fi else x, si, gs=int(model.stride.max()))
        yi = model(xi)[0].clone()      # the inference executor returns a view of static storage; descale_pred is in place
        yi = descale_pred(yi, fi, si, img_size)
        y.append(yi)
    y = clip_augmented(model, y)
    return torch.cat(y, 1), None
