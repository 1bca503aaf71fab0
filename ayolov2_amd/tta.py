"""Test-time augmentation with the reference's interface (scripts/utils/tta_utils.py:15-86; ``scale_img`` from
scripts/utils/torch_utils.py:305-331).

MI355X-first: the merged prediction is ONE tensor allocated up front.  For a ``YOLOModel`` on the GPU every augmented
forward runs on its cached inference plan and the head-decode kernel of each level stores its rows where the merged
prediction wants them, with the augmentation's inverse (xywh / scale, flip about the original image extent) applied in the
store and the clipped tails left out by a row window (``ayolo_head_decode_aug``) -- no per-augmentation copy, no in-place
fix-up pass, no concatenation.  Any other module (the CPU oracle, a stand-in model) goes through the same row bookkeeping
with tensor ops."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

UP_DOWN, LEFT_RIGHT = 2, 3           # flip codes of the reference = the image tensor's dimension that was flipped


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


def _level_rows(n_rows: int, nl: int) -> List[int]:
    """Rows of each detection level (P3 first) in a prediction of `n_rows` rows: level k has 4^(nl-1-k) shares of the
    (4^nl - 1) / 3 the pyramid splits into -- the reference's `g` / `4 ** x` arithmetic (tta_utils.py:52-58)."""
    share = n_rows // ((4 ** nl - 1) // 3)
    return [share * 4 ** (nl - 1 - k) for k in range(nl)]


def _kept(n_rows: int, nl: int, first: bool, last: bool) -> Tuple[int, int]:
    """[lo, hi) of the rows an augmentation contributes: the first one loses its coarsest level (its tail), the last one its
    finest level (its head).  A single augmentation is both: the reference cuts the tail first and then takes the head cut
    from the SHORTENED row count (tta_utils.py:54-58: `y[-1].shape[1] // g` after `y[0] = y[0][:, :-i]`), golden G9 `y_single`."""
    lv = _level_rows(n_rows, nl)
    hi = n_rows - lv[-1] if first else n_rows
    lo = _level_rows(hi, nl)[0] if last else 0
    return lo, hi


def _undo(dst: torch.Tensor, src: torch.Tensor, flip: Optional[int], scale: float, img_size: Sequence[int]) -> None:
    """dst = the augmentation's inverse of src (rows x [x, y, w, h, ...]); dst may be src."""
    torch.div(src[..., :4], scale, out=dst[..., :4])
    if dst.data_ptr() != src.data_ptr():
        dst[..., 4:] = src[..., 4:]
    if flip in (UP_DOWN, LEFT_RIGHT):
        axis, extent = (1, img_size[0]) if flip == UP_DOWN else (0, img_size[1])
        torch.sub(extent, dst[..., axis], out=dst[..., axis])


def descale_pred(p: torch.Tensor, flips: Optional[int], scale: float, img_size: Sequence[int]) -> torch.Tensor:
    """In-place inverse of one augmentation on decoded predictions (interface of tta_utils.py:15-37)."""
    _undo(p, p, flips, scale, img_size)
    return p


def clip_augmented(model: nn.Module, y: List[torch.Tensor]) -> List[torch.Tensor]:
    """The tail clip as views (interface of tta_utils.py:40-59): first entry without its coarsest level, last entry without
    its finest."""
    nl = model.model[-1].nl
    for k in sorted({0, len(y) - 1}):
        lo, hi = _kept(y[k].shape[1], nl, k == 0, k == len(y) - 1)
        y[k] = y[k][:, lo:hi]
    return y


def _device_plans(model: nn.Module, xs: Sequence[torch.Tensor]):
    """The inference plans of the augmented inputs if `model` runs on the HIP executor, else None."""
    if not (type(model).__name__ == "YOLOModel" and not model.training and getattr(model, "use_plan", True)
            and all(x.is_cuda and x.dim() == 4 for x in xs) and not torch.is_grad_enabled()
            and not getattr(model.model[-1], "out_xyxy", False)):
        return None
    from .infer_plan import eval_plan_for
    plans = [eval_plan_for(model, x) for x in xs]
    return None if any(p is None for p in plans) else plans


def inference_with_tta(model: nn.Module, x: torch.Tensor, s: Sequence[float], f: Sequence[Optional[int]]) -> Tuple[torch.Tensor, None]:
    """One eval forward per (scale, flip) pair; predictions mapped back to the original image and merged
    (tta_utils.py:62-86)."""
    img_size = tuple(x.shape[-2:])
    nl = model.model[-1].nl
    gs = int(model.stride.max())
    xs = [scale_img(x.flip(fi) if fi else x, si, gs=gs) for si, fi in zip(s, f)]
    n_aug = len(xs)
    plans = _device_plans(model, xs)
    if plans is not None:
        counts = [p.rows for p in plans]
        keep = [_kept(n, nl, k == 0, k == n_aug - 1) for k, n in enumerate(counts)]
        total = sum(hi - lo for lo, hi in keep)
        merged = torch.empty((x.shape[0], total, plans[0].out.shape[2]), dtype=torch.float32, device=x.device)
        base = 0
        for k, (plan, xi) in enumerate(zip(plans, xs)):
            lo, hi = keep[k]
            fi = f[k] or 0
            extent = img_size[0] if fi == UP_DOWN else img_size[1]
            plan.run_augmented(xi, merged, base - lo, (base, base + hi - lo), s[k], fi, extent)
            base += hi - lo
        return merged, None
    # (a model that returns views of plan-owned static storage -- YOLOModel.static_outputs -- would hand the SAME buffer back for
    # two augmentations of one input shape: keep a copy of each prediction then)
    static = bool(getattr(model, "static_outputs", False))
    preds = [model(xi)[0].clone() if static else model(xi)[0] for xi in xs]
    keep = [_kept(p.shape[1], nl, k == 0, k == n_aug - 1) for k, p in enumerate(preds)]
    total = sum(hi - lo for lo, hi in keep)
    merged = preds[0].new_empty((preds[0].shape[0], total, preds[0].shape[2]))
    base = 0
    for k, p in enumerate(preds):
        lo, hi = keep[k]
        _undo(merged[:, base:base + hi - lo], p[:, lo:hi], f[k], s[k], img_size)
        base += hi - lo
    return merged, None
