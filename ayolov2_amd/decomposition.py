"""Tucker-2 decomposition of conv weights with the reference's interface
(scripts/tensor_decomposition/decomposition.py): ``EVBMF``, ``estimate_ranks``, ``tucker_decomposition_conv_layer``,
``decompose_layer_evaluation``, ``decompose_model``.

The decomposition itself is an offline, SVD-bound host step (the reference runs it on the CPU,
decompose_model.py:131); what lands on the MI355X hot path is its RESULT: each eligible k x k conv becomes
``Sequential(1x1 Cin->r_in, k x k r_in->r_out, 1x1 r_out->Cout)``, which ``modules.Conv`` executes as three
launches of the MFMA conv kernel (BN + SiLU folded into the last one's epilogue).

* rank estimation: analytic EVBMF of the mode-0 / mode-1 unfoldings (decomposition.py:25-206, 342-360);
* ``partial_tucker``: tensorly-0.6.0 semantics (HOSVD init, HOOI sweeps until the reconstruction error moves
  by < 1e-4, at most 100 sweeps) written with torch.linalg so it runs wherever the weight lives;
* driver: per-layer probe ``torch.rand(1024, Cin, kh, kw)``, mean-abs-diff loss, binary search on the L1-unstructured
  prune ratio while the loss stays under ``loss_thr`` (decomposition.py:237-339); 1x1 convs are skipped.
There is no CP decomposition in the reference; only Tucker-2 on modes (0, 1).
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.utils.prune as prune
from scipy.optimize import minimize_scalar
from torch import nn


# --------------------------------------------------------------------------------------------------
# EVBMF
# --------------------------------------------------------------------------------------------------
def tau(x: np.ndarray, alpha: float) -> np.ndarray:
    return 0.5 * (x - (1 + alpha) + np.sqrt((x - (1 + alpha)) ** 2 - 4 * alpha))


def EVBsigma2(sigma2: float, L: int, M: int, s: np.ndarray, residual: float, xubar: float) -> float:
    """Free-energy objective whose minimiser is the noise variance estimate."""
    H = len(s)
    alpha = L / M
    x = s ** 2 / (M * sigma2)
    above, below = x[x > xubar], x[x <= xubar]
    t = tau(above, alpha)
    return float(np.sum(below - np.log(below)) + np.sum(above - t) + np.sum(np.log((t + 1) / above))
                 + alpha * np.sum(np.log(t / alpha + 1)) + residual / (M * sigma2) + (L - H) * np.log(sigma2))


def EVBMF(Y, sigma2: Optional[float] = None, H: Optional[int] = None):
    """Analytic empirical variational Bayes matrix factorisation (Nakajima et al. 2013) of Y (L x M, L <= M): the rank is
    the number of singular values above sqrt(M * sigma2 * (1 + tau_bar) * (1 + alpha / tau_bar)), tau_bar = 2.5129 sqrt(alpha),
    with the noise variance sigma2 found by a bounded scalar minimisation of the free energy when it is not given.
    Returns (U, diag(d), V, info) like the reference (decomposition.py:25-206); its callers only use diag's shape
    (estimate_ranks, decomposition.py:357-359), so the posterior moments the reference also fills in are not computed."""
    Y = Y.detach().cpu().numpy() if isinstance(Y, torch.Tensor) else np.asarray(Y)
    L, M = Y.shape
    H = L if H is None else H
    alpha = L / M
    tau_bar = 2.5129 * np.sqrt(alpha)
    U, s, Vt = np.linalg.svd(Y, full_matrices=False)
    U, s, V = U[:, :H], s[:H], Vt[:H].T
    residual = float(np.sum(Y ** 2) - np.sum(s ** 2)) if H < L else 0.0
    if sigma2 is None:
        x_bar = (1 + tau_bar) * (1 + alpha / tau_bar)
        h_ub = int(min(np.ceil(L / (1 + alpha)) - 1, H))
        hi = (np.sum(s ** 2) + residual) / (L * M)
        lo = max(s[h_ub] ** 2 / (M * x_bar), np.mean(s[h_ub:] ** 2) / M)
        sigma2 = minimize_scalar(EVBsigma2, args=(L, M, s, residual, x_bar), bounds=[lo, hi], method="Bounded").x
    rank = int(np.sum(s > np.sqrt(M * sigma2 * (1 + tau_bar) * (1 + alpha / tau_bar))))
    sp = s[:rank]
    shrink = 1 - (L + M) * sigma2 / sp ** 2
    d = sp / 2 * (shrink + np.sqrt(shrink ** 2 - 4 * L * M * sigma2 ** 2 / sp ** 4))        # EVB-shrunk singular values
    return U[:, :rank], np.diag(d), V[:, :rank], {"sigma2": float(sigma2)}


# --------------------------------------------------------------------------------------------------
# Tucker-2
# --------------------------------------------------------------------------------------------------
def unfold(t: torch.Tensor, mode: int) -> torch.Tensor:
    return torch.movedim(t, mode, 0).reshape(t.shape[mode], -1)


def _mode_dot(t: torch.Tensor, m: torch.Tensor, mode: int) -> torch.Tensor:
    """t x_mode m, m: (J, t.shape[mode])."""
    return torch.movedim(torch.tensordot(m, t, dims=([1], [mode])), 0, mode)


def partial_tucker(tensor: torch.Tensor, modes: Sequence[int], rank: Sequence[int], n_iter_max: int = 100,
                   tol: float = 1e-4, init: str = "svd") -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """core, factors with tensor ~= core x_{modes} factors (orthonormal columns).  Computed in float64."""
    if init != "svd":
        raise NotImplementedError("only init='svd' (the reference's call, decomposition.py:378-380)")
    out_dtype = tensor.dtype
    t = tensor.detach().to(torch.float64)
    modes, rank = list(modes), [int(r) for r in rank]
    factors = [torch.linalg.svd(unfold(t, m), full_matrices=False)[0][:, :r] for m, r in zip(modes, rank)]
    norm_t = torch.linalg.norm(t)
    errs: List[float] = []
    core = t
    for it in range(n_iter_max):
        for i, m in enumerate(modes):
            approx = t
            for j, m2 in enumerate(modes):
                if j != i:
                    approx = _mode_dot(approx, factors[j].T, m2)
            factors[i] = torch.linalg.svd(unfold(approx, m), full_matrices=False)[0][:, :rank[i]]
        core = t
        for j, m2 in enumerate(modes):
            core = _mode_dot(core, factors[j].T, m2)
        errs.append(float(torch.sqrt(torch.abs(norm_t ** 2 - torch.linalg.norm(core) ** 2)) / norm_t))
        if it > 1 and tol and abs(errs[-2] - errs[-1]) < tol:
            break
    return core.to(out_dtype), [f.to(out_dtype) for f in factors]


def estimate_ranks(layer: nn.Conv2d) -> List[int]:
    w = layer.weight.data
    _, d0, _, _ = EVBMF(unfold(w, 0))
    _, d1, _, _ = EVBMF(unfold(w, 1))
    return [d0.shape[0], d1.shape[1]]


def tucker_decomposition_conv_layer(layer: nn.Conv2d) -> nn.Sequential:
    """Conv2d(k x k) -> Sequential(1x1 Cin->r_in, k x k r_in->r_out (stride/pad/dilation of the original),
    1x1 r_out->Cout carrying the bias).  Raises ValueError when a rank is estimated as 0."""
    ranks = estimate_ranks(layer)
    if min(ranks) < 1:
        raise ValueError(f"estimated ranks {ranks}")
    core, (last, first) = partial_tucker(layer.weight.data, modes=[0, 1], rank=ranks, init="svd")
    dev, dt = layer.weight.device, layer.weight.dtype
    first_layer = nn.Conv2d(first.shape[0], first.shape[1], 1, 1, 0, dilation=layer.dilation, bias=False).to(dev, dt)
    core_layer = nn.Conv2d(core.shape[1], core.shape[0], layer.kernel_size, layer.stride, layer.padding,
                           dilation=layer.dilation, bias=False).to(dev, dt)
    last_layer = nn.Conv2d(last.shape[1], last.shape[0], 1, 1, 0, dilation=layer.dilation, bias=layer.bias is not None).to(dev, dt)
    if layer.bias is not None:
        last_layer.bias.data = layer.bias.data
    first_layer.weight.data = first.t().unsqueeze(-1).unsqueeze(-1).contiguous()
    last_layer.weight.data = last.unsqueeze(-1).unsqueeze(-1).contiguous()
    core_layer.weight.data = core.contiguous(memory_format=torch.channels_last)
    return nn.Sequential(first_layer, core_layer, last_layer)


def decompose_layer_evaluation(layer: nn.Conv2d, test_input: torch.Tensor, origin_out: torch.Tensor):
    try:
        dec = tucker_decomposition_conv_layer(deepcopy(layer))
    except ValueError:
        return None, float("inf")
    with torch.no_grad():
        out = dec(test_input)
    return dec, torch.abs(origin_out - out).sum() / origin_out.numel()


def decompose_model(model: nn.Module, loss_thr: float = 0.1, prune_step: float = 0.01) -> None:
    """In place: every conv that is ``parent.conv`` or an element of a ModuleList, with k > 1, is replaced by its
    Tucker-2 Sequential when the probe loss is under ``loss_thr`` (after the best prune ratio found by bisection)."""
    for i, (name, module) in enumerate(model.named_children()):
        if len(list(module.children())) > 0:
            decompose_model(module, loss_thr=loss_thr, prune_step=prune_step)
        if not isinstance(module, nn.Conv2d):
            continue
        conv = model[i] if isinstance(model, nn.ModuleList) else getattr(model, "conv", None)
        if conv is not module or conv.kernel_size == (1, 1):
            continue
        with torch.no_grad():
            test_input = torch.rand((1024, *conv.weight.shape[1:]), device=conv.weight.device, dtype=conv.weight.dtype)
            origin_out = conv(test_input)
        candidate, loss = decompose_layer_evaluation(conv, test_input, origin_out)
        best = candidate if loss < loss_thr else None
        search = best is not None and prune_step > 0
        lo, hi = 0.0, 1.0
        ratio = (lo + hi) / 2
        while search:
            pruned = deepcopy(conv)
            if ratio > 0.0:
                prune.l1_unstructured(pruned, name="weight", amount=ratio)
                prune.remove(pruned, "weight")
            candidate, loss = decompose_layer_evaluation(pruned, test_input, origin_out)
            if loss < loss_thr:
                lo, best = ratio, candidate
            else:
                hi = ratio
            nxt = (lo + hi) / 2
            if abs(ratio - nxt) == 0 or abs(ratio - nxt) < prune_step:
                break
            ratio = nxt
        if best is None:
            continue
        for attr in ("in_channels", "out_channels", "kernel_size"):
            setattr(best, attr, getattr(conv, attr))
        if isinstance(model, nn.ModuleList):
            model[i] = best
        else:
            model.conv = best


def count_param(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())


def run_decompose(model: nn.Module, validator, device, loss_thr: float = 0.1, prune_step: float = 0.01):
    """decompose_model.py:107-152 ``run_decompose``: decompose a copy of `model` on the CPU (the SVDs are host work, as in
    the reference), move it to `device` in eval mode and validate it with `validator` (a ``YoloValidator``; ``None`` skips
    the validation).  Returns (decomposed_model, validation result); the eval forward of the decomposed model runs on the
    inference executor (three launches per decomposed block, infer_plan.py)."""
    import time
    plans = model.__dict__.pop("_plans", None)               # cached executor plans are not part of the copy
    try:
        decomposed = deepcopy(model).cpu()
    finally:
        if plans is not None:
            model.__dict__["_plans"] = plans
    decompose_model(decomposed, loss_thr=loss_thr, prune_step=prune_step)
    n0, n1 = count_param(model), count_param(decomposed)
    decomposed.to(device).eval()
    result = None
    took = 0.0
    if validator is not None:
        validator.model = decomposed
        t0 = time.monotonic()
        result = validator.validation()
        took = time.monotonic() - t0
    decomposed.decompose_info = {"params_before": n0, "params_after": n1, "loss_thr": loss_thr, "prune_step": prune_step,
                                 "validation_s": took, "ranks": decomposed_ranks(decomposed)}
    return decomposed, result


def decomposed_ranks(model: nn.Module) -> Dict[str, Tuple[int, int]]:
    """{module name: (r_in, r_out)} of every Tucker-decomposed conv (the 3-conv Sequential of
    tucker_decomposition_conv_layer)."""
    out = {}
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Sequential) and len(mod) == 3 and all(isinstance(c, nn.Conv2d) for c in mod) \
                and mod[0].kernel_size == (1, 1) and mod[2].kernel_size == (1, 1):
            out[name] = (mod[0].out_channels, mod[1].out_channels)
    return out


def save_decomposed(model: nn.Module, path: str) -> None:
    """decompose_model.py:293-299: the decomposed model is stored in half precision."""
    plans = model.__dict__.pop("_plans", None)
    try:
        torch.save({"model": deepcopy(model).cpu().half()}, path)
    finally:
        if plans is not None:
            model.__dict__["_plans"] = plans
