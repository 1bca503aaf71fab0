"""CPU oracle for the Tucker-2 / EVBMF path (TEST INFRASTRUCTURE ONLY -- see oracle/ops_ref.py header).

* ``evbmf_rank`` / ``evb_sigma2`` restate the reference's analytic EVBMF rank estimator,
  scripts/tensor_decomposition/decomposition.py:25-206 (only the *rank* -- the shape of ``diag(d)`` -- is
  consumed by the reference, decomposition.py:357-359).
* ``unfold`` / ``partial_tucker`` restate the PUBLISHED algorithm of tensorly==0.6.0
  (environment.yml:50; NOT vendored in /root/reference, not installed here):
      unfold(t, m) = moveaxis(t, m, 0).reshape(t.shape[m], -1)
      partial_tucker: HOSVD init (leading left singular vectors of each mode unfolding), then HOOI sweeps
      until |rec_err[-2] - rec_err[-1]| < tol (tol = 1e-4, n_iter_max = 100, checked from the 3rd sweep).
  "parity unpinned" at this leaf: the only reference test that pins it
  (tests/test_tensor_decomposition.py:47-49) needs a weight blob that is absent (.MISSING_LARGE_BLOBS:5).
  Factors are unique only up to sign / rotation, so parity is checked on reconstructions and layer
  outputs, never on raw factors.
* ``tucker2_conv_weights`` follows decomposition.py:363-424 (first = 1x1 Cin->r_in with W = first^T,
  core = kxk r_in->r_out, last = 1x1 r_out->Cout carrying the bias).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
from scipy.optimize import minimize_scalar


def unfold(t: np.ndarray, mode: int) -> np.ndarray:
    t = np.asarray(t)
    return np.moveaxis(t, mode, 0).reshape(t.shape[mode], -1)


def _mode_dot(t: np.ndarray, m: np.ndarray, mode: int) -> np.ndarray:
    """t x_mode m  with m of shape (J, t.shape[mode])."""
    out = np.tensordot(m, t, axes=(1, mode))
    return np.moveaxis(out, 0, mode)


def partial_tucker(tensor, modes: Sequence[int], rank: Sequence[int], n_iter_max: int = 100,
                   tol: float = 1e-4) -> Tuple[np.ndarray, List[np.ndarray]]:
    t = np.asarray(tensor, dtype=np.float64)
    modes = list(modes)
    rank = [int(r) for r in rank]
    factors = []
    for r, mode in zip(rank, modes):
        u, _, _ = np.linalg.svd(unfold(t, mode), full_matrices=False)
        factors.append(u[:, :r])
    norm_t = np.linalg.norm(t)
    errs: List[float] = []
    core = t
    for it in range(n_iter_max):
        for idx, mode in enumerate(modes):
            approx = t
            for jdx, m2 in enumerate(modes):
                if jdx != idx:
                    approx = _mode_dot(approx, factors[jdx].T, m2)
            u, _, _ = np.linalg.svd(unfold(approx, mode), full_matrices=False)
            factors[idx] = u[:, :rank[idx]]
        core = t
        for jdx, m2 in enumerate(modes):
            core = _mode_dot(core, factors[jdx].T, m2)
        errs.append(np.sqrt(abs(norm_t ** 2 - np.linalg.norm(core) ** 2)) / norm_t)
        if it > 1 and tol and abs(errs[-2] - errs[-1]) < tol:
            break
    return core.astype(np.float32), [f.astype(np.float32) for f in factors]


def tucker_reconstruct(core, factors, modes) -> np.ndarray:
    out = np.asarray(core, np.float64)
    for f, m in zip(factors, modes):
        out = _mode_dot(out, np.asarray(f, np.float64), m)
    return out


# ---------------------------------------------------------------------------------------------
def _tau(x, alpha):
    return 0.5 * (x - (1 + alpha) + np.sqrt((x - (1 + alpha)) ** 2 - 4 * alpha))


def evb_sigma2(sigma2, L, M, s, residual, xubar):
    """decomposition.py:39-76."""
    H = len(s)
    alpha = L / M
    x = s ** 2 / (M * sigma2)
    z1, z2 = x[x > xubar], x[x <= xubar]
    tz1 = _tau(z1, alpha)
    return (np.sum(z2 - np.log(z2)) + np.sum(z1 - tz1) + np.sum(np.log((tz1 + 1) / z1))
            + alpha * np.sum(np.log(tz1 / alpha + 1)) + residual / (M * sigma2) + (L - H) * np.log(sigma2))


def evbmf_rank(Y) -> int:
    """Number of singular values above the EVBMF threshold (decomposition.py:79-206, sigma2=None, H=None)."""
    Y = np.asarray(Y)
    L, M = Y.shape
    H = L
    alpha = L / M
    tauubar = 2.5129 * np.sqrt(alpha)
    s = np.linalg.svd(Y, compute_uv=False)[:H]
    residual = 0.0
    xubar = (1 + tauubar) * (1 + alpha / tauubar)
    eH_ub = int(np.min([np.ceil(L / (1 + alpha)) - 1, H]))
    upper = (np.sum(s ** 2) + residual) / (L * M)
    lower = np.max([s[eH_ub] ** 2 / (M * xubar), np.mean(s[eH_ub:] ** 2) / M])
    res = minimize_scalar(evb_sigma2, args=(L, M, s, residual, xubar), bounds=[lower, upper], method="Bounded")
    sigma2 = res.x
    thr = np.sqrt(M * sigma2 * (1 + tauubar) * (1 + alpha / tauubar))
    return int(np.sum(s > thr))


def estimate_ranks(weight) -> List[int]:
    """decomposition.py:342-360: [rank of mode-0 unfolding, rank of mode-1 unfolding]."""
    w = np.asarray(weight)
    return [evbmf_rank(unfold(w, 0)), evbmf_rank(unfold(w, 1))]


def tucker2_conv_weights(weight, ranks=None):
    """decomposition.py:363-424 -> (first_w (r_in,Cin,1,1), core_w (r_out,r_in,kh,kw), last_w (Cout,r_out,1,1))."""
    w = np.asarray(weight, np.float32)
    if ranks is None:
        ranks = estimate_ranks(w)
    if min(ranks) < 1:
        raise ValueError("rank 0")
    core, (last, first) = partial_tucker(w, [0, 1], ranks)
    return first.T[:, :, None, None].copy(), core, last[:, :, None, None].copy()
