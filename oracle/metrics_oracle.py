"""CPU restatement of the parity metrics behind ``flocoder/metrics.py`` (TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``).

Parity status: **PARITY UNPINNED** for both functions.  The arithmetic lives in third-party packages that are absent from
``/root/reference`` and from this image, unpinned in ``pyproject.toml:49,51``:

* ``sinkhorn_loss`` (metrics.py:40-54) = ``geomloss.SamplesLoss("sinkhorn", p=2, blur=0.05)``.  Restated below from the published
  tensorized algorithm of geomloss 0.2.6 (``sinkhorn_samples.sinkhorn_tensorized`` -> ``sinkhorn_divergence.sinkhorn_loop`` /
  ``sinkhorn_cost``; defaults scaling=0.5, debias=True, reach=None, uniform weights): cost |x-y|^2/2, eps schedule
  [diam^2] + exp(arange(2 log diam, 2 log blur, 2 log scaling)) + [blur^2], symmetrised Jacobi updates of the four dual potentials,
  one final un-averaged update, value <a, f_ba - f_aa> + <b, g_ab - g_bb>.  ``dtype=float32`` follows the package's arithmetic
  (distances as |x|^2 - 2xy + |y|^2, everything in the input precision); ``float64`` is the same definition evaluated accurately and is
  what the HIP kernel is held to.
* ``fid_score`` (metrics.py:291-308) = ``torchmetrics.image.fid.FrechetInceptionDistance(feature=2048)``: Inception-v3 pool features of
  the uint8 images, then d^2 = |mu1-mu2|^2 + tr(S1) + tr(S2) - 2 sum sqrt(eig(S1 S2)) with unbiased covariances, in float64
  (torchmetrics ``_compute_fid``).  The network weights cannot be restated; ``frechet_distance`` below is the statistics half.

The reference's own call sites (metrics.py:493-555 compute_sample_metrics) anchor what is computed ON: latents flattened per sample,
decoded images after ``normalize_recon``, ``to_uint8`` before Inception.
"""
from __future__ import annotations

import math

import numpy as np
import torch

Tensor = torch.Tensor


def max_diameter(x: Tensor, y: Tensor) -> float:
    mins = torch.minimum(x.min(dim=0)[0], y.min(dim=0)[0])
    maxs = torch.maximum(x.max(dim=0)[0], y.max(dim=0)[0])
    return float((maxs - mins).double().norm())


def epsilon_schedule(diameter: float, blur: float, scaling: float, p: int = 2):
    return ([diameter ** p] + [float(np.exp(e)) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
            + [blur ** p])


def _half_sq_dist(x: Tensor, y: Tensor, exact: bool) -> Tensor:
    if exact:
        return 0.5 * torch.cdist(x, y, p=2, compute_mode="donot_use_mm_for_euclid_dist") ** 2
    return ((x * x).sum(-1).unsqueeze(1) - 2 * x @ y.t() + (y * y).sum(-1).unsqueeze(0)) / 2     # geomloss squared_distances / 2


def _softmin(eps: float, C: Tensor, h: Tensor) -> Tensor:
    return -eps * torch.logsumexp(h.view(1, -1) - C / eps, dim=1)


def sinkhorn_divergence(x: Tensor, y: Tensor, blur: float = 0.05, scaling: float = 0.5, diameter: float | None = None,
                        dtype=torch.float64, return_info: bool = False):
    """S_eps(x, y) for point clouds [N,D], [M,D] with uniform weights."""
    exact = dtype == torch.float64
    x, y = x.reshape(x.shape[0], -1).to(dtype), y.reshape(y.shape[0], -1).to(dtype)
    n, m = x.shape[0], y.shape[0]
    if diameter is None:
        diameter = max_diameter(x, y)
    if diameter <= 0:
        return (0.0, {"diameter": 0.0, "iterations": 0}) if return_info else 0.0
    eps_list = epsilon_schedule(diameter, blur, scaling)
    a_log = torch.full((n,), -math.log(n), dtype=dtype)
    b_log = torch.full((m,), -math.log(m), dtype=dtype)
    C_xy, C_yx = _half_sq_dist(x, y, exact), _half_sq_dist(y, x, exact)
    C_xx, C_yy = _half_sq_dist(x, x, exact), _half_sq_dist(y, y, exact)
    eps = eps_list[0]
    f_aa, g_bb = _softmin(eps, C_xx, a_log), _softmin(eps, C_yy, b_log)
    g_ab, f_ba = _softmin(eps, C_yx, a_log), _softmin(eps, C_xy, b_log)
    for eps in eps_list:
        ft_aa, gt_bb = _softmin(eps, C_xx, a_log + f_aa / eps), _softmin(eps, C_yy, b_log + g_bb / eps)
        ft_ba, gt_ab = _softmin(eps, C_xy, b_log + g_ab / eps), _softmin(eps, C_yx, a_log + f_ba / eps)
        f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)
        f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
    f_aa, g_bb = _softmin(eps, C_xx, a_log + f_aa / eps), _softmin(eps, C_yy, b_log + g_bb / eps)
    f_ba, g_ab = _softmin(eps, C_xy, b_log + g_ab / eps), _softmin(eps, C_yx, a_log + f_ba / eps)
    val = float((f_ba - f_aa).mean() + (g_ab - g_bb).mean())
    return (val, {"diameter": diameter, "iterations": len(eps_list)}) if return_info else val


def sinkhorn_loss(target: Tensor, gen: Tensor, max_B=None, dtype=torch.float64) -> float:
    """metrics.py:40-54 on flattened samples."""
    assert target.shape == gen.shape
    B = target.shape[0] if max_B is None else min(target.shape[0], max_B)
    return sinkhorn_divergence(target[:B].reshape(B, -1), gen[:B].reshape(B, -1), dtype=dtype)


def feature_statistics(feats: Tensor):
    """(mean, unbiased covariance) in float64, accumulated the way torchmetrics does (sum and sum of outer products)."""
    f = feats.double()
    n = f.shape[0]
    s, ss = f.sum(0), f.t() @ f
    mu = s / n
    return mu, (ss - n * torch.outer(mu, mu)) / (n - 1)


def frechet_distance(mu1: Tensor, sigma1: Tensor, mu2: Tensor, sigma2: Tensor) -> float:
    a = (mu1 - mu2).square().sum()
    b = sigma1.trace() + sigma2.trace()
    c = torch.linalg.eigvals(sigma1 @ sigma2).sqrt().real.sum()
    return float(a + b - 2 * c)


def to_uint8(x: Tensor) -> Tensor:
    x = x.clone().detach()
    x -= x.amin(dim=(1, 2, 3), keepdim=True)
    x /= x.amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-5)
    return (x * 255).clamp(0, 255).to(torch.uint8)
