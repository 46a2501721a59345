"""Parity metrics: host-side mirror of the sample metrics of ``flocoder/metrics.py``.

* ``g2rgb`` / ``rgb2g`` / ``to_uint8`` / ``normalize_recon`` -- the small tensor utilities (metrics.py:258-263,312-327,479-488);
* ``sinkhorn_loss`` (metrics.py:20-54) -- the debiased Sinkhorn divergence (p=2, blur 0.05, eps-scaling 0.5) on the device through
  ``fc_sinkhorn_divergence``: own HIP kernels restating geomloss's published algorithm (the package is absent: PARITY UNPINNED);
* ``fid_score`` (metrics.py:266-308) -- Frechet distance between Inception-pool features of real and generated images.  The
  statistics and the distance are computed here in float64; the feature network is a LOCAL TorchScript file (``inception=`` or
  ``$FLOCODER_FID_INCEPTION``) mapping uint8 images [B,3,H,W] to features [B,F] -- torchmetrics downloads its weights, this build
  never touches the network and raises FileNotFoundError without one;
* ``compute_sample_metrics`` (metrics.py:493-555) -- the same dictionary of numbers.
"""
import ctypes as C
import os

import torch

from . import _binding as B


def rgb2g(img_t):
    """metrics.py:312-317 -- RGB piano roll -> grey float: black 0, red 1.0, green 0.5."""
    red = (img_t[-3] > 0.5).float()
    green = (img_t[-2] > 0.5).float() * 0.5
    return (red + green).unsqueeze(-3)


def g2rgb(gf_img, keep_gray=False):
    """metrics.py:319-327 -- grey float -> quantised RGB (0 black, 1 red, 0.5 green) or binary b/w."""
    if gf_img.shape[-3] == 3:
        return gf_img
    gf = gf_img.squeeze(-3)
    if keep_gray:
        return (gf > 0.5).float().unsqueeze(-3).repeat(1, 3, 1, 1)
    return torch.stack([(gf >= 0.75).float(), (torch.abs(gf - 0.5) < 0.25).float(), torch.zeros_like(gf)], dim=-3)


def to_uint8(x):
    """metrics.py:258-263 -- per-image min/max stretch to uint8."""
    x = x.clone().detach()
    x -= x.amin(dim=(1, 2, 3), keepdim=True)
    x /= x.amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-5)
    return (x * 255).clamp(0, 255).to(torch.uint8)


def normalize_recon(orig, recon):
    """metrics.py:479-488 -- rescale each RGB channel of `recon` to the range of `orig` (vectorised; in place like upstream)."""
    o_min, o_max = orig[:, :3].amin(dim=(2, 3), keepdim=True), orig[:, :3].amax(dim=(2, 3), keepdim=True)
    r_min, r_max = recon[:, :3].amin(dim=(2, 3), keepdim=True), recon[:, :3].amax(dim=(2, 3), keepdim=True)
    ok = r_max > r_min
    scaled = (recon[:, :3] - r_min) / (r_max - r_min).clamp_min(1e-30) * (o_max - o_min) + o_min
    recon[:, :3] = torch.where(ok, scaled, recon[:, :3])
    return recon


def rel_l2(a, b):
    """||a-b|| / ||b|| in fp64 -- the parity gate of BASELINE.md section 4."""
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def sample_stats(pred_latents, target_latents, decoded_pred, decoded_target):
    """The third-party-free part of compute_sample_metrics (metrics.py:530-545)."""
    n = min(pred_latents.shape[0], target_latents.shape[0])
    f = torch.nn.functional.mse_loss
    return {
        'mse': f(pred_latents[:n], target_latents[:n]).item(), 'mse_px': f(decoded_pred, decoded_target).item(),
        'pred_mean': pred_latents.mean().item(), 'targ_mean': target_latents.mean().item(),
        'pred_std': pred_latents.std().item(), 'targ_std': target_latents.std().item(),
        'pred_px_mean': decoded_pred.mean().item(), 'targ_px_mean': decoded_target.mean().item(),
        'pred_px_std': decoded_pred.std().item(), 'targ_px_std': decoded_target.std().item(),
    }


# ------------------------------------------------------------------------------------------------ sinkhorn (metrics.py:20-54)
def sinkhorn_divergence(x, y, blur: float = 0.05, scaling: float = 0.5, diameter=None, return_info: bool = False):
    """S_eps between point clouds x [N,...] and y [M,...] (flattened per sample), uniform weights, cost |x-y|^2/2, on the GPU."""
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError("flocoder_amd.metrics.sinkhorn_divergence runs on MI355X (gfx950) only; there is no CPU path "
                           "(the CPU restatement under oracle/ is test infrastructure)")
    xf = x.reshape(x.shape[0], -1).float().contiguous()
    yf = y.reshape(y.shape[0], -1).float().contiguous()
    if xf.shape[1] != yf.shape[1]:
        raise ValueError("sinkhorn: both clouds must live in the same space")
    val, diam, its = C.c_double(0), C.c_double(0), C.c_int(0)
    B.check(B.lib().fc_sinkhorn_divergence(B.ptr(xf), B.ptr(yf), xf.shape[0], yf.shape[0], xf.shape[1], float(blur), float(scaling),
                                           float(diameter) if diameter else 0.0, C.byref(val), C.byref(diam), C.byref(its),
                                           B.current_stream(xf.device)))
    return (val.value, {"diameter": diam.value, "iterations": its.value}) if return_info else val.value


def sinkhorn_loss_chunked(target, gen, chunk_size=256, device='cuda'):
    """metrics.py:20-38: the mean of per-chunk divergences."""
    assert target.shape == gen.shape, f"target.shape {target.shape} != gen.shape {gen.shape}"
    vals = []
    for i in range(0, target.shape[0], chunk_size):
        vals.append(sinkhorn_divergence(target[i:i + chunk_size].to(device), gen[i:i + chunk_size].to(device)))
    return sum(vals) / len(vals)


def sinkhorn_loss(target, gen, max_B=None, device='cuda', chunk=False, debug=False):
    """metrics.py:40-54: SamplesLoss("sinkhorn", p=2, blur=0.05)(target, gen) on samples flattened to vectors; a Python float."""
    if chunk:
        return sinkhorn_loss_chunked(target, gen, device=device)
    assert target.shape == gen.shape, f"target.shape {target.shape} != gen.shape {gen.shape}"
    n = target.shape[0] if max_B is None else min(target.shape[0], max_B)
    return sinkhorn_divergence(target[:n].to(device), gen[:n].to(device))


# ------------------------------------------------------------------------------------------------ FID (metrics.py:266-308)
_INCEPTION = {}


def load_feature_extractor(path=None, device='cuda'):
    """The Inception-v3 pool3 network torchmetrics fetches, from a LOCAL TorchScript file (uint8 [B,3,H,W] -> float [B,F])."""
    path = path or os.environ.get("FLOCODER_FID_INCEPTION")
    if not path:
        raise FileNotFoundError("fid_score: no local Inception feature extractor.  Pass inception=<TorchScript file | callable> or set "
                                "FLOCODER_FID_INCEPTION (torchmetrics downloads its weights; this build never touches the network).")
    if not os.path.exists(path):
        raise FileNotFoundError(f"fid_score: feature extractor {path} does not exist")
    key = (os.path.abspath(path), str(device))
    if key not in _INCEPTION:
        _INCEPTION[key] = torch.jit.load(path, map_location=device).eval()
    return _INCEPTION[key]


class FrechetStatistics:
    """Running sum and sum of outer products of feature vectors in float64 (what torchmetrics' FrechetInceptionDistance keeps)."""

    def __init__(self):
        self.n, self.s, self.ss = 0, None, None

    def update(self, feats: torch.Tensor):
        f = feats.reshape(feats.shape[0], -1).double()
        if self.s is None:
            self.s = torch.zeros(f.shape[1], dtype=torch.float64, device=f.device)
            self.ss = torch.zeros(f.shape[1], f.shape[1], dtype=torch.float64, device=f.device)
        self.n += f.shape[0]
        self.s += f.sum(0)
        self.ss += f.t() @ f

    def moments(self):
        if self.n < 2:
            raise RuntimeError("FID needs at least two samples per side")
        mu = self.s / self.n
        return mu, (self.ss - self.n * torch.outer(mu, mu)) / (self.n - 1)


def frechet_distance(mu1, sigma1, mu2, sigma2) -> float:
    """|mu1 - mu2|^2 + tr(S1) + tr(S2) - 2 tr(sqrt(S1 S2)), the matrix square root through the eigenvalues of S1 S2."""
    a = (mu1 - mu2).square().sum()
    b = sigma1.trace() + sigma2.trace()
    c = torch.linalg.eigvals((sigma1 @ sigma2).cpu()).sqrt().real.sum().to(a.device)
    return float(a + b - 2 * c)


@torch.no_grad()
def fid_score(real, fake, device='cuda', chunk=False, inception=None, chunk_size=256):
    """metrics.py:266-308.  ``inception``: a callable uint8 [B,3,H,W] -> features, or a TorchScript path (see load_feature_extractor)."""
    net = inception if callable(inception) else load_feature_extractor(inception, device)
    stats = (FrechetStatistics(), FrechetStatistics())
    step = chunk_size if chunk else max(real.shape[0], fake.shape[0])
    for side, imgs in enumerate((real, fake)):
        for i in range(0, imgs.shape[0], step):
            u8 = to_uint8(imgs[i:i + step].to(device))
            if u8.shape[1] == 1:
                u8 = u8.repeat(1, 3, 1, 1)
            stats[side].update(net(u8))
    return frechet_distance(*stats[0].moments(), *stats[1].moments())


@torch.no_grad()
def compute_sample_metrics(pred_latents, target_latents, decoded_pred, decoded_target, debug=False, inception=None):
    """metrics.py:493-555: the dictionary evaluate_model logs.  'FID_px' is present only when a local feature extractor is (it is
    the one entry that needs third-party weights); everything else is computed unconditionally."""
    n = min(pred_latents.shape[0], target_latents.shape[0])
    decoded_pred = normalize_recon(decoded_target, decoded_pred)
    out = {}
    if inception is not None or os.environ.get("FLOCODER_FID_INCEPTION"):
        out['FID_px'] = fid_score(decoded_target, decoded_pred, device=decoded_pred.device, inception=inception)
    out['sinkhorn'] = sinkhorn_loss(target_latents[:n], pred_latents[:n], device=pred_latents.device)
    out['sinkhorn_px'] = sinkhorn_loss(decoded_target, decoded_pred, device=decoded_pred.device)
    out.update(sample_stats(pred_latents, target_latents, decoded_pred, decoded_target))
    return out
