"""Parity-harness helpers mirroring the small tensor utilities of ``flocoder/metrics.py``.

Only what the sampling path touches (``g2rgb``, sampling.py:166) and what the parity gates need (rel-L2, the
``normalize_recon`` / ``to_uint8`` pre-processing of compute_sample_metrics, metrics.py:258-263,479-488).  Sinkhorn
(geomloss) and FID (torchmetrics + Inception weights) are third-party and absent offline: SURVEY.md 8(f) N3.
"""
import torch


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
