"""CPU restatement of flocoder's latent-flow hot path (velocity U-Net, ODE
integrators, greedy OT pairing, mask encoder).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity status: PINNED.
Every function here is checked in ``tests/test_oracle_golden.py`` against
fixtures under ``tests/golden/`` that ``tools/make_golden.py`` produced by
importing the reference modules from ``/root/reference`` in the build container.

The restatement is functional: a model is a plain ``dict`` name -> tensor with the
reference's ``state_dict`` key names, evaluated with
``torch.nn.functional`` on the CPU in the dtype of the weights (fp32 for parity
and timing, fp64 when a test wants a tighter yardstick).  All ``file:line``
citations are into ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- U-Net

def unet_meta(sd: SD) -> dict:
    """Recover the constructor arguments from the weights, the way the reference's loader
    does (generate_samples.py:91-101): dim/channels from ``init_conv``, depth from ``downs``."""
    dim, channels = sd["init_conv.weight"].shape[:2]
    n_levels = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("downs."))
    chans = [dim] + [sd[f"downs.{i}.3.weight" if i == n_levels - 1 else f"downs.{i}.3.1.weight"].shape[0]
                     for i in range(n_levels)]
    return dict(dim=dim, channels=channels, n_levels=n_levels, chans=chans,
                time_dim=sd["time_mlp.1.weight"].shape[0],
                n_classes=sd["class_cond_mlp.0.weight"].shape[0] if "class_cond_mlp.0.weight" in sd else 0,
                mask_cond="mask_fusion_conv.0.weight" in sd,
                groups=4)  # resnet_block_groups default, unet.py:170


def sinusoidal_embedding(time: Tensor, dim: int) -> Tensor:
    """unet.py:18-30 -- frequencies exp(-k ln(1e4)/(dim/2-1)), sin half first."""
    half = dim // 2
    k = torch.arange(half, dtype=time.dtype)
    freqs = torch.exp(k * -(math.log(10000) / (half - 1)))
    arg = time[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def _conv(sd: SD, name: str, x: Tensor, padding: int = 0, stride: int = 1) -> Tensor:
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _gn(sd: SD, name: str, x: Tensor, groups: int) -> Tensor:
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def block(sd: SD, p: str, x: Tensor, groups: int, scale_shift=None) -> Tensor:
    """unet.py:57-73 -- conv3x3 -> GroupNorm -> x*(scale+1)+shift -> SiLU."""
    x = _gn(sd, p + ".norm", _conv(sd, p + ".proj", x, padding=1), groups)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(sd: SD, p: str, x: Tensor, temb: Tensor, groups: int, taps: Optional[dict] = None) -> Tensor:
    """unet.py:76-96 -- FiLM only on block1; 1x1 ``res_conv`` iff channel count changes."""
    ss = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, :, None, None]
    scale, shift = ss.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, groups, (scale, shift))
    if taps is not None:                      # raw conv outputs, as the GPU path stores them
        taps[p + ".h1"] = _conv(sd, p + ".block1.proj", x, padding=1)
        taps[p + ".h2"] = _conv(sd, p + ".block2.proj", h, padding=1)
    h = block(sd, p + ".block2", h, groups)
    res = _conv(sd, p + ".res_conv", x) if (p + ".res_conv.weight") in sd else x
    return h + res


def linear_attention(sd: SD, p: str, x: Tensor, heads: int = 4) -> Tensor:
    """unet.py:125-150 -- q softmax over d, k softmax over n, ctx = k v^T, out = ctx^T q."""
    b, c, h, w = x.shape
    qkv = _conv(sd, p + ".to_qkv", x).reshape(b, 3, heads, -1, h * w)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]                   # b heads d n
    d = q.shape[2]
    q = q.softmax(dim=-2) * d ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * d, h, w)
    out = _conv(sd, p + ".to_out.0", out)
    return _gn(sd, p + ".to_out.1", out, 1)


def full_attention(sd: SD, p: str, x: Tensor, heads: int = 4) -> Tensor:
    """unet.py:99-122 -- softmax(q^T k * d^-1/2) v, row-max subtracted."""
    b, c, h, w = x.shape
    qkv = _conv(sd, p + ".to_qkv", x).reshape(b, 3, heads, -1, h * w)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    d = q.shape[2]
    sim = torch.einsum("bhdi,bhdj->bhij", q * d ** -0.5, k)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)             # b heads n d
    out = out.permute(0, 1, 3, 2).reshape(b, heads * d, h, w)
    return _conv(sd, p + ".to_out", out)


def _prenorm_residual(sd: SD, p: str, x: Tensor, fn) -> Tensor:
    """Residual(PreNorm(fn)), unet.py:33-39,153-161: GroupNorm(1) then fn, plus x."""
    return fn(sd, p + ".fn.fn", _gn(sd, p + ".fn.norm", x, 1)) + x


def space_to_depth_conv(sd: SD, name: str, x: Tensor) -> Tensor:
    """Downsample, unet.py:49-54: 'b c (h p1) (w p2) -> b (c p1 p2) h w' then 1x1 conv."""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)
    return _conv(sd, name, x)


def time_embedding(sd: SD, time: Tensor, class_cond: Optional[Tensor]) -> Tensor:
    """unet.py:199-212,310-316 -- Linear/GELU(erf)/Linear on the sinusoid; class MLP added."""
    dim = sd["time_mlp.1.weight"].shape[1]
    t = sinusoidal_embedding(time, dim)
    t = F.linear(t, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    t = F.linear(F.gelu(t), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    if class_cond is not None and "class_cond_mlp.0.weight" in sd:
        c = F.embedding(class_cond, sd["class_cond_mlp.0.weight"])
        c = F.linear(c, sd["class_cond_mlp.1.weight"], sd["class_cond_mlp.1.bias"])
        c = F.linear(F.gelu(c), sd["class_cond_mlp.3.weight"], sd["class_cond_mlp.3.bias"])
        t = t + c
    return t


def unet_forward(sd: SD, x: Tensor, time: Tensor, cond: Optional[dict] = None, taps: Optional[dict] = None) -> Tensor:
    """Unet._forward, unet.py:289-372.  ``cond`` is None or a dict with optional
    'class_cond' (int64 [B]) and 'mask_cond' ([B,C,H,W]); non-dict cond is dead upstream
    (SURVEY Q15).  ``taps`` (tests only) collects block outputs by reference module name."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v
    m = unet_meta(sd)
    g, L = m["groups"], m["n_levels"]
    class_cond = cond.get("class_cond") if isinstance(cond, dict) else None
    mask = cond.get("mask_cond") if isinstance(cond, dict) else None
    use_mask = mask is not None and m["mask_cond"]

    x = _conv(sd, "init_conv", x)
    if use_mask and not torch.allclose(mask, torch.ones_like(mask)):       # unet.py:298-305
        f = torch.cat([x, mask], dim=1)
        f = F.silu(_conv(sd, "mask_fusion_conv.0", f, padding=2))
        f = F.silu(_conv(sd, "mask_fusion_conv.2", f, padding=1))
        x = _conv(sd, "mask_fusion_conv.4", f, padding=1)                  # replaces x, no residual
    r = tap("init", x)
    t = time_embedding(sd, time, class_cond)

    def inject(x, prefix, i):                                               # unet.py:336-340,360-364
        mr = F.interpolate(mask, size=x.shape[-2:], mode="bilinear")
        return x + F.silu(_conv(sd, f"{prefix}.{i}.0", torch.cat([x, mr], dim=1), padding=1))

    skips = []
    for i in range(L):
        p = f"downs.{i}"
        x = tap(p + ".0", resnet_block(sd, p + ".0", x, t, g, taps)); skips.append(x)
        x = tap(p + ".1", resnet_block(sd, p + ".1", x, t, g, taps))
        x = tap(p + ".2", _prenorm_residual(sd, p + ".2", x, linear_attention)); skips.append(x)
        if use_mask and i < 2:
            x = inject(x, "down_mask_fusions", i)
        x = _conv(sd, p + ".3", x, padding=1) if i == L - 1 else space_to_depth_conv(sd, p + ".3.1", x)
        tap(p + ".3", x)

    x = tap("mid_block1", resnet_block(sd, "mid_block1", x, t, g, taps))
    x = tap("mid_attn", _prenorm_residual(sd, "mid_attn", x, full_attention))
    x = tap("mid_block2", resnet_block(sd, "mid_block2", x, t, g, taps))

    for i in range(L):
        p = f"ups.{i}"
        x = tap(p + ".0", resnet_block(sd, p + ".0", torch.cat((x, skips.pop()), dim=1), t, g))
        x = tap(p + ".1", resnet_block(sd, p + ".1", torch.cat((x, skips.pop()), dim=1), t, g))
        x = tap(p + ".2", _prenorm_residual(sd, p + ".2", x, linear_attention))
        if use_mask and i < 2:
            x = inject(x, "up_mask_fusions", i)
        if i == L - 1:
            x = _conv(sd, p + ".3", x, padding=1)
        else:                                                               # Upsample, unet.py:42-46
            x = _conv(sd, p + ".3.1", F.interpolate(x, scale_factor=2, mode="nearest"), padding=1)
        tap(p + ".3", x)

    x = tap("final_res_block", resnet_block(sd, "final_res_block", torch.cat((x, r), dim=1), t, g))
    return _conv(sd, "final_conv", x)


# --------------------------------------------------------------------------- integrators

def warp_time(t: Tensor, s: float = 0.5) -> Tensor:
    """sampling.py:23-33 (the never-taken ``dt`` branch, SURVEY Q4, is not restated)."""
    if s < 0 or s > 1.5:
        raise ValueError(f"s={s} is out of bounds.")
    return 4 * (1 - s) * t ** 3 + 6 * (s - 1) * t ** 2 + (3 - 2 * s) * t


def rk4_time_grid(n_steps: int, dtype=torch.float32, init_strength: Optional[float] = None) -> Tensor:
    """sampling.py:102,104-111 -- linspace(0,1,n_steps) (n_steps-1 intervals), or
    the init-image variant linspace(s,1,max(1,int(n(1-s)))); always warped."""
    if init_strength is None:
        ts = torch.linspace(0, 1, n_steps, dtype=dtype)
    else:
        n_steps = max(1, int(n_steps * (1.0 - init_strength)))
        ts = torch.linspace(init_strength, 1.0, n_steps, dtype=dtype)
    return warp_time(ts)


def velocity_cfg(sd: SD, cond: Optional[dict], cfg_strength: float, x: Tensor, t: Tensor,
                 t_scale: float = 999) -> Tensor:
    """v_func_cfg, sampling.py:50-76: second pass drops only class_cond."""
    t_vec = torch.full((x.shape[0],), float(t), dtype=x.dtype)
    v = unet_forward(sd, x, t_vec * t_scale, cond)
    if cond and cond.get("class_cond") is not None and cfg_strength:
        nc = dict(cond); nc["class_cond"] = None
        v_nc = unet_forward(sd, x, t_vec * t_scale, nc)
        v = v_nc + cfg_strength * (v - v_nc)
    return v


def rk4_step(f, y: Tensor, t: Tensor, dt: Tensor) -> Tensor:
    """sampling.py:36-48."""
    k1 = f(y, t)
    th = t + dt / 2
    k2 = f(y + dt * k1 / 2, th)
    k3 = f(y + dt * k2 / 2, th)
    k4 = f(y + dt * k3, t + dt)
    return y + (dt / 6) * (k1 + 2 * k2 + 2 * k3 + k4)


def generate_latents_rk4(sd: SD, source: Tensor, n_steps: int = 50, cond: Optional[dict] = None,
                         cfg_strength: float = 3.0, init_latents: Optional[Tensor] = None,
                         init_strength: float = 0.0):
    """sampling.py:78-122 with the noise supplied by the caller (``source``); jitter is off on
    every live path (sampling.py:103, jitter_strength default 0).  Returns (latents, nfe)."""
    y = source
    if init_latents is None:
        ts = rk4_time_grid(n_steps, source.dtype)
    else:
        y = (1 - init_strength) * y + init_strength * init_latents
        ts = rk4_time_grid(n_steps, source.dtype, init_strength=init_strength)
        n_steps = max(1, int(n_steps * (1.0 - init_strength)))
    f = lambda yy, tt: velocity_cfg(sd, cond, cfg_strength, yy, tt)
    for i in range(len(ts) - 1):
        y = rk4_step(f, y, ts[i], ts[i + 1] - ts[i])
    return y, n_steps * 4


def euler_time_grid(n: int, eps: float = 1e-3) -> Tensor:
    """legacy/train_sd_flowers.py:59-62 -- t_i = i/N*(1-eps)+eps in Python floats, then the
    fp32 product ones*t."""
    return torch.tensor([i / n * (1 - eps) + eps for i in range(n)], dtype=torch.float64).to(torch.float32)


def euler_sampler(sd: SD, source: Tensor, n: int, class_ids: Optional[Tensor], eps: float = 1e-3):
    """legacy/train_sd_flowers.py:50-67 driving the live Unet: no warp, no CFG, nfe = N.  The
    legacy call passes the class-id tensor positionally; the live Unet only honours dict cond
    (SURVEY Q15), so the ids travel as {'class_cond': ids}."""
    x = source.clone()
    dt = 1.0 / n
    cond = {"class_cond": class_ids} if class_ids is not None else None
    for t in euler_time_grid(n, eps).to(source.dtype):
        t_vec = torch.ones(x.shape[0], dtype=x.dtype) * t
        x = x + unet_forward(sd, x, t_vec * 999, cond) * dt
    return x, n


# --------------------------------------------------------------------------- OT pairing

def ot_pairing_greedy(source: Tensor, target: Tensor) -> Tensor:
    """compute_ot_pairing_approximate, ot.py:63-78: L2 cdist, then for i in order the nearest
    still-unused target (argmin = first minimum).  int64 permutation."""
    b = source.shape[0]
    d = torch.cdist(source.reshape(b, -1), target.reshape(b, -1))
    used = torch.zeros(b, dtype=torch.bool)
    out = torch.zeros(b, dtype=torch.long)
    for i in range(b):
        row = d[i].masked_fill(used, float("inf"))
        j = int(row.argmin())
        out[i] = j
        used[j] = True
    return out


def ot_pairing_from_distances(d: Tensor) -> Tensor:
    """The greedy sweep alone, on a given BxB distance matrix (used to check the GPU sweep
    bit-exactly on the GPU's own distances)."""
    b = d.shape[0]
    used = torch.zeros(b, dtype=torch.bool)
    out = torch.zeros(b, dtype=torch.long)
    for i in range(b):
        j = int(d[i].masked_fill(used, float("inf")).argmin())
        out[i] = j
        used[j] = True
    return out


# --------------------------------------------------------------------------- inpainting

def mask_encoder_forward(sd: SD, mask_pixels: Tensor, shrink: int = 4) -> Tensor:
    """MaskEncoder(mode='pool', final_act=sigmoid), inpainting.py:161-245.  Each
    DownsampleBlock: silu(conv s4) -> silu(conv3x3), concatenated after the avg-pooled
    channel 0; then 1x1 conv, sigmoid, and the 16x avg-pooled raw mask in front."""
    x = mask_pixels.to(sd["layers.0.conv1.weight"].dtype)
    raw = x
    for i in (0, 1):
        skip = F.avg_pool2d(x[:, 0:1], shrink, shrink)
        h = F.silu(_conv(sd, f"layers.{i}.conv1", x, stride=shrink))
        h = F.silu(_conv(sd, f"layers.{i}.conv2", h, padding=1))
        x = torch.cat([skip, h], dim=1)
    learned = torch.sigmoid(_conv(sd, "layers.2", x))
    return torch.cat([F.avg_pool2d(raw, shrink ** 2, shrink ** 2), learned], dim=1)


def mask_blending(source: Tensor, mask: Tensor, noise: Tensor) -> Tensor:
    """inpainting.py:250-253."""
    return source + mask * (noise - source)


# --------------------------------------------------------------------------- train step

def flow_train_targets(source: Tensor, target: Tensor, t: Tensor):
    """train_flow.py:350-355: x = (1-t) s + t g ; v* = g - s, with t already warped."""
    te = t.view(-1, 1, 1, 1)
    return (1 - te) * source + te * target, target - source
