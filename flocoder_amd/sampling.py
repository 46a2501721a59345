"""ODE sampling: host-side mirror of ``flocoder/sampling.py`` (reference lines cited per function) plus the legacy
Euler sampler (``legacy/train_sd_flowers.py:50-67``) that BASELINE's "64-step Euler" refers to.

Same names, argument meaning and return values as the reference.  When ``model`` is a ``flocoder_amd.Unet`` on a
GPU the whole trajectory runs inside the library (one hipGraph replay per step, time grid on the device, no host
sync per velocity call -- SURVEY Q6); any other callable model goes through the same formulas with torch ops on
the model's own device.
"""
from __future__ import annotations

import gc
import random
from functools import partial
from typing import Optional

import torch

from .unet import Unet


def warp_time(t, dt=None, s=.5):
    """Parametric time warp, sampling.py:23-33.  ``dt`` is accepted for signature parity; the reference's
    derivative branch is never used and mis-parenthesised, so it is not offered here."""
    if s < 0 or s > 1.5:
        raise ValueError(f"s={s} is out of bounds.")
    if dt:
        raise NotImplementedError("warp_time(dt=...) is dead code upstream (operator-precedence bug, sampling.py:31-32)")
    return 4 * (1 - s) * t ** 3 + 6 * (s - 1) * t ** 2 + (3 - 2 * s) * t


def rk4_time_grid(n_steps: int, init_strength: Optional[float] = None, dtype=torch.float32) -> torch.Tensor:
    """The grid generate_latents_rk4 integrates on (sampling.py:102,108-111): warp_time(linspace(...)), computed with
    the same CPU torch ops as the reference so the values are bit-identical."""
    if init_strength is None:
        ts = torch.linspace(0, 1, n_steps, dtype=dtype)
    else:
        ts = torch.linspace(init_strength, 1.0, max(1, int(n_steps * (1.0 - init_strength))), dtype=dtype)
    return warp_time(ts)


def euler_time_grid(sample_N: int, eps: float = 1e-3) -> torch.Tensor:
    """t_i = i/N*(1-eps)+eps in Python floats, rounded to fp32 by ``ones * t`` (train_sd_flowers.py:59-61)."""
    return torch.tensor([i / sample_N * (1 - eps) + eps for i in range(sample_N)], dtype=torch.float64).to(torch.float32)


@torch.no_grad()
def rk4_step(f, y, t, dt, debug=False):
    """sampling.py:36-48 (generic-model path; the Unet path runs this inside the captured graph)."""
    k1 = f(y, t)
    tpdto2 = t + dt / 2
    k2 = f(y + dt * k1 / 2, tpdto2)
    k3 = f(y + dt * k2 / 2, tpdto2)
    k4 = f(y + dt * k3, t + dt)
    return y + (dt / 6) * (k1 + 2 * k2 + 2 * k3 + k4)


@torch.no_grad()
def v_func_cfg(model, cond, cfg_strength, t_vec_template, x, t, t_scale=999, debug=False):
    """sampling.py:50-76 without the three host syncs per call."""
    t_vec = t_vec_template.fill_(float(t))
    v = model(x, t_vec * t_scale, cond=cond)
    if cond and cond.get('class_cond') is not None and cfg_strength:
        cond_no_class = cond.copy()
        cond_no_class['class_cond'] = None
        v_no_class = model(x, t_vec * t_scale, cond=cond_no_class)
        v = v_no_class + cfg_strength * (v - v_no_class)
    return v


def _mask_flags(cond):
    mask = cond.get('mask_cond') if isinstance(cond, dict) else None
    ones = bool(torch.allclose(mask, torch.ones_like(mask))) if mask is not None else False   # once per call
    return mask, ones


@torch.no_grad()
def generate_latents_rk4(model, shape, n_steps=50, cond=None, cfg_strength=3.0, source=None, init_latents=None,
                         init_strength=0.0, jitter_strength=0, debug=False):
    """sampling.py:78-122.  Returns (latents, n_steps*4) -- the reference's nfe bookkeeping."""
    p0 = next(model.parameters())
    device, dtype = p0.device, p0.dtype
    current_points = source if source is not None else torch.randn(shape, device=device, dtype=dtype)
    if init_latents is None:
        ts = rk4_time_grid(n_steps, dtype=dtype)
        jitter_strength = 0
    else:
        current_points = (1 - init_strength) * current_points + init_strength * init_latents
        ts = rk4_time_grid(n_steps, init_strength, dtype=dtype)
        n_steps = max(1, int(n_steps * (1.0 - init_strength)))

    if isinstance(model, Unet) and not jitter_strength:
        x = current_points.to(device=device, dtype=torch.float32).contiguous().clone()
        if len(ts) > 1:
            cls = cond.get('class_cond') if isinstance(cond, dict) else None
            mask, ones = _mask_flags(cond)
            model.integrate("rk4", x, ts, class_ids=cls, cfg_strength=cfg_strength or 0.0, mask=mask, mask_is_ones=ones)
        return x, n_steps * 4

    ts = ts.to(device)
    t_vec_template = torch.zeros(shape[0], device=device, dtype=dtype)
    v_func = partial(v_func_cfg, model, cond, cfg_strength, t_vec_template)
    for i in range(len(ts) - 1):
        current_points = rk4_step(v_func, current_points, ts[i], ts[i + 1] - ts[i])
        if random.random() < 0.1 and jitter_strength > 0:
            current_points += torch.randn_like(current_points) * jitter_strength * (1 - ts[i])
    return current_points, n_steps * 4


@torch.no_grad()
def euler_sampler(model, shape, sample_N, device=None, cond=None, source=None, eps=1e-3, cfg_strength=0.0):
    """Legacy Euler sampler, train_sd_flowers.py:50-67: x += model(x, 999 t_i, cond)/N on the un-warped grid, nfe = N.
    ``cond`` is a class-id tensor as upstream (wrapped into the dict the live Unet needs) or a cond dict;
    ``source`` replaces the randn start for reproducible runs.  ``cfg_strength`` is an extension (upstream has no
    CFG here; 0 keeps upstream behaviour).  Returns (latents on `device`, nfe)."""
    p0 = next(model.parameters())
    device = p0.device if device is None else torch.device(device)
    if cond is not None and not isinstance(cond, dict):
        cond = {'class_cond': cond}
    x = (source if source is not None else torch.randn(shape, device=device)).to(device=device, dtype=torch.float32).contiguous().clone()
    ts = euler_time_grid(sample_N, eps)
    dt = 1.0 / sample_N
    if isinstance(model, Unet):
        cls = cond.get('class_cond') if cond else None
        mask, ones = _mask_flags(cond)
        model.integrate("euler", x, ts, dt_euler=dt, class_ids=cls, cfg_strength=cfg_strength, mask=mask, mask_is_ones=ones)
        return x, sample_N
    for t in ts.tolist():
        t_vec = torch.ones(shape[0], device=device) * t
        x = x + model(x, t_vec * 999, cond) * dt
    return x, sample_N


@torch.no_grad()
def generate_latents(model, shape, method='rk4', n_steps=50, cond=None, cfg_strength=3.0, device=None, source=None,
                     init_latents=None, init_strength=0.0, debug=False):
    """sampling.py:128-146.  'rk45' is undefined upstream; 'euler' selects the legacy sampler."""
    if method == "rk45":
        raise NameError("generate_latents_rk45 is not defined in the reference either (sampling.py:142-143)")
    if method == "euler":
        # (the legacy sampler has no guidance upstream: generate_latents keeps that; sample_many forwards its cfg_strength itself)
        return euler_sampler(model, shape, n_steps, device=device, cond=cond, source=source)
    return generate_latents_rk4(model, shape, n_steps, cond, cfg_strength, source=source, init_latents=init_latents,
                                init_strength=init_strength)


def _decode_latents(codec, latents, is_midi=False, keep_gray=False, device=None, debug=False):
    """sampling.py:150-166."""
    if device is None:
        try:
            device = next(codec.parameters()).device
        except Exception:
            device = latents.device
    decoded = codec.decode(latents.to(device))
    if is_midi:
        from .metrics import g2rgb
        return g2rgb(decoded, keep_gray=keep_gray)
    return decoded


def decode_latents(codec, latents, is_midi=False, keep_gray=False, device=None, chunk_size=128, debug=False):
    """sampling.py:169-183; chunks stay on the device (upstream bounces every chunk through the CPU)."""
    chunks = [_decode_latents(codec, latents[i:i + chunk_size], is_midi=is_midi, keep_gray=keep_gray, device=device)
              for i in range(0, latents.shape[0], chunk_size)]
    return torch.cat(chunks, dim=0).to(latents.device)


@torch.no_grad()
def sampler(model, codec, method='rk4', batch_size=256, n_steps=100, cond=None, n_classes=0, latent_shape=(4, 16, 16),
            cfg_strength=3.0, is_midi=False, keep_gray=False, device=None, source=None, init_image=None, init_strength=0.0,
            debug=False):
    """sampling.py:186-229: integrate, then decode.  Returns (pred_latents, decoded_pred, nfe).
    Tolerates parameter-less codecs and cond=None, which crash upstream."""
    if device is None:
        device = next(model.parameters()).device
    try:
        codec_device = next(codec.parameters()).device
        assert device == codec_device, f"sampler, device mismatch: device = {device}, but  codec_device {codec_device}"
    except StopIteration:
        pass
    cond = {} if cond is None else cond

    init_latents = None
    if init_image is not None:
        if isinstance(init_image, str):
            raise NameError("init_image as a path is dead upstream (Image is never imported, sampling.py:204)")
        init_tensor = init_image if torch.is_tensor(init_image) else _to_tensor(init_image)
        if init_tensor.dim() == 3:
            init_tensor = init_tensor.unsqueeze(0)
        init_latents = codec.encode(init_tensor.to(device))
        if init_latents.shape[0] == 1 and batch_size > 1:
            init_latents = init_latents.repeat(batch_size, 1, 1, 1)

    shape = (batch_size,) + tuple(latent_shape)
    if source is not None:
        source = source[:batch_size]
    if cond.get('class_cond') is None and n_classes > 0:
        cond['class_cond'] = torch.randint(n_classes, (10,)).repeat(batch_size // 10).to(device)
    elif cond.get('class_cond') is not None:
        cond['class_cond'] = cond['class_cond'][:batch_size]
    if cond.get('mask_cond') is not None:
        cond['mask_cond'] = cond['mask_cond'][:batch_size]

    pred_latents, nfe = generate_latents(model, shape, method, n_steps, cond, cfg_strength, device=device, source=source,
                                         init_latents=init_latents, init_strength=init_strength)
    decoded_pred = decode_latents(codec, pred_latents, is_midi, keep_gray, device=device)
    return pred_latents, decoded_pred, nfe


@torch.no_grad()
def sample_many(model, shape, batches, method="euler", n_steps=64, cfg_strength=0.0, in_flight=2):
    """Throughput mode for callers that generate MANY batches (the 50 k samples of an FID run, evaluate_model / generate_samples loops around
    sampling.py:186-229): ``batches`` is a sequence of ``(cond, source)`` pairs, one per call of ``generate_latents`` the reference would
    make; up to ``in_flight`` of them run at the same time, each on its own stream and its own replica of ``model`` (own activation arena and
    captured graphs, weights copied once).  Trajectories are independent, so results equal the one-at-a-time calls; one trajectory is a chain
    of ~4500 dependent launches with the chip mostly waiting on launch-to-launch latency, and a second chain fills those gaps: measured
    955-975 samples/s against 782-789 for one batch of 64 at a time (tools/inflight_sweep.py, under AMD_DIRECT_DISPATCH=0 --
    ``flocoder_amd.apply_runtime_defaults("sampling")``).  The replicas run the plan without cross-workgroup waits (``set_shared_device``).
    Returns the list of latents in the order of ``batches``."""
    if not isinstance(model, Unet):
        raise TypeError("sample_many drives flocoder_amd.Unet replicas")
    dev = next(model.parameters()).device
    in_flight = max(1, int(in_flight))
    reps = getattr(model, "_replicas", None) or []
    while len(reps) < in_flight - 1:
        reps.append(model.replica())
    model._replicas = reps
    models = [model] + reps[: in_flight - 1]
    was = [m._shared for m in models]
    for m in models:
        m.set_shared_device(True if in_flight > 1 else None)
    streams = [torch.cuda.Stream(dev) for _ in models]
    cur = torch.cuda.current_stream(dev)
    outs = []
    try:
        for st in streams:
            st.wait_stream(cur)
        for i, (cond, source) in enumerate(batches):
            k = i % len(models)
            with torch.cuda.stream(streams[k]):
                if method == "euler":       # generate_latents' euler branch has no guidance (as upstream); euler_sampler's extension does
                    lat, _ = euler_sampler(models[k], shape, n_steps, cond=cond, source=source, cfg_strength=cfg_strength)
                else:
                    lat, _ = generate_latents(models[k], shape, method=method, n_steps=n_steps, cond=cond, cfg_strength=cfg_strength, source=source)
                outs.append(lat)
        for st in streams:
            cur.wait_stream(st)
        for m in models:
            m.check_errors(synchronize=False)
    finally:
        for m, w in zip(models, was):
            m.set_shared_device(w)
    return outs


def _to_tensor(img):
    """PIL image -> float CHW in [0,1] (torchvision.transforms.ToTensor, absent here)."""
    import numpy as np
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a.copy()).permute(2, 0, 1)
    return t.float() / 255.0 if t.dtype == torch.uint8 else t.float()
