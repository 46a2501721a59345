"""Thin wrappers over the library's debug hooks, for kernel-level parity tests (not part of the drop-in surface)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _binding as B


def _nhwc(t: Optional[torch.Tensor]):
    return None if t is None else t.permute(0, 2, 3, 1).contiguous()


def conv_debug(x0: torch.Tensor, w: torch.Tensor, bias=None, x1=None, add=None, *, pad=0, stride=1, upsample=False,
               out_act=False, groups_out=0, tile="auto", repeats=0, precision="fp32"):
    """One implicit-GEMM launch.  NCHW in / NCHW out (converted with torch, test plumbing only).
    Returns (out, stats) with stats = (mean, var) per (b, group) reconstructed from the kernel's partials, or None;
    with repeats > 0 returns the average milliseconds of that many back-to-back launches instead."""
    dev = x0.device
    bsz, c0, hs, ws = x0.shape
    cout, cin, ks, _ = w.shape
    c1 = 0 if x1 is None else x1.shape[1]
    assert cin == c0 + c1
    ho = hs * 2 if upsample else (hs + 2 * pad - ks) // stride + 1
    wo = ws * 2 if upsample else (ws + 2 * pad - ks) // stride + 1
    x0n, x1n, addn = _nhwc(x0.float()), _nhwc(None if x1 is None else x1.float()), _nhwc(None if add is None else add.float())
    out = torch.empty(bsz, ho, wo, cout, device=dev, dtype=torch.float32)
    stats = torch.full((bsz * max(groups_out, 1) * 4096 * 2,), float("nan"), device=dev) if groups_out else None
    T, nt, ms = C.c_int(0), C.c_float(0), C.c_float(0)
    tile_id = B.TILE_AUTO if tile == "auto" else B.TILES[tile]
    wc = w.float().contiguous()
    bc = None if bias is None else bias.float().contiguous()
    B.check(B.lib().fc_debug_set_conv_precision(1 if precision == "bf16x3" else 0))
    try:
        B.check(B.lib().fc_debug_conv(B.ptr(x0n), c0, B.ptr(x1n), c1, B.ptr(wc), B.ptr(bc), B.ptr(addn), B.ptr(out), B.ptr(stats),
                                      groups_out, C.byref(T), C.byref(nt), bsz, hs, ws, cout, ks, pad, stride, int(upsample),
                                      int(out_act), tile_id, int(repeats), C.byref(ms), B.current_stream(dev)))
    finally:
        B.check(B.lib().fc_debug_set_conv_precision(0))
    if repeats:
        return ms.value
    res = out.permute(0, 3, 1, 2).contiguous()
    if not groups_out:
        return res, None
    part = stats[: bsz * groups_out * T.value * 2].view(bsz, groups_out, T.value, 2).double()
    mean = part[..., 0].mean(-1)
    m2 = part[..., 1].sum(-1) + nt.value * ((part[..., 0] - mean[..., None]) ** 2).sum(-1)
    var = m2 / (nt.value * T.value)
    return res, (mean, var)


def fetch_tap(model, name: str, batch: int) -> torch.Tensor:
    """Copy an internal NHWC activation of the model's last forward into a fresh NCHW tensor."""
    p, c, h, w = model.debug_tensor(name)
    dev = next(model.parameters()).device
    t = torch.empty(batch, h, w, c, device=dev, dtype=torch.float32)
    B.check(B.lib().fc_debug_copy(t.data_ptr(), p, t.numel() * 4, B.current_stream(dev)))
    torch.cuda.synchronize(dev)
    return t.permute(0, 3, 1, 2).contiguous()


def ot_pairing(source: torch.Tensor, target: torch.Tensor):
    """Greedy OT pairing on the GPU; returns (perm int64 [B], dist [B,B])."""
    bsz = source.shape[0]
    s = source.reshape(bsz, -1).float().contiguous()
    t = target.reshape(bsz, -1).float().contiguous()
    dist = torch.empty(bsz, bsz, device=s.device, dtype=torch.float32)
    perm = torch.empty(bsz, device=s.device, dtype=torch.int64)
    B.check(B.lib().fc_ot_pairing(B.ptr(s), B.ptr(t), bsz, s.shape[1], B.ptr(dist), B.ptr(perm), B.current_stream(s.device)))
    return perm, dist


def conv_wgrad_debug(x0: torch.Tensor, dy: torch.Tensor, ks: int, x1=None, *, pad=0, stride=1, upsample=False):
    """Weight / bias gradient of one convolution from its NCHW input(s) and NCHW output gradient: (dW [O,I,KH,KW], db [O])."""
    dev = x0.device
    bsz, c0, hs, ws = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cout = dy.shape[1]
    x0n, x1n, dyn = _nhwc(x0.float()), _nhwc(None if x1 is None else x1.float()), _nhwc(dy.float())
    dw = torch.full((cout, c0 + c1, ks, ks), float("nan"), device=dev, dtype=torch.float32)
    db = torch.full((cout,), float("nan"), device=dev, dtype=torch.float32)
    B.check(B.lib().fc_debug_conv_wgrad(B.ptr(x0n), c0, B.ptr(x1n), c1, B.ptr(dyn), cout, bsz, hs, ws, ks, pad, stride, int(upsample),
                                        B.ptr(dw), B.ptr(db), B.current_stream(dev)))
    return dw, db
