"""CPU restatement of flocoder's VQVAE codec encode / decode path (codecs.py:150-574), NATTEN-less.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity status: PINNED for ``encode`` and ``decode`` --
``tools/make_golden.py`` runs the reference's own ``VQVAE`` (with a stand-in for the absent ``vector_quantize_pytorch``
package, which neither method calls) on seeded weights and ``tests/test_oracle_golden.py`` holds this file to the result
(fixture g9).  ``quantize`` = ``vector_quantize_pytorch.ResidualVQ`` (``>=1.22.4``, pyproject.toml:41; call sites codecs.py:456-467,
504-521) is third-party and absent: ``residual_vq`` / ``quantize`` below restate its published INFERENCE behaviour -- PARITY
UNPINNED.  NATTEN is absent here, so in the pinned fixtures ``attention='natten'`` blocks have no attention (codecs.py:170-175,
SURVEY Q23); state dicts that DO carry ``attn.qkv`` weights run ``natten_block`` / ``na2d`` below -- a restatement of the package's
published neighbourhood-attention definition, PARITY UNPINNED;
eval mode: dropout is the identity and NoiseInjection is a no-op at noise_strength 0 (codecs.py:229-232).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def gn_groups(proposed: int, channels: int) -> int:
    """codecs.py:34-44."""
    if channels % proposed == 0:
        return proposed
    for c in range(proposed, channels):
        if channels % c == 0:
            return c
    return 1


def _conv(sd, n, x, padding=0, stride=1):
    return F.conv2d(x, sd[n + ".weight"], sd.get(n + ".bias"), stride=stride, padding=padding)


def _gn(sd, n, x, proposed, eps=1e-5):
    c = x.shape[1]
    return F.group_norm(x, gn_groups(proposed, c), sd[n + ".weight"], sd[n + ".bias"], eps=eps)


def attn_block(sd: SD, n: str, x: Tensor) -> Tensor:
    """AttnBlock, codecs.py:53-89: GroupNorm(32, eps 1e-6), 1x1 q/k/v, softmax(q^T k c^-1/2) over keys, proj_out, residual."""
    b, c, h, w = x.shape
    t = F.group_norm(x, gn_groups(32, c), sd[n + ".norm.norm.weight"], sd[n + ".norm.norm.bias"], eps=1e-6)
    q, k, v = _conv(sd, n + ".q", t), _conv(sd, n + ".k", t), _conv(sd, n + ".v", t)
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    a = torch.softmax(torch.bmm(q, k) * int(c) ** -0.5, dim=2)
    o = torch.bmm(v.reshape(b, c, h * w), a.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(sd, n + ".proj_out", o)


def na2d(q: Tensor, k: Tensor, v: Tensor, kernel_size: int = 7, scale: float | None = None) -> Tensor:
    """2-D neighbourhood attention on [B, X, Y, heads, d] tensors: query (x, y) attends to the kernel_size^2 keys of the window whose
    start along each axis is clamp(pos - kernel_size // 2, 0, L - kernel_size) (it shifts inwards at the borders: never padded), with
    softmax(q . k * scale) and scale = d^-0.5 by default.  Restates the PUBLISHED definition of ``natten.functional.na2d`` (dilation 1,
    non-causal, no relative position bias; natten >= 0.20.1, pyproject.toml:40).  PARITY UNPINNED: the package is absent."""
    B, X, Y, H, D = q.shape
    if X < kernel_size or Y < kernel_size:
        raise ValueError("neighbourhood larger than the attended axes")
    scale = D ** -0.5 if scale is None else scale
    r = kernel_size // 2
    sx = (torch.arange(X) - r).clamp(0, X - kernel_size)
    sy = (torch.arange(Y) - r).clamp(0, Y - kernel_size)
    ix = sx[:, None] + torch.arange(kernel_size)[None, :]                   # [X, K]
    iy = sy[:, None] + torch.arange(kernel_size)[None, :]                   # [Y, K]
    kw = k[:, ix][:, :, :, iy]                                              # [B, X, K, Y, K, H, D]
    vw = v[:, ix][:, :, :, iy]
    kw = kw.permute(0, 1, 3, 5, 2, 4, 6).reshape(B, X, Y, H, kernel_size * kernel_size, D)
    vw = vw.permute(0, 1, 3, 5, 2, 4, 6).reshape(B, X, Y, H, kernel_size * kernel_size, D)
    a = torch.softmax(torch.einsum("bxyhd,bxyhwd->bxyhw", q, kw) * scale, dim=-1)
    return torch.einsum("bxyhw,bxyhwd->bxyhd", a, vw)


def natten_block(sd: SD, n: str, x: Tensor, layout: int = 1, heads: int = 8, kernel_size: int = 7) -> Tensor:
    """NATTENBlock._forward, codecs.py:116-140: GroupNorm(gn_groups(8, C)) -> qkv Linear (no bias) -> neighbourhood attention ->
    proj Linear (no bias) -> identity + gamma * (.).  The reference builds q / k / v as [B, heads, H, W, d] (codecs.py:122-124).
    ``layout`` 1 reads that as intended (7x7 windows over image rows x columns, per head); ``layout`` 2 is what a natten >= 0.20
    ``na2d`` -- which documents [B, X, Y, heads, d] -- computes from the very same tensors: windows over (head index, image row), the
    image column playing the part of the head.  Which one a trained checkpoint needs can only be settled against the package."""
    B, C, H, W = x.shape
    d = C // heads
    t = F.group_norm(x, gn_groups(8, C), sd[n + ".norm.weight"], sd[n + ".norm.bias"]).permute(0, 2, 3, 1)      # B H W C
    qkv = F.linear(t, sd[n + ".qkv.weight"]).reshape(B, H, W, 3, heads, d).permute(3, 0, 4, 1, 2, 5)            # 3 B heads H W d
    q, k, v = qkv[0], qkv[1], qkv[2]
    if layout == 1:
        o = na2d(q.permute(0, 2, 3, 1, 4), k.permute(0, 2, 3, 1, 4), v.permute(0, 2, 3, 1, 4), kernel_size).permute(0, 3, 1, 2, 4)
    else:
        o = na2d(q, k, v, kernel_size)                                                                          # dims taken as [B, X, Y, heads, d]
    o = o.permute(0, 2, 3, 1, 4).reshape(B, H, W, C)                                                            # codecs.py:137
    o = F.linear(o, sd[n + ".proj.weight"]).permute(0, 3, 1, 2)
    return x + o * sd[n + ".gamma"]


def res_block(sd: SD, n: str, x: Tensor, stride: int = 1, natten_layout: int = 1) -> Tensor:
    """EncDecResidualBlock._forward, codecs.py:178-209 (eval): silu(norm1(conv1 x)) [-> attn] -> norm2(conv2 .) + identity -> silu."""
    out = F.silu(_gn(sd, n + ".norm1", _conv(sd, n + ".conv1", x, padding=1, stride=stride), 8))
    if (n + ".attn.q.weight") in sd:                       # attention='full'
        out = attn_block(sd, n + ".attn", out)
    elif (n + ".attn.qkv.weight") in sd:                   # attention='natten' with the package present (else: no attention at all)
        out = natten_block(sd, n + ".attn", out, natten_layout)
    out = _gn(sd, n + ".norm2", _conv(sd, n + ".conv2", out, padding=1), 8)
    if (n + ".downsample.0.weight") in sd:
        x = _gn(sd, n + ".downsample.1", _conv(sd, n + ".downsample.0", x, stride=stride), 8)
    return F.silu(out + x)


def rope(x: Tensor, scale: float = math.log(10000.0)) -> Tensor:
    """SpatialNonLocalAttention._apply_rope, codecs.py:351-369, on [b, hw, c]."""
    b, hw, c = x.shape
    if c % 2 != 0:
        x = F.pad(x, (0, 1)); c = x.shape[-1]
    pos = torch.arange(hw).unsqueeze(1)
    inv_freq = torch.exp(-torch.arange(0, c // 2) * scale / (c // 2))
    pe = pos * inv_freq.unsqueeze(0)
    ps, pc = torch.sin(pe).to(x.dtype), torch.cos(pe).to(x.dtype)
    xe, xo = x[..., 0::2], x[..., 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = xe * pc - xo * ps
    out[..., 1::2] = xo * pc + xe * ps
    return out


def spatial_nonlocal_attention(sd: SD, n: str, x: Tensor) -> Tensor:
    """codecs.py:371-383."""
    b, c, h, w = x.shape
    q = rope(_conv(sd, n + ".q_proj", x).reshape(b, -1, h * w).permute(0, 2, 1))
    k = rope(_conv(sd, n + ".k_proj", x).reshape(b, -1, h * w).permute(0, 2, 1))
    v = _conv(sd, n + ".v_proj", x).reshape(b, c, h * w).permute(0, 2, 1)
    a = F.softmax(torch.bmm(q, k.transpose(1, 2)) * (q.size(-1) ** -0.5), dim=-1)
    o = torch.bmm(a, v).permute(0, 2, 1).reshape(b, c, h, w)
    return x + _conv(sd, n + ".out_proj", o)


def n_downsamples(sd: SD) -> int:
    blocks = sorted({int(k.split(".")[1]) for k in sd if k.startswith("encoder.") and ".conv1.weight" in k})
    return (len(blocks) - 1) // 2


def encode(sd: SD, x: Tensor, natten_layout: int = 1) -> Tensor:
    """VQVAE.encode = self.encoder(x), codecs.py:414-443,492-502: 2 residual blocks per downsample (the first with stride 2), one
    more to internal_dim, a 1x1 conv, then the compress stack conv1x1 -> GroupNorm -> SiLU -> conv3x3."""
    nd = n_downsamples(sd)
    h = x
    for i in range(nd):
        h = res_block(sd, f"encoder.{2 * i}", h, stride=2, natten_layout=natten_layout)
        h = res_block(sd, f"encoder.{2 * i + 1}", h, natten_layout=natten_layout)
    h = res_block(sd, f"encoder.{2 * nd}", h, natten_layout=natten_layout)
    h = _conv(sd, f"encoder.{2 * nd + 1}", h)
    h = _conv(sd, f"encoder.{2 * nd + 2}", h)
    h = F.silu(_gn(sd, f"encoder.{2 * nd + 3}", h, 2))
    return _conv(sd, f"encoder.{2 * nd + 5}", h, padding=1)


def decode(sd: SD, z: Tensor, natten_layout: int = 1) -> Tensor:
    """VQVAE.decode -> Decoder.forward at noise_strength 0, codecs.py:245-316,523-525."""
    nd = n_downsamples(sd)
    L = "decoder.layers."
    i = 0
    h = z
    if (L + "0.q_proj.weight") in sd:
        h = spatial_nonlocal_attention(sd, L + "0", h); i = 1
    h = _conv(sd, L + str(i), h)
    emb = z.shape[1]
    h = F.silu(_gn(sd, L + str(i + 1), h, emb))
    h = _conv(sd, L + str(i + 3), h)
    h = res_block(sd, L + str(i + 5), h, natten_layout=natten_layout)   # i+4 is a NoiseInjection
    i += 6
    for _ in range(nd):
        h = F.pixel_shuffle(F.silu(_conv(sd, L + str(i), h, padding=1)), 2)      # conv, SiLU, PixelShuffle(2), NoiseInjection
        h = res_block(sd, L + str(i + 4), h, natten_layout=natten_layout)
        h = res_block(sd, L + str(i + 6), h)                                       # i+5 is a NoiseInjection
        i += 7
    h = F.silu(_conv(sd, L + str(i + 1), h, padding=1))   # i is a NoiseInjection
    return _conv(sd, L + str(i + 4), h, padding=1)        # i+2 SiLU, i+3 NoiseInjection


def residual_vq(codebooks: Tensor, x: Tensor):
    """ResidualVQ.forward in eval mode on flattened vectors x [N, D] with codebooks [L, K, D] (vector_quantize_pytorch: per layer
    ``embed_ind = argmax(-cdist(residual, embed))``, ``quantize = embed[embed_ind]``, ``residual -= quantize``; outputs the sum of the
    layers' codewords, indices [N, L] and -- nothing is learnt in eval -- zero losses [1, L]).  PARITY UNPINNED (third party)."""
    residual, out, idx = x.clone(), torch.zeros_like(x), []
    for cb in codebooks:
        d = ((residual[:, None, :] - cb[None, :, :]) ** 2).sum(-1)
        i = d.argmin(dim=1)
        q = cb[i]
        residual = residual - q
        out = out + q
        idx.append(i)
    return out, torch.stack(idx, dim=-1), torch.zeros(1, codebooks.shape[0])


def quantize(sd: SD, z: Tensor):
    """VQVAE.quantize, codecs.py:504-521: permute to [N, C], ResidualVQ, back to NCHW.  Codebooks = ``vq.layers.{i}._codebook.embed``
    ([1, K, D] each).  Returns (z_q, commit_loss [1, L], indices [N, L])."""
    L = len({k.split(".")[2] for k in sd if k.startswith("vq.layers.") and k.endswith("._codebook.embed")})
    cbs = torch.stack([sd[f"vq.layers.{i}._codebook.embed"].reshape(-1, z.shape[1]) for i in range(L)])
    zp = z.permute(0, 2, 3, 1)
    zq, idx, loss = residual_vq(cbs, zp.reshape(-1, z.shape[1]))
    return zq.view(zp.shape).permute(0, 3, 1, 2).contiguous(), loss, idx
