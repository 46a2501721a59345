"""CPU restatement of the SD-VAE codec behind flocoder's ``SD_VAE_Wrapper`` (codecs.py:631-663).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Parity status: **PARITY UNPINNED.**  The arithmetic lives in a third-party dependency that is absent from
``/root/reference`` and from this image: ``diffusers`` (``pyproject.toml:34``, unpinned), class
``diffusers.models.AutoencoderKL``, weights ``stabilityai/sd-vae-ft-mse`` (call sites codecs.py:635-636,642,651).  The
reference holds no test, golden vector or fixture for it.  What follows restates the published
architecture of that class for the sd-vae-ft-mse config (block_out_channels [128,256,512,512], layers_per_block 2,
latent_channels 4, norm_num_groups 32, GroupNorm eps 1e-6, SiLU, single-head 512-d mid attention, encoder downsample =
pad (0,1,0,1) + conv3x3 stride 2, decoder upsample = nearest x2 + conv3x3) with the upstream state_dict key names; its
parameter count (83 653 863) matches the published model.  It anchors GPU<->CPU self-consistency on seeded random
weights, nothing more, until a local copy of diffusers + weights can pin it.

The wrapper semantics that ARE the reference's own (and are restated exactly): ``encode`` returns
``latent_dist.mean`` (channels 0-3 of quant_conv's output), ``decode`` returns ``.sample``, and neither applies the
0.18215 scaling factor (codecs.py:639-652).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
BLOCK_OUT = (128, 256, 512, 512)
GROUPS, EPS = 32, 1e-6


def shapes(in_channels: int = 3, latent_channels: int = 4, block_out=BLOCK_OUT, layers_per_block: int = 2) -> Dict[str, tuple]:
    """Upstream state_dict layout (name -> shape) for AutoencoderKL with this config."""
    s: Dict[str, tuple] = {}

    def conv(n, o, i, k):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def norm(n, c):
        s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)

    def lin(n, o, i):
        s[n + ".weight"] = (o, i); s[n + ".bias"] = (o,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", o, i, 3); norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", o, i, 1)

    def mid(n, c):
        resnet(n + ".resnets.0", c, c)
        a = n + ".attentions.0"
        norm(a + ".group_norm", c); lin(a + ".to_q", c, c); lin(a + ".to_k", c, c); lin(a + ".to_v", c, c); lin(a + ".to_out.0", c, c)
        resnet(n + ".resnets.1", c, c)

    L = len(block_out)
    conv("encoder.conv_in", block_out[0], in_channels, 3)
    ci = block_out[0]
    for i, co in enumerate(block_out):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if i < L - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        ci = co
    mid("encoder.mid_block", block_out[-1])
    norm("encoder.conv_norm_out", block_out[-1]); conv("encoder.conv_out", 2 * latent_channels, block_out[-1], 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    conv("post_quant_conv", latent_channels, latent_channels, 1)
    rev = block_out[::-1]
    conv("decoder.conv_in", rev[0], latent_channels, 3)
    mid("decoder.mid_block", rev[0])
    ci = rev[0]
    for i, co in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if i < L - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        ci = co
    norm("decoder.conv_norm_out", rev[-1]); conv("decoder.conv_out", in_channels, rev[-1], 3)
    return s


def _conv(sd, n, x, padding=0, stride=1):
    return F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=stride, padding=padding)


def _gn_silu(sd, n, x):
    return F.silu(F.group_norm(x, GROUPS, sd[n + ".weight"], sd[n + ".bias"], eps=EPS))


def resnet(sd: SD, n: str, x: Tensor) -> Tensor:
    h = _conv(sd, n + ".conv1", _gn_silu(sd, n + ".norm1", x), padding=1)
    h = _conv(sd, n + ".conv2", _gn_silu(sd, n + ".norm2", h), padding=1)
    if (n + ".conv_shortcut.weight") in sd:
        x = _conv(sd, n + ".conv_shortcut", x)
    return x + h                                            # output_scale_factor = 1


def attention(sd: SD, n: str, x: Tensor) -> Tensor:
    """Single head over all h*w tokens, d = C: softmax(q k^T / sqrt(C)) v, projected, plus the residual."""
    b, c, h, w = x.shape
    t = F.group_norm(x, GROUPS, sd[n + ".group_norm.weight"], sd[n + ".group_norm.bias"], eps=EPS)
    t = t.reshape(b, c, h * w).transpose(1, 2)             # b n c
    q = F.linear(t, sd[n + ".to_q.weight"], sd[n + ".to_q.bias"])
    k = F.linear(t, sd[n + ".to_k.weight"], sd[n + ".to_k.bias"])
    v = F.linear(t, sd[n + ".to_v.weight"], sd[n + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
    o = F.linear(a @ v, sd[n + ".to_out.0.weight"], sd[n + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def _mid(sd, n, x):
    x = resnet(sd, n + ".resnets.0", x)
    x = attention(sd, n + ".attentions.0", x)
    return resnet(sd, n + ".resnets.1", x)


def encode_mean(sd: SD, x: Tensor) -> Tensor:
    """vae.encode(x).latent_dist.mean (codecs.py:642): no sampling, no scaling factor."""
    n_blocks = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.down_blocks."))
    h = _conv(sd, "encoder.conv_in", x, padding=1)
    for i in range(n_blocks):
        j = 0
        while f"encoder.down_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h); j += 1
        d = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if d + ".weight" in sd:
            h = _conv(sd, d, F.pad(h, (0, 1, 0, 1)), stride=2)
    h = _mid(sd, "encoder.mid_block", h)
    h = _conv(sd, "encoder.conv_out", _gn_silu(sd, "encoder.conv_norm_out", h), padding=1)
    moments = _conv(sd, "quant_conv", h)
    return moments[:, : moments.shape[1] // 2]


def decode(sd: SD, z: Tensor) -> Tensor:
    """vae.decode(z).sample (codecs.py:651)."""
    n_blocks = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.up_blocks."))
    h = _conv(sd, "post_quant_conv", z)
    h = _conv(sd, "decoder.conv_in", h, padding=1)
    h = _mid(sd, "decoder.mid_block", h)
    for i in range(n_blocks):
        j = 0
        while f"decoder.up_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h); j += 1
        u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        if u + ".weight" in sd:
            h = _conv(sd, u, F.interpolate(h, scale_factor=2.0, mode="nearest"), padding=1)
    return _conv(sd, "decoder.conv_out", _gn_silu(sd, "decoder.conv_norm_out", h), padding=1)
