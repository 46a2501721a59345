"""Codec plugin API: host-side mirror of ``flocoder/codecs.py``'s factory and codec wrappers (codecs.py:578-741).

``setup_codec(config, device, no_natten=False, load_checkpoint=True, eval=True)`` returns an object with the protocol the
flow path consumes: ``encode(x) -> z``, ``decode(z, orig_size=None, noise_strength=0.0) -> x``,
``forward(x, noise_strength, minval, get_stats)``, ``in_channels``, ``parameters()``.  The SD codec runs on the gfx950
library (``fc_vae_*``) and the VQVAE's encode / decode on ``fc_vqvae_*``; there is no CPU path for either.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _binding as B
from .general import ldcfg


class SimpleResizeAE(nn.Module):
    """codecs.py:578-619 -- interpolation 'codec' (encode: resize to latent_shape, extra channels = channel mean)."""

    def __init__(self, in_channels=3, latent_shape=(4, 16, 16), mode='bicubic'):
        super().__init__()
        self.in_channels, self.latent_shape, self.orig_shape, self.mode = in_channels, latent_shape, None, mode

    def encode(self, x):
        self.orig_shape = x.shape[1:]
        if self.latent_shape is None or self.orig_shape == self.latent_shape:
            return x
        c, h, w = self.latent_shape
        small = F.interpolate(x, size=(h, w), mode=self.mode, align_corners=False)
        if c == x.shape[-3]:
            return small
        return torch.cat([small, small.mean(dim=1, keepdim=True).repeat(1, c - x.shape[-3], 1, 1)], dim=1)

    def decode(self, z, orig_shape=None, noise_strength=0.0):
        target = orig_shape if orig_shape is not None else self.orig_shape
        if self.latent_shape is None or target is None or target == self.latent_shape:
            return z
        return F.interpolate(z[:, :3], size=(target[-2], target[-1]), mode=self.mode, align_corners=False)

    def forward(self, x, noise_strength=0.0, minval=0, get_stats=False):
        recon = self.decode(self.encode(x))
        return (recon, 0.0, {'codebook_mean_dist': 0.0, 'codebook_max_dist': 0.0}) if get_stats else (recon, 0.0)


class NoOpAE(SimpleResizeAE):
    """codecs.py:621-627.  As upstream, the parent constructor resets latent_shape to (4,16,16), so 'noop' resizes
     -- kept, because checkpoints trained against that behaviour expect it."""

    def __init__(self):
        self.latent_shape = None
        super().__init__()


# legacy (pre-0.14 diffusers) attention parameter names still present in the published sd-vae-ft-mse checkpoint
_LEGACY = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


def vae_param_table():
    """(name, shape, offset) of the native AutoencoderKL parameter table (no GPU needed)."""
    lib = B.lib()
    h = C.c_void_p()
    B.check(lib.fc_vae_create(-1, C.byref(h)))
    try:
        out = []
        for i in range(lib.fc_vae_param_count(h)):
            name, shape, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int64()
            B.check(lib.fc_vae_param_info(h, i, C.byref(name), C.byref(shape), C.byref(off)))
            out.append((name.value.decode(), tuple(int(s) for s in shape if s), int(off.value)))
        return out, int(lib.fc_vae_param_numel(h))
    finally:
        lib.fc_vae_destroy(h)


class _Node(nn.Module):
    pass


_PRECISIONS = {"fp32": 0, "bf16x3": 1}


class SD_VAE_Wrapper(nn.Module):
    """codecs.py:631-663 over the native AutoencoderKL.  Parameters live under ``self.vae.*`` with the upstream key names, so a
    ``state_dict`` saved from the reference's wrapper loads here unchanged.

    The reference calls ``AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-mse")`` (a network fetch).  Here weights come from a
    LOCAL copy: ``weights`` = a state_dict, a ``.safetensors`` / ``.pt`` file, or a directory holding
    ``diffusion_pytorch_model.safetensors`` (also taken from ``$FLOCODER_SD_VAE_PATH``); ``weights="random"`` draws seeded random
    weights (benchmarks / tests).  Without any of these it raises FileNotFoundError -- it never downloads."""

    def __init__(self, pretrained_model_name="stabilityai/sd-vae-ft-mse", weights=None, seed: int = 0):
        super().__init__()
        self.in_channels = 3
        self.pretrained_model_name = pretrained_model_name
        self._table, self._flat_numel = vae_param_table()
        self.vae = _Node()
        for name, shape, _ in self._table:
            node = self.vae
            *path, leaf = name.split(".")
            for part in path:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            node.register_parameter(leaf, nn.Parameter(torch.empty(shape, dtype=torch.float32), requires_grad=False))
        self._handle, self._handle_device, self._synced = None, None, None
        if weights is None:
            weights = os.environ.get("FLOCODER_SD_VAE_PATH")
        if weights is None:
            raise FileNotFoundError(
                f"SD_VAE_Wrapper: no local weights for '{pretrained_model_name}'. Pass weights=<dir|file|state_dict> or set "
                "FLOCODER_SD_VAE_PATH (the reference downloads them; this build never touches the network).")
        if isinstance(weights, str) and weights == "random":
            self._init_random(seed)
        else:
            self.load_vae_state_dict(weights if isinstance(weights, dict) else _read_weights(weights))

    # ---- weights
    def _init_random(self, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.vae.named_parameters():
                leaf = name.rsplit(".", 1)[1]
                if p.dim() == 1:
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if leaf == "weight" else 0.05 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.prod(p.shape[1:])) ** 0.5)

    def load_vae_state_dict(self, sd: Dict[str, torch.Tensor]):
        sd = dict(sd)
        for k in list(sd):
            nk = k[4:] if k.startswith("vae.") else k
            for old, new in _LEGACY.items():
                nk = nk.replace(old, new)
            if nk != k:
                sd[nk] = sd.pop(k)
        own = dict(self.vae.named_parameters())
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"SD-VAE weights are missing {len(missing)} tensors, e.g. {missing[:3]}")
        with torch.no_grad():
            for k, p in own.items():
                p.copy_(sd[k].reshape(p.shape).to(p.dtype))      # legacy attention weights are [C,C,1,1]
        self._synced = None

    # ---- native object
    def mark_dirty(self) -> None:
        """Re-upload the weights on the next use (for writes that bypass the (data_ptr, _version) key, e.g. ``p.data.copy_``)."""
        self._synced = None

    def _native(self, device):
        lib = B.lib()
        if self._handle is None or self._handle_device != device:
            self._release()
            h = C.c_void_p()
            B.check(lib.fc_vae_create(device.index or 0, C.byref(h)))
            self._handle, self._handle_device, self._synced = h, device, None
        B.check(lib.fc_vae_set_precision(self._handle, _PRECISIONS[getattr(self, "_precision", "fp32")]))
        ver = tuple((p.data_ptr(), p._version) for p in self.vae.parameters())
        if ver != self._synced:
            flat = torch.zeros(self._flat_numel, dtype=torch.float32, device=device)
            sd = dict(self.vae.named_parameters())
            for name, shape, off in self._table:
                flat[off:off + math.prod(shape)] = sd[name].detach().reshape(-1).to(device)
            B.check(lib.fc_vae_load_params(self._handle, flat.data_ptr(), flat.numel(), 1, B.current_stream(device)))
            torch.cuda.current_stream(device).synchronize()
            self._synced = ver
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            B.lib().fc_vae_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def set_precision(self, mode: str = "fp32") -> None:
        """Arithmetic of the codec's convolutions: "fp32" (default, exact fp32 on the matrix pipe -- what parity tests and the headline use)
        or "bf16x3" (every operand as bf16 hi + lo, three bf16 MFMAs per product, fp32 accumulation: ~1e-5 relative per layer; an opt-in for
        callers that decode many images, ``fc_vae_set_precision``)."""
        if mode not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self._precision = mode

    @staticmethod
    def _need_gpu(t):
        if not t.is_cuda:
            raise RuntimeError("flocoder_amd.SD_VAE_Wrapper runs on MI355X (gfx950) only; there is no CPU path")

    # ---- codec protocol
    @torch.no_grad()
    def encode(self, x):
        """vae.encode(x).latent_dist.mean.detach() (codecs.py:639-642): deterministic, no 0.18215 scaling."""
        self._need_gpu(x)
        bsz, ch, h, w = x.shape
        if ch != 3:
            raise ValueError("SD-VAE expects 3-channel images")
        x = x.contiguous().float()
        hnd = self._native(x.device)
        B.check(B.lib().fc_vae_reserve_encode(hnd, bsz, h, w))
        out = torch.empty(bsz, 4, h // 8, w // 8, device=x.device, dtype=torch.float32)
        B.check(B.lib().fc_vae_encode(hnd, B.ptr(x), B.ptr(out), bsz, h, w, B.current_stream(x.device)))
        return out

    @torch.no_grad()
    def decode(self, z, orig_size=None, noise_strength=0.0):
        """vae.decode(z).sample (codecs.py:649-652)."""
        self._need_gpu(z)
        bsz, ch, h, w = z.shape
        if ch != 4:
            raise ValueError("SD-VAE latents have 4 channels")
        z = z.contiguous().float()
        hnd = self._native(z.device)
        B.check(B.lib().fc_vae_reserve_decode(hnd, bsz, h, w))
        out = torch.empty(bsz, 3, 8 * h, 8 * w, device=z.device, dtype=torch.float32)
        B.check(B.lib().fc_vae_decode(hnd, B.ptr(z), B.ptr(out), bsz, h, w, B.current_stream(z.device)))
        return out

    def forward(self, x, noise_strength=0.0, minval=0, get_stats=False):
        """codecs.py:657-663."""
        self.last_min, self.last_max = x.min(), x.max()
        recon = self.decode(self.encode(x))
        return (recon, 0.0, {'codebook_mean_dist': 0.0, 'codebook_max_dist': 0.0}) if get_stats else (recon, 0.0)

    def flops_per_sample(self, decode=True) -> float:
        return float(B.lib().fc_vae_flops_per_sample(self._handle, int(decode))) if self._handle else 0.0

    def plan_ops(self, decode=True):
        """(kernel family, reference module, algorithmic FLOPs per sample) of every launch of the current decode / encode plan."""
        lib, h = B.lib(), self._handle
        out = []
        for i in range(lib.fc_vae_plan_launches(h, int(decode)) if h else 0):
            k, m, f = C.c_char_p(), C.c_char_p(), C.c_double()
            B.check(lib.fc_vae_op_info(h, int(decode), i, C.byref(k), C.byref(m), C.byref(f)))
            out.append((k.value.decode(), m.value.decode(), f.value))
        return out

    def plan_kernels(self, decode=True):
        return [k for k, _, _ in self.plan_ops(decode)]

    def profile_ops(self, inp: torch.Tensor, out: torch.Tensor, decode=True, repeats: int = 5):
        """Per-launch device milliseconds of the decode (or encode) plan for the batch ``inp`` -> ``out`` (bench.py's live
        roofline measurement for the codec).  Run decode()/encode() at this batch first so the plan exists."""
        lib, h = B.lib(), self._handle
        ops = self.plan_ops(decode)
        ms = (C.c_float * len(ops))()
        B.check(lib.fc_vae_profile_ops(h, int(decode), B.ptr(inp.contiguous()), B.ptr(out), inp.shape[0], repeats, ms, len(ops),
                                       B.current_stream(inp.device)))
        rows = []
        for i, (k, m, f) in enumerate(ops):
            bp, bf = C.c_double(), C.c_double()
            B.check(lib.fc_vae_op_bytes(h, int(decode), i, C.byref(bp), C.byref(bf)))
            rows.append(dict(kernel=k, module=m, flops_per_sample=f, ms=float(ms[i]), rows=inp.shape[0], bytes=bp.value * inp.shape[0] + bf.value))
        return rows


class _NoiseInjectionParams(nn.Module):
    """Parameter holder for NoiseInjection (codecs.py:217-241): a no-op at noise_strength 0, which is all the flow path uses;
    the tensors exist so reference checkpoints round-trip through ``state_dict``."""

    def __init__(self, channels):
        super().__init__()
        self.to_noise_scale = nn.Conv2d(channels, channels, 1)
        self.to_noise_bias = nn.Conv2d(channels, channels, 1)
        nn.init.zeros_(self.to_noise_scale.weight)
        nn.init.zeros_(self.to_noise_bias.weight)


class _Codebook(nn.Module):
    def __init__(self, size, dim):
        super().__init__()
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("cluster_size", torch.zeros(1, size))
        self.register_buffer("embed_avg", torch.zeros(1, size, dim))
        self.register_buffer("embed", torch.zeros(1, size, dim))


class _VQLayer(nn.Module):
    def __init__(self, size, dim):
        super().__init__()
        self._codebook = _Codebook(size, dim)


class ResidualVQ(nn.Module):
    """Inference form of ``vector_quantize_pytorch.ResidualVQ`` as VQVAE builds it (codecs.py:456-467), holding the buffers under the
    package's state_dict names (``layers.{i}._codebook.{initted,cluster_size,embed_avg,embed}``) so trained checkpoints load.
    THIRD-PARTY ALGORITHM, PARITY UNPINNED (the package is absent offline): per level the nearest codeword of the running
    residual, output = their sum; k-means initialisation, EMA updates and the training losses are not built -- calling it with
    codebooks that were never initialised (``initted`` false) or in training mode raises."""

    def __init__(self, dim, codebook_size, num_quantizers, **unused):
        super().__init__()
        self.dim, self.codebook_size, self.num_quantizers = dim, codebook_size, num_quantizers
        self.layers = nn.ModuleList([_VQLayer(codebook_size, dim) for _ in range(num_quantizers)])

    @property
    def codebooks(self):
        return torch.stack([l._codebook.embed[0] for l in self.layers])

    def _check(self, t):
        if self.training:
            raise NotImplementedError("ResidualVQ: only the inference form is built (codec training is out of scope)")
        if not all(bool(l._codebook.initted) for l in self.layers):
            raise NotImplementedError("ResidualVQ: codebooks are uninitialised (k-means init is not built); load a trained checkpoint")
        if not t.is_cuda:
            raise RuntimeError("flocoder_amd.ResidualVQ runs on MI355X (gfx950) only; there is no CPU path")

    def quantize_nchw(self, z):
        self._check(z)
        bsz, d, h, w = z.shape
        z = z.contiguous().float()
        cb = self.codebooks.to(z.device).contiguous()
        zq = torch.empty_like(z)
        idx = torch.empty(bsz * h * w, self.num_quantizers, dtype=torch.int64, device=z.device)
        B.check(B.lib().fc_rvq_quantize(B.ptr(z), B.ptr(cb), B.ptr(zq), B.ptr(idx), bsz, d, h * w, self.codebook_size, self.num_quantizers,
                                        B.current_stream(z.device)))
        return zq, idx, torch.zeros(1, self.num_quantizers, device=z.device)

    def forward(self, x):
        """x [N, dim] (as VQVAE.quantize passes it) or [B, n, dim] -> (quantized, indices [..., levels], losses [1, levels])."""
        flat = x.reshape(-1, x.shape[-1])
        zq, idx, loss = self.quantize_nchw(flat.t().contiguous().view(1, x.shape[-1], flat.shape[0], 1))
        return zq.view(x.shape[-1], flat.shape[0]).t().reshape(x.shape), idx.view(*x.shape[:-1], self.num_quantizers), loss


class VQVAE(nn.Module):
    """codecs.py:395-574 -- encode / decode on the gfx950 library (``fc_vqvae_*``), NATTEN-less, eval mode.

    Parameters carry the reference's state_dict names (``encoder.0.conv1.weight`` ... ``decoder.layers.0.q_proj.weight`` ...), so
    ``load_state_dict(ckpt['model_state_dict'], strict=False)`` takes a reference checkpoint as is.  ``quantize`` / ``forward`` run the
    inference form of ResidualVQ (third party, parity unpinned -- see ``ResidualVQ`` above); ``decode`` with a non-zero
    ``noise_strength`` (training-time NoiseInjection) raises NotImplementedError."""

    def __init__(self, in_channels=3, hidden_channels=256, num_downsamples=3, vq_num_embeddings=512, internal_dim=256,
                 codebook_levels=3, vq_embedding_dim=4, commitment_weight=0.25, use_checkpoint=False, no_natten=False,
                 encoder_nonlocal=False, decoder_nonlocal=True, natten_layout=0):
        """``natten_layout``: 0 = no NATTENBlocks (what the reference builds when the natten package is missing or ``no_natten``),
        1 / 2 = with them (needed to load a checkpoint TRAINED with NATTEN): 1 reads the 7x7 windows over image rows x
        columns per head, 2 reproduces what natten >= 0.20 makes of the reference's [B, heads, H, W, d] call (include/flocoder_amd.h,
        fc_vqvae_create_ex).  Third-party arithmetic, parity unpinned."""
        super().__init__()
        if encoder_nonlocal:
            raise NotImplementedError("VQVAE(encoder_nonlocal=True) is not built (no reference config uses it)")
        natten_layout = 0 if no_natten else int(natten_layout)
        self.in_channels, self.num_downsamples = in_channels, num_downsamples
        self.codebook_levels, self.vq_num_embeddings = codebook_levels, vq_num_embeddings
        self.vq_embedding_dim, self.indices, self.info = vq_embedding_dim, None, None
        self._cfg = (in_channels, hidden_channels, num_downsamples, internal_dim, vq_embedding_dim, int(bool(decoder_nonlocal)), natten_layout)
        self.natten_layout = natten_layout
        lib = B.lib()
        h = C.c_void_p()
        B.check(lib.fc_vqvae_create_ex(*self._cfg, -1, C.byref(h)))
        try:
            self._table = []
            for i in range(lib.fc_vqvae_param_count(h)):
                name, shape, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int64()
                B.check(lib.fc_vqvae_param_info(h, i, C.byref(name), C.byref(shape), C.byref(off)))
                self._table.append((name.value.decode(), tuple(int(s) for s in shape if s), int(off.value)))
            self._flat_numel = int(lib.fc_vqvae_param_numel(h))
        finally:
            lib.fc_vqvae_destroy(h)
        for name, shape, _ in self._table:
            node = self
            *path, leaf = name.split(".")
            for part in path:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            p = torch.empty(shape, dtype=torch.float32)
            if leaf == "gamma":
                p.zero_()                                                 # NATTENBlock's residual gate starts closed (codecs.py:105)
                node.register_parameter(leaf, nn.Parameter(p, requires_grad=False))
                continue
            if name.endswith((".attn.qkv.weight", ".attn.proj.weight")):
                p.normal_(0.0, 0.02)                                      # codecs.py:108-109
                node.register_parameter(leaf, nn.Parameter(p, requires_grad=False))
                continue
            if leaf == "weight" and len(shape) > 1:                       # nn.Conv2d default init (kaiming_uniform, a=sqrt(5))
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            elif leaf == "weight":
                p.fill_(1.0)                                              # GroupNorm
            else:
                wname = name[:-4] + "weight"
                wshape = next(s for n, s, _ in self._table if n == wname)
                bound = 1.0 / math.sqrt(math.prod(wshape[1:])) if len(wshape) > 1 else 0.0
                p.uniform_(-bound, bound) if bound else p.zero_()
            node.register_parameter(leaf, nn.Parameter(p, requires_grad=False))
        # NoiseInjection tensors of the reference's Decoder (codecs.py:259,283-300): present in checkpoints, unused at strength 0
        i0 = 1 if decoder_nonlocal else 0
        cur = hidden_channels * 2 ** (num_downsamples - 1)
        self.decoder.layers.add_module(str(i0 + 4), _NoiseInjectionParams(cur))
        i = i0 + 6
        for lvl in range(num_downsamples - 1, -1, -1):
            co = hidden_channels * 2 ** max(0, lvl - 1) if lvl else hidden_channels
            self.decoder.layers.add_module(str(i + 3), _NoiseInjectionParams(cur))
            self.decoder.layers.add_module(str(i + 5), _NoiseInjectionParams(co))
            cur, i = co, i + 7
        self.decoder.layers.add_module(str(i), _NoiseInjectionParams(cur))
        self.decoder.layers.add_module(str(i + 3), _NoiseInjectionParams(64))
        self.vq = ResidualVQ(dim=vq_embedding_dim, codebook_size=vq_num_embeddings, num_quantizers=codebook_levels)
        self.register_buffer('codebook_usage', torch.zeros(codebook_levels, vq_num_embeddings))
        self.usage_count = 0
        self._handle, self._handle_device, self._synced = None, None, None

    def mark_dirty(self) -> None:
        """Re-upload the weights on the next use (for writes that bypass the (data_ptr, _version) key, e.g. ``p.data.copy_``)."""
        self._synced = None

    def set_precision(self, mode: str = "fp32") -> None:
        """"fp32" (default) or "bf16x3" -- see ``SD_VAE_Wrapper.set_precision`` (``fc_vqvae_set_precision``)."""
        if mode not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self._precision = mode

    def _native(self, device):
        lib = B.lib()
        if self._handle is None or self._handle_device != device:
            self._release()
            h = C.c_void_p()
            B.check(lib.fc_vqvae_create_ex(*self._cfg, device.index or 0, C.byref(h)))
            self._handle, self._handle_device, self._synced = h, device, None
        B.check(lib.fc_vqvae_set_precision(self._handle, _PRECISIONS[getattr(self, "_precision", "fp32")]))
        sd = dict(self.named_parameters())
        ver = tuple((sd[n].data_ptr(), sd[n]._version) for n, _, _ in self._table)
        if ver != self._synced:
            flat = torch.zeros(self._flat_numel, dtype=torch.float32, device=device)
            for name, shape, off in self._table:
                flat[off:off + math.prod(shape)] = sd[name].detach().reshape(-1).to(device)
            B.check(lib.fc_vqvae_load_params(self._handle, flat.data_ptr(), flat.numel(), 1, B.current_stream(device)))
            torch.cuda.current_stream(device).synchronize()
            self._synced = ver
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            B.lib().fc_vqvae_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    @torch.no_grad()
    def encode(self, x, debug=False):
        """z = self.encoder(x) (codecs.py:492-502): [B,in_channels,H,W] -> [B,vq_embedding_dim,H>>nd,W>>nd], pre-quantisation."""
        if not x.is_cuda:
            raise RuntimeError("flocoder_amd.VQVAE runs on MI355X (gfx950) only; there is no CPU path")
        bsz, ch, h, w = x.shape
        if ch != self.in_channels:
            raise ValueError(f"VQVAE expects {self.in_channels}-channel input, got {ch}")
        x = x.contiguous().float()
        hnd = self._native(x.device)
        B.check(B.lib().fc_vqvae_reserve_encode(hnd, bsz, h, w))
        nd = self.num_downsamples
        out = torch.empty(bsz, self.vq_embedding_dim, h >> nd, w >> nd, device=x.device, dtype=torch.float32)
        B.check(B.lib().fc_vqvae_encode(hnd, B.ptr(x), B.ptr(out), bsz, h, w, B.current_stream(x.device)))
        return out

    @torch.no_grad()
    def decode(self, z_q, noise_strength=0.0):
        """self.decoder(z_q, noise_strength) (codecs.py:523-525) at noise_strength 0."""
        if noise_strength:
            raise NotImplementedError("VQVAE.decode: noise_strength != 0 (training-time NoiseInjection) is not built")
        if not z_q.is_cuda:
            raise RuntimeError("flocoder_amd.VQVAE runs on MI355X (gfx950) only; there is no CPU path")
        bsz, ch, h, w = z_q.shape
        if ch != self.vq_embedding_dim:
            raise ValueError(f"VQVAE latents have {self.vq_embedding_dim} channels, got {ch}")
        z = z_q.contiguous().float()
        hnd = self._native(z.device)
        B.check(B.lib().fc_vqvae_reserve_decode(hnd, bsz, h, w))
        nd = self.num_downsamples
        out = torch.empty(bsz, self.in_channels, h << nd, w << nd, device=z.device, dtype=torch.float32)
        B.check(B.lib().fc_vqvae_decode(hnd, B.ptr(z), B.ptr(out), bsz, h, w, B.current_stream(z.device)))
        return out

    def quantize(self, z, debug=False):
        """codecs.py:504-521: z [B,C,h,w] -> (z_q [B,C,h,w], commit_loss); the indices of the last call stay in ``self.indices``."""
        z_q, self.indices, commit_loss = self.vq.quantize_nchw(z)
        return z_q, commit_loss

    @torch.no_grad()
    def calc_distance_stats(self, z, z_q):
        """codecs.py:527-536 (diagnostic)."""
        z_flat = z.reshape(-1, z.shape[1])
        distances = torch.norm(z_flat.unsqueeze(1) - self.vq.codebooks[0].to(z.device), dim=-1)
        return {'codebook_mean_dist': distances.mean().item(), 'codebook_max_dist': distances.max().item()}

    def forward(self, x, noise_strength=None, minval=0, get_stats=False):
        """codecs.py:545-574 in eval mode: encode -> quantize -> decode; returns (recon, commit_loss.mean()[, stats])."""
        if self.training:
            raise NotImplementedError("VQVAE.forward: training mode (codec training) is out of scope; call .eval()")
        z = self.encode(x)
        if noise_strength is None:
            noise_strength = 0.0
        if self.info is None:
            self.info = z.shape
        z_q, commit_loss = self.quantize(z)
        x_recon = self.decode(z_q, noise_strength=noise_strength)
        if get_stats:
            return x_recon, commit_loss.mean(), self.calc_distance_stats(z, z_q)
        return x_recon, commit_loss.mean()

    def flops_per_sample(self, decode=True) -> float:
        return float(B.lib().fc_vqvae_flops_per_sample(self._handle, int(decode))) if self._handle else 0.0

    def profile_ops(self, inp: torch.Tensor, out: torch.Tensor, decode=True, repeats: int = 5):
        """Per-launch device milliseconds of the decode (or encode) plan for the batch ``inp`` -> ``out`` with each launch's kernel
        family, algorithmic FLOPs per sample and algorithmic HBM bytes (bench.py's config-5 leg).  Run decode()/encode() at this
        batch first so the plan exists."""
        lib, h = B.lib(), self._handle
        n = lib.fc_vqvae_plan_launches(h, int(decode)) if h else 0
        ms = (C.c_float * max(n, 1))()
        B.check(lib.fc_vqvae_profile_ops(h, int(decode), B.ptr(inp.contiguous()), B.ptr(out), inp.shape[0], repeats, ms, n,
                                         B.current_stream(inp.device)))
        rows = []
        for i in range(n):
            k, m, f, bp, bf = C.c_char_p(), C.c_char_p(), C.c_double(), C.c_double(), C.c_double()
            B.check(lib.fc_vqvae_op_info(h, int(decode), i, C.byref(k), C.byref(m), C.byref(f), C.byref(bp), C.byref(bf)))
            rows.append(dict(kernel=k.value.decode(), module=m.value.decode(), flops_per_sample=f.value, ms=float(ms[i]), rows=inp.shape[0],
                             bytes=bp.value * inp.shape[0] + bf.value))
        return rows


def _read_weights(path: str) -> Dict[str, torch.Tensor]:
    path = os.path.expanduser(path)
    if os.path.isdir(path):
        for cand in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin", "vae.safetensors", "vae.pt"):
            if os.path.exists(os.path.join(path, cand)):
                path = os.path.join(path, cand)
                break
        else:
            raise FileNotFoundError(f"no SD-VAE weight file under {path}")
    if not os.path.exists(path):
        raise FileNotFoundError(f"SD-VAE weights not found: {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd.get("state_dict", sd.get("model_state_dict", sd))


def setup_codec(config, device, no_natten=False, load_checkpoint=True, eval=True):
    """codecs.py:668-741 -- codec factory keyed on ``config.codec.choice``."""
    choice = config.codec.choice if 'choice' in config.codec else None
    if choice is None or choice == "noop":
        print("Using NoOpAE")
        codec = NoOpAE().eval().to(device)
    elif choice == "resize":
        print("Using SimpleResizeAE")
        codec = SimpleResizeAE(latent_shape=tuple(config.codec.get('latent_shape', (4, 16, 16)))).eval().to(device)
    elif choice == "sd":
        print("Loading SD VAE via SD_VAE_Wrapper")
        weights = config.codec.get('sd_vae_path') or os.environ.get("FLOCODER_SD_VAE_PATH")
        codec = SD_VAE_Wrapper(pretrained_model_name="stabilityai/sd-vae-ft-mse", weights=weights).eval().to(device)
        if 'image_size' in config and config.image_size % 8 != 0:
            print(f"Warning: SD VAE works best with image sizes divisible by 8. Current size: {config.image_size}")
    elif choice == "vqgan_plus":
        raise NotImplementedError("codec 'vqgan_plus' is codec-training territory and outside the flow hot path (SURVEY.md 2)")
    else:
        print("Loading VQVAE model")
        checkpoint = None
        if load_checkpoint:
            if 'vqgan_checkpoint' in config:
                path = config.vqgan_checkpoint
            elif 'codec' in config and 'checkpoint' in config.codec:
                path = config.codec.checkpoint
            else:
                raise ValueError("Could not find codec checkpoint path in config")
            if path.lower() != "sd" and not os.path.exists(path):
                raise FileNotFoundError(f"Codec checkpoint file {path} not found.")
            print(f"Loading codec checkpoint from {path}")
            try:
                checkpoint = torch.load(path, map_location=device, weights_only=True)
            except Exception:
                checkpoint = torch.load(path, map_location=device, weights_only=False)
        # A checkpoint trained WITH the natten package carries NATTENBlock weights ('...attn.qkv.weight', codecs.py:93-145).  Upstream
        # such a model silently loses its attention when the package is missing; here the blocks are native
        # (fc_vqvae_create_ex), so they are built whenever the checkpoint has them -- unless no_natten asks for upstream's fallback.
        natten_layout = 0
        if checkpoint is not None and any(k.endswith('.attn.qkv.weight') for k in checkpoint['model_state_dict']):
            if no_natten:
                print("Warning: the checkpoint holds NATTEN attention blocks and no_natten=True drops them (as upstream without the package)")
            else:
                natten_layout = int(ldcfg(config, 'natten_layout', 2, verbose=False) or 2)
                print(f"Codec checkpoint holds NATTEN blocks: building them natively (natten_layout={natten_layout}; 2 = what natten >= 0.20, "
                      "the reference's pinned minimum, computes from its call; set codec.natten_layout=1 for windows over image rows x columns)")
        codec = VQVAE(                                                    # codecs.py:707-718
            in_channels=ldcfg(config, 'in_channels', 3, verbose=False),
            hidden_channels=ldcfg(config, 'hidden_channels', 256, verbose=False),
            num_downsamples=ldcfg(config, 'num_downsamples', 3, verbose=False),
            internal_dim=ldcfg(config, 'internal_dim', 256, verbose=False),
            vq_embedding_dim=ldcfg(config, 'vq_embedding_dim', 4, verbose=False),
            codebook_levels=ldcfg(config, 'codebook_levels', 4, verbose=False),
            vq_num_embeddings=ldcfg(config, 'vq_num_embeddings', 512, verbose=False),
            commitment_weight=ldcfg(config, 'commitment_weight', 0.5, verbose=False),
            use_checkpoint=not config.get('no_grad_ckpt', False),
            no_natten=no_natten, natten_layout=natten_layout,
        ).to(device)
        if checkpoint is not None:
            codec.load_state_dict(checkpoint['model_state_dict'], strict=False)   # strict=False: vq.* training buffers may be absent
    if eval:
        codec = codec.eval()
    print("Codec model ready")
    return codec
