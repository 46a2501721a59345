"""Inpainting conditioning: host-side mirror of ``flocoder/inpainting.py``'s ``MaskEncoder`` / ``mask_blending``
(inpainting.py:161-253) over the gfx950 library.  The mask *generators*, ``InpaintingDataset`` and the diagnostics of that
file are data preparation and stay out of scope."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _binding as B


class _Node(nn.Module):
    pass


class _MaskEncoderFunction(torch.autograd.Function):
    """Autograd bridge for MaskEncoder training (train_flow.py:312-318,361-371).  The reference calls the encoder three times per
    step (the batch's masks, all ones, all zeros) before ``loss.backward()``; the native object keeps the activations of its LAST
    forward only, so the backward re-runs the (3 MFLOP) forward of its own call first."""

    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.model = model
        ctx.save_for_backward(x)
        return model._forward_native(x)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        model = ctx.model
        flat = model.backward_native(x, d_out)
        return (None, None, *[flat[off:off + math.prod(shape)].view(shape).clone() for _, shape, off in model._table])


class MaskEncoder(nn.Module):
    """inpainting.py:182-245 with the defaults the flow trainer uses (output_channels=4, shrink_fac=4, mode='pool', sigmoid):
    pixel mask [B,1,H,W] -> [B,4,H/16,W/16]; channel 0 is the 16x average-pooled raw mask, channels 1-3 are learned.
    Same ``state_dict`` keys as upstream (``layers.0.conv1.weight`` ...) and the same default init / RNG order."""

    def __init__(self, output_channels=4, shrink_fac=4, mode='pool', final_act=torch.sigmoid):
        super().__init__()
        if output_channels != 4 or shrink_fac != 4 or mode != 'pool':
            raise NotImplementedError("only the configuration train_flow.py instantiates (MaskEncoder()) is built")
        lib = B.lib()
        h = C.c_void_p()
        B.check(lib.fc_mask_encoder_create(-1, C.byref(h)))
        self._table = []
        for i in range(lib.fc_mask_encoder_param_count(h)):
            name, shape, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int64()
            B.check(lib.fc_mask_encoder_param_info(h, i, C.byref(name), C.byref(shape), C.byref(off)))
            self._table.append((name.value.decode(), tuple(int(s) for s in shape if s), int(off.value)))
        self._flat_numel = int(lib.fc_mask_encoder_param_numel(h))
        lib.fc_mask_encoder_destroy(h)
        for name, shape, _ in self._table:                     # registration order = upstream construction order
            node = self
            *path, leaf = name.split(".")
            for part in path:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            p = nn.Parameter(torch.empty(shape))
            with torch.no_grad():                              # nn.Conv2d defaults, weight then bias
                if len(shape) > 1:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                    fan_in = math.prod(shape[1:])
                else:
                    p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            node.register_parameter(leaf, p)
        self._handle, self._handle_device, self._synced = None, None, None

    def mark_dirty(self) -> None:
        """Re-upload the weights on the next use (for writes that bypass the (data_ptr, _version) key, e.g. ``p.data.copy_``)."""
        self._synced = None

    def _native(self, device):
        lib = B.lib()
        if self._handle is None or self._handle_device != device:
            self._release()
            h = C.c_void_p()
            B.check(lib.fc_mask_encoder_create(device.index or 0, C.byref(h)))
            self._handle, self._handle_device, self._synced = h, device, None
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if ver != self._synced:
            flat = torch.zeros(self._flat_numel, device=device)
            sd = dict(self.named_parameters())
            for name, shape, off in self._table:
                flat[off:off + math.prod(shape)] = sd[name].detach().reshape(-1).to(device)
            B.check(lib.fc_mask_encoder_load_params(self._handle, flat.data_ptr(), flat.numel(), 1, B.current_stream(device)))
            torch.cuda.current_stream(device).synchronize()
            self._synced = ver
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            B.lib().fc_mask_encoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def forward(self, mask_pixels):
        if not mask_pixels.is_cuda:
            raise RuntimeError("flocoder_amd.MaskEncoder runs on MI355X (gfx950) only; there is no CPU path")
        if mask_pixels.dtype in (torch.uint8, torch.int32, torch.int64, torch.bool):      # inpainting.py:236-237
            mask_pixels = mask_pixels.float()
        x = mask_pixels.detach().contiguous().float()
        if x.shape[1] != 1:
            raise ValueError("mask_pixels must have one channel")
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            return _MaskEncoderFunction.apply(self, x, *[self.get_parameter(n) for n, _, _ in self._table])
        with torch.no_grad():
            return self._forward_native(x)

    def _forward_native(self, x):
        bsz, _, h, w = x.shape
        hnd = self._native(x.device)
        B.check(B.lib().fc_mask_encoder_reserve(hnd, bsz, h, w))
        out = torch.empty(bsz, 4, h // 16, w // 16, device=x.device)
        B.check(B.lib().fc_mask_encoder_forward(hnd, B.ptr(x), B.ptr(out), bsz, h, w, B.current_stream(x.device)))
        return out

    def backward_native(self, x, d_out, grads=None, accumulate=False):
        """Parameter gradients (flat, table layout) for d(mask_latents) = ``d_out`` of the forward on ``x`` -- which is re-run first."""
        bsz, _, h, w = x.shape
        out = self._forward_native(x)
        if grads is None:
            grads = torch.zeros(self._flat_numel, device=x.device)
            accumulate = False
        B.check(B.lib().fc_mask_encoder_backward(self._native(x.device), B.ptr(x), B.ptr(out), B.ptr(d_out.contiguous().float()), B.ptr(grads),
                                                 grads.numel(), int(accumulate), bsz, h, w, B.current_stream(x.device)))
        return grads

    def grad_views(self, flat):
        return {name: flat[off:off + math.prod(shape)].view(shape) for name, shape, off in self._table}


def mask_blending(source, mask, noise=None):
    """inpainting.py:250-253: source + mask*(noise - source)."""
    if noise is None:
        noise = torch.randn_like(source)
    if not source.is_cuda:
        raise RuntimeError("flocoder_amd.mask_blending runs on MI355X (gfx950) only; there is no CPU path")
    if torch.is_grad_enabled() and any(t.requires_grad for t in (source, mask, noise)):
        return source + mask * (noise - source)             # under autograd (MaskEncoder training): three device elementwise ops on a latent
    s, m, n = (t.contiguous().float() for t in (source, mask.expand_as(source), noise))
    out = torch.empty_like(s)
    B.check(B.lib().fc_mask_blend(B.ptr(s), B.ptr(m), B.ptr(n), B.ptr(out), s.numel(), B.current_stream(s.device)))
    return out
