"""Velocity U-Net: host-side mirror of ``flocoder.unet.Unet`` (reference unet.py:164-377) over the gfx950 library.

Same constructor, same ``forward(x, time, cond)`` protocol, same ``state_dict`` key names and shapes (so the
reference's checkpoints load with ``load_state_dict``), same default initialisation *and RNG consumption order*
(``torch.manual_seed(s); Unet(...)`` gives the reference's weights).  The arithmetic happens in
``libflocoder_amd.so``; this class owns the parameters and hands them over.  There is no CPU path: calling it
with CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _binding as B

# construction order of the reference's Unet.__init__ (unet.py:185-286); state_dict order differs (ups before mid)
_CTOR_ORDER = ["init_conv", "time_mlp", "class_cond_mlp", "mask_fusion_conv", "down_mask_fusions", "up_mask_fusions",
               "downs", "mid_block1", "mid_attn", "mid_block2", "ups", "final_res_block", "final_conv"]
_STATE_ORDER = ["init_conv", "time_mlp", "class_cond_mlp", "mask_fusion_conv", "down_mask_fusions", "up_mask_fusions",
                "downs", "ups", "mid_block1", "mid_attn", "mid_block2", "final_res_block", "final_conv"]


def _make_config(dim, dim_mults, channels, groups, n_classes, mask_cond) -> B.fc_unet_config:
    if len(dim_mults) > 8:
        raise ValueError("at most 8 resolution levels")
    cfg = B.fc_unet_config(dim=int(dim), channels=int(channels), n_levels=len(dim_mults), groups=int(groups),
                           n_classes=max(0, int(n_classes)), mask_cond=int(bool(mask_cond)))
    for i, m in enumerate(dim_mults):
        cfg.dim_mults[i] = int(m)
    return cfg


def param_table(cfg: B.fc_unet_config) -> List[Tuple[str, Tuple[int, ...], int]]:
    """(name, shape, offset into the flat padded vector) as the library lays parameters out.  Needs no GPU."""
    lib = B.lib()
    h = C.c_void_p()
    B.check(lib.fc_unet_create(C.byref(cfg), -1, C.byref(h)))
    try:
        out = []
        for i in range(lib.fc_unet_param_count(h)):
            name, shape, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int64()
            B.check(lib.fc_unet_param_info(h, i, C.byref(name), C.byref(shape), C.byref(off)))
            out.append((name.value.decode(), tuple(int(s) for s in shape if s), int(off.value)))
        return out
    finally:
        lib.fc_unet_destroy(h)


class _Node(nn.Module):
    """Parameter container; attribute names reproduce the reference's module tree."""


class _UnetFunction(torch.autograd.Function):
    """Autograd bridge: forward and backward both run in the library; parameters receive ``.grad`` as torch expects, and so do
    ``x`` and the mask when they require it (the inpainting step reaches the MaskEncoder through both, train_flow.py:146-147)."""

    @staticmethod
    def forward(ctx, model, x, time, cls, mask, *params):
        ctx.model, ctx.cls = model, cls
        ctx.save_for_backward(x, time, mask if mask is not None else x.new_empty(0))
        ctx.has_mask = mask is not None
        out = model._forward_native(x, time, cls, mask, train=True)
        ctx.serial = model.arena_serial()
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, time, mask = ctx.saved_tensors
        model = ctx.model
        mask = mask if ctx.has_mask else None
        need_dx, need_dm = ctx.needs_input_grad[1], ctx.has_mask and ctx.needs_input_grad[4]
        if model.arena_serial() != ctx.serial:
            # the library keeps ONE activation arena per model and something wrote it since this graph's forward (a second
            # micro-batch, a validation / sampler call, a re-plan): bring this forward's activations back before differentiating
            model._forward_native(x, time, ctx.cls, mask, train=True)
        flat, dx, dm = model.backward_native(x, time, ctx.cls, d_out, mask=mask, want_dx=need_dx, want_dmask=need_dm)
        grads = []
        for name, shape, off in model._table:
            if ctx.cls is None and name.startswith("class_cond_mlp."):
                grads.append(None)                                # unused this step: p.grad stays None, as in the reference
            elif mask is None and (name.startswith("mask_fusion_conv.") or "_mask_fusions." in name):
                grads.append(None)
            else:
                grads.append(flat[off:off + math.prod(shape)].view(shape).clone())
        return (None, dx, None, None, dm, *grads)


class Unet(nn.Module):
    def __init__(self, dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=4, n_classes=10, mask_cond=False,
                 use_checkpoint=False):
        super().__init__()
        self.use_checkpoint = use_checkpoint      # accepted for signature parity; activations are never stored
        self.channels = channels
        self.out_dim = channels
        self.class_condition = n_classes > 0
        self.dim, self.dim_mults = int(dim), tuple(int(m) for m in dim_mults)
        self._cfg = _make_config(dim, self.dim_mults, channels, resnet_block_groups, n_classes, mask_cond)
        table = param_table(self._cfg)
        self._table = table
        self._flat_numel = max(off + int(math.prod(shape)) for _, shape, off in table)
        self._flat_numel = (self._flat_numel + 3) // 4 * 4

        # registration in state_dict order, initialisation in constructor order (RNG parity with the reference)
        by_top: Dict[str, List[Tuple[str, Tuple[int, ...]]]] = {}
        for name, shape, _ in table:
            by_top.setdefault(name.split(".")[0], []).append((name, shape))
        for top in _STATE_ORDER:
            for name, shape in by_top.get(top, []):
                self._register(name, shape)
        with torch.no_grad():
            for top in _CTOR_ORDER:
                for name, shape in by_top.get(top, []):
                    self._init(name, self.get_parameter(name), by_top[top])

        self._handle: Optional[C.c_void_p] = None
        self._handle_device: Optional[torch.device] = None
        self._synced_version = None
        self._shared = None          # None: decide per call (see _device_is_shared); True / False: set_shared_device

    # ------------------------------------------------------------------ parameters
    def _register(self, name: str, shape: Tuple[int, ...]) -> None:
        node = self
        *path, leaf = name.split(".")
        for part in path:
            if not hasattr(node, part):
                node.add_module(part, _Node())
            node = getattr(node, part)
        node.register_parameter(leaf, nn.Parameter(torch.empty(shape, dtype=torch.float32)))

    @staticmethod
    def _init(name: str, p: nn.Parameter, siblings) -> None:
        """nn.Conv2d / nn.Linear / nn.Embedding / nn.GroupNorm defaults, drawn in the reference's order."""
        leaf = name.rsplit(".", 1)[1]
        if p.dim() == 1 and leaf == "weight":
            p.fill_(1.0)                                        # GroupNorm gain
        elif p.dim() == 1:
            wname = name[: -len("bias")] + "weight"
            wshape = next(s for n, s in siblings if n == wname)
            if len(wshape) == 1:
                p.zero_()                                       # GroupNorm bias
            else:
                bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
                p.uniform_(-bound, bound)                       # Conv2d / Linear bias
        elif name == "class_cond_mlp.0.weight":
            p.normal_(0.0, 1.0)                                 # nn.Embedding
        else:
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))         # Conv2d / Linear weight

    def _flat_params(self, device) -> torch.Tensor:
        flat = torch.zeros(self._flat_numel, dtype=torch.float32, device=device)
        sd = dict(self.named_parameters())
        for name, shape, off in self._table:
            flat[off:off + math.prod(shape)] = sd[name].detach().reshape(-1)
        return flat

    def _version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def mark_dirty(self) -> None:
        """Force a re-upload of the parameters on the next use.  Needed after writes the (data_ptr, _version) key cannot see:
        ``param.data.copy_(...)`` changes neither (the reference's EMA swaps weights that way, train_flow.py:56-71)."""
        self._synced_version = None

    # ------------------------------------------------------------------ native object
    def _native(self, device: torch.device):
        lib = B.lib()
        if self._handle is None or self._handle_device != device:
            self._release()
            h = C.c_void_p()
            B.check(lib.fc_unet_create(C.byref(self._cfg), device.index or 0, C.byref(h)))
            self._handle, self._handle_device, self._synced_version = h, device, None
            half = self.dim // 2    # frequency table exactly as torch computes it (unet.py:26-27)
            fr = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).contiguous()
            B.check(lib.fc_unet_set_time_freqs(h, fr.numpy().ctypes.data_as(C.POINTER(C.c_float)), half))
        B.check(lib.fc_unet_set_shared(self._handle, int(self._device_is_shared(device))))
        B.check(lib.fc_unet_set_grad_buckets(self._handle, int(getattr(self, "_grad_buckets", False))))
        ver = self._version()
        if ver != self._synced_version:
            flat = self._flat_params(device)
            B.check(lib.fc_unet_load_params(self._handle, flat.data_ptr(), flat.numel(), 1, B.current_stream(device)))
            torch.cuda.current_stream(device).synchronize()      # `flat` dies when this frame returns
            self._synced_version = ver
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            B.lib().fc_unet_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def arena_serial(self) -> int:
        """Counter of writes to the library's activation arena (fc_unet_arena_serial)."""
        return int(B.lib().fc_unet_arena_serial(self._handle)) if self._handle else 0

    def reserve(self, rows: int, height: int, width: int, device=None) -> None:
        """Build the launch plan / activation arena for up to `rows` U-Net rows (a CFG sampler needs 2x batch)."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        B.check(B.lib().fc_unet_reserve(self._native(device), rows, height, width))

    @property
    def flops_per_sample(self) -> float:
        return float(B.lib().fc_unet_flops_per_sample(self._handle)) if self._handle else 0.0

    @property
    def chains(self):
        """(number of concurrent row-range chains, rows per chain) of the current plan."""
        rows = C.c_int(0)
        n = B.lib().fc_unet_chains(self._handle, C.byref(rows)) if self._handle else 0
        return n, rows.value

    def replica(self) -> "Unet":
        """A second model object with the same architecture and (copied) weights on the same device: its own native handle, activation
        arena and captured graphs, so that it can run beside this one on another stream (``sampling.sample_many``)."""
        c = self._cfg
        twin = Unet(self.dim, self.dim_mults, channels=self.channels, resnet_block_groups=int(c.groups), n_classes=int(c.n_classes),
                    mask_cond=bool(c.mask_cond), use_checkpoint=self.use_checkpoint)
        twin.load_state_dict(self.state_dict())
        twin.train(self.training)
        twin = twin.to(next(self.parameters()).device)
        if self._handle:          # same reservation -> same tiles -> bit-identical results (the plan is built for the reserved row count)
            rows, h, w = C.c_int(0), C.c_int(0), C.c_int(0)
            B.check(B.lib().fc_unet_reserved(self._handle, C.byref(rows), C.byref(h), C.byref(w)))
            if rows.value > 0:
                twin.reserve(rows.value, h.value, w.value)
        return twin

    # ------------------------------------------------------------------ device sharing (fc_unet_set_shared)
    def set_shared_device(self, shared: Optional[bool]) -> None:
        """Tell the library whether this model's GPU work runs beside other work it is not ordered against (a second replica meant to
        overlap, the collectives of a training job, another process on the same GPU).  ``True`` selects the plan without
        cross-workgroup waits (same results to fp32 rounding, one more launch per Block); ``False`` insists on the exclusive plan;
        ``None`` (default) decides per call: shared when a process group with more than one rank is live or the caller works on a
        non-default stream, exclusive otherwise."""
        self._shared = shared

    def _device_is_shared(self, device) -> bool:
        if self._shared is not None:
            return bool(self._shared)
        import os
        if os.environ.get("FLOCODER_AMD_SHARED_DEVICE") in ("0", "1"):
            return os.environ["FLOCODER_AMD_SHARED_DEVICE"] == "1"
        d = torch.distributed
        if d.is_available() and d.is_initialized() and d.get_world_size() > 1 and self.training:
            return True          # gradient collectives (RCCL kernels) run beside this model's launches
        return torch.cuda.current_stream(device) != torch.cuda.default_stream(device)

    @property
    def meeting_launches(self) -> int:
        """Launches of the current plan whose workgroups wait for each other (0 on a shared device)."""
        return int(B.lib().fc_unet_meeting_launches(self._handle)) if self._handle else 0

    def check_errors(self, synchronize: bool = True) -> None:
        """Raise RuntimeError if a cross-workgroup wait of a fused Block tail ever timed out on this model (its samples are NaN and
        every later call fails too, until the plan is rebuilt).  With ``synchronize`` the current stream is waited for first, so
        the answer covers everything queued so far."""
        if self._handle:
            B.check(B.lib().fc_unet_check(self._handle, B.current_stream(self._handle_device), int(synchronize)))

    def fused_tail_errors(self) -> int:
        """Timed-out waits of the fused Block tails since the plan was built (must be 0; synchronises)."""
        if not self._handle:
            return 0
        n = C.c_int(0)
        B.check(B.lib().fc_unet_fused_tail_errors(self._handle, C.byref(n)))
        return n.value

    @property
    def launches_per_forward(self) -> int:
        return int(B.lib().fc_unet_plan_launches(self._handle)) if self._handle else 0

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _split_cond(cond):
        if cond is None:
            return None, None
        if not isinstance(cond, dict):
            # the reference dies here too (warnings.DeprecationWarning does not exist)
            raise AttributeError("Non-dict cond signals are dead in the reference; use cond={'class_cond': ids}")
        return cond.get("class_cond"), cond.get("mask_cond")

    def check_class_ids(self, cls: Optional[torch.Tensor]) -> None:
        """nn.Embedding raises IndexError for an id outside [0, n_classes) (unet.py:205,313); the kernels would silently treat such a
        row as unconditional, so the host mirror raises like the reference.  One small host sync per call."""
        if cls is None or cls.numel() == 0:
            return
        lo, hi = (int(v) for v in torch.stack((cls.min(), cls.max())).tolist())
        if lo < 0 or hi >= self._cfg.n_classes:
            raise IndexError(f"class_cond ids must lie in [0, {self._cfg.n_classes}); got min {lo}, max {hi}")

    def forward(self, x: torch.Tensor, time: torch.Tensor, cond=None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("flocoder_amd.Unet runs on MI355X (gfx950) only; there is no CPU path "
                               "(the CPU restatement under oracle/ is test infrastructure).")
        dev = x.device
        bsz, ch, h, w = x.shape
        if ch != self.channels:
            raise ValueError(f"expected {self.channels} input channels, got {ch}")
        cls, mask = self._split_cond(cond)
        x = x.contiguous().float()
        time = time.to(device=dev, dtype=torch.float32).contiguous()
        if time.shape != (bsz,):
            raise ValueError("time must have shape [batch]")
        if cls is not None and not self.class_condition:
            cls = None                                            # hasattr(self,'class_cond_mlp') is False, unet.py:315
        if cls is not None:
            cls = cls.to(device=dev, dtype=torch.int64).contiguous()
            if cls.shape != (bsz,):
                raise ValueError("class_cond must have shape [batch]")
            self.check_class_ids(cls)
        if mask is not None and not self._cfg.mask_cond:
            mask = None                                           # hasattr(self,'mask_fusion_conv') is False, unet.py:298
        if mask is not None:
            mask = mask.to(device=dev, dtype=torch.float32).contiguous()
            if mask.shape != x.shape:
                raise ValueError("mask_cond must have the shape of x (unet.py:302)")
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            # loss.backward() support (train_flow.py:358-371): gradients come from the library's backward plan
            params = [self.get_parameter(n) for n, _, _ in self._table]
            return _UnetFunction.apply(self, x, time, cls, mask, *params)
        return self._forward_native(x, time, cls, mask, train=False)

    def _forward_native(self, x, time, cls, mask, train: bool) -> torch.Tensor:
        dev = x.device
        bsz, _, h, w = x.shape
        ones = 0
        if mask is not None:
            ones = int(torch.allclose(mask, torch.ones_like(mask)))   # unet.py:301 (one host sync, as upstream)
        hnd = self._native(dev)
        B.check((B.lib().fc_unet_train_reserve if train else B.lib().fc_unet_reserve)(hnd, bsz, h, w))
        out = torch.empty_like(x)
        B.check(B.lib().fc_unet_forward(hnd, B.ptr(x), B.ptr(time), B.ptr(cls), B.ptr(mask), ones, B.ptr(out), bsz, h, w,
                                        B.current_stream(dev)))
        return out

    def set_grad_buckets(self, on: bool) -> None:
        """Build the backward plan in its two-bucket form (``fc_unet_set_grad_buckets``): data-parallel trainers, so that the all-reduce of
        the late layers' gradients can start in the middle of the backward."""
        self._grad_buckets = bool(on)

    def grad_buckets(self) -> Tuple[int, int]:
        """(number of gradient buckets of the current backward plan, flat offset where the early bucket starts): after part 0 of
        ``backward_native`` the range [offset, numel) -- final_*, mid_*, ups.* -- is final, part 1 completes [0, offset)."""
        off = C.c_int64(0)
        n = B.lib().fc_unet_grad_buckets(self._handle, C.byref(off)) if self._handle else 0
        return int(n), int(off.value)

    def backward_native(self, x, time, cls, d_out, grads: Optional[torch.Tensor] = None, mask=None, want_dx=False, want_dmask=False,
                        parts: Tuple[int, int] = (0, 1), dx=None, dm=None):
        """Parameter gradients of the LAST training forward (same x / time / class ids / mask) for d(out) = ``d_out``: a flat fp32
        vector in the library's table layout (``grad_views`` splits it), plus d(x) / d(mask) on request.  Returns
        ``(flat, dx | None, dmask | None)``.  train_flow.py:371 loss.backward().  ``parts`` = (first, last) of the two halves of the
        backward plan (``fc_unet_backward_parts``): (0, 0) stops behind mid_block1 with the late-layer gradients final, (1, 1) runs the
        rest -- a data-parallel trainer all-reduces the first bucket in between."""
        dev = x.device
        bsz, _, h, w = x.shape
        if grads is None:
            grads = torch.empty(self._flat_numel, dtype=torch.float32, device=dev)
        ones = int(torch.allclose(mask, torch.ones_like(mask))) if mask is not None else 0
        if dx is None:
            dx = torch.empty_like(x) if want_dx else None
        if dm is None:
            dm = torch.empty_like(x) if (want_dmask and mask is not None) else None
        B.check(B.lib().fc_unet_backward_parts(self._native(dev), B.ptr(x), B.ptr(time), B.ptr(cls), B.ptr(mask), ones, B.ptr(d_out.contiguous()),
                                               B.ptr(grads), grads.numel(), B.ptr(dx), B.ptr(dm), bsz, h, w, int(parts[0]), int(parts[1]),
                                               B.current_stream(dev)))
        return grads, dx, dm

    def grad_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {name: flat[off:off + math.prod(shape)].view(shape) for name, shape, off in self._table}

    def param_range(self, *prefixes) -> Tuple[int, int]:
        """[lo, hi) inside the flat table of the (contiguous) parameters whose names start with one of ``prefixes``."""
        names = [(off, off + (math.prod(shape) + 3) // 4 * 4) for name, shape, off in self._table if name.startswith(tuple(prefixes))]
        return (min(a for a, _ in names), max(b for _, b in names)) if names else (0, 0)

    def class_param_range(self) -> Tuple[int, int]:
        """class_cond_mlp.*: no gradient when a step runs without conditioning."""
        return self.param_range("class_cond_mlp.")

    def adopt_flat(self, flat: torch.Tensor) -> None:
        """Make every parameter a view into ``flat`` (table layout) so that an optimiser working on the flat vector updates the
        module in place; ``sync_flat`` then hands the new values to the library without a gather."""
        with torch.no_grad():
            for name, shape, off in self._table:
                p = self.get_parameter(name)
                flat[off:off + math.prod(shape)].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + math.prod(shape)].view(shape)
        self._flat = flat

    def sync_flat(self) -> None:
        flat = self._flat
        hnd = self._native(flat.device) if self._handle is None else self._handle
        B.check(B.lib().fc_unet_load_params(hnd, flat.data_ptr(), flat.numel(), 1, B.current_stream(flat.device)))
        self._synced_version = self._version()

    # ------------------------------------------------------------------ integrators (used by flocoder_amd.sampling)
    def integrate(self, method: str, x: torch.Tensor, ts: torch.Tensor, *, dt_euler: float = 0.0, t_scale: float = 999.0,
                  class_ids: Optional[torch.Tensor] = None, cfg_strength: float = 0.0, mask: Optional[torch.Tensor] = None,
                  mask_is_ones: bool = False, check: bool = True) -> torch.Tensor:
        """Integrate ``x`` in place along the fp32 grid ``ts`` with the hipGraph-captured step; returns ``x``.  With ``check`` (default)
        the call waits for the trajectory when the plan contains cross-workgroup waits and raises if one timed out -- a caller never
        receives samples from a plan whose residency assumption broke.  ``check=False`` keeps the call asynchronous; the error then
        surfaces at the next call on the model or at ``check_errors()``."""
        if not x.is_cuda:
            raise RuntimeError("flocoder_amd integrators run on MI355X (gfx950) only")
        dev = x.device
        bsz, ch, h, w = x.shape
        if not x.is_contiguous() or x.dtype != torch.float32:
            raise ValueError("x must be a contiguous fp32 tensor (it is updated in place)")
        code = {"euler": B.FC_METHOD_EULER, "rk4": B.FC_METHOD_RK4}[method]
        if class_ids is not None and not self.class_condition:
            class_ids = None
        if class_ids is not None:
            class_ids = class_ids.to(device=dev, dtype=torch.int64).contiguous()
            if class_ids.shape != (bsz,):
                raise ValueError("class ids must have shape [batch]")
            self.check_class_ids(class_ids)
        if mask is not None and not self._cfg.mask_cond:
            mask = None
        if mask is not None:
            mask = mask.to(device=dev, dtype=torch.float32).contiguous()
            if mask.shape != x.shape:
                raise ValueError("mask_cond must have the shape of x")
        rows = bsz * (2 if (class_ids is not None and cfg_strength) else 1)
        hnd = self._native(dev)
        B.check(B.lib().fc_unet_reserve(hnd, rows, h, w))
        ts_host = ts.detach().to("cpu", torch.float32).contiguous()
        B.check(B.lib().fc_unet_integrate(hnd, code, B.ptr(x), bsz, h, w, ts_host.numpy().ctypes.data_as(C.POINTER(C.c_float)),
                                          ts_host.numel(), float(dt_euler), float(t_scale), B.ptr(class_ids),
                                          float(cfg_strength or 0.0), B.ptr(mask), int(mask_is_ones), B.current_stream(dev)))
        if check and B.lib().fc_unet_meeting_launches(hnd) > 0:
            B.check(B.lib().fc_unet_check(hnd, B.current_stream(dev), 1))
        return x

    def profile_ops(self, batch: int, repeats: int = 20):
        """Per-launch device time of the current plan (bench.py's live roofline measurement).  Run a forward or an
        integration first so the internal state holds finite data.  Returns a list of dicts."""
        lib, h = B.lib(), self._handle
        rows = C.c_int(0)
        lib.fc_unet_chains(h, C.byref(rows))
        batch = min(batch, rows.value)                 # launches are timed at the rows one chain carries
        n = lib.fc_unet_plan_launches(h)
        ms = (C.c_float * n)()
        dev = self._handle_device
        B.check(lib.fc_unet_profile_ops(h, batch, repeats, ms, n, B.current_stream(dev)))
        out = []
        for i in range(n):
            k, m, f = C.c_char_p(), C.c_char_p(), C.c_double()
            B.check(lib.fc_unet_op_info(h, i, C.byref(k), C.byref(m), C.byref(f)))
            bp, bf = C.c_double(), C.c_double()
            B.check(lib.fc_unet_op_bytes(h, i, C.byref(bp), C.byref(bf)))
            out.append(dict(kernel=k.value.decode(), module=m.value.decode(), flops_per_sample=f.value, ms=float(ms[i]), rows=batch,
                            bytes=bp.value * batch + bf.value))
        return out

    def debug_tensor(self, name: str) -> torch.Tensor:
        """NHWC copy of an internal activation of the last forward (tests only)."""
        p, c, h, w = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        B.check(B.lib().fc_unet_debug_tensor(self._handle, name.encode(), C.byref(p), C.byref(c), C.byref(h), C.byref(w)))
        return p.value, c.value, h.value, w.value
