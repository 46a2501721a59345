"""Flow training step: host-side mirror of the step in ``train_flow.py`` (reference train_flow.py:33-71 EMA, :90-182 batch_to_data,
:338-397 the step) over the gfx950 library.

Two ways in:

* ``FlowTrainer.step(source, target, cond)`` -- the whole step on the device through the C ABI: interpolation (``fc_flow_interp``),
  U-Net forward + backward (``fc_unet_forward`` / ``fc_unet_backward``), MSE loss and its gradient (``fc_mse_loss_grad``),
  gradient-norm clipping (``fc_grad_clip_coef``), Adam and the EMA in one pass over flat vectors (``fc_adam_ema_step``).  No host
  synchronisation inside a step; data-parallel ranks add one all-reduce of the flat gradient vector (RCCL).
* the reference's own loop shape -- ``loss_fn(model(x, t*999, cond), v).backward(); clip_grad_norm_; optimizer.step(); ema.update()`` --
  works unchanged too: ``Unet.forward`` joins autograd through the same native backward, ``EMA`` below mirrors the reference class
  (kept on the device instead of round-tripping through host memory every step, train_flow.py:52-54).

There is no CPU path.
"""
from __future__ import annotations

import os
import random
from typing import Dict, Optional

import torch

from . import _binding as B
from .dist import average_gradients
from .inpainting import mask_blending
from .ot import compute_ot_pairing
from .sampling import warp_time


class EMA:
    """train_flow.py:33-71 with the shadow weights resident on the device (same recurrence, same eval()/train() swap)."""

    def __init__(self, model, decay=0.99, device=None):
        self.model, self.decay = model, decay
        self.device = device if device is not None else 'cuda'
        self.shadow: Dict[str, torch.Tensor] = {}
        self.backup: Dict[str, Optional[torch.Tensor]] = {}
        for name, param in model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()

    def update(self):
        with torch.no_grad():
            for name, param in self.model.named_parameters():
                if param.requires_grad:
                    assert name in self.shadow
                    self.shadow[name] = self.decay * self.shadow[name] + (1.0 - self.decay) * param.data

    def eval(self):
        """Averaged weights into the model (train_flow.py:56-63).  Written with ``param.copy_`` under no_grad so the tensor version
        moves, and the model is told explicitly as well: the library keeps its own packed copy of the weights and must re-upload."""
        with torch.no_grad():
            for name, param in self.model.named_parameters():
                if param.requires_grad:
                    self.backup[name] = param.detach().clone()
                    param.copy_(self.shadow[name])
        _mark_dirty(self.model)

    def train(self):
        """Live weights back (train_flow.py:65-71)."""
        with torch.no_grad():
            for name, param in self.model.named_parameters():
                if param.requires_grad:
                    param.copy_(self.backup[name])
                    self.backup[name] = None
        _mark_dirty(self.model)


def _mark_dirty(module) -> None:
    for m in module.modules():
        if hasattr(m, "mark_dirty"):
            m.mark_dirty()


def batch_to_data(batch, device, pre_encoded=True, mask_encoder=None, epoch=None, curriculum_epochs=10, extend_epochs=20,
                  blank_latents=None):
    """train_flow.py:90-182: unpack a (latents | dict, class) batch, draw the source noise, encode / blend the inpainting mask and
    re-index the TARGET by the greedy OT pairing.  The on-the-fly mask augmentation is a no-op upstream (p_ones = p_zeros = 0,
    train_flow.py:129-133) and is therefore absent.  Returns (source, target, class_cond, mask, mask_pixels)."""
    source, mask, mask_pixels = None, None, None
    if not pre_encoded:
        raise NotImplementedError("batch_to_data: only pre-encoded latents are supported (pre_encoded=True is hard-wired upstream, train_flow.py:213)")
    data, class_cond = batch
    if isinstance(data, dict):
        target = data['target_latents'].to(device)
        class_cond = class_cond.to(device)
        if mask_encoder is not None:
            mask_pixels = data['mask_pixels'].float()
            if len(mask_pixels.shape) < 4:
                mask_pixels = mask_pixels.unsqueeze(1)
            mask = mask_pixels.to(device)
            source = data['source_latents'].to(device)
    else:
        target, class_cond = data.to(device), class_cond.to(device)
    noise = torch.randn_like(target)
    if mask is None:
        source = noise
    elif mask_pixels is not None and mask_encoder is not None:
        mask = mask_encoder(mask_pixels.to(device))
        source = mask_blending(source, mask, noise)
    else:
        raise AssertionError("Unintended edge case in batch_to_data (train_flow.py:149)")
    ot_indices = compute_ot_pairing(source, target)
    target = target[ot_indices]
    return source, target, class_cond, mask, mask_pixels


class FlowTrainer:
    """One object per rank.  Owns the flat parameter / gradient / Adam / EMA vectors (library table layout); the model's
    parameters become views into the flat parameter vector, so ``model.state_dict()`` always shows the trained weights."""

    FLAG_TAIL = 8          # floats behind the gradient table (five used; keeps the buffer's length a multiple of four)

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0, ema_decay: float = 0.999,
                 t_eps: float = 1e-3, t_scale: float = 999.0, device=None, process_group=None, distributed: Optional[bool] = None):
        device = torch.device(device) if device is not None else next(model.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("flocoder_amd.FlowTrainer runs on MI355X (gfx950) only; there is no CPU path")
        self.model, self.device = model, device
        self.lr, self.betas, self.eps, self.max_norm, self.ema_decay = lr, betas, eps, max_norm, ema_decay
        self.t_eps, self.t_scale = t_eps, t_scale
        n = model._flat_numel
        z = lambda: torch.zeros(n, dtype=torch.float32, device=device)
        self.params, self.exp_avg, self.exp_avg_sq = z(), z(), z()
        # the gradient vector carries FLAG_TAIL extra floats behind the library's table: the replicas' agreement flags (three parameter-group
        # presences, two input-validity bits) ride in the early gradient bucket of a data-parallel step as SUMs instead of an all-reduce of
        # their own (``_flag_tail`` / ``_agree_from_tail``); the library never sees them (it is given ``self.grads``, the first n floats)
        self._gbuf = torch.zeros(n + self.FLAG_TAIL, dtype=torch.float32, device=device)
        self.grads = self._gbuf[:n]
        model.to(device)
        model.adopt_flat(self.params)
        self.ema = self.params.clone()
        # parameter groups that can go without a gradient in a step (torch's optimiser skips p.grad is None, step count included):
        # class embedding path (cond=None), mask_fusion_conv (no mask, or a mask that is all ones: unet.py:301), the per-scale injections
        self._groups = {"class": model.param_range("class_cond_mlp."), "fusion": model.param_range("mask_fusion_conv."),
                        "inject": model.param_range("down_mask_fusions.", "up_mask_fusions.")}
        self._lo, self._hi = self._groups["class"]
        self.step_main, self.steps = 0, {k: 0 for k in self._groups}
        self._scal = torch.zeros(4, dtype=torch.float32, device=device)      # loss | grad norm | clip coefficient
        self._id_flag = torch.zeros(1, dtype=torch.int32, device=device)     # sticky: bit 0 = a class id, bit 1 = a pairing entry out of range (step prologue)
        self._id_flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()   # its mirror, refreshed asynchronously behind every prologue
        self._ws = torch.zeros(256, dtype=torch.float32, device=device)
        self.pg = process_group
        dist = torch.distributed
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1) if distributed is None else distributed
        # the 10 % conditioning drop (train_flow.py:343-345) is drawn from a generator every rank seeds alike, so replicas drop together
        self._drop_rng = random.Random(0x5EED) if self.distributed else random
        # data-parallel steps start the all-reduce of the late layers' gradients in the middle of the backward (loss_and_grads / finish_gradients)
        self.overlap_comm = os.environ.get("FLOCODER_AMD_NO_COMM_OVERLAP") is None
        self._pending = None
        model.set_grad_buckets(self.distributed and self.overlap_comm)
        model.train()
        model.sync_flat()

    # ---- pieces, each one library call -----------------------------------------------------------------------------
    def interpolate(self, source, target, t):
        x, v = torch.empty_like(target), torch.empty_like(target)
        B.check(B.lib().fc_flow_interp(B.ptr(source), B.ptr(target), B.ptr(t), B.ptr(x), B.ptr(v), target.shape[0], target[0].numel(),
                                       B.current_stream(self.device)))
        return x, v

    def prepare(self, source, target, u, cls=None, pairing=None):
        """The step's prologue as ONE library call (``fc_flow_prepare``): t = warp_time(u (1 - t_eps) + t_eps), the U-Net's time input
        t * t_scale, x = (1 - t) source + t target[pairing], v* = target[pairing] - source, and the range check of the class ids
        into a sticky device flag (``check_class_ids`` reads it).  Returns (t, time, x, v*)."""
        dev, bsz = self.device, target.shape[0]
        t, time = torch.empty(bsz, device=dev), torch.empty(bsz, device=dev)
        x, v = torch.empty_like(source), torch.empty_like(source)
        ncls = self.model._cfg.n_classes if cls is not None else 0
        check = cls is not None or pairing is not None
        B.check(B.lib().fc_flow_prepare(B.ptr(source), B.ptr(target), B.ptr(pairing), B.ptr(u), self.t_eps, 0.5, self.t_scale, B.ptr(cls), ncls,
                                        B.ptr(t), B.ptr(time), B.ptr(x), B.ptr(v), self._id_flag.data_ptr() if check else None, bsz,
                                        target[0].numel(), B.current_stream(dev)))
        if check:
            self._id_flag_host.copy_(self._id_flag, non_blocking=True)     # the host looks at it when it next passes by: no sync
        return t, time, x, v

    def check_class_ids(self) -> None:
        """Raise IndexError if any step so far was given a class id outside [0, n_classes) (nn.Embedding raises for those, unet.py:205)
        or a pairing entry outside [0, batch) (``target[ot_indices]`` raises, train_flow.py:350).  The optimiser launches are guarded
        by the same device flag (``fc_adam_ema_step_guarded``): from the offending step on NO update is applied -- parameters, Adam
        moments and EMA are those of the last valid step -- until this raises.  One host sync: ``step`` calls it as soon as the pinned
        mirror of the flag shows it (usually the next step), every 64 steps in any case, and the state-dict methods always."""
        flag = int(self._id_flag.item())
        if flag:
            self._id_flag.zero_()
            self._id_flag_host.zero_()
            what = []
            if flag & 1:
                what.append(f"class_cond ids must lie in [0, {self.model._cfg.n_classes})")
            if flag & 2:
                what.append("pairing indices must lie in [0, batch)")
            raise IndexError("; ".join(what) + " (a training step since the last check was given others; no optimiser update has been "
                                               "applied from that step on)")

    def loss_and_grads(self, x, t, cls, v_target, mask=None, time=None, overlap: bool = False, tail: Optional[torch.Tensor] = None):
        """forward -> loss -> backward; leaves the gradients in ``self.grads`` and returns (loss 0-d tensor, v_model).  With ``overlap``
        (data-parallel steps) the backward runs in its two parts and the SUM all-reduce of the early bucket -- final_*, mid_*, ups.*:
        parameters every step has gradients for, so it needs no agreement between the ranks first -- is started in between: RCCL moves
        it over xGMI while the second half of the data-gradient chain runs.  The handle is left in ``self._pending``;
        ``finish_gradients`` waits for it and reduces the rest."""
        m, lib, st = self.model, B.lib(), B.current_stream(self.device)
        if time is None:
            time = (t * self.t_scale).contiguous()
        v = m._forward_native(x, time, cls, mask, train=True)
        dv = torch.empty_like(v)
        B.check(lib.fc_mse_loss_grad(B.ptr(v), B.ptr(v_target), B.ptr(dv), self._scal.data_ptr(), self._ws.data_ptr(), v.numel(), st))
        self._drain_pending()      # a collective an aborted step left in flight still writes self.grads: wait before the backward zeroes it
        nb, split = m.grad_buckets() if overlap else (1, 0)
        if overlap and nb == 2 and 0 < split < self.grads.numel() and all(hi <= split for _, hi in self._groups.values()):
            m.backward_native(x, time, cls, dv, self.grads, mask=mask, parts=(0, 0))
            n, hi = self.grads.numel(), self.grads.numel()
            if tail is not None:                # the agreement flags ride behind the late layers' gradients: one collective fewer per step
                self._gbuf[n:n + tail.numel()].copy_(tail)
                hi = n + tail.numel()
            self._pending = (torch.distributed.all_reduce(self._gbuf[split:hi], op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True), split,
                             tail is not None)
            m.backward_native(x, time, cls, dv, self.grads, mask=mask, parts=(1, 1))
        else:
            m.backward_native(x, time, cls, dv, self.grads, mask=mask)
        return self._scal[0], v

    def _drain_pending(self) -> None:
        """Wait for an early-bucket all-reduce that was started but never finished (a step that raised between ``loss_and_grads`` and
        ``finish_gradients``, e.g. the input check of ``_agree``): its communication stream may still be writing ``self.grads``."""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is not None and pend[0] is not None:
            pend[0].wait()

    def finish_gradients(self) -> None:
        """DDP's gradient averaging over the flat vector: what ``loss_and_grads(overlap=True)`` left in flight is waited for, the remaining
        range is reduced now (the ranks have agreed on its parameter groups by then, ``_agree``), everything is divided by the world size."""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is None:
            average_gradients(self.grads, self.pg)
            return
        work, split = pend[0], pend[1]
        torch.distributed.all_reduce(self.grads[:split], op=torch.distributed.ReduceOp.SUM, group=self.pg)
        if work is not None:
            work.wait()
        self.grads.div_(torch.distributed.get_world_size(self.pg))

    @property
    def step_class(self):
        return self.steps["class"]

    def optimizer_step(self, has_class_grads: bool, has_mask_grads: bool = False, has_fusion_grads: Optional[bool] = None):
        """clip_grad_norm_ -> Adam -> EMA (train_flow.py:392-397).  Parameters without a gradient in this step (class_cond_mlp.*
        without conditioning, the mask branches without a mask, mask_fusion_conv with an all-ones mask) are skipped by Adam, step
        count included, exactly as torch.optim skips ``p.grad is None``; the EMA still averages them."""
        lib, st = B.lib(), B.current_stream(self.device)
        B.check(lib.fc_grad_clip_coef(self.grads.data_ptr(), self.params.numel(), None, 0, self.max_norm, self._scal.data_ptr() + 4,
                                      self._ws.data_ptr(), st))
        self._adam_all(has_class_grads, has_mask_grads, has_fusion_grads)

    def _adam_all(self, has_class_grads: bool, has_mask_grads: bool = False, has_fusion_grads: Optional[bool] = None):
        """Adam + EMA over the U-Net's flat vectors with the clip coefficient already in ``_scal[2]``."""
        lib, st = B.lib(), B.current_stream(self.device)
        n = self.params.numel()
        P, G, M, V, E = (t.data_ptr() for t in (self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.ema))
        coef = self._scal.data_ptr() + 8
        b1, b2 = self.betas
        self.step_main += 1
        present = {"class": has_class_grads, "inject": has_mask_grads,
                   "fusion": has_mask_grads if has_fusion_grads is None else has_fusion_grads}

        def run(a, b, step, adam):
            if b > a:
                B.check(lib.fc_adam_ema_step_guarded(P + 4 * a, G + 4 * a, M + 4 * a, V + 4 * a, E + 4 * a, b - a, coef, self.lr, b1, b2, self.eps,
                                                     max(step, 1), self.ema_decay, int(adam), self._id_flag.data_ptr(), st))
        cuts = sorted((lo, hi, k) for k, (lo, hi) in self._groups.items() if hi > lo)
        pos = 0
        for lo, hi, k in cuts:
            run(pos, lo, self.step_main, True)
            if present[k]:
                self.steps[k] += 1
            run(lo, hi, self.steps[k], present[k])
            pos = hi
        run(pos, n, self.step_main, True)
        self.model.sync_flat()

    # ---- data-parallel agreement -------------------------------------------------------------------------------------
    def _flag_tail(self, present: Dict[str, bool]) -> torch.Tensor:
        """What ``_agree`` exchanges, as five non-negative floats whose SUM over the ranks answers the same questions (a group is stepped
        if ANY rank has a gradient for it <=> the sum of the 0/1 presences is positive; SOME rank was given bad inputs <=> the sum of that
        bit is positive): they can then ride in the early gradient bucket, a SUM all-reduce, instead of a MAX all-reduce of their own."""
        keys = sorted(self._groups)
        bits = torch.stack((self._id_flag & 1, (self._id_flag >> 1) & 1)).flatten().to(torch.float32)
        return torch.cat([torch.tensor([float(present[k]) for k in keys], device=self.device), bits])

    def _agree_from_tail(self, present: Dict[str, bool]) -> Dict[str, bool]:
        """``_agree`` for a step whose flags travelled with the early gradient bucket (``loss_and_grads(tail=...)``): wait for that bucket,
        zero the groups this rank has no gradient for (they lie in the bucket still to be reduced), read the summed flags."""
        work, split, _ = self._pending
        work.wait()
        self._pending = (None, split, True)
        keys = sorted(self._groups)
        for k in keys:
            lo, hi = self._groups[k]
            if hi > lo and not present[k]:
                self.grads[lo:hi].zero_()
        n = self.grads.numel()
        vals = self._gbuf[n:n + len(keys) + 2].tolist()
        bad = (1 if vals[-2] > 0 else 0) | (2 if vals[-1] > 0 else 0)
        if bad:
            self._pending = None                             # every rank raises here, none enters the second bucket
            if not int(self._id_flag.item()):
                self._id_flag.fill_(bad)                     # another rank's batch: raise the same error here
            self.check_class_ids()
        return {k: v > 0 for k, v in zip(keys, vals)}

    def _agree(self, present: Dict[str, bool]) -> Dict[str, bool]:
        """Replicas must take the SAME Adam decisions.  Which parameter groups received a gradient is a per-rank fact (one rank may
        have dropped its conditioning, another not); Adam over a group on one rank but not the other would let parameters, moments
        and step counters drift apart for good.  So: a group is stepped everywhere if ANY rank produced a gradient for it (DDP's
        zero-gradient-but-stepped semantics) -- one MAX all-reduce of three flags -- and a rank that had none contributes zeros."""
        if not self.distributed:
            return present
        keys = sorted(self._groups)
        for k in keys:
            lo, hi = self._groups[k]
            if hi > lo and not present[k]:
                self.grads[lo:hi].zero_()
        # the input-validity flag (class ids / pairing, set by the step prologue on the device) rides in the same collective: every rank
        # learns that SOME rank was given bad inputs and all raise together, before any of them enters the gradient all-reduce
        flags = torch.cat([torch.tensor([float(present[k]) for k in keys], device=self.device), self._id_flag.to(torch.float32)])
        torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        vals = flags.tolist()
        if vals[-1]:
            if not int(self._id_flag.item()):
                self._id_flag.fill_(int(vals[-1]))           # another rank's batch: raise the same error here
            self.check_class_ids()
        return {k: bool(v) for k, v in zip(keys, vals)}

    # ---- the step -------------------------------------------------------------------------------------------------
    def step(self, source, target, cond=None, u: Optional[torch.Tensor] = None, pairing: Optional[torch.Tensor] = None):
        """train_flow.py:346-397 for one batch: returns the loss as a 0-d device tensor (no host sync).  ``pairing`` (int64 [B], e.g.
        from ``compute_ot_pairing``) trains against ``target[pairing]`` without materialising the gather (train_flow.py:350)."""
        dev = self.device
        source = source.to(dev, torch.float32).contiguous()
        target = target.to(dev, torch.float32).contiguous()
        bsz = target.shape[0]
        if u is None:
            u = torch.rand(bsz, device=dev)
        u = u.to(dev, torch.float32).contiguous()
        cls = cond.get('class_cond') if isinstance(cond, dict) else None
        mask = cond.get('mask_cond') if isinstance(cond, dict) else None
        if mask is not None and not self.model._cfg.mask_cond:
            mask = None
        if mask is not None:
            # the mask latents enter as data: the U-Net's mask branches train, the MaskEncoder that produced them does not
            mask = mask.detach().to(dev, torch.float32).contiguous()
        if cls is not None and not self.model.class_condition:
            cls = None
        if cls is not None:
            cls = cls.to(dev, torch.int64).contiguous()
            if cls.shape != (bsz,):
                raise ValueError("class_cond must have shape [batch]")
        if pairing is not None:
            pairing = pairing.to(dev, torch.int64).contiguous()
            if pairing.shape != (bsz,):
                raise ValueError(f"pairing must have shape [batch] = ({bsz},), got {tuple(pairing.shape)}")
        if int(self._id_flag_host[0]):
            self.check_class_ids()                                   # an earlier step's inputs were out of range: raise now
        t, time, x, v_target = self.prepare(source, target, u, cls, pairing)
        fused = mask is not None and not bool(torch.allclose(mask, torch.ones_like(mask)))    # unet.py:301 (host sync, as upstream)
        mine = {"class": cls is not None, "inject": mask is not None, "fusion": fused}
        overlap = self.distributed and self.overlap_comm
        loss, _ = self.loss_and_grads(x, t, cls, v_target, mask, time=time, overlap=overlap, tail=self._flag_tail(mine) if overlap else None)
        loss = loss.clone()
        if (cls is not None or pairing is not None) and (self.step_main & 63) == 0 and not self.distributed:
            self.check_class_ids()                                   # every 64th step (and from the state-dict methods): one host sync
        # (data-parallel steps learn about bad inputs from the agreement below, on every rank at once: a rank that raised alone here
        # would leave the others waiting in the gradient all-reduce)
        rode = self._pending is not None and len(self._pending) > 2 and self._pending[2]
        present = self._agree_from_tail(mine) if rode else self._agree(mine)
        if self.distributed:                                     # DDP semantics: average the gradients over ranks
            self.finish_gradients()
        self.optimizer_step(has_class_grads=present["class"], has_mask_grads=present["inject"], has_fusion_grads=present["fusion"])
        return loss

    # ---- inpainting: the MaskEncoder trains with the U-Net (train_flow.py:312-318,361-395) ----------------------------------
    def attach_mask_encoder(self, mask_encoder, lr_scale: float = 0.1, max_norm: float = 0.5):
        """The second parameter group of the reference's optimiser (lr x 0.1, train_flow.py:313-318) with its own flat vectors."""
        me = mask_encoder.to(self.device).train()
        n = me._flat_numel
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.me, self.me_lr, self.me_max_norm = me, self.lr * lr_scale, max_norm
        self.me_params, self.me_grads, self.me_m, self.me_v = z(), z(), z(), z()
        with torch.no_grad():
            for name, shape, off in me._table:
                p = me.get_parameter(name)
                self.me_params[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.me_params[off:off + p.numel()].view(shape)
        self.me_ema = self.me_params.clone()
        self.me_step = 0
        self._me_scal = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.model.mask_encoder = me                          # as upstream (train_flow.py:333): checkpoints / EMA see it through the model

    def inpaint_step(self, source_latents, target, mask_pixels, class_cond=None, noise=None, u=None, drop_cond=False, ot=False):
        """One inpainting training step entirely on the device (batch_to_data's mask branch + train_flow.py:346-397):
        mask = MaskEncoder(mask_pixels); source = blend(source_latents, mask, noise); flow loss through the mask-conditioned U-Net;
        + MSE(MaskEncoder(1), 1) + MSE(MaskEncoder(0), 0); backward into both networks (the encoder is reached through the U-Net's
        mask input AND through the blended source); joint clip at 1.0, the encoder's own at 0.5, Adam (lr, lr x 0.1), EMA of both."""
        lib, dev, st = B.lib(), self.device, B.current_stream(self.device)
        me = self.me
        f = lambda t_: t_.to(dev, torch.float32).contiguous()
        s0, tgt, mp = f(source_latents), f(target), f(mask_pixels)
        if mp.dim() < 4:
            mp = mp.unsqueeze(1)
        bsz = tgt.shape[0]
        noise = f(noise) if noise is not None else torch.randn_like(tgt)
        if u is None:
            u = torch.rand(bsz, device=dev)
        t = warp_time(f(u) * (1 - self.t_eps) + self.t_eps).contiguous()
        cls = class_cond.to(dev, torch.int64).contiguous() if (class_cond is not None and self.model.class_condition) else None
        self.model.check_class_ids(cls)                       # this step does not go through fc_flow_prepare: check on the host (it synchronises anyway)
        with torch.no_grad():
            time = (t * self.t_scale).contiguous()
            if drop_cond:                                     # the 10 % classifier-free-guidance drop (train_flow.py:343-345): no cond, pure noise source
                mask, cls = None, None
                x, v_target = self.interpolate(torch.randn_like(tgt), tgt, t)
                loss_t, _ = self.loss_and_grads(x, t, None, v_target, None)
                self.me_grads.zero_()
            else:
                mask = me._forward_native(mp)
                source = mask_blending(s0, mask, noise)
                if ot:                                        # batch_to_data re-indexes the TARGET by the greedy pairing (train_flow.py:160-163)
                    tgt = tgt[compute_ot_pairing(source, tgt)].contiguous()
                x, v_target = self.interpolate(source, tgt, t)
                v = self.model._forward_native(x, time, cls, mask, train=True)
                dv = torch.empty_like(v)
                B.check(lib.fc_mse_loss_grad(B.ptr(v), B.ptr(v_target), B.ptr(dv), self._scal.data_ptr(), self._ws.data_ptr(), v.numel(), st))
                _, dx, dmask = self.model.backward_native(x, time, cls, dv, self.grads, mask=mask, want_dx=True, want_dmask=True)
                # chain rule through x = (1-t) s' + t g,  v* = g - s',  s' = s + m (noise - s)
                d_src = (1 - t).view(-1, 1, 1, 1) * dx + dv
                d_mask = dmask + d_src * (noise - s0)
                me.backward_native(mp, d_mask, self.me_grads, accumulate=False)
            loss = self._scal[0].clone()
            for fill in (1.0, 0.0):                           # the 0/1 anchors of the mask latents (train_flow.py:361-369)
                pix = torch.full_like(mp, fill)
                y = me._forward_native(pix)
                want = torch.full_like(y, fill)
                dy = torch.empty_like(y)
                B.check(lib.fc_mse_loss_grad(B.ptr(y), B.ptr(want), B.ptr(dy), self._me_scal.data_ptr(), self._ws.data_ptr(), y.numel(), st))
                me.backward_native(pix, dy, self.me_grads, accumulate=True)
                loss = loss + self._me_scal[0]
            fused = mask is not None and not bool(torch.allclose(mask, torch.ones_like(mask)))
            present = self._agree({"class": cls is not None, "inject": mask is not None, "fusion": fused})
            if self.distributed:
                average_gradients(self.grads, self.pg)
                average_gradients(self.me_grads, self.pg)
            # clip_grad_norm_(model.parameters(), 1.0) covers the attached encoder too (train_flow.py:333,392), then the encoder alone at 0.5
            n, nm = self.params.numel(), self.me_params.numel()
            B.check(lib.fc_grad_clip_coef(self.grads.data_ptr(), n, self.me_grads.data_ptr(), nm, self.max_norm, self._scal.data_ptr() + 4,
                                          self._ws.data_ptr(), st))
            B.check(lib.fc_grad_clip_coef(self.me_grads.data_ptr(), nm, None, 0, 1e30, self._me_scal.data_ptr() + 4, self._ws.data_ptr(), st))
            c1 = self._scal[2]
            self._me_scal[2] = c1 * torch.clamp(self.me_max_norm / (c1 * self._me_scal[1] + 1e-6), max=1.0)
        self._adam_all(has_class_grads=present["class"], has_mask_grads=present["inject"], has_fusion_grads=present["fusion"])
        self.me_step += 1
        b1, b2 = self.betas
        B.check(lib.fc_adam_ema_step(self.me_params.data_ptr(), self.me_grads.data_ptr(), self.me_m.data_ptr(), self.me_v.data_ptr(),
                                     self.me_ema.data_ptr(), nm, self._me_scal.data_ptr() + 8, self.me_lr, b1, b2, self.eps, self.me_step,
                                     self.ema_decay, 1, st))
        me._synced = None                                     # its flat vector changed under the views: re-upload on next use
        return loss

    def train_batch(self, batch, epoch=None, cfg_drop: float = 0.1, mask_encoder=None, blank_latents=None):
        """batch_to_data + the 10 % conditioning drop of train_flow.py:338-345 + step.  With an attached MaskEncoder and an
        inpainting batch (dict with mask_pixels) the encoder trains too (``inpaint_step``)."""
        data = batch[0]
        if getattr(self, "me", None) is not None and isinstance(data, dict) and 'mask_pixels' in data:
            target = data['target_latents'].to(self.device)
            return self.inpaint_step(data['source_latents'], target, data['mask_pixels'].float(), class_cond=batch[1],
                                     drop_cond=self._drop_rng.random() < cfg_drop, ot=True)
        source, target, class_cond, mask_cond, _ = batch_to_data(batch, self.device, True, mask_encoder, epoch=epoch, blank_latents=blank_latents)
        cond = {'class_cond': class_cond, 'mask_cond': mask_cond}
        if self._drop_rng.random() < cfg_drop:
            cond = None
            source = torch.randn_like(source)
        return self.step(source, target, cond)

    # ---- state ----------------------------------------------------------------------------------------------------
    @property
    def grad_norm(self):
        return self._scal[1]

    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self.model.grad_views(self.ema).items()}

    def swap_ema(self):
        """EMA.eval() / EMA.train() (train_flow.py:56-71): exchange the live and the averaged weights."""
        tmp = self.params.clone()
        self.params.copy_(self.ema)
        self.ema.copy_(tmp)
        self.model.sync_flat()

    def state_dict(self):
        self.check_class_ids()
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "ema": self.ema.clone(),
                "step_main": self.step_main, "steps": dict(self.steps)}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.ema.copy_(sd["ema"])
        self.step_main, self.steps = int(sd["step_main"]), {k: int(v) for k, v in sd["steps"].items()}
