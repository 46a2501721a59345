"""CPU restatement of flocoder's flow training step (train_flow.py:33-71 EMA, :338-397 step), torch-CPU autograd over the
functional U-Net of ``flow_oracle``.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity status: PINNED -- ``tools/make_golden.py`` (fixture g10) drives the
reference's own ``Unet`` through the same lines with ``torch.optim.Adam`` and ``tests/test_oracle_golden.py`` holds this file to
the recorded loss, gradient norms, per-parameter gradient / parameter / EMA checksums over two consecutive steps.

The optimiser arithmetic below is written out (no ``torch.optim``): Adam as PyTorch's single-tensor path computes it
(torch/optim/adam.py ``_single_tensor_adam``: lerp of the first moment, bias corrections from a per-parameter step count, ``denom =
sqrt(v)/sqrt(bc2) + eps``), ``clip_grad_norm_`` (coefficient ``max_norm / (norm + 1e-6)`` clamped to 1) and the EMA recurrence.
Parameters that receive no gradient in a step (``class_cond_mlp.*`` when ``cond is None``, train_flow.py:343-345) are skipped by
Adam -- no moment decay, no step-count increment -- exactly as ``optimizer.step()`` skips ``p.grad is None``; the EMA still
averages them.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import flow_oracle as fo

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def train_time(u: Tensor, eps: float = 1e-3) -> Tensor:
    """train_flow.py:347-348: t = warp_time(u (1-eps) + eps) for u ~ U[0,1)."""
    return fo.warp_time(u * (1 - eps) + eps)


def loss_and_grads(sd: SD, source: Tensor, target: Tensor, t: Tensor, cond: Optional[dict]):
    """train_flow.py:350-371: x = (1-t)s + t g, v* = g - s, loss = MSE(model(x, 999 t, cond), v*) (mean over all elements);
    returns (loss, {name: grad or None}, v_model).  Buffers (none in this model) and unused parameters get None."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    x, v_star = fo.flow_train_targets(source, target, t)
    v = fo.unet_forward(leaves, x, t * 999, cond)
    loss = torch.nn.functional.mse_loss(v, v_star)
    loss.backward()
    return loss.detach(), {k: p.grad for k, p in leaves.items()}, v.detach()


def grad_norm(grads: Dict[str, Optional[Tensor]]) -> Tensor:
    """clip_grad_norm_'s total norm: the 2-norm of the per-tensor 2-norms (tensors with a gradient only)."""
    norms = [g.norm(2) for g in grads.values() if g is not None]
    return torch.stack(norms).norm(2)


def clip_coef(total_norm: Tensor, max_norm: float = 1.0) -> Tensor:
    return torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)


def new_state(sd: SD, ema_decay: float = 0.999) -> dict:
    return {"m": {k: torch.zeros_like(v) for k, v in sd.items()}, "v": {k: torch.zeros_like(v) for k, v in sd.items()},
            "step": {k: 0 for k in sd}, "ema": {k: v.clone() for k, v in sd.items()}, "ema_decay": ema_decay}


def adam_ema_step(sd: SD, grads: Dict[str, Optional[Tensor]], state: dict, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                  max_norm: float = 1.0) -> Tensor:
    """clip (train_flow.py:392) -> Adam (:396) -> EMA (:397), in place on ``sd`` / ``state``; returns the pre-clip gradient norm."""
    total = grad_norm(grads)
    coef = clip_coef(total, max_norm)
    b1, b2 = betas
    for k, p in sd.items():
        g = grads.get(k)
        if g is not None:
            g = g * coef
            state["step"][k] += 1
            n = state["step"][k]
            m, v = state["m"][k], state["v"][k]
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** n, 1 - b2 ** n
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-(lr / bc1))
        d = state["ema_decay"]
        state["ema"][k] = d * state["ema"][k] + (1.0 - d) * p
    return total
