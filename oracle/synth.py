"""Deterministic synthetic weights for parity fixtures (TEST INFRASTRUCTURE ONLY).

The golden fixtures cannot carry 40 MB of U-Net weights, so both sides regenerate them:
``tools/make_golden.py`` loads ``synth_state_dict(shapes, seed)`` into the *reference*
modules, the tests load the same tensors into the oracle and into the HIP path.  The recipe
depends only on (name, shape, seed), never on construction order or on torch's init code.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import torch


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight" and len(shape) == 1:            # GroupNorm gain
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) == 2 and "class_cond_mlp.0" in name:   # nn.Embedding table
        return torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / max(1, fan_in)) ** 0.5


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, v, seed) for k, v in shapes.items()}


def synth_input(tag: str, shape: Sequence[int], seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(("in:" + tag).encode()) + 104729 * seed) & 0x7FFFFFFF)
    return scale * torch.randn(tuple(shape), generator=g)
