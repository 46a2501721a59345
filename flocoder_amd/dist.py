"""One process per GPU on one node, RCCL over xGMI through ``torch.distributed`` (backend "nccl" is RCCL on ROCm).

The sampling path shards by sample: every trajectory depends only on its own noise, class id and mask (GroupNorm is
per sample, there is no cross-sample op in the reference's unet.py / sampling.py), so the only collective is one
broadcast of the frozen weights from rank 0 before the loop; there is no per-step communication.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str | None = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver
        if backend is None:
            # FLOCODER_AMD_DIST_BACKEND=gloo rehearses the multi-rank code path where the ranks cannot each have a GPU
            backend = os.environ.get("FLOCODER_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n samples for `rank`; the first n % world ranks take one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_weights(module: torch.nn.Module, src: int = 0) -> int:
    """One flat broadcast of every parameter and buffer from `src` (frozen weights: U-Net 39.7 MB, SD-VAE 334.6 MB).
    Returns the number of bytes moved.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    # the writes above go through ``.data`` and move neither data_ptr nor the version counter: a module that already ran once
    # (its weights packed inside the library) must be told to upload again
    for m in module.modules():
        if hasattr(m, "mark_dirty"):
            m.mark_dirty()
    return flat.numel() * 4


def gather_samples(local: torch.Tensor, n_total: int) -> torch.Tensor | None:
    """Optional final all-gather of per-rank results (16 KiB per latent); rank order = sample order of shard_range."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def average_gradients(flat: torch.Tensor, group=None, bucket_bytes: int = 64 << 20) -> int:
    """Data-parallel training (BASELINE config 4): all-reduce(sum) the flat gradient vector in place and divide by the world
    size -- DDP's gradient averaging on the library's one contiguous vector (39.7 MB at dim=32, so one or two buckets; on xGMI a
    ring all-reduce is per-link bound and large buckets are the right shape).  Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    per = max(1, bucket_bytes // 4)
    n = 0
    for lo in range(0, flat.numel(), per):
        dist.all_reduce(flat[lo:lo + per], op=dist.ReduceOp.SUM, group=group)
        n += 1
    flat.div_(world)
    return n
