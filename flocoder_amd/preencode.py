"""Pre-encoding pipeline: host-side mirror of ``preencode_data.py``'s ``encode_batch`` / ``process_dataset``
(preencode_data.py:34-42,84-181) over the gfx950 codec kernels.

``process_dataset`` keeps the reference's contract -- batches from any DataLoader of ``(images, classes)`` (or the inpainting dict
batches), ``codec.encode`` on the device, one sample per output item, numeric class directories, the storage / batch budget -- and
adds two things: ``packed=True`` writes the run as ONE memory-mappable file (``flocoder_amd.data`` format) instead of one pickle
per latent, and ``rank / world`` give every GPU of a node its own share of the batches and its own shard
(``<out>/shard_<rank>.fcl``); ``merge_shards`` concatenates them.  The image datasets themselves (torchvision / MIDI
rendering, preencode_data.py:45-62) are the caller's business.
"""
from __future__ import annotations

import concurrent.futures
import os
import random
import string
from pathlib import Path
from typing import Iterable, Optional

import numpy as np
import torch

from .data import PackedLatentDataset, pack_latents


def generate_random_string(length=6):
    """preencode_data.py:29-31."""
    return ''.join(random.choices(string.ascii_lowercase + string.digits, k=length))


def encode_batch(codec, x, device, quantize=False):
    """preencode_data.py:34-42: z = codec.encode(x) (optionally quantised) without gradients."""
    with torch.no_grad():
        x = x.to(device, non_blocking=True)
        z = codec.encode(x)
        if quantize:
            z_q, _ = codec.quantize(z)
            return z_q
        return z


class _ListDataset(torch.utils.data.Dataset):
    def __init__(self, items, labels, n_classes):
        self.items, self.labels, self.n_classes = items, labels, n_classes

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i], torch.tensor(self.labels[i], dtype=torch.long)


def process_dataset(codec, dataloader: Iterable, output_dir, device, max_batches: Optional[int] = None,
                    max_storage_bytes: float = float("inf"), n_classes: int = 0, quantize=False, inpainting=False, packed=False,
                    rank: int = 0, world: int = 1, io_workers: int = 16) -> dict:
    """preencode_data.py:84-181.  Returns {'samples', 'bytes', 'path'}.  With ``packed`` the latents of this rank are collected
    (host memory: 16 KiB per 4x32x32 latent) and written once at the end; otherwise every sample becomes
    ``<out>/<class>/sample_<batch>_<i>_<rand>.pt`` (``<out>/<batch % 100>/...`` without classes), as upstream."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    has_classes = n_classes > 0
    if has_classes and not packed:
        for c in range(n_classes):
            (output_dir / str(c)).mkdir(exist_ok=True)
    items, labels, stored, count = [], [], 0, 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=io_workers) as pool:
        for batch_idx, batch in enumerate(dataloader):
            if (max_batches is not None and batch_idx >= max_batches) or stored >= max_storage_bytes:
                break
            if batch_idx % world != rank:
                continue                                               # another GPU's batch
            if isinstance(batch, (tuple, list)):
                target_images, classes = batch
                source_latents = mask_pixels = None
            elif isinstance(batch, dict) and inpainting:
                target_images, mask_pixels, classes = batch['target_image'], batch['mask_pixels'], batch['label']
                source_latents = encode_batch(codec, batch['source_image'], device, quantize=quantize).cpu()
            else:
                raise ValueError(f"Unexpected batch format: {type(batch)}")
            target_latents = encode_batch(codec, target_images, device, quantize=quantize).cpu()
            futures = {}
            for i in range(target_latents.shape[0]):
                class_idx = classes[i].item() if isinstance(classes, torch.Tensor) else classes
                if inpainting:
                    data = {'target_latents': target_latents[i].clone(), 'source_latents': source_latents[i].clone(),
                            'mask_pixels': mask_pixels[i].cpu().bool()}
                else:
                    data = target_latents[i].clone()
                count += 1
                if packed:
                    items.append(data)
                    labels.append(int(class_idx) if has_classes else 0)
                    stored += target_latents[i].numel() * 4 * (2 if inpainting else 1)
                    continue
                name = f"sample_{batch_idx}_{i}_{generate_random_string(4)}.pt"
                sub = str(class_idx) if has_classes else f"{batch_idx % 100:02d}"
                (output_dir / sub).mkdir(exist_ok=True)
                path = output_dir / sub / name
                futures[pool.submit(torch.save, data, path)] = path
            for fut in concurrent.futures.as_completed(futures):
                fut.result()
                stored += os.path.getsize(futures[fut])
    path = str(output_dir)
    if packed:
        path = str(output_dir / (f"shard_{rank}.fcl" if world > 1 else "latents.fcl"))
        if items:
            stored = pack_latents(_ListDataset(items, labels, n_classes), path)["bytes"]
    return {"samples": count, "bytes": stored, "path": path}


def merge_shards(paths, out_path: str) -> dict:
    """Concatenate packed shards (same latent shape / fields) into one file."""
    shards = [PackedLatentDataset(p) for p in paths]

    class _Cat(torch.utils.data.Dataset):
        n_classes = max(s.n_classes for s in shards)

        def __len__(self):
            return sum(len(s) for s in shards)

        def __getitem__(self, i):
            for s in shards:
                if i < len(s):
                    return s[i]
                i -= len(s)
            raise IndexError(i)
    return pack_latents(_Cat(), out_path)
