"""Pre-encoded latent datasets: host-side mirror of ``flocoder/data.py``'s ``PreEncodedDataset`` /
``InfiniteDataset`` (data.py:287-384; writer preencode_data.py:130-156), written against the behaviour fixture g11 pins, plus a packed
single-file format.

The reference stores ONE ``torch.save`` file per latent (a tensor ``[C,h,w]``, or a dict ``{target_latents, source_latents,
mask_pixels(bool)}`` for inpainting) under numeric class directories and reads them back with 12 DataLoader workers -- a
``torch.load`` (unpickle + small read) per sample, 256 per step: once the step runs in a few milliseconds that loader is the
bottleneck.  ``pack_latents`` converts such a directory once into one contiguous file; ``PackedLatentDataset`` memory-maps it
and hands out the same items; ``PackedLatentLoader`` builds whole shuffled batches with one gather per field into pinned
memory (no worker processes, rank-sharded for data-parallel training).

File layout (little endian): 64-byte header ``b"FCLATENT"``, u32 version, u32 flags (bit 0: inpainting fields), u64 count,
u32 C, h, w, u32 mask H, mask W, u32 n_classes, padding; then ``labels`` i64[count]; ``target`` f32[count][C][h][w];
with flag 0: ``source`` f32[count][C][h][w]; ``mask`` u8[count][ceil(H*W/8)] (bit-packed bool, numpy.packbits order).
Every section starts on a 64-byte boundary.
"""
from __future__ import annotations

import os
import random
import struct
from pathlib import Path
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset, IterableDataset

MAGIC, VERSION, HEADER = b"FCLATENT", 1, 64
_HDR = "<IIQIIIIII"


def _latent_files(root: Path) -> List[Path]:
    """All ``*.pt`` files (extension compared case-insensitively) at any depth below ``root``, in sorted path order.  The reference
    walks with ``os.scandir`` and keeps the file system's enumeration order (data.py:17-43); which file gets which label does not
    depend on that order, and a sorted list makes ``pack_latents`` reproducible."""
    found = []
    for dirpath, dirnames, filenames in os.walk(root, onerror=lambda e: None):
        dirnames.sort()
        found += [Path(dirpath, f) for f in sorted(filenames) if os.path.splitext(f)[1].lower() == ".pt"]
    return found


class PreEncodedDataset(Dataset):
    """One ``torch.save`` file per latent, as the reference's pre-encoder writes them (reader: data.py:311-384).

    Behaviour pinned by fixture g11 (the reference class on the same tree):
      * sub-directories whose names are all digits are classes.  A class's label is its POSITION among those directories sorted as
        paths -- lexicographic, so "10" comes before "2" -- not its numeric value; ``class_to_idx`` maps the numeric name to it.
      * ``n_classes=0`` switches class handling off: every file is labelled 0.  Without class directories the same happens; files are
        then taken from the sub-directories if there are any, else from ``data_dir`` itself.
      * an item is ``(payload, label)`` with ``payload`` whatever the file holds (a ``[C,h,w]`` tensor, or the inpainting dict) and
        ``label`` a 0-d int64 tensor; items are memoised up to ``max_cache_items`` and then replaced at random, one in a hundred.
      * an unreadable file yields zeros shaped like a cached payload (``[4,16,16]`` when nothing is cached) with label 0 instead of
        raising, so a damaged sample does not end a long training run.
    """

    def __init__(self, data_dir, max_cache_items=10000, n_classes=None):
        self.data_dir = Path(os.path.expanduser(str(data_dir)))
        children = [d for d in self.data_dir.iterdir() if d.is_dir()]
        numbered = sorted(d for d in children if d.name.isdigit())
        self.has_classes = bool(numbered) and n_classes != 0
        self.files: List[Path] = []
        self._labels: List[int] = []
        if self.has_classes:
            self.n_classes = len(numbered)
            self.class_to_idx = {int(d.name): pos for pos, d in enumerate(numbered)}
            for pos, d in enumerate(numbered):
                mine = _latent_files(d)
                self.files += mine
                self._labels += [pos] * len(mine)
        else:
            self.n_classes = 0
            for top in (children if children else [self.data_dir]):
                self.files += _latent_files(top)
            self._labels = [0] * len(self.files)
        self.actual_len = len(self.files)
        self.max_cache_items = max_cache_items
        self.cache = {}
        print(f"PreEncodedDataset({self.data_dir}): {self.actual_len} latents" + (f", {self.n_classes} classes" if self.has_classes else ""))

    def __len__(self):
        return self.actual_len

    def _remember(self, idx, item):
        if len(self.cache) < self.max_cache_items:
            self.cache[idx] = item
        elif self.cache and random.random() < 0.01:
            self.cache.pop(random.choice(list(self.cache)))
            self.cache[idx] = item

    def __getitem__(self, idx):
        hit = self.cache.get(idx)
        if hit is not None:
            return hit
        path = self.files[idx]
        try:
            payload = torch.load(path, map_location='cpu')
        except Exception as exc:
            print(f"PreEncodedDataset: cannot read {path} ({exc}); substituting zeros")
            like = next(iter(self.cache.values()))[0] if self.cache else torch.zeros(4, 16, 16)
            return torch.zeros_like(like), torch.tensor(0)
        item = (payload, torch.tensor(self._labels[idx], dtype=torch.long))
        self._remember(idx, item)
        return item


class InfiniteDataset(IterableDataset):
    """Endless stream of uniformly drawn items of a map-style dataset (data.py:287-308); the wrapped dataset's plain data attributes
    (``n_classes``, ``has_classes``, ...) are visible on the wrapper, which is how the training script reads them."""

    def __init__(self, base_dataset, shuffle=True):
        super().__init__()
        if not shuffle:
            raise AssertionError("InfiniteDataset only supports shuffle=True for now")
        self.dataset = base_dataset
        self.actual_len = len(base_dataset)
        for name, value in vars(base_dataset).items():
            if not name.startswith('__') and not callable(value) and not hasattr(self, name):
                setattr(self, name, value)

    def __iter__(self):
        n = self.actual_len
        while True:
            yield self.dataset[random.randrange(n)]


# ------------------------------------------------------------------------------------------------ packed format
def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def pack_latents(src, out_path: str, n_classes=None, chunk: int = 4096) -> dict:
    """Convert a reference-format directory (or any map-style dataset yielding ``(tensor | dict, label)``) into one packed file.
    Sample order = the dataset's index order.  Returns the header fields."""
    ds = src if isinstance(src, Dataset) else PreEncodedDataset(src, max_cache_items=0, n_classes=n_classes)
    n = len(ds)
    if n == 0:
        raise ValueError(f"pack_latents: no samples under {src}")
    first, _ = ds[0]
    inpaint = isinstance(first, dict)
    tgt0 = first['target_latents'] if inpaint else first
    C, h, w = (int(s) for s in tgt0.shape)
    mh = mw = 0
    if inpaint:
        m0 = first['mask_pixels']
        mh, mw = int(m0.shape[-2]), int(m0.shape[-1])
    mbytes = (mh * mw + 7) // 8
    off_lab = HEADER
    off_tgt = _align(off_lab + 8 * n)
    off_src = _align(off_tgt + 4 * n * C * h * w)
    off_msk = _align(off_src + (4 * n * C * h * w if inpaint else 0))
    total = off_msk + (n * mbytes if inpaint else 0)
    ncls = int(getattr(ds, "n_classes", 0) or 0)
    tmp = str(out_path) + ".tmp"
    with open(tmp, "wb") as f:
        f.write((MAGIC + struct.pack(_HDR, VERSION, int(inpaint), n, C, h, w, mh, mw, ncls)).ljust(HEADER, b"\0"))
        f.truncate(total)
    mm = np.memmap(tmp, dtype=np.uint8, mode="r+")
    lab = mm[off_lab:off_lab + 8 * n].view(np.int64)
    tgt = mm[off_tgt:off_tgt + 4 * n * C * h * w].view(np.float32).reshape(n, C, h, w)
    srcv = mm[off_src:off_src + 4 * n * C * h * w].view(np.float32).reshape(n, C, h, w) if inpaint else None
    msk = mm[off_msk:off_msk + n * mbytes].reshape(n, mbytes) if inpaint else None
    for i in range(n):
        item, label = ds[i]
        lab[i] = int(label)
        if inpaint:
            t = item['target_latents']
            srcv[i] = item['source_latents'].float().numpy()
            msk[i] = np.packbits(item['mask_pixels'].reshape(-1).numpy().astype(bool))
        else:
            t = item
        if tuple(t.shape) != (C, h, w):
            raise ValueError(f"pack_latents: sample {i} has shape {tuple(t.shape)}, expected {(C, h, w)}")
        tgt[i] = t.float().numpy()
    mm.flush()
    del lab, tgt, srcv, msk, mm
    os.replace(tmp, out_path)
    return {"count": n, "shape": (C, h, w), "inpainting": inpaint, "mask_shape": (mh, mw), "n_classes": ncls, "bytes": total}


class PackedLatentDataset(Dataset):
    """Memory-mapped reader of a ``pack_latents`` file; items are what ``PreEncodedDataset`` yields for the same samples."""

    def __init__(self, path: str):
        self.path = str(path)
        with open(self.path, "rb") as f:
            head = f.read(HEADER)
        if head[:8] != MAGIC:
            raise ValueError(f"{path}: not a packed latent file")
        ver, flags, n, C, h, w, mh, mw, ncls = struct.unpack(_HDR, head[8:8 + struct.calcsize(_HDR)])
        if ver != VERSION:
            raise ValueError(f"{path}: format version {ver}, this reader handles {VERSION}")
        self.count, self.shape, self.inpainting, self.mask_shape, self.n_classes = n, (C, h, w), bool(flags & 1), (mh, mw), ncls
        self.has_classes = ncls > 0
        self.actual_len = n
        mbytes = (mh * mw + 7) // 8
        off_lab = HEADER
        off_tgt = _align(off_lab + 8 * n)
        off_src = _align(off_tgt + 4 * n * C * h * w)
        off_msk = _align(off_src + (4 * n * C * h * w if self.inpainting else 0))
        mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        self._mm = mm
        self.labels = mm[off_lab:off_lab + 8 * n].view(np.int64)
        self.target = mm[off_tgt:off_tgt + 4 * n * C * h * w].view(np.float32).reshape(n, C, h, w)
        self.source = mm[off_src:off_src + 4 * n * C * h * w].view(np.float32).reshape(n, C, h, w) if self.inpainting else None
        self.mask = mm[off_msk:off_msk + n * mbytes].reshape(n, mbytes) if self.inpainting else None

    def __len__(self):
        return self.count

    def _mask_rows(self, idx) -> np.ndarray:
        mh, mw = self.mask_shape
        bits = np.unpackbits(self.mask[idx], axis=-1)[..., :mh * mw]
        return bits.reshape(bits.shape[:-1] + (1, mh, mw)).astype(bool)

    def __getitem__(self, idx):
        label = torch.tensor(int(self.labels[idx]), dtype=torch.long)
        tgt = torch.from_numpy(np.array(self.target[idx]))
        if not self.inpainting:
            return tgt, label
        return {'target_latents': tgt, 'source_latents': torch.from_numpy(np.array(self.source[idx])),
                'mask_pixels': torch.from_numpy(self._mask_rows(idx))}, label


class PackedLatentLoader:
    """Batches straight from the memory map: one shuffled permutation per epoch (seeded, identical on every rank), this
    rank's strided share of it, one fancy-index gather per field into pinned memory.  Yields what a DataLoader over
    ``PreEncodedDataset`` yields after collation: ``(tensor[B,C,h,w] | dict of batched tensors, LongTensor[B])``."""

    def __init__(self, dataset: PackedLatentDataset, batch_size: int, shuffle: bool = True, drop_last: bool = False, seed: int = 0,
                 rank: int = 0, world: int = 1, pin_memory: Optional[bool] = None):
        self.ds, self.batch_size, self.shuffle, self.drop_last, self.seed = dataset, int(batch_size), shuffle, drop_last, seed
        self.rank, self.world, self.epoch = rank, world, 0
        self.pin = torch.cuda.is_available() if pin_memory is None else pin_memory

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _indices(self) -> np.ndarray:
        n = len(self.ds)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        per = n // self.world if self.drop_last else -(-n // self.world)
        if not self.drop_last and per * self.world > n:              # pad by wrapping, as DistributedSampler does
            order = np.concatenate([order, order[:per * self.world - n]])
        return order[self.rank:per * self.world:self.world]

    def __len__(self):
        m = len(self._indices())
        return m // self.batch_size if self.drop_last else -(-m // self.batch_size)

    def _tensor(self, arr: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.pin_memory() if self.pin else t

    def __iter__(self) -> Iterator[Tuple[object, torch.Tensor]]:
        idx = self._indices()
        self.epoch += 1
        for lo in range(0, len(idx), self.batch_size):
            sel = np.sort(idx[lo:lo + self.batch_size])              # ascending offsets: sequential page faults within a batch
            if len(sel) < self.batch_size and self.drop_last:
                break
            labels = self._tensor(self.ds.labels[sel])
            tgt = self._tensor(self.ds.target[sel])
            if not self.ds.inpainting:
                yield tgt, labels
            else:
                yield {'target_latents': tgt, 'source_latents': self._tensor(self.ds.source[sel]),
                       'mask_pixels': self._tensor(self.ds._mask_rows(sel))}, labels


def latent_loader(data_path: str, batch_size: int, n_classes=None, shuffle=True, seed=0, rank=0, world=1, num_workers=12):
    """The training loop's loader (train_flow.py:215-224): ``<data_path>.fcl`` / ``<data_path>/latents.fcl`` if a packed file
    exists, else the reference-format directory through a torch DataLoader exactly as upstream."""
    for cand in (str(data_path) + ".fcl", os.path.join(str(data_path), "latents.fcl")):
        if os.path.exists(cand):
            return PackedLatentLoader(PackedLatentDataset(cand), batch_size, shuffle=shuffle, seed=seed, rank=rank, world=world)
    from torch.utils.data import DataLoader
    ds = PreEncodedDataset(data_path, n_classes=n_classes)
    return DataLoader(dataset=ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, pin_memory=True,
                      persistent_workers=num_workers > 0)
