"""Pre-encoded latent datasets, CPU only.  `PreEncodedDataset` against what the reference's own class reported for the
same directory tree (fixture g11: file -> label maps, class indexing by PATH order, class-free modes, inpainting dict items), then the
packed format: lossless round trip, identical items, loader batches that partition the data across ranks."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from tools.make_golden import make_latent_tree


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("latents"))
    make_latent_tree(root)
    return root


@pytest.mark.parametrize("tag,sub,kw", [("classes", "cls", {}), ("classes_off", "cls", {"n_classes": 0}), ("subdirs", "sub", {}),
                                        ("flat", "flat", {}), ("inpaint", "inp", {})])
def test_preencoded_dataset_matches_reference(tree, tag, sub, kw):
    from flocoder_amd.data import PreEncodedDataset
    g = load_golden("g11_latent_dataset")["layout"][tag]
    ds = PreEncodedDataset(os.path.join(tree, sub), **kw)
    assert ds.n_classes == g["n_classes"] and bool(ds.has_classes) == g["has_classes"] and len(ds) == g["len"]
    got = {os.path.relpath(str(f), os.path.join(tree, sub)): int(l) for f, l in zip(ds.files, ds._labels)}
    assert got == g["labels"]
    assert {str(k): v for k, v in getattr(ds, "class_to_idx", {}).items()} == g["class_to_idx"]
    if tag == "classes":
        assert ds.class_to_idx[10] == 2 and ds.class_to_idx[2] == 4          # '10' sorts before '2': path order, not numeric (data.py:330)
    item, lab = ds[0]
    assert lab.dtype == torch.long and lab.dim() == 0
    if tag == "inpaint":
        assert sorted(item.keys()) == g["item_keys"] and str(item["mask_pixels"].dtype) == g["mask_dtype"] == "torch.bool"
    else:
        assert item.shape == (4, 8, 8) and item.dtype == torch.float32
        assert ds[0][0] is item                                               # second access comes from the cache


@pytest.mark.parametrize("sub", ["cls", "inp", "flat"])
def test_packed_format_round_trip(tree, tmp_path, sub):
    from flocoder_amd.data import PackedLatentDataset, PreEncodedDataset, pack_latents
    src = PreEncodedDataset(os.path.join(tree, sub))
    out = str(tmp_path / f"{sub}.fcl")
    info = pack_latents(src, out)
    ds = PackedLatentDataset(out)
    assert len(ds) == len(src) == info["count"] and ds.n_classes == src.n_classes and ds.shape == (4, 8, 8)
    assert os.path.getsize(out) == info["bytes"]
    for i in range(len(src)):
        a, la = src[i]
        b, lb = ds[i]
        assert int(la) == int(lb) and lb.dtype == torch.long
        if isinstance(a, dict):
            assert set(a) == set(b)
            for k in a:
                assert torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k   # bool masks survive the bit packing
        else:
            assert torch.equal(a, b)
    with pytest.raises(ValueError, match="not a packed latent file"):
        bad = tmp_path / "bad.fcl"
        bad.write_bytes(b"x" * 128)
        PackedLatentDataset(str(bad))


def test_packed_loader_partitions_and_collates(tree, tmp_path):
    from torch.utils.data import DataLoader
    from flocoder_amd.data import PackedLatentDataset, PackedLatentLoader, PreEncodedDataset, latent_loader, pack_latents
    src = PreEncodedDataset(os.path.join(tree, "cls"))
    out = os.path.join(tree, "cls.fcl")
    pack_latents(src, out)
    ds = PackedLatentDataset(out)
    n = len(ds)
    key = lambda t: round(float(t.double().sum()), 6)
    want = sorted(key(src[i][0]) for i in range(n))
    seen = []
    for rank in range(2):                                                      # two ranks, same seed: disjoint shares of one permutation
        ld = PackedLatentLoader(ds, batch_size=3, seed=5, rank=rank, world=2, pin_memory=False)
        batches = list(ld)
        assert len(batches) == len(ld)
        for x, y in batches:
            assert x.dim() == 4 and x.shape[1:] == (4, 8, 8) and y.dtype == torch.long and y.shape == (x.shape[0],)
            seen += [key(t) for t in x]
    assert sorted(set(seen)) == sorted(set(want)) and len(seen) == -(-n // 2) * 2      # everything once (+ wrap-around padding)
    ld = PackedLatentLoader(ds, batch_size=4, seed=1, pin_memory=False)
    e0 = torch.cat([y for _, y in ld]); e1 = torch.cat([y for _, y in ld])
    assert sorted(e0.tolist()) == sorted(e1.tolist()) == sorted(int(src[i][1]) for i in range(n))   # each epoch a full pass
    # collation equals torch's default_collate of the same samples (what the reference's DataLoader hands to batch_to_data)
    inp = PreEncodedDataset(os.path.join(tree, "inp"))
    pack_latents(inp, os.path.join(tree, "inp.fcl"))
    x, y = next(iter(PackedLatentLoader(PackedLatentDataset(os.path.join(tree, "inp.fcl")), batch_size=4, shuffle=False, pin_memory=False)))
    xr, yr = next(iter(DataLoader(inp, batch_size=4, shuffle=False)))
    assert torch.equal(y, yr) and all(torch.equal(x[k], xr[k]) and x[k].dtype == xr[k].dtype for k in xr)
    assert isinstance(latent_loader(os.path.join(tree, "cls"), 4, num_workers=0), PackedLatentLoader)   # picks up <dir>.fcl
    assert isinstance(latent_loader(os.path.join(tree, "flat"), 2, num_workers=0), DataLoader)


def test_preencode_pipeline_with_resize_codec(tmp_path):
    """process_dataset (preencode_data.py:84-181) end to end with the CPU-capable SimpleResizeAE: reference directory layout when
    unpacked, one memory-mappable shard per rank when packed, same latents either way."""
    from flocoder_amd.codecs import SimpleResizeAE
    from flocoder_amd.data import PackedLatentDataset, PreEncodedDataset
    from flocoder_amd.preencode import merge_shards, process_dataset
    codec = SimpleResizeAE(latent_shape=(4, 8, 8)).eval()
    g = torch.Generator().manual_seed(0)
    batches = [(torch.rand(5, 3, 32, 32, generator=g), torch.randint(0, 3, (5,), generator=g)) for _ in range(4)]
    want = torch.cat([codec.encode(x) for x, _ in batches])
    r = process_dataset(codec, batches, tmp_path / "files", "cpu", n_classes=3)
    assert r["samples"] == 20 and r["bytes"] > 0
    ds = PreEncodedDataset(str(tmp_path / "files"))
    assert len(ds) == 20 and ds.n_classes == 3
    assert all(p.parent.name in "012" and p.name.startswith("sample_") and p.suffix == ".pt" for p in ds.files)
    key = lambda t: round(float(t.double().sum()), 5)
    assert sorted(key(ds[i][0]) for i in range(20)) == sorted(key(t) for t in want)
    labels = torch.cat([y for _, y in batches])
    assert sorted((key(ds[i][0]), int(ds[i][1])) for i in range(20)) == sorted((key(t), int(l)) for t, l in zip(want, labels))
    # packed, two ranks -> two shards -> merged file holding every sample once
    shards = [process_dataset(codec, batches, tmp_path / "packed", "cpu", n_classes=3, packed=True, rank=r_, world=2) for r_ in range(2)]
    assert [s["samples"] for s in shards] == [10, 10]
    info = merge_shards([s["path"] for s in shards], str(tmp_path / "packed" / "latents.fcl"))
    pk = PackedLatentDataset(str(tmp_path / "packed" / "latents.fcl"))
    assert info["count"] == len(pk) == 20 and pk.n_classes == 3
    assert sorted((key(pk[i][0]), int(pk[i][1])) for i in range(20)) == sorted((key(t), int(l)) for t, l in zip(want, labels))
    r = process_dataset(codec, batches, tmp_path / "budget", "cpu", n_classes=0, max_batches=2)
    assert r["samples"] == 10 and sorted(p.name for p in (tmp_path / "budget").iterdir()) == ["00", "01"]
