"""N2 on the GPU: the pre-encoding pipeline (preencode_data.py:34-42,84-181) through the HIP SD-VAE and VQVAE encoders -- every
latent that lands on disk equals the CPU oracle's encoding of its image, in the reference's directory layout and in rank-sharded
packed shards (two ranks rehearsed one after the other; the split itself is the CPU-tested `batch_idx % world`)."""
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import sdvae_oracle as vo
from oracle import vqvae_oracle as vq
from oracle.synth import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batches(n_batches, bsz, shape, n_classes, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand((bsz,) + shape, generator=g) * 2 - 1, torch.randint(0, n_classes, (bsz,), generator=g)) for _ in range(n_batches)]


def _check(files_ds, want, labels, tol):
    """every stored latent matches exactly one expected latent (order is free: the writer's file names are random) with its label"""
    got = [(files_ds[i][0], int(files_ds[i][1])) for i in range(len(files_ds))]
    assert len(got) == len(want)
    used = set()
    for z, lab in got:
        errs = [(rel_l2(z, w), j) for j, w in enumerate(want) if j not in used]
        e, j = min(errs)
        assert e < tol, e
        assert lab == int(labels[j])
        used.add(j)


def test_sdvae_preencode_files_and_rank_shards(tmp_path):
    from flocoder_amd.codecs import SD_VAE_Wrapper
    from flocoder_amd.data import PackedLatentDataset, PreEncodedDataset
    from flocoder_amd.preencode import merge_shards, process_dataset
    codec = SD_VAE_Wrapper(weights="random", seed=7).eval().to(DEV)
    vsd = {k[4:]: v.detach().cpu() for k, v in codec.state_dict().items()}
    batches = _batches(4, 3, (3, 64, 64), 5, 1)
    want = torch.cat([vo.encode_mean(vsd, x) for x, _ in batches])
    labels = torch.cat([y for _, y in batches])
    r = process_dataset(codec, batches, tmp_path / "files", DEV, n_classes=5)
    assert r["samples"] == 12
    ds = PreEncodedDataset(str(tmp_path / "files"))
    assert ds.n_classes == 5 and ds[0][0].shape == (4, 8, 8) and ds[0][0].device.type == "cpu"
    # class directories are indexed by path order of the digit names ("0".."4" here: identity)
    _check(ds, list(want), labels, 2e-4)
    shards = [process_dataset(codec, batches, tmp_path / "packed", DEV, n_classes=5, packed=True, rank=rk, world=2) for rk in range(2)]
    assert [s["samples"] for s in shards] == [6, 6]
    merge_shards([s["path"] for s in shards], str(tmp_path / "packed" / "latents.fcl"))
    pk = PackedLatentDataset(str(tmp_path / "packed" / "latents.fcl"))
    assert len(pk) == 12 and pk.shape == (4, 8, 8)
    _check(pk, list(want), labels, 2e-4)
    # rank 0 took batches 0 and 2, in order: packed shards keep the encode order
    s0 = PackedLatentDataset(shards[0]["path"])
    assert rel_l2(torch.stack([s0[i][0] for i in range(6)]), torch.cat([want[0:3], want[6:9]])) < 2e-4


def test_vqvae_preencode_with_quantize(tmp_path):
    from flocoder_amd.codecs import VQVAE
    from flocoder_amd.data import PreEncodedDataset
    from flocoder_amd.preencode import encode_batch, process_dataset
    g = load_golden("g9_vqvae")
    sd = synth_state_dict(g["gray_nd4_small_shapes"], 9)
    m = VQVAE(in_channels=1, hidden_channels=32, num_downsamples=4, internal_dim=32, vq_embedding_dim=4).eval()
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV)
    batches = [(torch.rand(2, 1, 128, 128, generator=torch.Generator().manual_seed(i)), torch.zeros(2, dtype=torch.long)) for i in range(3)]
    want = torch.cat([vq.encode(sd, x) for x, _ in batches])
    r = process_dataset(m, batches, tmp_path / "midi", DEV, n_classes=0, max_batches=2)
    assert r["samples"] == 4                                                  # the batch budget of preencode_data.py:100
    ds = PreEncodedDataset(str(tmp_path / "midi"), n_classes=0)   # the "00".."99" bucket directories are digits: class handling must be off
    assert ds.n_classes == 0 and sorted(p.parent.name for p in ds.files) == ["00", "00", "01", "01"]
    _check(ds, list(want[:4]), torch.zeros(4), 2e-5)
    z = encode_batch(m, batches[0][0], DEV)
    assert z.is_cuda and rel_l2(z.cpu(), want[:2]) < 2e-5
