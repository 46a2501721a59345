"""GPU parity of the VQVAE codec's encode / decode (through the C ABI) against the golden vectors the reference itself produced
(fixture g9, tools/make_golden.py) and against oracle/vqvae_oracle.py at other batch sizes / resolutions.
Tolerance rel-L2 <= 2e-5: ~25 stacked fp32 convolutions + GroupNorms; measured values are printed by -s."""
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import vqvae_oracle as vq
from oracle.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5
CFG = {"midi_vqgan": dict(in_channels=3, hidden_channels=256, num_downsamples=3, internal_dim=128, vq_embedding_dim=4),
       "gray_nd4_small": dict(in_channels=1, hidden_channels=32, num_downsamples=4, internal_dim=32, vq_embedding_dim=4)}


def build(tag):
    from flocoder_amd.codecs import VQVAE
    g = load_golden("g9_vqvae")
    sd = synth_state_dict(g[tag + "_shapes"], 9)
    m = VQVAE(**CFG[tag]).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k == "codebook_usage" or k.startswith("vq.") for k in missing)
    return m.to(DEV), sd, g


@pytest.mark.parametrize("tag", list(CFG))
def test_golden_encode_decode(tag):
    m, sd, g = build(tag)
    ic = CFG[tag]["in_channels"]
    x = torch.sigmoid(synth_input("g9.x." + tag, (1, ic, 128, 128), 9, scale=2.0))
    z = m.encode(x.to(DEV))
    assert tuple(z.shape) == tuple(g[tag + "_z"].shape)
    e1 = rel_l2(z.cpu(), g[tag + "_z"])
    zin = synth_input("g9.z." + tag, tuple(z.shape), 9)
    y = m.decode(zin.to(DEV))
    assert tuple(y.shape) == (1, ic, 128, 128)
    e2 = rel_l2(y.cpu(), g[tag + "_recon"])
    print(tag, "encode", e1, "decode", e2)
    assert e1 < TOL and e2 < TOL
    assert m.flops_per_sample(decode=True) > 0 and m.flops_per_sample(decode=False) > 0


def test_batch_and_resolution_vs_oracle():
    m, sd, _ = build("gray_nd4_small")
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(3, 1, 64, 128, generator=gen)
    z = m.encode(x.to(DEV))
    assert z.shape == (3, 4, 4, 8) and rel_l2(z.cpu(), vq.encode(sd, x)) < TOL
    zq = torch.randn(3, 4, 4, 8, generator=gen)
    y = m.decode(zq.to(DEV))
    assert y.shape == (3, 1, 64, 128) and rel_l2(y.cpu(), vq.decode(sd, zq)) < TOL
    assert rel_l2(m.decode(zq[2:3].to(DEV)).cpu(), y[2:3].cpu()) < 1e-5          # batch independence
    assert torch.equal(m.decode(zq.to(DEV)), y)                                    # bit-reproducible launches


def test_quantize_and_forward_inference_form():
    """VQVAE.quantize / forward on the inference form of ResidualVQ (third party, parity unpinned): the GPU kernel against the
    oracle's restatement, plus the structural properties any residual quantiser has."""
    from flocoder_amd.codecs import VQVAE
    g = load_golden("g9_vqvae")
    m = VQVAE(vq_num_embeddings=64, codebook_levels=3, **CFG["gray_nd4_small"]).eval()
    m.load_state_dict(synth_state_dict(g["gray_nd4_small_shapes"], 9), strict=False)
    gen = torch.Generator().manual_seed(1)
    sdv = {}
    for i in range(3):
        cb = torch.randn(1, 64, 4, generator=gen) * (0.5 ** i)
        m.vq.layers[i]._codebook.embed.copy_(cb)
        m.vq.layers[i]._codebook.initted.fill_(True)
        sdv[f"vq.layers.{i}._codebook.embed"] = cb
    m = m.to(DEV)
    z = torch.randn(3, 4, 8, 8, generator=gen)
    zq, loss = m.quantize(z.to(DEV))
    ref_q, ref_loss, ref_idx = vq.quantize(sdv, z)
    assert torch.equal(m.indices.cpu(), ref_idx) and m.indices.shape == (3 * 64, 3)
    assert rel_l2(zq.cpu(), ref_q) < 1e-6 and loss.shape == (1, 3) and float(loss.abs().sum()) == 0.0
    assert float((z - zq.cpu()).norm()) < float(z.norm())                        # quantisation reduces the residual
    zq2, _ = m.quantize(zq)                                                      # re-quantising stays within the codebook lattice
    assert float((zq2 - zq).abs().max()) < float(zq.abs().max())
    flat = z.permute(0, 2, 3, 1).reshape(-1, 4).to(DEV)                          # the `.vq(flat)` protocol of sampling.py:281
    q2, idx2, _ = m.vq(flat)
    assert torch.equal(idx2.cpu(), ref_idx) and rel_l2(q2.cpu(), ref_q.permute(0, 2, 3, 1).reshape(-1, 4)) < 1e-6
    x = torch.rand(2, 1, 128, 128, generator=gen)
    recon, closs, stats = m(x.to(DEV), get_stats=True)
    sd_all = synth_state_dict(g["gray_nd4_small_shapes"], 9)
    zr = vq.encode(sd_all, x)
    assert recon.shape == x.shape and float(closs) == 0.0 and stats["codebook_mean_dist"] > 0
    assert rel_l2(recon.cpu(), vq.decode(sd_all, vq.quantize(sdv, zr)[0])) < 1e-3   # an encoder rounding flip may move one codeword
    m.train()
    with pytest.raises(NotImplementedError):
        m(x.to(DEV))


def test_protocol_errors():
    m, _, _ = build("gray_nd4_small")
    with pytest.raises(NotImplementedError):
        m.quantize(torch.zeros(1, 4, 8, 8, device=DEV))
    with pytest.raises(NotImplementedError):
        m.decode(torch.zeros(1, 4, 8, 8, device=DEV), noise_strength=0.05)
    with pytest.raises(ValueError):
        m.encode(torch.zeros(1, 3, 64, 64, device=DEV))
    with pytest.raises(RuntimeError):
        m.encode(torch.zeros(1, 1, 64, 64))
    with pytest.raises(ValueError, match="latent pixels"):                          # latent side < 4: no plan
        m.encode(torch.zeros(1, 1, 32, 32, device=DEV))


@pytest.mark.timeout(1500)
def test_full_midi_inpainting_shape_vs_oracle_and_batch_independence():
    """The codec of BASELINE config 5 at its FULL midi_inpainting.yaml shape (codecs.py:395-574 with in_channels=1, 4 downsamples,
    hidden 256, internal 128 -> 4x8x8 latents; 563 M parameters): B=1 against the CPU oracle, then B=64 at 128x128 (the size SURVEY
    8(d) config (5) names) through batch independence -- every sample of the big batch equals its own B=1 pass.  Weights: seeded
    synthetic tensors over the library's own parameter table (names / shapes as the reference's state_dict, pinned for two other
    configurations by fixture g9)."""
    from flocoder_amd.codecs import VQVAE
    cfg = dict(in_channels=1, hidden_channels=256, num_downsamples=4, internal_dim=128, vq_embedding_dim=4)
    m = VQVAE(codebook_levels=2, vq_num_embeddings=32, **cfg).eval()
    shapes = {name: list(shape) for name, shape, _ in m._table}
    nparam = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    assert nparam > 5.0e8, nparam
    sd = synth_state_dict(shapes, 21)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not (set(missing) & set(shapes))     # (what stays missing: the quantiser and the NoiseInjection holders, neither on this path)
    m = m.to(DEV)
    gen = torch.Generator().manual_seed(8)
    x1 = torch.rand(1, 1, 128, 128, generator=gen)
    z1 = m.encode(x1.to(DEV))
    assert z1.shape == (1, 4, 8, 8)
    e_enc = rel_l2(z1.cpu(), vq.encode(sd, x1))
    zq1 = torch.randn(1, 4, 8, 8, generator=gen)
    y1 = m.decode(zq1.to(DEV))
    e_dec = rel_l2(y1.cpu(), vq.decode(sd, zq1))
    print("full midi_inpainting VQVAE (%.0f M params): encode %.2e decode %.2e" % (nparam / 1e6, e_enc, e_dec))
    assert e_enc < 5e-5 and e_dec < 5e-5
    # B = 64 at 128 x 128
    x = torch.rand(64, 1, 128, 128, generator=gen)
    x[17] = x1[0]
    z = m.encode(x.to(DEV))
    assert z.shape == (64, 4, 8, 8) and torch.isfinite(z).all()
    assert rel_l2(z[17:18].cpu(), z1.cpu()) < 1e-5
    for k in (0, 40, 63):
        assert rel_l2(z[k:k + 1].cpu(), m.encode(x[k:k + 1].to(DEV)).cpu()) < 1e-5
    zq = torch.randn(64, 4, 8, 8, generator=gen)
    zq[5] = zq1[0]
    y = m.decode(zq.to(DEV))
    assert y.shape == (64, 1, 128, 128) and torch.isfinite(y).all()
    assert rel_l2(y[5:6].cpu(), y1.cpu()) < 1e-5
    for k in (0, 33, 63):
        assert rel_l2(y[k:k + 1].cpu(), m.decode(zq[k:k + 1].to(DEV)).cpu()) < 1e-5
    assert torch.equal(m.decode(zq.to(DEV)), y)                                    # bit-reproducible
