"""CPU checks of the boundary code around the hot path: YAML composition and ldcfg precedence, the codec
factory's branches and errors (codecs.py:668-741), the parameter-free codecs, checkpoint dict layout."""
import os

import pytest
import torch

from conftest import ROOT

CFG = os.path.join(ROOT, "configs")


def test_compose_flowers_sd_and_ldcfg_precedence():
    from flocoder_amd.general import load_config, ldcfg
    c = load_config(os.path.join(CFG, "flowers_sd.yaml"))
    assert c.codec.choice == "sd" and c.flow.unet.n_classes == 102 and c.preencoding.augs_per == 128
    assert c.flow.dim_mults == [1, 2, 4, 8] and c.flow.batch_size == 256
    # flow -> preencoding -> codec -> top level: batch_size comes from flow (256), not preencoding (32) or codec (64)
    assert ldcfg(c, "batch_size", verbose=False) == 256
    # image_size resolves to codec.image_size before the top-level key (Q21); both are 128 by default ...
    assert ldcfg(c, "image_size", verbose=False) == 128
    # ... and a top-level override alone does not change what ldcfg returns
    c2 = load_config(os.path.join(CFG, "flowers_sd.yaml"), ["image_size=256"])
    assert c2.image_size == 256 and ldcfg(c2, "image_size", verbose=False) == 128
    c3 = load_config(os.path.join(CFG, "flowers_sd.yaml"), ["codec.image_size=256", "image_size=256", "+flow.lambda_lowres=0.2"])
    assert ldcfg(c3, "image_size", verbose=False) == 256 and c3.flow.get("lambda_lowres", 0.1) == 0.2
    # a miss returns None unless supply_defaults (general.py:69-70)
    assert ldcfg(c, "nope", default=7, verbose=False) is None
    assert ldcfg(c, "nope", default=7, supply_defaults=True, verbose=False) == 7
    assert ldcfg(c.flow.unet, "n_classes", 0, verbose=False) == 102


def test_other_configs_compose():
    from flocoder_amd.general import load_config
    stl = load_config(os.path.join(CFG, "stl_sd.yaml"))
    assert stl.flow.unet.n_classes == 10 and stl.codec.choice == "sd" and stl.batch_size == 64
    midi = load_config(os.path.join(CFG, "midi_inpainting.yaml"))
    assert midi.codec.in_channels == 1 and midi.codec.num_downsamples == 4 and midi.flow.unet.n_classes == 0
    assert midi.preencoding.quantize is True and midi.data == "~/datasets/POP909_images"


def test_config_name_is_mandatory_and_argv_rewrite(monkeypatch, tmp_path):
    import sys
    from flocoder_amd import general as G
    with pytest.raises(ValueError):
        G.config_from_argv([])
    monkeypatch.setattr(sys, "argv", ["prog", "--config-name", os.path.join(CFG, "stl_sd.yaml"), "flow.batch_size=8"])
    G.handle_config_path()
    assert sys.argv[1] == f"--config-path={CFG}" and sys.argv[2] == "--config-name=stl_sd"
    c = G.config_from_argv()
    assert c.flow.batch_size == 8 and c.flow.unet.n_classes == 10


def test_setup_codec_branches():
    from flocoder_amd.codecs import NoOpAE, SimpleResizeAE, setup_codec
    from flocoder_amd.general import load_config
    c = load_config(os.path.join(CFG, "flowers_resize.yaml"))
    codec = setup_codec(c, "cpu")
    assert isinstance(codec, SimpleResizeAE) and codec.in_channels == 3
    x = torch.rand(2, 3, 128, 128)
    z = codec.encode(x)
    assert z.shape == (2, 4, 16, 16) and torch.allclose(z[:, 3], z[:, :3].mean(1))
    assert codec.decode(z).shape == (2, 3, 128, 128)
    recon, loss = codec(x)
    assert recon.shape == x.shape and loss == 0.0
    c.codec.choice = "noop"
    noop = setup_codec(c, "cpu")
    assert isinstance(noop, NoOpAE) and noop.encode(x).shape == (2, 4, 16, 16)      # 'noop' resizes
    c.codec.choice = None
    assert isinstance(setup_codec(c, "cpu"), NoOpAE)
    # SD codec: local weights only, never a download
    sd = load_config(os.path.join(CFG, "flowers_sd.yaml"))
    os.environ.pop("FLOCODER_SD_VAE_PATH", None)
    with pytest.raises(FileNotFoundError, match="never touches the network"):
        setup_codec(sd, "cpu")
    # VQVAE branch: checkpoint lookup errors as upstream (codecs.py:725-728)
    vq = load_config(os.path.join(CFG, "midi_vqgan.yaml"))
    with pytest.raises(FileNotFoundError, match="vqgan_best.pt"):
        setup_codec(vq, "cpu")


def test_vqvae_mirror_state_dict_and_checkpoint(tmp_path):
    """The VQVAE mirror carries the reference's state_dict keys / shapes (fixture g9 lists them from the reference itself), loads a
    reference-style checkpoint through setup_codec (strict=False: vq.* ignored), and has no CPU compute path."""
    from conftest import load_golden
    from flocoder_amd.codecs import VQVAE, setup_codec
    from flocoder_amd.general import load_config
    from oracle.synth import synth_state_dict
    g = load_golden("g9_vqvae")
    vq = load_config(os.path.join(CFG, "midi_vqgan.yaml"))
    codec = setup_codec(vq, "cpu", load_checkpoint=False)
    assert isinstance(codec, VQVAE) and codec.in_channels == 3 and not codec.training
    ref = g["midi_vqgan_shapes"]
    own = {k: tuple(v.shape) for k, v in codec.state_dict().items() if k != "codebook_usage" and not k.startswith("vq.")}
    assert own == {k: tuple(v) for k, v in ref.items()}
    assert list(own) != [] and sum(1 for _ in codec.parameters()) == len(ref)
    # ResidualVQ buffers under vector_quantize_pytorch's names (codecs.py:456-467), 4 per level
    vqk = [k for k in codec.state_dict() if k.startswith("vq.")]
    assert len(vqk) == 4 * codec.codebook_levels and "vq.layers.0._codebook.embed" in vqk
    assert tuple(codec.state_dict()["vq.layers.0._codebook.embed"].shape) == (1, codec.vq_num_embeddings, 4)
    sd = synth_state_dict(ref, 9)
    sd["vq.layers.0._codebook.embed"] = torch.ones(1, codec.vq_num_embeddings, 4)     # real checkpoints carry the codebooks
    sd["vq.layers.0._codebook.initted"] = torch.tensor(True)
    path = str(tmp_path / "vqgan_best.pt")
    torch.save({"model_state_dict": sd, "epoch": 3}, path)
    vq.vqgan_checkpoint = path
    codec = setup_codec(vq, "cpu")
    assert torch.equal(codec.state_dict()["encoder.0.conv1.weight"], sd["encoder.0.conv1.weight"])
    assert torch.equal(codec.state_dict()["decoder.layers.0.q_proj.bias"], sd["decoder.layers.0.q_proj.bias"])
    assert bool(codec.vq.layers[0]._codebook.initted) and float(codec.vq.codebooks[0].sum()) == codec.vq_num_embeddings * 4
    with pytest.raises(RuntimeError, match="no CPU path"):
        codec.encode(torch.zeros(1, 3, 128, 128))
    with pytest.raises(NotImplementedError):
        codec.quantize(torch.zeros(1, 4, 16, 16))
    small = VQVAE(in_channels=1, hidden_channels=32, num_downsamples=4, internal_dim=32, vq_embedding_dim=4)
    ref = g["gray_nd4_small_shapes"]
    assert {k: tuple(v.shape) for k, v in small.state_dict().items() if k != "codebook_usage" and not k.startswith("vq.")} == {k: tuple(v) for k, v in ref.items()}


def test_sd_vae_wrapper_state_dict_layout_and_no_cpu_path():
    from flocoder_amd.codecs import SD_VAE_Wrapper
    from oracle import sdvae_oracle as vo
    w = SD_VAE_Wrapper(weights="random", seed=1)
    sd = w.state_dict()
    ref = vo.shapes()
    assert set(sd) == {"vae." + k for k in ref} and all(tuple(sd["vae." + k].shape) == tuple(v) for k, v in ref.items())
    assert sum(v.numel() for v in sd.values()) == 83_653_863                 # the published sd-vae-ft-mse parameter count
    with pytest.raises(RuntimeError, match="no CPU path"):
        w.encode(torch.zeros(1, 3, 64, 64))
    # legacy attention names + [C,C,1,1] weights (as in the published checkpoint) load too
    legacy = {}
    for k, v in sd.items():
        k = k[4:]
        for new, old in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            if new in k and "attentions" in k:
                k = k.replace(new, old)
                v = v.reshape(*v.shape, 1, 1) if v.dim() == 2 else v
        legacy[k] = v
    w2 = SD_VAE_Wrapper(weights=legacy)
    assert all(torch.equal(a, b) for a, b in zip(w.state_dict().values(), w2.state_dict().values()))


def test_checkpoint_dict_layout(tmp_path):
    from flocoder_amd.general import load_flow_model, save_checkpoint, load_config
    from flocoder_amd.unet import Unet
    torch.manual_seed(3)
    m = Unet(dim=8, channels=4, n_classes=5)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    cfg = load_config(os.path.join(CFG, "stl_sd.yaml"))
    path = save_checkpoint(m, epoch=25, optimizer=opt, prefix="flow_", ckpt_dir=str(tmp_path), config=cfg)
    assert path.endswith("flow__25.pt")
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model_state_dict", "epoch", "optimizer_state_dict", "config"} and ck["epoch"] == 25
    m2 = load_flow_model(path, cfg, "cpu")
    assert m2.dim == 8 and m2.class_condition and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_setup_codec_builds_natten_blocks_when_the_checkpoint_has_them(tmp_path):
    """upstream a NATTEN-trained VQGAN silently loses its attention without the package; here the blocks are native and are
    built whenever the checkpoint carries their weights (natten_layout from the config, default 2), unless no_natten asks otherwise."""
    from flocoder_amd.codecs import VQVAE, setup_codec
    from flocoder_amd.general import Config, _wrap
    kw = dict(in_channels=1, hidden_channels=32, num_downsamples=3, internal_dim=32, vq_embedding_dim=4)
    src = VQVAE(natten_layout=1, decoder_nonlocal=True, **kw)
    sd = src.state_dict()
    assert any(k.endswith(".attn.qkv.weight") for k in sd) and sd["encoder.2.attn.qkv.weight"].shape == (96 * 2, 64)
    path = str(tmp_path / "vq.pt")
    torch.save({"model_state_dict": sd}, path)
    cfg = _wrap({"codec": {"choice": "vqgan", "checkpoint": path, "codebook_levels": 3, "vq_num_embeddings": 512, "commitment_weight": 0.5, **kw}})
    c = setup_codec(cfg, "cpu")
    assert c.natten_layout == 2 and torch.equal(c.state_dict()["encoder.2.attn.qkv.weight"], sd["encoder.2.attn.qkv.weight"])
    cfg.codec["natten_layout"] = 1
    assert setup_codec(cfg, "cpu").natten_layout == 1
    c0 = setup_codec(cfg, "cpu", no_natten=True)
    assert c0.natten_layout == 0 and not any(k.endswith(".attn.qkv.weight") for k in c0.state_dict())
    plain = VQVAE(**kw)
    torch.save({"model_state_dict": plain.state_dict()}, path)
    assert setup_codec(cfg, "cpu").natten_layout == 0
