"""Pins oracle/flow_oracle.py to the golden vectors produced by the reference itself
(tools/make_golden.py).  CPU only.  Tolerance: fp32 rel-L2 <= 2e-6 per op / forward
(same ops, same order as the reference, so the difference is summation-order noise);
trajectories <= 2e-5; integer outputs exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import flow_oracle as fo
from oracle.synth import synth_input, synth_state_dict

torch.set_num_threads(4)
TOL = 2e-6


def test_g1_sinusoidal():
    g = load_golden("g1_sinusoidal")
    t = torch.from_numpy(g["times"])
    assert rel_l2(fo.sinusoidal_embedding(t, 32), g["emb32"]) < 1e-6
    assert rel_l2(fo.sinusoidal_embedding(t, 16), g["emb16"]) < 1e-6


def test_g2_modules():
    g = load_golden("g2_modules")
    B = 2
    sds = {tag: synth_state_dict(sh, 11) for tag, sh in g["shapes"].items()}
    pre = lambda sd, p: {p + "." + k: v for k, v in sd.items()}
    x = lambda tag, shape: synth_input("g2." + tag, shape, 11)

    sc = synth_input("g2.scale", (B, 64, 1, 1), 11, 0.3)
    sh = synth_input("g2.shift", (B, 64, 1, 1), 11, 0.3)
    xb = x("block", (B, 32, 16, 16))
    assert rel_l2(fo.block(pre(sds["block"], "b"), "b", xb, 4, (sc, sh)), g["block"]) < TOL
    assert rel_l2(fo.block(pre(sds["block"], "b"), "b", xb, 4), g["block_noss"]) < TOL
    for tag, shape in (("resnet_same", (B, 32, 32, 32)), ("resnet_proj", (B, 96, 16, 16))):
        temb = synth_input("g2.temb." + tag, (B, 256), 11)
        y = fo.resnet_block(pre(sds[tag], "r"), "r", x(tag, shape), temb, 4)
        assert rel_l2(y, g[tag]) < TOL, tag
    for tag, shape in (("linattn", (B, 32, 32, 32)), ("linattn_small", (B, 128, 4, 4))):
        assert rel_l2(fo.linear_attention(pre(sds[tag], "a"), "a", x(tag, shape)), g[tag]) < TOL, tag
    assert rel_l2(fo.full_attention(pre(sds["attn"], "a"), "a", x("attn", (B, 256, 4, 4))), g["attn"]) < TOL
    assert rel_l2(fo.space_to_depth_conv(pre(sds["down"], "d"), "d.1", x("down", (B, 32, 32, 32))), g["down"]) < TOL
    xu = torch.nn.functional.interpolate(x("up", (B, 64, 8, 8)), scale_factor=2, mode="nearest")
    assert rel_l2(fo._conv(pre(sds["up"], "u"), "u.1", xu, padding=1), g["up"]) < TOL


@pytest.mark.parametrize("tag,dim,seed,B", [("d32c102", 32, 1, 2), ("d16c10", 16, 2, 3), ("d8mask", 8, 3, 2)])
def test_g3_unet_forward(tag, dim, seed, B):
    g = load_golden("g3_unet_" + tag)
    sd = synth_state_dict(g["shapes"], seed)
    x = synth_input("g3.x." + tag, (B, 4, dim, dim), seed)
    t = torch.from_numpy(g["t"])
    meta = fo.unet_meta(sd)
    assert meta["dim"] == dim and meta["chans"] == [dim, dim, 2 * dim, 4 * dim, 8 * dim]
    if "cls" in g:
        cls = torch.from_numpy(g["cls"])
        assert rel_l2(fo.unet_forward(sd, x, t, {"class_cond": cls}), g["v_class"]) < TOL
        assert rel_l2(fo.unet_forward(sd, x, t, {"class_cond": None}), g["v_noclass"]) < TOL
        assert rel_l2(fo.unet_forward(sd, x, t, None), g["v_none"]) < TOL
        assert rel_l2(g["v_class"], g["v_noclass"]) > 1e-3      # the class path is live
    if "mask" in g:
        mask = torch.from_numpy(g["mask"])
        assert rel_l2(fo.unet_forward(sd, x, t, {"class_cond": None, "mask_cond": mask}), g["v_mask"]) < TOL
        assert rel_l2(fo.unet_forward(sd, x, t, {"mask_cond": torch.ones_like(mask)}), g["v_ones"]) < TOL
        assert rel_l2(fo.unet_forward(sd, x, t, None), g["v_none"]) < TOL
        assert rel_l2(g["v_mask"], g["v_none"]) > 1e-3
        assert rel_l2(g["v_ones"], g["v_none"]) > 1e-3          # per-scale injections stay on when mask == 1


def test_g4_time_grids():
    g = load_golden("g4_timegrids")
    for n in (3, 5, 16, 64, 100):
        assert np.array_equal(fo.rk4_time_grid(n).numpy(), g[f"rk4_{n}"]), n
    assert np.array_equal(fo.warp_time(torch.from_numpy(g["rand_in"])).numpy(), g["rand_out"])
    assert np.allclose(g["rk4_5"], [0, .34375, .5, .65625, 1])           # SURVEY Q3 probe
    with pytest.raises(ValueError):
        fo.warp_time(torch.zeros(2), s=1.6)


def test_g5_rk4_and_euler_trajectories():
    g = load_golden("g5_trajectories")
    sd = synth_state_dict(g["shapes"], 5)
    src = synth_input("g5.src", (2, 4, 16, 16), 5)
    cls = torch.from_numpy(g["cls"])
    for cfg in (0, 3):
        lat, nfe = fo.generate_latents_rk4(sd, src.clone(), 5, {"class_cond": cls}, float(cfg))
        assert nfe == int(g[f"rk4_n5_cfg{cfg}_nfe"]) == 20
        assert rel_l2(lat, g[f"rk4_n5_cfg{cfg}"]) < 2e-5, cfg
    assert rel_l2(g["rk4_n5_cfg0"], g["rk4_n5_cfg3"]) > 1e-3
    lat, _ = fo.generate_latents_rk4(sd, src.clone(), 4, {}, 3.0)
    assert rel_l2(lat, g["rk4_n4_nocond"]) < 2e-5
    init = synth_input("g5.init", (2, 4, 16, 16), 5)
    lat, nfe = fo.generate_latents_rk4(sd, src.clone(), 8, {"class_cond": cls}, 3.0, init_latents=init, init_strength=0.5)
    assert nfe == int(g["rk4_n8_init05_nfe"]) == 16
    assert rel_l2(lat, g["rk4_n8_init05"]) < 2e-5
    for n in (4, 16):
        lat, nfe = fo.euler_sampler(sd, src, n, cls)
        assert nfe == n and rel_l2(lat, g[f"euler_n{n}"]) < 2e-5, n


def test_g7_ot_pairing():
    g = load_golden("g7_ot")
    for B, D in ((8, 64), (64, 64), (256, 1024)):
        s, t = synth_input(f"g7.s{B}", (B, D), 7), synth_input(f"g7.t{B}", (B, D), 7)
        perm = fo.ot_pairing_greedy(s, t)
        assert perm.dtype == torch.int64 and np.array_equal(perm.numpy(), g[f"perm_{B}_{D}"])
        assert sorted(perm.tolist()) == list(range(B))
    perm = fo.ot_pairing_greedy(torch.from_numpy(g["tie_src"]), torch.from_numpy(g["tie_tgt"]))
    assert np.array_equal(perm.numpy(), g["tie_perm"])


def test_g8_mask_encoder():
    g = load_golden("g8_mask_encoder")
    sd = synth_state_dict(g["shapes"], 8)
    mp = (synth_input("g8.mask", (2, 1, 128, 128), 8) > 0.3).float()
    ml = fo.mask_encoder_forward(sd, mp)
    assert ml.shape == (2, 4, 8, 8) and rel_l2(ml, g["mask_latents"]) < TOL
    assert rel_l2(fo.mask_encoder_forward(sd, mp.bool()), g["mask_latents_bool"]) < TOL
    src, noise = synth_input("g8.src", (2, 4, 8, 8), 8), synth_input("g8.noise", (2, 4, 8, 8), 8)
    assert rel_l2(fo.mask_blending(src, ml, noise), g["blended"]) < TOL


@pytest.mark.parametrize("tag,in_ch", [("midi_vqgan", 3), ("gray_nd4_small", 1)])
def test_g9_vqvae_encode_decode(tag, in_ch):
    """oracle/vqvae_oracle.py against the reference's own VQVAE.encode / .decode (NATTEN-less, eval)."""
    from oracle import vqvae_oracle as vq
    g = load_golden("g9_vqvae")
    shapes = g[tag + "_shapes"]
    sd = synth_state_dict(shapes, 9)
    x = torch.sigmoid(synth_input("g9.x." + tag, (1, in_ch, 128, 128), 9, scale=2.0))
    z = vq.encode(sd, x)
    assert tuple(z.shape) == tuple(g[tag + "_z"].shape) and rel_l2(z, g[tag + "_z"]) < 5e-6
    y = vq.decode(sd, synth_input("g9.z." + tag, tuple(z.shape), 9))
    assert tuple(y.shape) == (1, in_ch, 128, 128) and rel_l2(y, g[tag + "_recon"]) < 5e-6


def test_g10_train_step():
    """oracle/train_oracle.py against three consecutive training steps of the reference Unet under torch.optim.Adam (step 2
    without conditioning: class_cond_mlp.* has no gradient and is skipped by Adam).  Gradients: per-tensor sum / abs-sum within
    1e-4 relative of the abs-sum (fp32 backward, different reduction order); parameters after the step within 2e-6."""
    from oracle import train_oracle as to
    g = load_golden("g10_train_step")
    names = list(g["names"])
    sd = synth_state_dict(g["shapes"], 10)
    assert list(sd) == names
    state = to.new_state(sd)
    cls = torch.from_numpy(g["cls"])
    for step in (1, 2, 3):
        src, tgt = synth_input(f"g10.src{step}", (8, 4, 16, 16), 10), synth_input(f"g10.tgt{step}", (8, 4, 16, 16), 10)
        t = to.train_time(torch.sigmoid(synth_input(f"g10.u{step}", (8,), 10, scale=1.5)))
        assert np.array_equal(t.numpy(), g[f"s{step}_t"])
        cond = {"class_cond": cls, "mask_cond": None} if step != 2 else None
        loss, grads, v = to.loss_and_grads(sd, src, tgt, t, cond)
        assert rel_l2(v, g[f"s{step}_v"]) < TOL
        assert abs(float(loss) - float(g[f"s{step}_loss"])) < 1e-6 * abs(float(g[f"s{step}_loss"]))
        gsum, gabs = g[f"s{step}_gsum"], g[f"s{step}_gabs"]
        for i, k in enumerate(names):
            if np.isnan(gabs[i]):
                assert grads[k] is None and k.startswith("class_cond_mlp"), k
                continue
            assert abs(float(grads[k].double().abs().sum()) - gabs[i]) <= 1e-4 * gabs[i] + 1e-12, k
            assert abs(float(grads[k].double().sum()) - gsum[i]) <= 1e-4 * gabs[i] + 1e-12, k
        for k in g["small"]:
            if f"s{step}_grad_{k}" in g:
                assert rel_l2(grads[k], g[f"s{step}_grad_{k}"]) < 1e-5, k
        total = to.adam_ema_step(sd, grads, state)
        assert abs(float(total) - float(g[f"s{step}_norm"])) < 1e-5 * float(g[f"s{step}_norm"])
        for i, k in enumerate(names):
            assert abs(float(sd[k].double().sum()) - g[f"s{step}_psum"][i]) <= 2e-6 * g[f"s{step}_pabs"][i] + 1e-9, k
            assert abs(float(state["ema"][k].double().sum()) - g[f"s{step}_esum"][i]) <= 2e-6 * g[f"s{step}_eabs"][i] + 1e-9, k
        for k in g["small"]:
            assert rel_l2(sd[k], g[f"s{step}_param_{k}"]) < 1e-6, k
    assert state["step"]["class_cond_mlp.0.weight"] == 2 and state["step"]["init_conv.weight"] == 3
