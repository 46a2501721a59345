#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *reference* (read-only at /root/reference).

Run in the build container only (``python tools/make_golden.py``); the GPU box never sees
the reference.  The reference's own files are imported, never copied; the only stand-ins
are for third-party modules that are absent here and that the hot path never executes
(omegaconf via general.py:5; wandb/torchvision/viz/metrics via sampling.py:3-11).
Weights and inputs come from oracle/synth.py so the tests can regenerate them.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FLOCODER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
from oracle.synth import synth_input, synth_state_dict  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def _stand_in(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stand_in("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    _stand_in("wandb")
    tv = _stand_in("torchvision")
    tv.transforms = _stand_in("torchvision.transforms", ToTensor=object)
    sys.path.insert(0, REF)
    import flocoder  # noqa: F401  (package __init__ is empty)
    _stand_in("flocoder.viz", save_img_grid=None)
    _stand_in("flocoder.metrics", g2rgb=None, compute_sample_metrics=None)
    from flocoder import unet, sampling
    # vector_quantize_pytorch is absent; VQVAE.__init__ builds a ResidualVQ that encode()/decode() never call
    class _RVQ(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    _stand_in("vector_quantize_pytorch", VectorQuantize=_RVQ, ResidualVQ=_RVQ)

    def load(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, "flocoder", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    from flocoder import codecs
    return unet, sampling, load("ot"), load("inpainting"), codecs


def shapes_of(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed):
    shapes = shapes_of(module)
    module.load_state_dict(synth_state_dict(shapes, seed))
    return shapes


def npz(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        elif isinstance(v, (dict, list)):
            v = np.array(json.dumps(v))
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


UNET_CASES = [  # tag, dim, n_classes, mask_cond, B, seed
    ("d32c102", 32, 102, False, 2, 1),
    ("d16c10", 16, 10, False, 3, 2),
    ("d8mask", 8, 0, True, 2, 3),
]


def make_latent_tree(root):
    """A tiny pre-encoded dataset in the reference's on-disk format (preencode_data.py:130-156), deterministic content:
    cls/  numeric class directories whose path order differs from their numeric order, plus a non-numeric directory
    sub/  non-numeric sub-directories (no classes)      flat/  files at the top      inp/  inpainting dicts with bool masks"""
    def save(path, obj):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(obj, path)
    k = 0
    for cname, count in (("0", 2), ("1", 1), ("2", 3), ("10", 2), ("11", 1), ("notes", 1)):
        for i in range(count):
            save(os.path.join(root, "cls", cname, f"sample_{k}_{i}_abcd.pt"), synth_input(f"g11.cls.{cname}.{i}", (4, 8, 8), 11))
            k += 1
    for sub, count in (("00", 2), ("01", 3)):
        for i in range(count):
            save(os.path.join(root, "sub", sub, f"sample_{sub}_{i}.pt"), synth_input(f"g11.sub.{sub}.{i}", (4, 8, 8), 11))
    for i in range(3):
        save(os.path.join(root, "flat", f"sample_{i}.pt"), synth_input(f"g11.flat.{i}", (4, 8, 8), 11))
    for cname, count in (("0", 2), ("1", 2)):
        for i in range(count):
            save(os.path.join(root, "inp", cname, f"sample_{cname}_{i}.pt"),
                 {"target_latents": synth_input(f"g11.inp.t.{cname}.{i}", (4, 8, 8), 11), "source_latents": synth_input(f"g11.inp.s.{cname}.{i}", (4, 8, 8), 11),
                  "mask_pixels": synth_input(f"g11.inp.m.{cname}.{i}", (1, 20, 20), 11) > 0.2})


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    unet, sampling, ot, inpainting, codecs = import_reference()

    # ---- G0: state_dict layout and default initialisation under a fixed seed -----------------
    g0 = {}
    for tag, kw in (("d16c10", dict(dim=16, channels=4, n_classes=10)),
                    ("d8mask", dict(dim=8, channels=4, n_classes=0, mask_cond=True)),
                    ("d32c102", dict(dim=32, channels=4, n_classes=102))):
        torch.manual_seed(0)
        m = unet.Unet(dim_mults=(1, 2, 4, 8), **kw)
        sd = m.state_dict()
        g0[tag] = {"keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()],
                   "sum": [float(v.double().sum()) for v in sd.values()],
                   "abssum": [float(v.double().abs().sum()) for v in sd.values()]}
    npz("g0_state_dict", layout=g0)

    # ---- G1: sinusoidal embedding ---------------------------------------------------
    times = torch.tensor([0.999, 1.0, 37.5, 250.0, 499.5, 750.25, 998.0, 999.0])
    npz("g1_sinusoidal", times=times, emb32=unet.SinusoidalPositionEmbeddings(32)(times),
        emb16=unet.SinusoidalPositionEmbeddings(16)(times))

    # ---- G2: individual modules -----------------------------------------------------
    out = {}
    B = 2
    mods = {
        "block": (unet.Block(32, 64, groups=4), (B, 32, 16, 16)),
        "resnet_same": (unet.ResnetBlock(32, 32, time_emb_dim=256, groups=4), (B, 32, 32, 32)),
        "resnet_proj": (unet.ResnetBlock(96, 64, time_emb_dim=256, groups=4), (B, 96, 16, 16)),
        "linattn": (unet.LinearAttention(32), (B, 32, 32, 32)),
        "linattn_small": (unet.LinearAttention(128), (B, 128, 4, 4)),
        "attn": (unet.Attention(256), (B, 256, 4, 4)),
        "down": (unet.Downsample(32, 64), (B, 32, 32, 32)),
        "up": (unet.Upsample(64, 32), (B, 64, 8, 8)),
    }
    shapes_all = {}
    for tag, (mod, xs) in mods.items():
        mod.eval()
        shapes_all[tag] = load_synth(mod, seed=11)
        x = synth_input("g2." + tag, xs, seed=11)
        if tag.startswith("resnet"):
            temb = synth_input("g2.temb." + tag, (B, 256), seed=11)
            y = mod(x, temb)
        elif tag == "block":
            sc = synth_input("g2.scale", (B, 64, 1, 1), seed=11, scale=0.3)
            sh = synth_input("g2.shift", (B, 64, 1, 1), seed=11, scale=0.3)
            y = mod(x, (sc, sh))
            out["block_noss"] = mod(x)
        else:
            y = mod(x)
        out[tag] = y
    npz("g2_modules", shapes=shapes_all, **out)

    # ---- G3: full U-Net forward -----------------------------------------------------
    for tag, dim, ncls, mask_cond, B, seed in UNET_CASES:
        H = dim
        m = unet.Unet(dim=dim, channels=4, dim_mults=(1, 2, 4, 8), n_classes=ncls, mask_cond=mask_cond).eval()
        shapes = load_synth(m, seed)
        x = synth_input("g3.x." + tag, (B, 4, H, H), seed)
        t = torch.tensor([0.001, 0.37, 0.9][:B]) * 999
        arrays = dict(shapes=shapes, t=t)
        if ncls:
            cls = torch.tensor([3, ncls - 1, 0][:B])
            arrays["cls"] = cls
            arrays["v_class"] = m(x, t, {"class_cond": cls})
            arrays["v_noclass"] = m(x, t, {"class_cond": None})
            arrays["v_none"] = m(x, t, None)
        if mask_cond:
            mask = torch.sigmoid(synth_input("g3.mask." + tag, (B, 4, H, H), seed, scale=2.0))
            arrays["mask"] = mask
            arrays["v_mask"] = m(x, t, {"class_cond": None, "mask_cond": mask})
            arrays["v_ones"] = m(x, t, {"mask_cond": torch.ones_like(mask)})
            arrays["v_none"] = m(x, t, None)
        npz("g3_unet_" + tag, **arrays)

    # ---- G4: time grids -------------------------------------------------------------
    grids = {f"rk4_{n}": sampling.warp_time(torch.linspace(0, 1, n)) for n in (3, 5, 16, 64, 100)}
    grids["rand_in"] = synth_input("g4.rand", (64,), 0).abs().clamp(max=1.0)
    grids["rand_out"] = sampling.warp_time(grids["rand_in"])
    npz("g4_timegrids", **grids)

    # ---- G5 / G6: RK4 (live) and Euler (legacy formula) trajectories, dim=16 model --------
    m = unet.Unet(dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10).eval()
    shapes = load_synth(m, seed=5)
    B = 2
    src = synth_input("g5.src", (B, 4, 16, 16), 5)
    cls = torch.tensor([7, 2])
    arrays = dict(shapes=shapes, cls=cls)
    for cfg in (0.0, 3.0):
        lat, nfe = sampling.generate_latents_rk4(m, (B, 4, 16, 16), n_steps=5, cond={"class_cond": cls},
                                                 cfg_strength=cfg, source=src.clone())
        arrays[f"rk4_n5_cfg{int(cfg)}"] = lat
        arrays[f"rk4_n5_cfg{int(cfg)}_nfe"] = np.int64(nfe)
    lat, nfe = sampling.generate_latents_rk4(m, (B, 4, 16, 16), n_steps=4, cond={}, cfg_strength=3.0,
                                             source=src.clone())
    arrays["rk4_n4_nocond"] = lat
    init = synth_input("g5.init", (B, 4, 16, 16), 5)
    lat, nfe = sampling.generate_latents_rk4(m, (B, 4, 16, 16), n_steps=8, cond={"class_cond": cls}, cfg_strength=3.0,
                                             source=src.clone(), init_latents=init, init_strength=0.5)
    arrays["rk4_n8_init05"] = lat
    arrays["rk4_n8_init05_nfe"] = np.int64(nfe)

    def legacy_euler(model, x0, n, ids, eps=0.001):   # legacy/train_sd_flowers.py:50-67, cond as dict (Q15)
        x = x0.detach().clone()
        dt = 1.0 / n
        for i in range(n):
            num_t = i / n * (1 - eps) + eps
            t = torch.ones(x.shape[0]) * num_t
            pred = model(x, t * 999, {"class_cond": ids})
            x = x.detach().clone() + pred * dt
        return x
    for n in (4, 16):
        arrays[f"euler_n{n}"] = legacy_euler(m, src, n, cls)
    npz("g5_trajectories", **arrays)

    # ---- G7: greedy OT pairing ------------------------------------------------------
    arrays = {}
    for B, D in ((8, 64), (64, 64), (256, 1024)):
        s = synth_input(f"g7.s{B}", (B, D), 7)
        t = synth_input(f"g7.t{B}", (B, D), 7)
        arrays[f"perm_{B}_{D}"] = ot.compute_ot_pairing(s, t)
    # ties: duplicated targets must resolve to the first unused index
    s = synth_input("g7.tie.s", (6, 16), 7)
    t = s[[2, 2, 0, 0, 5, 1]].clone()
    arrays["tie_src"], arrays["tie_tgt"], arrays["tie_perm"] = s, t, ot.compute_ot_pairing(s, t)
    npz("g7_ot", **arrays)

    # ---- G8: mask encoder + blending ------------------------------------------------
    me = inpainting.MaskEncoder().eval()
    shapes = load_synth(me, seed=8)
    mp = (synth_input("g8.mask", (2, 1, 128, 128), 8) > 0.3).float()
    ml = me(mp)
    srcl, noise = synth_input("g8.src", (2, 4, 8, 8), 8), synth_input("g8.noise", (2, 4, 8, 8), 8)
    npz("g8_mask_encoder", shapes=shapes, mask_latents=ml, mask_latents_bool=me(mp.bool()),
        blended=inpainting.mask_blending(srcl, ml, noise))

    # ---- G9: VQVAE encode / decode (NATTEN-less), the midi_vqgan.yaml shape and a reduced 4-downsample grayscale one ----
    import contextlib, io
    arrays = {}
    for tag, kw, hw in (("midi_vqgan", dict(in_channels=3, hidden_channels=256, num_downsamples=3, internal_dim=128, vq_embedding_dim=4), 128),
                        ("gray_nd4_small", dict(in_channels=1, hidden_channels=32, num_downsamples=4, internal_dim=32, vq_embedding_dim=4), 128)):
        with contextlib.redirect_stdout(io.StringIO()):
            m = codecs.VQVAE(vq_num_embeddings=32, codebook_levels=2, no_natten=True, **kw).eval()
        shapes = {k: list(v.shape) for k, v in m.state_dict().items() if k != "codebook_usage"}
        sd = synth_state_dict(shapes, 9)
        m.load_state_dict(sd, strict=False)
        x = torch.sigmoid(synth_input("g9.x." + tag, (1, kw["in_channels"], hw, hw), 9, scale=2.0))
        with contextlib.redirect_stdout(io.StringIO()):
            z = m.encode(x, debug=False)
            y = m.decode(synth_input("g9.z." + tag, tuple(z.shape), 9))
        arrays[tag + "_shapes"] = shapes
        arrays[tag + "_z"] = z
        arrays[tag + "_recon"] = y
    npz("g9_vqvae", **arrays)

    # ---- G10: flow training step (train_flow.py:338-397 restated around the reference Unet, torch autograd, torch.optim.Adam) ----
    # three consecutive steps on the stl_sd.yaml shape (dim=16, n_classes=10, latents 4x16x16), B=8; step 2 drops the conditioning
    # (train_flow.py:343-345), so class_cond_mlp.* gets no gradient there and Adam skips it.
    torch.manual_seed(0)
    m = unet.Unet(dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10).train()
    shapes = load_synth(m, seed=10)
    B = 8
    cls = torch.tensor([3, 0, 9, 9, 1, 4, 7, 2])
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    shadow = {k: p.data.clone() for k, p in m.named_parameters()}            # EMA.__init__, train_flow.py:41-44 (decay 0.999, :334)
    loss_fn = torch.nn.MSELoss()
    eps = 0.001
    small = ("final_conv.weight", "final_conv.bias", "init_conv.weight", "class_cond_mlp.0.weight", "time_mlp.1.bias",
             "downs.0.0.block1.norm.weight", "downs.0.0.mlp.1.bias", "downs.3.2.fn.norm.bias", "mid_attn.fn.fn.to_out.bias",
             "ups.3.1.block2.proj.bias", "ups.0.2.fn.fn.to_out.1.weight")
    arrays = {"shapes": shapes, "cls": cls, "small": list(small)}
    for step in (1, 2, 3):
      with torch.enable_grad():
          source = synth_input(f"g10.src{step}", (B, 4, 16, 16), 10)
          target = synth_input(f"g10.tgt{step}", (B, 4, 16, 16), 10)
          u = torch.sigmoid(synth_input(f"g10.u{step}", (B,), 10, scale=1.5))
          cond = {"class_cond": cls, "mask_cond": None} if step != 2 else None
          opt.zero_grad()
          t = sampling.warp_time(u * (1 - eps) + eps)
          t_expand = t.view(-1, 1, 1, 1).repeat(1, target.shape[1], target.shape[2], target.shape[3])
          x = (1 - t_expand) * source + t_expand * target
          v_guess = target - source
          v_model = m(x, t * 999, cond)
          loss = loss_fn(v_model, v_guess)
          loss.backward()
          names = [k for k, _ in m.named_parameters()]
          gsum = np.array([float("nan") if p.grad is None else float(p.grad.double().sum()) for p in m.parameters()])
          gabs = np.array([float("nan") if p.grad is None else float(p.grad.double().abs().sum()) for p in m.parameters()])
          for k, p in m.named_parameters():
              if k in small and p.grad is not None:
                  arrays[f"s{step}_grad_{k}"] = p.grad.clone()
          total = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
          opt.step()
          for k, p in m.named_parameters():                                    # EMA.update, train_flow.py:46-54
              shadow[k] = 0.999 * shadow[k] + (1.0 - 0.999) * p.data
          arrays.update({f"s{step}_t": t, f"s{step}_loss": loss.detach(), f"s{step}_norm": total, f"s{step}_v": v_model.detach(),
                         f"s{step}_gsum": gsum, f"s{step}_gabs": gabs,
                         f"s{step}_psum": np.array([float(p.double().sum()) for p in m.parameters()]),
                         f"s{step}_pabs": np.array([float(p.double().abs().sum()) for p in m.parameters()]),
                         f"s{step}_esum": np.array([float(shadow[k].double().sum()) for k in names]),
                         f"s{step}_eabs": np.array([float(shadow[k].double().abs().sum()) for k in names])})
          for k, p in m.named_parameters():
              if k in small:
                  arrays[f"s{step}_param_{k}"] = p.data.clone()
    arrays["names"] = names
    npz("g10_train_step", **arrays)

    # ---- G11: PreEncodedDataset (data.py:311-384) on a small tree in the on-disk format of preencode_data.py:130-156 ----
    import tempfile
    tvm = sys.modules["torchvision"]
    tvm.datasets = _stand_in("torchvision.datasets")
    spec = importlib.util.spec_from_file_location("flocoder.data", os.path.join(REF, "flocoder", "data.py"))
    refdata = importlib.util.module_from_spec(spec)
    refdata.__package__ = "flocoder"
    spec.loader.exec_module(refdata)
    g11 = {}
    with tempfile.TemporaryDirectory() as td:
        make_latent_tree(td)
        with contextlib.redirect_stdout(io.StringIO()):
            for tag, sub, kw in (("classes", "cls", {}), ("classes_off", "cls", {"n_classes": 0}), ("subdirs", "sub", {}), ("flat", "flat", {}),
                                 ("inpaint", "inp", {})):
                ds = refdata.PreEncodedDataset(os.path.join(td, sub), **kw)
                g11[tag] = {"n_classes": ds.n_classes, "has_classes": bool(ds.has_classes), "len": len(ds),
                            "labels": {os.path.relpath(str(f), os.path.join(td, sub)): int(l) for f, l in zip(ds.files, ds._labels)},
                            "class_to_idx": {str(k): int(v) for k, v in getattr(ds, "class_to_idx", {}).items()}}
            item, lab = refdata.PreEncodedDataset(os.path.join(td, "inp"))[0]
        g11["inpaint"]["item_keys"] = sorted(item.keys())
        g11["inpaint"]["mask_dtype"] = str(item["mask_pixels"].dtype)
    npz("g11_latent_dataset", layout=g11)


if __name__ == "__main__":
    main()
