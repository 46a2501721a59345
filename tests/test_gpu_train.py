"""GPU parity of the flow training step (through the C ABI) against oracle/train_oracle.py and the golden vectors the reference
produced (fixture g10: three consecutive steps of the reference Unet under torch.optim.Adam, the second without conditioning).

Tolerances (fp32 throughout; the backward re-associates every reduction, so these are rounding-level bounds, measured values are
printed with -s): per-parameter gradients rel-L2 <= 2e-5 against the oracle's autograd (measured worst 3.4e-6; <= 5e-3 for tensors whose
gradient is itself cancellation noise, flagged by a tiny norm); loss 2e-6 relative; parameters / EMA after a step 2e-6 of their magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_l2
from oracle import train_oracle as to
from oracle.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev():
    return torch.device(DEV)


WG_CASES = [  # (B, C0, C1, Cout, H, W, KS, pad, stride, ups)
    (2, 32, 0, 32, 16, 16, 3, 1, 1, 0),
    (3, 64, 32, 64, 8, 8, 3, 1, 1, 0),        # concat sources, batch not a multiple of the tile's samples
    (2, 32, 0, 96, 16, 16, 1, 0, 1, 0),       # 1x1 (to_qkv shape)
    (5, 96, 0, 32, 4, 4, 1, 0, 1, 0),
    (2, 16, 0, 32, 16, 16, 2, 0, 2, 0),       # Downsample as a 2x2 stride-2 conv
    (2, 32, 0, 16, 8, 8, 3, 1, 1, 1),         # Upsample: nearest x2 in the loader
    (4, 4, 0, 16, 16, 16, 1, 0, 1, 0),        # init_conv
    (4, 16, 0, 4, 16, 16, 1, 0, 1, 0),        # final_conv
    (9, 128, 0, 128, 2, 2, 3, 1, 1, 0),
    (40, 64, 0, 64, 1, 1, 3, 1, 1, 0),        # 1x1 images: small-tile path
    (2, 48, 16, 40, 32, 32, 3, 1, 1, 0),      # channel counts that are not multiples of 32
    (2, 8, 4, 12, 8, 8, 5, 2, 1, 0),
]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad_matches_autograd(case):
    from flocoder_amd._ops import conv_wgrad_debug
    B, c0, c1, co, H, W, ks, pad, stride, ups = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x0 = torch.randn(B, c0, H, W, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    w = torch.zeros(co, c0 + c1, ks, ks, requires_grad=True)
    b = torch.zeros(co, requires_grad=True)
    y = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dw, db = conv_wgrad_debug(x0.to(dev()), dy.to(dev()), ks, None if x1 is None else x1.to(dev()), pad=pad, stride=stride, upsample=bool(ups))
    assert torch.isfinite(dw).all() and torch.isfinite(db).all()
    assert rel_l2(dw.cpu(), w.grad) < 2e-6 and rel_l2(db.cpu(), b.grad) < 2e-6


def _model(tag_shapes, seed, **kw):
    from flocoder_amd.unet import Unet
    m = Unet(**kw)
    m.load_state_dict(synth_state_dict(tag_shapes, seed))
    return m.to(dev()).train()


def _check_grads(m, flat, grads_ref, tol=2e-5):
    worst = ("", 0.0)
    views = m.grad_views(flat)
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads_ref.values() if g is not None)))
    for k, gr in grads_ref.items():
        got = views[k].cpu()
        if gr is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        e = rel_l2(got, gr)
        lim = tol if float(gr.norm()) > 1e-4 * total else 5e-3        # cancellation-dominated tensors
        if e > worst[1]:
            worst = (k, e)
        assert e < lim, (k, e, float(gr.norm()), total)
    return worst


@pytest.mark.parametrize("with_cls", [True, False])
def test_unet_gradients_match_oracle_g10_shape(with_cls):
    g = load_golden("g10_train_step")
    sd = synth_state_dict(g["shapes"], 10)
    m = _model(g["shapes"], 10, dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10)
    src, tgt = synth_input("g10.src1", (8, 4, 16, 16), 10), synth_input("g10.tgt1", (8, 4, 16, 16), 10)
    t = to.train_time(torch.sigmoid(synth_input("g10.u1", (8,), 10, scale=1.5)))
    cls = torch.from_numpy(g["cls"])
    cond = {"class_cond": cls, "mask_cond": None} if with_cls else None
    loss_ref, grads_ref, v_ref = to.loss_and_grads(sd, src, tgt, t, cond)
    from flocoder_amd.train import FlowTrainer
    tr = FlowTrainer(m)
    x, vt = tr.interpolate(src.to(dev()), tgt.to(dev()), t.to(dev()))
    loss, v = tr.loss_and_grads(x, t.to(dev()), cls.to(dev()) if with_cls else None, vt)
    assert rel_l2(v.cpu(), v_ref) < 2e-5
    assert abs(float(loss) - float(loss_ref)) < 2e-6 * float(loss_ref)
    worst = _check_grads(m, tr.grads, grads_ref)
    print("worst parameter gradient:", worst)
    flat2 = tr.grads.clone()
    tr.loss_and_grads(x, t.to(dev()), cls.to(dev()) if with_cls else None, vt)
    assert torch.equal(flat2, tr.grads)                         # bit-reproducible backward


@pytest.mark.parametrize("kw,B,hw", [(dict(dim=32, channels=4, dim_mults=(1, 2, 4, 8), n_classes=102), 3, 32),
                                     (dict(dim=8, channels=4, dim_mults=(1, 2, 4, 8), n_classes=0), 5, 8),
                                     (dict(dim=16, channels=4, dim_mults=(1, 2, 4), n_classes=3), 2, 16)])
def test_unet_gradients_other_shapes(kw, B, hw):
    """flowers_sd (dim 32, 102 classes), a class-free dim-8 model whose deepest level is 1x1, and a 3-level model."""
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    torch.manual_seed(3)
    m = Unet(**kw)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev()).train()
    gen = torch.Generator().manual_seed(5)
    src, tgt = torch.randn(B, 4, hw, hw, generator=gen), torch.randn(B, 4, hw, hw, generator=gen)
    t = to.train_time(torch.rand(B, generator=gen))
    ncls = kw["n_classes"]
    cls = torch.randint(0, ncls, (B,), generator=gen) if ncls else None
    loss_ref, grads_ref, v_ref = to.loss_and_grads(sd, src, tgt, t, {"class_cond": cls} if ncls else None)
    tr = FlowTrainer(m)
    x, vt = tr.interpolate(src.to(dev()), tgt.to(dev()), t.to(dev()))
    loss, v = tr.loss_and_grads(x, t.to(dev()), None if cls is None else cls.to(dev()), vt)
    assert rel_l2(v.cpu(), v_ref) < 2e-5 and abs(float(loss) - float(loss_ref)) < 2e-6 * float(loss_ref)
    print("worst parameter gradient:", _check_grads(m, tr.grads, grads_ref))


def test_three_training_steps_match_reference_golden():
    """FlowTrainer.step against fixture g10 (reference Unet + torch.optim.Adam + EMA 0.999, three steps, step 2 unconditioned)."""
    from flocoder_amd.train import FlowTrainer
    g = load_golden("g10_train_step")
    names = list(g["names"])
    m = _model(g["shapes"], 10, dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10)
    tr = FlowTrainer(m, lr=1e-4, ema_decay=0.999)
    cls = torch.from_numpy(g["cls"]).to(dev())
    for step in (1, 2, 3):
        src, tgt = synth_input(f"g10.src{step}", (8, 4, 16, 16), 10), synth_input(f"g10.tgt{step}", (8, 4, 16, 16), 10)
        u = torch.sigmoid(synth_input(f"g10.u{step}", (8,), 10, scale=1.5))
        cond = {"class_cond": cls, "mask_cond": None} if step != 2 else None
        loss = tr.step(src, tgt, cond, u=u)
        assert abs(float(loss) - float(g[f"s{step}_loss"])) < 5e-6 * float(g[f"s{step}_loss"]), step
        assert abs(float(tr.grad_norm) - float(g[f"s{step}_norm"])) < 1e-4 * float(g[f"s{step}_norm"]), step
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ema = {k: v.cpu() for k, v in tr.ema_state_dict().items()}
        for i, k in enumerate(names):
            assert abs(float(sd[k].double().sum()) - g[f"s{step}_psum"][i]) <= 3e-6 * g[f"s{step}_pabs"][i] + 1e-9, (step, k)
            assert abs(float(ema[k].double().sum()) - g[f"s{step}_esum"][i]) <= 3e-6 * g[f"s{step}_eabs"][i] + 1e-9, (step, k)
        for k in g["small"]:
            assert rel_l2(sd[k], g[f"s{step}_param_{k}"]) < 2e-6, (step, k)
    assert tr.step_main == 3 and tr.step_class == 2
    # the trained weights are what the sampler sees
    m.eval()
    x = synth_input("g10.src1", (8, 4, 16, 16), 10).to(dev())
    with torch.no_grad():
        v = m(x, torch.full((8,), 500.0, device=dev()), {"class_cond": cls})
    from oracle import flow_oracle as fo
    ref = fo.unet_forward({k: v_.cpu() for k, v_ in m.state_dict().items()}, x.cpu(), torch.full((8,), 500.0), {"class_cond": cls.cpu()})
    assert rel_l2(v.cpu(), ref) < 2e-5


def test_reference_loop_shape_through_autograd():
    """The reference's own step lines (train_flow.py:346-397) run unchanged on the mirror: autograd reaches the native backward,
    torch.optim.Adam / clip_grad_norm_ / EMA behave as upstream, and the result equals FlowTrainer's fused step."""
    from flocoder_amd.sampling import warp_time
    from flocoder_amd.train import EMA, FlowTrainer
    g = load_golden("g10_train_step")
    a = _model(g["shapes"], 10, dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10)
    b = _model(g["shapes"], 10, dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10)
    tr = FlowTrainer(b)
    opt = torch.optim.Adam(a.parameters(), lr=1e-4)
    ema = EMA(a, decay=0.999, device=dev())
    cls = torch.from_numpy(g["cls"]).to(dev())
    for step in (1, 2):
        source = synth_input(f"g10.src{step}", (8, 4, 16, 16), 10).to(dev())
        target = synth_input(f"g10.tgt{step}", (8, 4, 16, 16), 10).to(dev())
        u = torch.sigmoid(synth_input(f"g10.u{step}", (8,), 10, scale=1.5)).to(dev())
        cond = {"class_cond": cls, "mask_cond": None} if step == 1 else None
        opt.zero_grad()
        t = warp_time(u * (1 - 0.001) + 0.001)
        t_expand = t.view(-1, 1, 1, 1).repeat(1, target.shape[1], target.shape[2], target.shape[3])
        x = (1 - t_expand) * source + t_expand * target
        v_guess = target - source
        v_model = a(x, t * 999, cond)
        loss = torch.nn.MSELoss()(v_model, v_guess)
        loss.backward()
        if step == 2:
            assert a.get_parameter("class_cond_mlp.0.weight").grad is None
        torch.nn.utils.clip_grad_norm_(a.parameters(), max_norm=1.0)
        opt.step()
        ema.update()
        loss_b = tr.step(source, target, cond, u=u)
        assert abs(float(loss.detach()) - float(loss_b)) < 1e-6 * float(loss.detach())
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert rel_l2(sa[k].cpu(), sb[k].cpu()) < 1e-6, k
    eb = tr.ema_state_dict()
    for k, v in ema.shadow.items():
        assert rel_l2(v.cpu(), eb[k].cpu()) < 1e-6, k
    ema.eval()
    assert torch.equal(a.get_parameter("final_conv.weight").data, ema.shadow["final_conv.weight"])
    ema.train()


def test_batch_to_data_and_train_batch():
    from flocoder_amd.train import FlowTrainer, batch_to_data
    from flocoder_amd.unet import Unet
    torch.manual_seed(0)
    lat = torch.randn(16, 4, 16, 16)
    cls = torch.randint(0, 10, (16,))
    src, tgt, cc, mask, mp = batch_to_data((lat, cls), dev())
    assert mask is None and mp is None and src.shape == tgt.shape == (16, 4, 16, 16) and cc.device.type == "cuda"
    assert sorted(tgt.flatten(1).sum(1).cpu().tolist()) == pytest.approx(sorted(lat.flatten(1).sum(1).tolist()), rel=1e-5)   # a permutation of the batch
    m = Unet(dim=16, channels=4, dim_mults=(1, 2, 4, 8), n_classes=10).to(dev())
    tr = FlowTrainer(m, lr=1e-3)
    losses = [float(tr.train_batch((lat, cls))) for _ in range(12)]
    assert all(np.isfinite(losses)) and np.mean(losses[-4:]) < np.mean(losses[:4])       # it learns


@pytest.mark.parametrize("variant", ["mask", "ones", "nomask"])
@pytest.mark.parametrize("dim,hw,ncls", [(8, 8, 0), (16, 16, 5)])
def test_mask_conditioned_gradients(variant, dim, hw, ncls):
    """Backward of the mask-conditioning branches (mask_fusion_conv, per-scale injections; unet.py:298-305,336-340,360-364): every
    parameter gradient plus d(x) and d(mask) against the oracle's autograd, for a real mask, an all-ones mask (fusion bypassed,
    unet.py:301) and no mask at all (cond dropped)."""
    from flocoder_amd.unet import Unet
    torch.manual_seed(11)
    m = Unet(dim=dim, channels=4, dim_mults=(1, 2, 4, 8), n_classes=ncls, mask_cond=True)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev()).train()
    gen = torch.Generator().manual_seed(6)
    B = 3
    x = torch.randn(B, 4, hw, hw, generator=gen)
    t = torch.rand(B, generator=gen) * 999
    cls = torch.randint(0, ncls, (B,), generator=gen) if ncls else None
    mask = {"mask": torch.rand(B, 4, hw, hw, generator=gen), "ones": torch.ones(B, 4, hw, hw), "nomask": None}[variant]
    dv = torch.randn(B, 4, hw, hw, generator=gen)
    # oracle: autograd through the functional U-Net, x and mask as leaves
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    mr = mask.clone().requires_grad_(True) if mask is not None else None
    from oracle import flow_oracle as fo
    cond = {"class_cond": cls, "mask_cond": mr}
    v_ref = fo.unet_forward(leaves, xr, t, cond)
    (v_ref * dv).sum().backward()
    # library
    xd, td = x.to(dev()), t.to(dev())
    md = mask.to(dev()) if mask is not None else None
    cd = cls.to(dev()) if cls is not None else None
    v = m._forward_native(xd, td, cd, md, train=True)
    assert rel_l2(v.cpu(), v_ref.detach()) < 2e-5
    flat, dx, dm = m.backward_native(xd, td, cd, dv.to(dev()), mask=md, want_dx=True, want_dmask=True)
    views = m.grad_views(flat)
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in leaves.values() if p.grad is not None)))
    worst = ("", 0.0)
    for k, p in leaves.items():
        got = views[k].cpu()
        if p.grad is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        e = rel_l2(got, p.grad)
        lim = 2e-5 if float(p.grad.norm()) > 1e-4 * total else 5e-3
        worst = max(worst, (k, e), key=lambda kv: kv[1])
        assert e < lim, (k, e)
    assert rel_l2(dx.cpu(), xr.grad) < 2e-5
    if mask is not None:
        ref_dm = mr.grad if mr.grad is not None else torch.zeros_like(mask)
        if float(ref_dm.norm()) > 0:
            assert rel_l2(dm.cpu(), ref_dm) < 2e-5
        else:
            assert float(dm.abs().max()) == 0.0
    unused = [k for k, p in leaves.items() if p.grad is None]
    if variant == "nomask":
        assert any(k.startswith("mask_fusion_conv") for k in unused) and any("mask_fusions" in k for k in unused)
    if variant == "ones":
        assert any(k.startswith("mask_fusion_conv") for k in unused) and not any("mask_fusions" in k for k in unused)
    print(variant, "worst parameter gradient:", worst)


def test_mask_conditioned_training_steps_match_torch_adam():
    """Three FlowTrainer steps on a mask-conditioned model (mask, all-ones mask, cond dropped) against the same steps taken with the
    oracle's gradients and its explicit Adam: the mask parameter groups are skipped exactly when torch would skip them."""
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    torch.manual_seed(12)
    m = Unet(dim=8, channels=4, dim_mults=(1, 2, 4, 8), n_classes=0, mask_cond=True)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev())
    tr = FlowTrainer(m, lr=1e-3)
    state = to.new_state(sd)
    gen = torch.Generator().manual_seed(7)
    well = {}
    for step, kind in enumerate(["mask", "ones", "nomask"]):
        src, tgt = torch.randn(4, 4, 8, 8, generator=gen), torch.randn(4, 4, 8, 8, generator=gen)
        u = torch.rand(4, generator=gen)
        mask = {"mask": torch.rand(4, 4, 8, 8, generator=gen), "ones": torch.ones(4, 4, 8, 8), "nomask": None}[kind]
        cond = {"class_cond": None, "mask_cond": mask} if mask is not None else None
        loss = tr.step(src, tgt, cond, u=u)
        t = to.train_time(u)
        loss_ref, grads, _ = to.loss_and_grads(sd, src, tgt, t, cond)
        to.adam_ema_step(sd, grads, state, lr=1e-3)
        assert abs(float(loss) - float(loss_ref)) < 5e-6 * float(loss_ref)
        got = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        # Adam's update g/(|g|+eps)-like ratio is ill-conditioned where a gradient is ~eps (1e-8): hold the well-conditioned elements
        # (|g| > 1e-5 in every step so far) to 1e-3 of the learning rate, the rest to the learning rate itself
        for k in sd:
            if grads[k] is not None:
                well[k] = well.get(k, torch.ones_like(sd[k], dtype=torch.bool)) & (grads[k].abs() > 1e-5)
            d = (got[k] - sd[k]).abs()
            w = well.get(k, torch.ones_like(sd[k], dtype=torch.bool))
            assert float(d[w].max()) < 1e-3 * 1e-3 if bool(w.any()) else True, (kind, k, float(d[w].max()))
            assert float(d.max()) < 1.01e-3 * (step + 1), (kind, k)
    assert tr.steps == {"class": 0, "fusion": 1, "inject": 2} and tr.step_main == 3


def test_mask_encoder_gradients_match_oracle():
    """MaskEncoder backward (five direct convolutions, SiLU / sigmoid) against autograd through the oracle's restatement, including
    the accumulate mode the three-pass inpainting step uses."""
    from flocoder_amd.inpainting import MaskEncoder
    from oracle import flow_oracle as fo
    torch.manual_seed(3)
    me = MaskEncoder()
    sd = {k: v.detach().clone() for k, v in me.state_dict().items()}
    me = me.to(dev()).train()
    gen = torch.Generator().manual_seed(9)
    mp = (torch.rand(3, 1, 128, 128, generator=gen) > 0.6).float()
    dl = torch.randn(3, 4, 8, 8, generator=gen)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ml = fo.mask_encoder_forward(leaves, mp)
    (ml * dl).sum().backward()
    out = me(mp.to(dev()))
    assert out.requires_grad and rel_l2(out.detach().cpu(), ml.detach()) < 2e-6
    flat = me.backward_native(mp.to(dev()), dl.to(dev()))
    for k, v in me.grad_views(flat).items():
        assert rel_l2(v.cpu(), leaves[k].grad) < 2e-5, k
    flat2 = me.backward_native(mp.to(dev()), dl.to(dev()), grads=flat.clone(), accumulate=True)
    assert rel_l2(flat2.cpu(), 2 * flat.cpu()) < 1e-6
    (out * dl.to(dev())).sum().backward()                                     # and through autograd
    for k, p in me.named_parameters():
        assert rel_l2(p.grad.cpu(), leaves[k].grad) < 2e-5, k


def test_inpainting_step_in_the_reference_loop_shape():
    """train_flow.py:338-397 with a MaskEncoder, verbatim on the mirrors: mask latents from the encoder, blended source, mask-conditioned
    U-Net, the two mask losses, loss.backward() -- every gradient (U-Net and MaskEncoder) against the oracle's autograd."""
    from flocoder_amd.inpainting import MaskEncoder, mask_blending
    from flocoder_amd.sampling import warp_time
    from flocoder_amd.unet import Unet
    from oracle import flow_oracle as fo
    torch.manual_seed(21)
    model, me = Unet(dim=8, channels=4, dim_mults=(1, 2, 4, 8), n_classes=0, mask_cond=True), MaskEncoder()
    sdm = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sde = {k: v.detach().clone() for k, v in me.state_dict().items()}
    model, me = model.to(dev()).train(), me.to(dev()).train()
    gen = torch.Generator().manual_seed(22)
    B = 3
    target, src0, noise = (torch.randn(B, 4, 8, 8, generator=gen) for _ in range(3))
    mask_pixels = (torch.rand(B, 1, 128, 128, generator=gen) > 0.5).float()
    u = torch.rand(B, generator=gen)

    def step(unet_fn, enc_fn, blend, to_dev):
        mp, tg, s0, nz, uu = (to_dev(t) for t in (mask_pixels, target, src0, noise, u))
        mask = enc_fn(mp)
        source = blend(s0, mask, nz)
        t = warp_time(uu * (1 - 0.001) + 0.001)
        te = t.view(-1, 1, 1, 1).repeat(1, tg.shape[1], tg.shape[2], tg.shape[3])
        x = (1 - te) * source + te * tg
        v_guess = tg - source
        v_model = unet_fn(x, t * 999, {"class_cond": None, "mask_cond": mask})
        loss = torch.nn.functional.mse_loss(v_model, v_guess)
        mask_loss = torch.nn.functional.mse_loss(enc_fn(torch.ones_like(mp)), torch.ones_like(mask))
        mask_loss = mask_loss + torch.nn.functional.mse_loss(enc_fn(torch.zeros_like(mp)), torch.zeros_like(mask))
        loss = loss + 1.0 * mask_loss
        loss.backward()
        return loss.detach()

    lm = {k: v.clone().requires_grad_(True) for k, v in sdm.items()}
    le = {k: v.clone().requires_grad_(True) for k, v in sde.items()}
    loss_ref = step(lambda x, t, c: fo.unet_forward(lm, x, t, c), lambda mp: fo.mask_encoder_forward(le, mp), fo.mask_blending, lambda t: t)
    loss = step(model, me, mask_blending, lambda t: t.to(dev()))
    assert abs(float(loss) - float(loss_ref)) < 5e-6 * float(loss_ref)
    tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in list(lm.values()) + list(le.values()) if p.grad is not None)))
    for name, p in list(model.named_parameters()) + [("ME." + k, v) for k, v in me.named_parameters()]:
        ref = le[name[3:]].grad if name.startswith("ME.") else lm[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        lim = 5e-5 if float(ref.norm()) > 1e-4 * tot else 1e-2
        assert rel_l2(p.grad.cpu(), ref) < lim, (name, rel_l2(p.grad.cpu(), ref))


def test_fused_inpaint_step_equals_reference_loop_with_two_param_groups():
    """FlowTrainer.inpaint_step (everything through the C ABI, no autograd) against train_flow.py:312-397 run verbatim on the mirrors
    with torch.optim.Adam's two parameter groups, the joint clip at 1.0 and the encoder's clip at 0.5, over two steps."""
    from flocoder_amd.inpainting import MaskEncoder, mask_blending
    from flocoder_amd.sampling import warp_time
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet

    def make():
        torch.manual_seed(31)
        return Unet(dim=8, channels=4, dim_mults=(1, 2, 4, 8), n_classes=0, mask_cond=True).to(dev()).train(), MaskEncoder().to(dev()).train()
    model, me = make()
    model_b, me_b = make()
    lr = 1e-3
    opt = torch.optim.Adam([{'params': model.parameters(), 'lr': lr}, {'params': me.parameters(), 'lr': lr * 0.1}])
    model.mask_encoder = me                                  # train_flow.py:333 -- from here on model.parameters() includes the encoder
    tr = FlowTrainer(model_b, lr=lr)
    tr.attach_mask_encoder(me_b)
    gen = torch.Generator().manual_seed(32)
    well = {}
    for step in range(2):
        tgt, s0, noise = (torch.randn(3, 4, 8, 8, generator=gen).to(dev()) for _ in range(3))
        mp = (torch.rand(3, 1, 128, 128, generator=gen) > 0.5).float().to(dev())
        u = torch.rand(3, generator=gen).to(dev())
        # -- reference loop shape
        opt.zero_grad()
        mask = me(mp)
        source = mask_blending(s0, mask, noise)
        t = warp_time(u * (1 - 0.001) + 0.001)
        te = t.view(-1, 1, 1, 1).repeat(1, 4, 8, 8)
        x = (1 - te) * source + te * tgt
        loss = torch.nn.functional.mse_loss(model(x, t * 999, {'class_cond': None, 'mask_cond': mask}), tgt - source)
        mask_loss = torch.nn.functional.mse_loss(me(torch.ones_like(mp)), torch.ones_like(mask))
        mask_loss = mask_loss + torch.nn.functional.mse_loss(me(torch.zeros_like(mp)), torch.zeros_like(mask))
        loss = loss + 1.0 * mask_loss
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        torch.nn.utils.clip_grad_norm_(me.parameters(), max_norm=0.5)
        opt.step()
        # -- fused step
        loss_b = tr.inpaint_step(s0, tgt, mp, noise=noise, u=u)
        assert abs(float(loss.detach()) - float(loss_b)) < 2e-6 * float(loss.detach())
        pa, pb = dict(model.named_parameters()), dict(model_b.named_parameters())
        assert set(pa) == set(pb) and any(k.startswith("mask_encoder.") for k in pa)
        for k in pa:
            well[k] = well.get(k, torch.ones_like(pa[k], dtype=torch.bool)) & (grads[k].abs() > 1e-5)
            d = (pa[k].detach() - pb[k].detach()).abs()
            if bool(well[k].any()):
                assert float(d[well[k]].max()) < 2e-3 * lr, (step, k, float(d[well[k]].max()))
            assert float(d.max()) < 1.01 * lr * (step + 1), (step, k)
    assert tr.me_step == 2 and tr.step_main == 2


def test_train_batch_inpainting_dict_batches():
    """train_batch on the dict batches of the inpainting dataset (data.py / preencode_data.py:130-156): encoder attached, OT pairing,
    the 10 % condition drop -- finite losses and both networks move."""
    from flocoder_amd.inpainting import MaskEncoder
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    torch.manual_seed(41)
    model, me = Unet(dim=8, channels=4, dim_mults=(1, 2, 4, 8), n_classes=0, mask_cond=True).to(dev()), MaskEncoder()
    tr = FlowTrainer(model, lr=1e-3)
    tr.attach_mask_encoder(me)
    before = tr.me_params.clone(), tr.params.clone()
    gen = torch.Generator().manual_seed(42)
    batch = ({"target_latents": torch.randn(6, 4, 8, 8, generator=gen), "source_latents": torch.randn(6, 4, 8, 8, generator=gen),
              "mask_pixels": torch.rand(6, 1, 128, 128, generator=gen) > 0.5}, torch.zeros(6, dtype=torch.long))
    import random
    random.seed(0)
    losses = [float(tr.train_batch(batch, cfg_drop=0.5)) for _ in range(12)]
    assert all(np.isfinite(losses)) and not torch.equal(before[0], tr.me_params) and not torch.equal(before[1], tr.params)
    assert tr.me_step == 12 and tr.steps["inject"] < 12                     # some steps dropped the condition
    assert "mask_encoder.layers.0.conv1.weight" in dict(model.named_parameters())


_SPLIT_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from flocoder_amd.unet import Unet
torch.manual_seed(3)
m = Unet(dim=16, dim_mults=(1, 2, 4, 8), channels=4, n_classes=10).to("cuda:0").train()
g = torch.Generator().manual_seed(4)
x = torch.randn(6, 4, 16, 16, generator=g).to("cuda:0"); t = (torch.rand(6, generator=g) * 999).to("cuda:0")
ids = torch.randint(10, (6,), generator=g).to("cuda:0"); dv = torch.randn(6, 4, 16, 16, generator=g).to("cuda:0")
m._forward_native(x, t, ids, None, train=True)
flat, _, _ = m.backward_native(x, t, ids, dv)
torch.save(flat.cpu(), sys.argv[1])
"""


@pytest.mark.timeout(900)
def test_weight_gradients_do_not_depend_on_which_entries_join_the_table_launch(tmp_path):
    """ADVICE r2 (unet_backward.hip): an activation recomputed only for a weight gradient leaves the data-gradient chain exactly when
    that weight gradient joined the end-of-plan table launch -- one decision for both.  Different table splits (and no table at all)
    move entries in and out of the table; the gradients must not care."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("default", {}), ("split64", {"FLOCODER_AMD_WGRAD_TABLE_SPLIT": "64"}), ("split4096", {"FLOCODER_AMD_WGRAD_TABLE_SPLIT": "4096"}),
                     ("each", {"FLOCODER_AMD_WGRAD_EACH": "1"})):
        f = str(tmp_path / (tag + ".pt"))
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", _SPLIT_SCRIPT % root, f], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = torch.load(f)
    for tag, v in outs.items():
        assert torch.isfinite(v).all()
        assert rel_l2(v, outs["default"]) < 2e-6, tag


def test_backward_in_two_parts_equals_the_whole_and_finishes_the_late_layers_first():
    """fc_unet_backward_parts (round 3): part 0 runs the backward through mid_block1 and leaves [split, numel) of the flat gradient vector --
    ups.*, mid_*, final_* -- final, so a data-parallel trainer can all-reduce that bucket while part 1 runs; the two parts together are the
    one-call backward bit for bit (same launches, same order)."""
    from flocoder_amd.unet import Unet
    torch.manual_seed(11)
    m = Unet(dim=16, dim_mults=(1, 2, 4, 8), channels=4, n_classes=10).to(dev()).train()
    m.set_grad_buckets(True)                   # what a data-parallel FlowTrainer asks for; a single process keeps the one-bucket plan
    g = torch.Generator().manual_seed(12)
    x = torch.randn(6, 4, 16, 16, generator=g).to(dev()); t = (torch.rand(6, generator=g) * 999).to(dev())
    ids = torch.randint(10, (6,), generator=g).to(dev()); dv = torch.randn(6, 4, 16, 16, generator=g).to(dev())
    m._forward_native(x, t, ids, None, train=True)
    whole, _, _ = m.backward_native(x, t, ids, dv)
    whole = whole.clone()
    nb, split = m.grad_buckets()
    assert nb == 2 and 0 < split < whole.numel()
    names = {n: off for n, _, off in m._table}
    assert split == names["ups.0.0.mlp.1.weight"] and all(off >= split for n, off in names.items() if n.startswith(("ups.", "mid_", "final_")))
    assert all(off < split for n, off in names.items() if n.startswith(("downs.", "init_conv", "time_mlp", "class_cond_mlp")))
    m._forward_native(x, t, ids, None, train=True)
    flat = torch.full_like(whole, float("nan"))
    m.backward_native(x, t, ids, dv, flat, parts=(0, 0))
    torch.cuda.synchronize()
    assert torch.equal(flat[split:], whole[split:]), "the late layers' gradients are final after part 0"
    assert not torch.equal(flat[:split], whole[:split])
    m.backward_native(x, t, ids, dv, flat, parts=(1, 1))
    assert torch.equal(flat, whole)
