"""GPU parity AT THE SIZES THAT ARE BENCHMARKED (BASELINE configs 2 and 3), through the C ABI, against the CPU oracle.

  (a) the north_star gate: 64-step Euler on the dim-32 / 102-class model, then SD-VAE decode 4x32x32 -> 3x256x256; decoded images
      within 1e-3 rel-L2 of the oracle's (flow + sdvae restatements run end to end on the same noise);
  (b) the SD-VAE alone at 256x256 <-> 4x32x32 -- the shapes whose plan picks the M256N64 tile and gn_fold;
  (c) the DEFAULT plan at B=64 (no debug switch): rows 0-7 against the oracle, and the launch list is the one bench.py times;
  (d) "100-step" RK4 (99 intervals, 396 evaluations) with CFG 3.0 on the dim-32 model: error growth over BASELINE config 3's length;
  (e) decode_latents with chunk_size < batch (sampling.py:169-183).

Tolerances: single forward rel-L2 <= 2e-5; trajectories <= 2e-4 (64 evaluations) / 1e-3 (396 evaluations with CFG); decoded images
<= 1e-3 (north_star).  The SD-VAE oracle is PARITY UNPINNED (diffusers + weights absent): those cases are GPU <-> own restatement."""
import os
import sys

import pytest
import torch

from conftest import ROOT, load_golden, rel_l2
from oracle import flow_oracle as fo
from oracle import sdvae_oracle as vo
from oracle.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def unet32():
    from flocoder_amd.unet import Unet
    g = load_golden("g3_unet_d32c102")
    sd = synth_state_dict(g["shapes"], 1)
    m = Unet(dim=32, dim_mults=(1, 2, 4, 8), channels=4, n_classes=102).eval()
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), sd


@pytest.fixture(scope="module")
def sdvae():
    from flocoder_amd.codecs import SD_VAE_Wrapper
    w = SD_VAE_Wrapper(weights="random", seed=7).eval().to(DEV)
    sd = {k[4:]: v.detach().cpu() for k, v in w.state_dict().items()}
    return w, sd


def test_a_euler64_then_sdvae_decode_meets_the_1e3_gate(unet32, sdvae):
    from flocoder_amd import sampling as S
    model, sd = unet32
    codec, vsd = sdvae
    B = 2
    src = synth_input("gate.src", (B, 4, 32, 32), 1)
    cls = torch.tensor([17, 93])
    ref_lat, _ = fo.euler_sampler(sd, src, 64, cls)
    # SD latents are unscaled VAE means, std ~4.5: bring the flow's output to that range before decoding (both sides alike)
    scale = 4.5 / float(ref_lat.std())
    ref_img = vo.decode(vsd, ref_lat * scale)
    lat, nfe = S.euler_sampler(model, (B, 4, 32, 32), 64, cond=cls.to(DEV), source=src.to(DEV))
    img = S.decode_latents(codec, lat * scale)
    assert nfe == 64 and img.shape == (B, 3, 256, 256)
    e_lat, e_img = rel_l2(lat.cpu(), ref_lat), rel_l2(img.cpu(), ref_img)
    max_abs = float((img.cpu() - ref_img).abs().max())
    print(f"\n[gate] 64-step Euler latents rel-L2 {e_lat:.3e}; decoded 3x256x256 rel-L2 {e_img:.3e}, max-abs {max_abs:.3e} "
          f"(image range {float(ref_img.min()):.2f}..{float(ref_img.max()):.2f})")
    assert e_lat < 2e-4
    assert e_img < 1e-3, f"north_star gate: decoded images rel-L2 {e_img:.3e} > 1e-3"


def test_b_sdvae_at_256_matches_oracle(sdvae):
    codec, vsd = sdvae
    g = torch.Generator().manual_seed(21)
    z = torch.randn(2, 4, 32, 32, generator=g) * 4.5
    ref = vo.decode(vsd, z)
    y = codec.decode(z.to(DEV))
    assert y.shape == (2, 3, 256, 256)
    e = rel_l2(y.cpu(), ref)
    print(f"\n[sdvae 256] decode rel-L2 {e:.3e}, max-abs {float((y.cpu() - ref).abs().max()):.3e}")
    assert e < 2e-4
    assert torch.equal(codec.decode(z.to(DEV)), y)                       # bit-reproducible
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    refz = vo.encode_mean(vsd, x)
    zz = codec.encode(x.to(DEV))
    assert zz.shape == (2, 4, 32, 32)
    e = rel_l2(zz.cpu(), refz)
    print(f"[sdvae 256] encode rel-L2 {e:.3e}")
    assert e < 2e-4
    # the decode plan at this size uses the 256-row tile and folds many-tile GroupNorm partials once: both must be on the path tested
    kernels = codec.plan_kernels(decode=True)
    assert any("M256,N64" in k for k in kernels) and "gn_fold" in kernels, kernels
    assert abs(codec.flops_per_sample(decode=True) - 622.2e9) / 622.2e9 < 0.02


def test_c_default_plan_forward_at_bench_batch(unet32):
    """bench.py's launch configuration itself: B=64, default plan, no debug switch."""
    from flocoder_amd import sampling as S
    sys.path.insert(0, ROOT)
    import bench
    assert (bench.BATCH, bench.LATENT, bench.DIM, bench.NCLS, bench.N_EULER) == (64, (4, 32, 32), 32, 102, 64)
    assert not os.environ.get("FLOCODER_AMD_FUSED_TAIL"), "this test is about the DEFAULT plan"
    model, sd = unet32
    B = bench.BATCH
    x = synth_input("b64.x", (B, 4, 32, 32), 1)
    t = torch.linspace(1.0, 998.0, B)
    ids = (torch.arange(B) * 7) % 102
    with torch.no_grad():
        v = model(x.to(DEV), t.to(DEV), {"class_cond": ids.to(DEV)})
        v2 = model(x.to(DEV), t.to(DEV), {"class_cond": ids.to(DEV)})
    assert torch.equal(v, v2)
    ref = fo.unet_forward(sd, x[:8], t[:8], {"class_cond": ids[:8]})
    e = rel_l2(v[:8].cpu(), ref)
    assert e < 2e-5, f"B=64 default plan, rows 0-7: {e:.3e}"
    plan_fwd = [r["kernel"] for r in model.profile_ops(B, repeats=1)]
    # ... and the sampler bench.py times replays exactly this plan
    S.euler_sampler(model, (B, 4, 32, 32), 2, cond=ids.to(DEV), source=x.to(DEV))
    plan_bench = [r["kernel"] for r in model.profile_ops(B, repeats=1)]
    assert plan_fwd == plan_bench and model.chains == (1, B)
    assert model.fused_tail_errors() == 0
    # sample 5 of the 64 integrates the same alone as in the batch (size-independent property at the benchmarked size)
    full, _ = S.euler_sampler(model, (B, 4, 32, 32), 8, cond=ids.to(DEV), source=x.to(DEV))
    ref8, _ = fo.euler_sampler(sd, x[5:6], 8, ids[5:6])
    assert rel_l2(full[5:6].cpu(), ref8) < 2e-4


def test_d_rk4_100_with_cfg_error_growth(unet32):
    """BASELINE config 3's length: warp_time(linspace(0,1,100)) -> 99 intervals x 4 stages x 2 CFG rows = 792 U-Net rows per sample."""
    from flocoder_amd import sampling as S
    model, sd = unet32
    B = 2
    src = synth_input("rk100.src", (B, 4, 32, 32), 1)
    cls = torch.tensor([3, 64])
    ref, nfe_ref = fo.generate_latents_rk4(sd, src.clone(), 100, {"class_cond": cls}, 3.0)
    lat, nfe = S.generate_latents_rk4(model, (B, 4, 32, 32), 100, {"class_cond": cls.to(DEV)}, 3.0, source=src.to(DEV))
    assert nfe == nfe_ref == 400
    e = rel_l2(lat.cpu(), ref)
    print(f"\n[rk4-100 cfg3] latents rel-L2 {e:.3e} after 396 evaluations x 2 rows, max-abs {float((lat.cpu() - ref).abs().max()):.3e}")
    assert e < 1e-3


def test_e_decode_latents_in_chunks(sdvae):
    from flocoder_amd import sampling as S
    codec, vsd = sdvae
    z = synth_input("chunks.z", (5, 4, 8, 8), 3, scale=4.5)
    whole = S.decode_latents(codec, z.to(DEV), chunk_size=128)
    parts = S.decode_latents(codec, z.to(DEV), chunk_size=2)              # 2 + 2 + 1
    assert parts.shape == whole.shape == (5, 3, 64, 64) and parts.device == whole.device
    assert rel_l2(parts.cpu(), whole.cpu()) < 1e-6                        # a sample decodes the same whatever chunk it rides in
    assert rel_l2(parts.cpu(), vo.decode(vsd, z)) < 2e-4
    one = S.decode_latents(codec, z.to(DEV), chunk_size=1)
    assert rel_l2(one.cpu(), whole.cpu()) < 1e-6
