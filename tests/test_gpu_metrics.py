"""GPU parity of the Sinkhorn divergence kernels (fc_sinkhorn_divergence, through the C ABI) against the CPU restatement
(oracle/metrics_oracle.py, float64).  PARITY UNPINNED w.r.t. geomloss itself (absent): tolerance 1e-6 relative on the value --
both sides evaluate the same definition in float64 apart from the fp32 pairwise blocks of the cost kernel."""
import pytest
import torch

from oracle import metrics_oracle as mo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,m,d,scale,shift", [(64, 64, 1024, 1.0, 0.0), (37, 90, 4096, 1.0, 0.2), (256, 256, 64, 3.0, 1.0),
                                               (16, 16, 3 * 64 * 64, 0.5, 0.1), (1, 5, 8, 1.0, 0.0), (300, 128, 12, 1.0, 0.0)])
def test_sinkhorn_matches_oracle(n, m, d, scale, shift):
    from flocoder_amd.metrics import sinkhorn_divergence
    g = torch.Generator().manual_seed(n * 1000 + m + d)
    x = torch.randn(n, d, generator=g)
    y = torch.randn(m, d, generator=g) * scale + shift
    want, winfo = mo.sinkhorn_divergence(x, y, return_info=True)
    got, info = sinkhorn_divergence(x.to(DEV), y.to(DEV), return_info=True)
    assert info["iterations"] == winfo["iterations"] and abs(info["diameter"] - winfo["diameter"]) < 1e-7 * winfo["diameter"]
    assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (got, want)


def test_sinkhorn_surface_and_edge_cases():
    from flocoder_amd import metrics as M
    g = torch.Generator().manual_seed(5)
    lat_t, lat_p = torch.randn(48, 4, 16, 16, generator=g), torch.randn(48, 4, 16, 16, generator=g) * 1.1
    s = M.sinkhorn_loss(lat_t.to(DEV), lat_p.to(DEV))
    assert isinstance(s, float) and abs(s - mo.sinkhorn_loss(lat_t, lat_p)) < 1e-6 * max(1.0, s)
    assert abs(M.sinkhorn_loss(lat_t.to(DEV), lat_p.to(DEV), max_B=20) - mo.sinkhorn_loss(lat_t, lat_p, max_B=20)) < 1e-6 * max(1.0, s)
    assert abs(M.sinkhorn_loss(lat_t.to(DEV), lat_t.to(DEV))) < 1e-9                       # debiased: identical clouds -> 0
    same = torch.ones(4, 7, device=DEV)
    assert M.sinkhorn_divergence(same, same.clone()) == 0.0                                # zero diameter: nothing to transport
    ch = M.sinkhorn_loss(lat_t, lat_p, chunk=True, device=DEV)                             # metrics.py:20-38 (one 48-sample chunk)
    assert abs(ch - s) < 1e-12
    two = M.sinkhorn_loss_chunked(lat_t, lat_p, chunk_size=24, device=DEV)
    want = 0.5 * (mo.sinkhorn_loss(lat_t[:24], lat_p[:24]) + mo.sinkhorn_loss(lat_t[24:], lat_p[24:]))
    assert abs(two - want) < 1e-6 * max(1.0, want)
    with pytest.raises(ValueError):
        M.sinkhorn_divergence(torch.zeros(3, 4, device=DEV), torch.zeros(3, 5, device=DEV))
    assert M.sinkhorn_divergence(lat_t.to(DEV), lat_p.to(DEV)) == M.sinkhorn_divergence(lat_t.to(DEV), lat_p.to(DEV))   # bit-reproducible


def test_compute_sample_metrics_on_sampler_output_vs_oracle():
    """The north_star's "sinkhorn from flocoder/metrics.py" on a real hot-path result: GPU sampler latents / decoded images against
    the CPU oracle's for the same noise -- the divergence between the two sets is ~0, and each side's divergence to a common target
    set agrees."""
    from flocoder_amd import metrics as M
    from flocoder_amd import sampling as S
    from flocoder_amd.codecs import SimpleResizeAE
    from flocoder_amd.unet import Unet
    from oracle import flow_oracle as fo
    torch.manual_seed(0)
    model = Unet(dim=16, dim_mults=(1, 2, 4, 8), channels=4, n_classes=10).eval().to(DEV)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    B = 12
    src, cls = torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 10, (B,), generator=g)
    lat, _ = S.euler_sampler(model, (B, 4, 16, 16), 8, cond=cls.to(DEV), source=src.to(DEV))
    ref, _ = fo.euler_sampler(sd, src, 8, cls)
    codec = SimpleResizeAE(latent_shape=(4, 16, 16)).eval()
    img, img_ref = codec.decode(lat.cpu()), codec.decode(ref)
    target_lat = torch.randn(B, 4, 16, 16, generator=g)
    target_img = codec.decode(target_lat)
    m = M.compute_sample_metrics(lat, target_lat.to(DEV), img.to(DEV), target_img.to(DEV))
    assert {"sinkhorn", "sinkhorn_px", "mse", "mse_px", "pred_mean", "targ_std", "pred_px_std"} <= set(m) and "FID_px" not in m
    want = mo.sinkhorn_loss(target_lat, ref)
    print(f"\\n[metrics] sinkhorn(target, GPU samples) {m['sinkhorn']:.6f} vs sinkhorn(target, oracle samples) {want:.6f}; "
          f"sinkhorn(GPU samples, oracle samples) {M.sinkhorn_loss(ref.to(DEV), lat):.3e}")
    assert abs(m["sinkhorn"] - want) < 1e-4 * max(1.0, want)
    assert abs(M.sinkhorn_loss(ref.to(DEV), lat)) < 1e-6
