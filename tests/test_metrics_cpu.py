"""The metrics restatement (oracle/metrics_oracle.py; geomloss / torchmetrics are absent: PARITY UNPINNED) against closed forms, CPU only."""
import math

import numpy as np
import pytest
import scipy.linalg
import torch

from oracle import metrics_oracle as mo


def test_epsilon_schedule_follows_the_published_rule():
    e = mo.epsilon_schedule(4.0, 0.05, 0.5)
    assert e[0] == 16.0 and e[-1] == 0.05 ** 2
    mid = e[1:-1]
    assert abs(mid[0] - 16.0) < 1e-12 and all(abs(b / a - 0.25) < 1e-12 for a, b in zip(mid, mid[1:]))      # eps falls by scaling^p
    assert mid[-1] > 0.05 ** 2 and mid[-1] * 0.25 <= 0.05 ** 2 + 1e-15


def test_sinkhorn_divergence_properties_and_translation_limit():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(60, 5, generator=g)
    assert abs(mo.sinkhorn_divergence(x, x.clone())) < 1e-10                       # debiased: S(x, x) = 0
    y = torch.rand(45, 5, generator=g) + 0.3
    sxy, syx = mo.sinkhorn_divergence(x, y), mo.sinkhorn_divergence(y, x)
    assert sxy > 0 and abs(sxy - syx) < 1e-9 * max(1.0, sxy)                       # positive, symmetric
    # a pure translation by v: W2^2 / 2 = |v|^2 / 2 exactly, and the blur-0.05 divergence sits within O(blur^2) of it
    v = torch.tensor([0.7, -0.2, 0.1, 0.0, 0.4])
    s = mo.sinkhorn_divergence(x, x + v)
    assert abs(s - 0.5 * float(v.square().sum())) < 5e-3
    # fp32 in the package's own arithmetic agrees with the accurate evaluation on unit-cube data
    s32 = mo.sinkhorn_divergence(x, y, dtype=torch.float32)
    assert abs(s32 - sxy) < 2e-4 * max(1.0, sxy)


def test_frechet_distance_matches_matrix_square_root():
    g = torch.Generator().manual_seed(1)
    d = 24
    a, b = torch.randn(200, d, generator=g).double(), (torch.randn(300, d, generator=g) * 1.3 + 0.5).double()
    m1, s1 = mo.feature_statistics(a)
    m2, s2 = mo.feature_statistics(b)
    assert torch.allclose(s1, torch.cov(a.t()), atol=1e-12) and torch.allclose(m2, b.mean(0), atol=1e-12)
    want = float((m1 - m2).square().sum() + s1.trace() + s2.trace()
                 - 2 * np.trace(scipy.linalg.sqrtm((s1 @ s2).numpy()).real))
    got = mo.frechet_distance(m1, s1, m2, s2)
    assert abs(got - want) < 1e-8 * max(1.0, abs(want))
    assert abs(mo.frechet_distance(m1, s1, m1, s1)) < 1e-8
    # commuting case in closed form: S1 = I, S2 = 4 I, means 0 / 1  ->  d + (1 + 4 - 2 * 2) d = 2 d
    eye = torch.eye(d, dtype=torch.float64)
    assert abs(mo.frechet_distance(torch.zeros(d).double(), eye, torch.ones(d).double(), 4 * eye) - 2 * d) < 1e-9


def test_to_uint8_is_a_per_image_stretch():
    x = torch.tensor([[[[0.0, 0.5], [1.0, 2.0]]], [[[3.0, 3.0], [3.0, 3.0]]]])
    u = mo.to_uint8(x)
    assert u.dtype == torch.uint8 and u[0].flatten().tolist() == [0, 63, 127, 255] and u[1].max() == 0


def test_product_metrics_module_fails_loudly_without_gpu_or_weights(monkeypatch):
    from flocoder_amd import metrics as M
    with pytest.raises(RuntimeError, match="no CPU path"):
        M.sinkhorn_divergence(torch.zeros(3, 2), torch.zeros(3, 2))
    monkeypatch.delenv("FLOCODER_FID_INCEPTION", raising=False)
    with pytest.raises(FileNotFoundError, match="never touches the network"):
        M.fid_score(torch.zeros(2, 3, 8, 8), torch.zeros(2, 3, 8, 8), device="cpu")
    # the statistics half runs anywhere: a callable extractor, CPU tensors
    g = torch.Generator().manual_seed(3)
    proj = torch.randn(3 * 8 * 8, 16, generator=g)
    net = lambda u8: u8.float().flatten(1) @ proj
    real, fake = torch.rand(40, 3, 8, 8, generator=g), torch.rand(40, 3, 8, 8, generator=g) * 0.7
    got = M.fid_score(real, fake, device="cpu", inception=net)
    want = mo.frechet_distance(*mo.feature_statistics(net(mo.to_uint8(real))), *mo.feature_statistics(net(mo.to_uint8(fake))))
    assert abs(got - want) < 1e-6 * max(1.0, abs(want))
    chunked = M.fid_score(real, fake, device="cpu", inception=net, chunk=True, chunk_size=16)
    assert abs(chunked - want) < 1e-6 * max(1.0, abs(want))
