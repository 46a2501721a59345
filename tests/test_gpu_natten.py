"""N4: 2-D neighbourhood attention (fc_na2d) and the VQVAE with NATTENBlocks, through the C ABI, against the CPU restatement
(oracle/vqvae_oracle.py na2d / natten_block).  PARITY UNPINNED: the natten package (pyproject.toml:40) is absent; what is held equal is
the HIP path and the restated published definition, for BOTH readings of the reference's call."""
import pytest
import torch

from conftest import rel_l2
from oracle import vqvae_oracle as vq
from oracle.synth import synth_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _na2d_gpu(qkv_nhwc, C, heads, k, mode, gamma=None):
    from flocoder_amd import _binding as B
    Bn, H, W, _ = qkv_nhwc.shape
    out = torch.full((Bn, H, W, C), float("nan"), device=DEV)
    B.check(B.lib().fc_na2d(B.ptr(qkv_nhwc), B.ptr(out), B.ptr(gamma), Bn, H, W, C, heads, k, mode, B.current_stream(qkv_nhwc.device)))
    return out


@pytest.mark.parametrize("Bn,H,W,C,k", [(2, 16, 16, 64, 7), (1, 8, 32, 128, 7), (3, 7, 9, 32, 7), (2, 16, 16, 1024, 7), (2, 8, 8, 32, 3)])
@pytest.mark.parametrize("mode", [1, 2])
def test_na2d_matches_restated_definition(Bn, H, W, C, k, mode):
    heads, d = 8, C // 8
    if mode == 2 and (heads < k or H < k):
        pytest.skip("layout 2 attends over (head index, image row): needs both >= kernel size")
    g = torch.Generator().manual_seed(Bn * 1000 + H * 10 + C + mode)
    qkv = torch.randn(Bn, H, W, 3 * C, generator=g)
    t = qkv.reshape(Bn, H, W, 3, heads, d).permute(3, 0, 4, 1, 2, 5)                 # 3 B heads H W d, as codecs.py:122-124 builds them
    q, kk, v = t[0], t[1], t[2]
    if mode == 1:
        ref = vq.na2d(q.permute(0, 2, 3, 1, 4), kk.permute(0, 2, 3, 1, 4), v.permute(0, 2, 3, 1, 4), k).permute(0, 3, 1, 2, 4)
    else:
        ref = vq.na2d(q, kk, v, k)
    ref = ref.permute(0, 2, 3, 1, 4).reshape(Bn, H, W, C)
    out = _na2d_gpu(qkv.to(DEV).contiguous(), C, heads, k, mode)
    assert rel_l2(out.cpu(), ref) < 2e-6
    gam = torch.tensor([0.37], device=DEV)
    assert rel_l2(_na2d_gpu(qkv.to(DEV).contiguous(), C, heads, k, mode, gam).cpu(), 0.37 * ref) < 2e-6


def test_na2d_rejects_windows_larger_than_the_image():
    from flocoder_amd import _binding as B
    qkv = torch.zeros(1, 4, 4, 96, device=DEV)
    out = torch.zeros(1, 4, 4, 32, device=DEV)
    with pytest.raises(ValueError, match="larger than the attended axes"):
        B.check(B.lib().fc_na2d(B.ptr(qkv), B.ptr(out), None, 1, 4, 4, 32, 8, 7, 1, B.current_stream(qkv.device)))


@pytest.mark.parametrize("layout", [1, 2])
def test_vqvae_with_natten_blocks_vs_oracle(layout):
    from flocoder_amd.codecs import VQVAE
    cfg = dict(in_channels=1, hidden_channels=32, num_downsamples=3, internal_dim=32, vq_embedding_dim=4, decoder_nonlocal=False)
    m = VQVAE(natten_layout=layout, **cfg).eval()
    own = {k: v for k, v in m.state_dict().items() if k != "codebook_usage" and not k.startswith("vq.") and "noise" not in k.lower()}
    natten_keys = [k for k in own if ".attn." in k]
    # the reference's NATTENBlock names (codecs.py:101-105), in the blocks it gives them to (codecs.py:266,275-278,414-429)
    assert {k.rsplit(".attn.", 1)[1] for k in natten_keys} == {"gamma", "norm.weight", "norm.bias", "qkv.weight", "proj.weight"}
    blocks = sorted({k.rsplit(".attn.", 1)[0] for k in natten_keys})
    assert blocks == ["decoder.layers.10", "decoder.layers.5", "encoder.2", "encoder.3", "encoder.4", "encoder.5", "encoder.6"]
    assert all(float(own[k]) == 0.0 for k in natten_keys if k.endswith("gamma"))    # the gate starts closed, as upstream
    sd = {k: synth_tensor(k, v.shape, 4) for k, v in own.items()}
    for k in natten_keys:
        if k.endswith("gamma"):
            sd[k] = torch.tensor([0.8])                                              # open the gate so the attention matters
        elif k.endswith(("qkv.weight", "proj.weight")):
            sd[k] = sd[k] * 1.5
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(layout)
    x = torch.rand(2, 1, 128, 128, generator=g)
    z = m.encode(x.to(DEV))
    sd64 = {k: v.double() for k, v in sd.items()}                                   # float64 reference: sharp softmaxes amplify fp32 rounding
    ref = vq.encode(sd64, x.double(), natten_layout=layout)
    assert z.shape == ref.shape == (2, 4, 16, 16)
    e = rel_l2(z.cpu(), ref)
    zz = torch.randn(2, 4, 16, 16, generator=g)
    y = m.decode(zz.to(DEV))
    ry = vq.decode(sd64, zz.double(), natten_layout=layout)
    e2 = rel_l2(y.cpu(), ry)
    print(f"\n[natten layout {layout}] encode {e:.2e} decode {e2:.2e}")
    assert e < 5e-5 and e2 < 5e-5
    # and the attention is really on the path: closing the gates changes the result
    sd0 = dict(sd)
    for k in natten_keys:
        if k.endswith("gamma"):
            sd0[k] = torch.zeros(1)
    assert rel_l2(vq.encode(sd0, x, natten_layout=layout).double(), ref) > 1e-3
    other = 3 - layout
    assert rel_l2(vq.encode(sd, x, natten_layout=other).double(), ref) > 1e-4           # the two readings of the call are different functions
