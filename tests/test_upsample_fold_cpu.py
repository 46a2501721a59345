"""The algebra behind the folded upsampling (flocoder_amd/csrc/pack.hip kinds 9 / 10, plan.h conv_up2), checked on the CPU in float64.

Reference modules: unet.py:42-46 ``Upsample`` = nn.Upsample(scale_factor=2, mode="nearest") + nn.Conv2d(dim, dim_out, 3, padding=1); the
SD-VAE decoder's Upsample2D behind codecs.py:631-652 is the same pair.  Output pixel (2y + a, 2x + b) reads, through its nine taps, only a
2x2 block of source pixels: rows {y - 1, y} for a = 0, {y, y + 1} for a = 1 (columns likewise), each through the SUM of the taps that
land on it.  So the layer equals four 2x2 convolutions on the low-resolution tensor, one per parity class, interleaved into the output.
"""
import torch
import torch.nn.functional as F


def fold_weights(w):
    """[O, I, 3, 3] -> [4 parity classes (2a + b)][O, I, 2, 2], tap (ty, tx); row sets per (a, ty): a=0: {0}, {1, 2}; a=1: {0, 1}, {2}."""
    rows = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    out = torch.zeros(4, w.shape[0], w.shape[1], 2, 2, dtype=w.dtype)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    for ky in rows[(a, ty)]:
                        for kx in rows[(b, tx)]:
                            out[2 * a + b, :, :, ty, tx] += w[:, :, ky, kx]
    return out


def folded_upsample_conv(x, w, bias):
    B, C, H, W = x.shape
    w4 = fold_weights(w)
    y = torch.zeros(B, w.shape[0], 2 * H, 2 * W, dtype=x.dtype)
    for a in range(2):
        for b in range(2):
            # the 2x2 window of class (a, b) starts (1 - a) rows above and (1 - b) columns left of the output pixel: ConvArgs::pad_y / pad_x
            xp = F.pad(x, (1 - b, b, 1 - a, a))
            y[:, :, a::2, b::2] = F.conv2d(xp, w4[2 * a + b], bias)
    return y


def test_four_parity_kernels_equal_nearest_upsampling_plus_conv3x3():
    g = torch.Generator().manual_seed(3)
    for (B, Cin, Cout, H, W) in ((2, 5, 7, 4, 6), (1, 3, 2, 1, 1), (3, 8, 8, 8, 8)):
        x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
        bias = torch.randn(Cout, generator=g, dtype=torch.float64)
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
        got = folded_upsample_conv(x, w, bias)
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < 1e-12 * ref.abs().max().clamp(min=1.0)


def test_folding_saves_five_ninths_of_the_multiply_adds():
    # per 2x2 block of output pixels and (ci, co) pair: 4 pixels x 9 taps against 4 classes x 4 taps
    assert 4 * 9 == 36 and 4 * 4 == 16 and fold_weights(torch.ones(1, 1, 3, 3)).sum() == 4 * 9    # every tap is used exactly once per class
