#!/usr/bin/env python3
"""Smallest run that exercises every kernel of a codec pass once per launch site, for `rocprofv3 --pmc <counter> --kernel-trace`
(one counter group per pass): tools/pmc_summary.py turns the FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

    python tools/pmc_codec.py sdvae_decode|vqvae_encode|vqvae_decode [n_passes] [fp32|bf16x3]

sdvae_decode: SD-VAE decode of one chunk of 16 latents 4x32x32 -> 3x256x256 (bench.py's DECODE_CHUNK, seeded random weights);
vqvae_*: the midi_vqgan.yaml VQVAE at B=64, 128x128 (bench.py config5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    what = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(55)
    if what == "sdvae_decode":
        from flocoder_amd.codecs import SD_VAE_Wrapper
        codec = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
        z = (torch.randn(bench.DECODE_CHUNK, 4, 32, 32, generator=g) * 4.5).to(dev)
        fn = lambda: codec.decode(z)
    else:
        from flocoder_amd.codecs import VQVAE
        torch.manual_seed(5)
        codec = VQVAE(in_channels=3, hidden_channels=256, num_downsamples=3, internal_dim=128, vq_embedding_dim=4, codebook_levels=4,
                      vq_num_embeddings=96).eval().to(dev)
        x = torch.rand(bench.BATCH, 3, 128, 128, generator=g).to(dev)
        if what == "vqvae_encode":
            fn = lambda: codec.encode(x)
        else:
            z = torch.randn(bench.BATCH, 4, 16, 16, generator=g).to(dev)      # (no encode here: the counters should see decode launches only)
            fn = lambda: codec.decode(z)
    codec.set_precision(prec)
    with torch.no_grad():
        for _ in range(n):
            out = fn()
    torch.cuda.synchronize()
    print("ok", what, float(out.abs().mean()))


if __name__ == "__main__":
    main()
