"""Codecs in fp32 and in the opt-in split-bf16 form, alternating on one box: SD-VAE decode / encode at B=64 in chunks of 16, the
per-kernel table of the split-bf16 decode plan, VQVAE (midi_vqgan shape) encode / decode with the rel-L2 of the split-bf16 results
(profiles/r03_split_bf16.txt)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flocoder_amd.codecs import SD_VAE_Wrapper, VQVAE
from flocoder_amd.sampling import decode_latents
dev = torch.device("cuda", 0)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
g = torch.Generator().manual_seed(1)
z = (torch.randn(64, 4, 32, 32, generator=g) * 4.5).to(dev)
for mode in ("fp32", "bf16x3", "fp32", "bf16x3"):
    vae.set_precision(mode)
    t, img = bench._gpu_time(lambda: decode_latents(vae, z, chunk_size=16), dev, 2)
    te, lat = bench._gpu_time(lambda: torch.cat([vae.encode(img[i:i + 16]) for i in range(0, 64, 16)]), dev, 2)
    print(mode, "decode %.1f ms = %.1f images/s; encode %.1f images/s" % (t * 1e3, 64 / t, 64 / te), flush=True)
vae.set_precision("bf16x3")
rows = vae.profile_ops(z[:16].contiguous(), torch.empty_like(img[:16]), decode=True, repeats=3)
by = bench._by_kernel(rows)
for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-28s %8.3f ms %3d launches %7.1f TFLOP/s(equiv)" % (k, v["ms"], v["launches"], v["flops"] / max(v["ms"], 1e-9) / 1e9))
torch.manual_seed(5)
vq = VQVAE(in_channels=3, hidden_channels=256, num_downsamples=3, internal_dim=128, vq_embedding_dim=4, codebook_levels=4, vq_num_embeddings=96).eval().to(dev)
x = torch.rand(64, 3, 128, 128, generator=g).to(dev)
for mode in ("fp32", "bf16x3"):
    vq.set_precision(mode)
    te, zz = bench._gpu_time(lambda: vq.encode(x), dev, 3)
    td, yy = bench._gpu_time(lambda: vq.decode(zz), dev, 3)
    print("vqvae", mode, "encode %.1f images/s decode %.1f images/s" % (64 / te, 64 / td), flush=True)
    if mode == "fp32": y0, z0 = yy.clone(), zz.clone()
    else: print("   rel-L2 vs fp32: encode %.2e decode %.2e" % (float((zz - z0).norm() / z0.norm()), float((vq.decode(z0) - y0).norm() / y0.norm())))
