"""Per-launch table of the SD-VAE decode plan (16 latents) in fp32 and in split-bf16: where does the 3-MFMA form fall short of 3x?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents
dev = torch.device("cuda", 0)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
g = torch.Generator().manual_seed(1)
z = (torch.randn(16, 4, 32, 32, generator=g) * 4.5).to(dev)
tabs = {}
for mode in ("fp32", "bf16x3"):
    vae.set_precision(mode)
    img = decode_latents(vae, z, chunk_size=16)
    tabs[mode] = vae.profile_ops(z, torch.empty_like(img), decode=True, repeats=5)
print("%-3s %-26s %-34s %9s %9s %6s %8s" % ("#", "kernel", "module", "fp32 ms", "bf16x3 ms", "x", "TF equiv"))
for i, (a, b) in enumerate(zip(tabs["fp32"], tabs["bf16x3"])):
    fl = b["flops_per_sample"] * b["rows"]
    print("%-3d %-26s %-34s %9.3f %9.3f %6.2f %8.1f" % (i, b["kernel"], b["module"][:34], a["ms"], b["ms"], a["ms"] / max(b["ms"], 1e-9), fl / max(b["ms"], 1e-9) / 1e9))
print("total %.2f -> %.2f ms" % (sum(r["ms"] for r in tabs["fp32"]), sum(r["ms"] for r in tabs["bf16x3"])))
