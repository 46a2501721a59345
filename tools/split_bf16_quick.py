import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents
dev = torch.device("cuda", 0)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
g = torch.Generator().manual_seed(1)
z = (torch.randn(64, 4, 32, 32, generator=g) * 4.5).to(dev)
vae.set_precision("bf16x3")
t, img = bench._gpu_time(lambda: decode_latents(vae, z, chunk_size=16), dev, 2)
print("bf16x3 decode %.1f ms = %.1f images/s" % (t * 1e3, 64 / t), flush=True)
