"""Folded upsampling (four 2x2 parity convolutions, plan.h conv_up2) against the 3x3 convolution over the upsampled window
(FLOCODER_AMD_UPS_FOLD=0 in a second process is the other way to compare; here: against the CPU oracle and timing)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents
from oracle import sdvae_oracle as vo
dev = torch.device("cuda", 0)
w = SD_VAE_Wrapper(weights="random", seed=7).eval().to(dev)
sd = {k[4:]: v.detach().cpu() for k, v in w.state_dict().items()}
g = torch.Generator().manual_seed(14)
z = torch.randn(2, 4, 32, 32, generator=g) * 4.5
img = w.decode(z.to(dev))
ref = vo.decode(sd, z[:1])
print("decode vs oracle rel-L2 %.3e  finite %s" % (float((img[:1].cpu() - ref).norm() / ref.norm()), bool(torch.isfinite(img).all())), flush=True)
w.set_precision("bf16x3")
img3 = w.decode(z.to(dev))
print("split-bf16 decode vs fp32 rel-L2 %.3e" % float((img3 - img).norm() / img.norm()), flush=True)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
z64 = (torch.randn(64, 4, 32, 32, generator=g) * 4.5).to(dev)
for mode in ("fp32", "bf16x3"):
    vae.set_precision(mode)
    t, _ = bench._gpu_time(lambda: decode_latents(vae, z64, chunk_size=16), dev, 2)
    print(mode, "decode %.1f ms = %.1f images/s" % (t * 1e3, 64 / t), flush=True)
