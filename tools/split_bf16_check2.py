import os, sys
os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
import torch
sys.path.insert(0, "/root/repo")
import bench
from flocoder_amd.unet import Unet
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents, euler_sampler
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = Unet(dim=32, dim_mults=(1, 2, 4, 8), channels=4, n_classes=102).eval().to(dev)
g = torch.Generator().manual_seed(1)
noise = torch.randn(64, 4, 32, 32, generator=g).to(dev); ids = torch.randint(102, (64,), generator=g).to(dev)
lat = euler_sampler(model, (64, 4, 32, 32), 64, cond=ids, source=noise)[0]
print("lat std %.3e absmax %.3e" % (float(lat.std()), float(lat.abs().max())))
z = lat * (4.5 / float(lat.std()))
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
img = decode_latents(vae, z, chunk_size=16).clone()
vae.set_precision("bf16x3")
img3 = decode_latents(vae, z, chunk_size=16)
d = (img3.double() - img.double())
print("err %.3e  img absmax %.3e  diff absmax %.3e  finite %s" % (float(d.norm() / img.double().norm()), float(img.abs().max()), float(d.abs().max()), bool(torch.isfinite(img3).all())))
per = d.flatten(1).norm(dim=1) / img.double().flatten(1).norm(dim=1)
print("per-sample err:", ["%.1e" % float(v) for v in per])
