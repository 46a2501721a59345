"""Split-bf16 against fp32, layer by layer (conv_debug) and for the whole SD-VAE decode at the bench's batch: rel-L2 of every case."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flocoder_amd._ops import conv_debug
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
for (B, ci, co, H, ks, ups) in [(16, 512, 512, 32, 3, 0), (16, 256, 256, 64, 3, 0), (16, 128, 128, 128, 3, 0), (16, 512, 512, 32, 3, 1), (16, 256, 128, 64, 3, 0),
                                (16, 256, 128, 64, 1, 0), (16, 512, 512, 32, 1, 0), (2, 512, 512, 32, 3, 0), (16, 128, 3, 64, 3, 0), (16, 4, 512, 32, 3, 0)]:
    x = torch.randn(B, ci, H, H, generator=g).to(dev)
    w = (torch.randn(co, ci, ks, ks, generator=g) * 0.05).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    for tile in ("auto", "M256N64", "M128N64", "M128N32"):
        try:
            y0, _ = conv_debug(x, w, b, pad=ks // 2, upsample=bool(ups), tile=tile)
            y1, _ = conv_debug(x, w, b, pad=ks // 2, upsample=bool(ups), tile=tile, precision="bf16x3")
        except ValueError as e:
            print(B, ci, co, H, ks, ups, tile, "n/a"); continue
        print(B, ci, co, H, ks, ups, tile, "rel-L2 %.3e  same-bits %s" % (float((y1 - y0).norm() / y0.norm()), bool(torch.equal(y0, y1))), flush=True)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
for B in (2, 16, 64):
    z = (torch.randn(B, 4, 32, 32, generator=g) * 4.5).to(dev)
    vae.set_precision("fp32")
    a = decode_latents(vae, z, chunk_size=16).clone()
    vae.set_precision("bf16x3")
    c = decode_latents(vae, z, chunk_size=16)
    print("decode B=%d rel-L2 %.3e finite %s" % (B, float((c - a).norm() / a.norm()), bool(torch.isfinite(c).all())), flush=True)
