#!/usr/bin/env python3
"""BASELINE config 5's flow leg alone: the mask-conditioned RK4 inpainting sampler at the midi_inpainting.yaml flow shape (U-Net dim 8, latents
4x8x8, masks through the MaskEncoder), B=64, 100 grid points = 396 evaluations.  One JSON line.  (bench.py's `secondary.config5_midi.inpaint_rk4`.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("FLOCODER_AMD_KEEP_ENV"):
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
import torch  # noqa: E402

from flocoder_amd.inpainting import MaskEncoder, mask_blending  # noqa: E402
from flocoder_amd.sampling import generate_latents_rk4  # noqa: E402
from flocoder_amd.unet import Unet  # noqa: E402

B, N = 64, 100
dev = torch.device("cuda:0")
torch.manual_seed(6)
unet = Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=4, n_classes=0, mask_cond=True).eval().to(dev)
me = MaskEncoder().eval().to(dev)
g = torch.Generator().manual_seed(55)
pix = torch.zeros(B, 1, 128, 128)
for b in range(B):
    h0, w0 = 8 + (b * 7) % 64, 8 + (b * 13) % 64
    pix[b, :, h0:h0 + 24 + b % 32, w0:w0 + 24 + (3 * b) % 32] = 1.0
pix = pix.to(dev)
src_lat = torch.randn(B, 4, 8, 8, generator=g).to(dev)
noise = torch.randn(B, 4, 8, 8, generator=g).to(dev)


def run():
    with torch.no_grad():
        mask = me(pix)
        source = mask_blending(src_lat, mask, noise)
        return generate_latents_rk4(unet, (B, 4, 8, 8), N, {"mask_cond": mask}, 0.0, source=source)[0]


lat = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    lat = run()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 3
evals = (N - 1) * 4
print(json.dumps({"ms": round(1e3 * t, 1), "samples_per_s": round(B / t, 1), "us_per_evaluation": round(1e6 * t / evals, 1), "finite": bool(torch.isfinite(lat).all()),
                  "checksum": float(lat.double().abs().sum()), "chains": os.environ.get("FLOCODER_AMD_CHAINS"), "launches": unet.launches_per_forward}))

# the same 64 samples as K independent trajectories in flight (sampling.sample_many: K replicas, K streams)
from flocoder_amd.sampling import sample_many  # noqa: E402
for K in (2, 4, 8):
    with torch.no_grad():
        mask = me(pix)
        source = mask_blending(src_lat, mask, noise)
    n = B // K
    batches = [({"mask_cond": mask[i * n:(i + 1) * n].contiguous()}, source[i * n:(i + 1) * n].contiguous()) for i in range(K)]
    outs = sample_many(unet, (n, 4, 8, 8), batches, method="rk4", n_steps=N, cfg_strength=0.0, in_flight=K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        outs = sample_many(unet, (n, 4, 8, 8), batches, method="rk4", n_steps=N, cfg_strength=0.0, in_flight=K)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    got = torch.cat(outs)
    print(json.dumps({"in_flight": K, "ms": round(1e3 * t, 1), "samples_per_s": round(B / t, 1), "rel_to_single": float((got.double() - lat.double()).norm() / lat.double().norm())}))
