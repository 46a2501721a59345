#!/usr/bin/env python3
"""Secondary metric: decoded images/s = 64-step Euler + SD-VAE decode to 3x256x256, and decode / encode alone.
GPU box only.  Seeded random VAE weights (no real weights offline)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import euler_sampler, decode_latents
from flocoder_amd.unet import Unet

dev = torch.device("cuda:0")
B, chunk = int(os.environ.get("B", 64)), int(os.environ.get("CHUNK", 16))
torch.manual_seed(0)
model = Unet(dim=32, channels=4, n_classes=102).eval().to(dev)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
noise = torch.randn(B, 4, 32, 32, device=dev); ids = torch.randint(102, (B,), device=dev)

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out

z = noise * 4.5
t_dec, img = timeit(lambda: decode_latents(vae, z, chunk_size=chunk))
t_enc, _ = timeit(lambda: torch.cat([vae.encode(img[i:i + chunk]) for i in range(0, B, chunk)]))
t_ode, lat = timeit(lambda: euler_sampler(model, (B, 4, 32, 32), 64, cond=ids, source=noise)[0])
t_cfg, _ = timeit(lambda: euler_sampler(model, (B, 4, 32, 32), 64, cond=ids, source=noise, cfg_strength=3.0)[0])
from flocoder_amd.sampling import generate_latents_rk4
t_rk4, _ = timeit(lambda: generate_latents_rk4(model, (B, 4, 32, 32), 100, {"class_cond": ids}, 3.0, source=noise)[0], n=1)
gf_dec, gf_enc = vae.flops_per_sample(True) / 1e9, 272.7
print(json.dumps({"batch": B, "chunk": chunk, "decode_ms": round(t_dec * 1e3, 1), "decode_images_per_s": round(B / t_dec, 1),
                  "decode_tflops": round(B * gf_dec / t_dec / 1e3, 1), "encode_ms": round(t_enc * 1e3, 1), "encode_images_per_s": round(B / t_enc, 1),
                  "ode_ms": round(t_ode * 1e3, 1), "ode_cfg_ms": round(t_cfg * 1e3, 1), "ode_cfg_samples_per_s": round(B / t_cfg, 1),
                  "rk4_100_cfg_ms": round(t_rk4 * 1e3, 1), "rk4_100_cfg_samples_per_s": round(B / t_rk4, 2),
                  "ode_plus_decode_images_per_s": round(B / (t_ode + t_dec), 1),
                  "decode_gflop_per_image": round(gf_dec, 1), "frac_fp32_mfma_peak_decode": round(B * gf_dec / t_dec / 1e3 / 157.3, 3)}))
