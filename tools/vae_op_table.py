#!/usr/bin/env python3
"""Per-launch table of the SD-VAE decoder (or encoder with --encode) at the sampler's shape: latents 4x32x32 -> 3x256x256."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flocoder_amd import _binding as B
from flocoder_amd.codecs import SD_VAE_Wrapper

dev = torch.device("cuda", 0)
enc = "--encode" in sys.argv
bsz = int(os.environ.get("B", 16))
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
z = torch.randn(bsz, 4, 32, 32, device=dev) * 4.5
img = vae.decode(z)
inp, out = (img, vae.encode(img)) if enc else (z, img)
lib, h = B.lib(), vae._handle
n = lib.fc_vae_plan_launches(h, int(not enc))
ms = (C.c_float * n)()
B.check(lib.fc_vae_profile_ops(h, int(not enc), B.ptr(inp.contiguous()), B.ptr(out), bsz, 5, ms, n, B.current_stream(dev)))
tot, rows = 0.0, []
for i in range(n):
    k, m, f = C.c_char_p(), C.c_char_p(), C.c_double()
    B.check(lib.fc_vae_op_info(h, int(not enc), i, C.byref(k), C.byref(m), C.byref(f)))
    rows.append((m.value.decode(), k.value.decode(), ms[i], f.value * bsz / max(ms[i], 1e-9) / 1e9))
    tot += ms[i]
print(f"{'module':46s} {'kernel':26s} {'ms':>8s} {'TFLOP/s':>8s}")
for r in rows:
    print(f"{r[0]:46s} {r[1]:26s} {r[2]:8.3f} {r[3]:8.1f}")
print("total ms", tot, "batch", bsz, "images/s", bsz / tot * 1e3)
