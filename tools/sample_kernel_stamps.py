#!/usr/bin/env python3
"""Per-step time of the one-workgroup-per-sample U-Net kernel (unet_sample.hip): workgroup 0 stamps the 100 MHz clock at every step."""
import ctypes as C
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from flocoder_amd import _binding as B  # noqa: E402
from flocoder_amd.unet import Unet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(6)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=4, n_classes=0, mask_cond=True).eval().to(dev)
x = torch.randn(Bn, 4, 8, 8, device=dev)
mask = torch.rand(Bn, 4, 8, 8, device=dev)
t = torch.full((Bn,), 300.0, device=dev)
with torch.no_grad():
    m(x, t, {"mask_cond": mask})
    buf = torch.zeros(4096, dtype=torch.int64, device=dev)
    B.check(B.lib().fc_debug_set_conv_stamps(buf.data_ptr()))
    for _ in range(3):
        m(x, t, {"mask_cond": mask})
    torch.cuda.synchronize()
    B.check(B.lib().fc_debug_set_conv_stamps(None))
v = buf.cpu().tolist()
names = {0: "conv", 1: "norm", 2: "bilinear", 3: "linattn", 4: "attn", 5: "copy", 6: "attn1", 7: "linattn_w", 8: "linattn_g", 255: "end"}
rows, i = [], 0
while v[2 * i + 1] != 255 and i < 2000:
    code = v[2 * i + 1]
    rows.append((names[code & 15], (code >> 4) & 15, (code >> 8) & 4095, (code >> 20) & 4095, (code >> 32) & 255, (code >> 40) & 15, (v[2 * i + 2] - v[2 * i]) / 100.0))
    i += 1
tot = sum(r[-1] for r in rows)
ph = [v[2 * i + 2 + k] / 100.0 for k in range(3)]
print(f"{len(rows)} steps, {tot:.1f} us in the step loop of workgroup 0 (B={Bn})")
print(f"  convolution steps: {ph[0]:.1f} us staging the first weights + the barrier, {ph[1]:.1f} us multiply-add, {ph[2]:.1f} us epilogue")
by = defaultdict(lambda: [0, 0.0])
for r in rows:
    by[(r[0], r[1])][0] += 1; by[(r[0], r[1])][1] += r[-1]
for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[0]:9s} KS={k[1]}  x{n:3d}  {us:8.1f} us  ({us / n:6.2f} each)")
ns = len(rows)
for k, r in enumerate(rows):
    split = ""
    if r[0] == "conv":
        split = "   stage %.2f  multiply-add %.2f  epilogue %.2f" % tuple(v[2 * ns + 8 + 3 * k + j] / 100.0 for j in range(3))
    print(f"{r[0]:9s} KS={r[1]} Cout={r[2]:3d} Cin={r[3]:3d} H={r[4]:2d} guard={r[5]}  {r[6]:7.2f} us{split}")
