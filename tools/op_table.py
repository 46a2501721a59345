#!/usr/bin/env python3
"""Per-launch table of one U-Net forward at the bench shape (fc_unet_profile_ops: each launch timed alone, 20 repeats)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
model.reserve(bench.BATCH, 32, 32, dev)
rows = model.profile_ops(bench.BATCH, repeats=20)
tot = sum(r["ms"] for r in rows)
print(f"{'module':34s} {'kernel':28s} {'us':>8s} {'TFLOP/s':>8s}")
for r in rows:
    tf = r["flops_per_sample"] * r["rows"] / max(r["ms"], 1e-9) / 1e9
    print(f"{r['module']:34s} {r['kernel']:28s} {1e3 * r['ms']:8.1f} {tf:8.1f}")
print("total ms", tot)
att = sum(r["ms"] for r in rows if ".2" in r["module"] and ("downs" in r["module"] or "ups" in r["module"]))
print("linear-attention modules ms", att)
