#!/usr/bin/env python3
"""Isolated timing of the implicit-GEMM kernel on the U-Net's layer shapes (GPU box only).

    python tools/conv_microbench.py [--batch 64] [--only NAME] [--tiles M128N32,M64N32K2,...]

Prints, per layer shape and tile, microseconds per launch (back-to-back launches, warm caches) and TFLOP/s against
the 157.3 TFLOP/s fp32-MFMA peak.  FLOCODER_AMD_CONV=simple selects the synchronous kernel for an A/B.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flocoder_amd._ops import conv_debug  # noqa: E402

# name, Cin0, Cin1, Cout, H, ks, pad, stride, ups, groups_out
LAYERS = [
    ("L0 32>32 3x3 @32", 32, 0, 32, 32, 3, 1, 1, 0, 4),
    ("L0 64>32 3x3 @32 cat", 32, 32, 32, 32, 3, 1, 1, 0, 4),
    ("L0 qkv 32>384 1x1 @32", 32, 0, 384, 32, 1, 0, 1, 0, 0),
    ("L0 out 128>32 1x1 @32", 128, 0, 32, 32, 1, 0, 1, 0, 1),
    ("L1 32>32 3x3 @16", 32, 0, 32, 16, 3, 1, 1, 0, 4),
    ("L1 96>64 3x3 @16 cat", 64, 32, 64, 16, 3, 1, 1, 0, 4),
    ("L1 64>64 3x3 @16", 64, 0, 64, 16, 3, 1, 1, 0, 4),
    ("L2 64>64 3x3 @8", 64, 0, 64, 8, 3, 1, 1, 0, 4),
    ("L2 192>128 3x3 @8 cat", 128, 64, 128, 8, 3, 1, 1, 0, 4),
    ("L2 128>128 3x3 @8", 128, 0, 128, 8, 3, 1, 1, 0, 4),
    ("L3 128>128 3x3 @4", 128, 0, 128, 4, 3, 1, 1, 0, 4),
    ("L3 256>256 3x3 @4", 256, 0, 256, 4, 3, 1, 1, 0, 4),
    ("L3 384>256 3x3 @4 cat", 256, 128, 256, 4, 3, 1, 1, 0, 4),
    ("down 32>32 s2d @32>16", 32, 0, 32, 32, 2, 0, 2, 0, 0),
    ("up 64>32 3x3 ups @16>32", 64, 0, 32, 16, 3, 1, 1, 1, 0),
    ("up 256>128 3x3 ups @4>8", 256, 0, 128, 4, 3, 1, 1, 1, 0),
]
VAE_LAYERS = [  # SD-VAE decoder shapes (--vae, use --batch 16)
    ("V 512>512 3x3 @64", 512, 0, 512, 64, 3, 1, 1, 0, 32),
    ("V 256>256 3x3 @128", 256, 0, 256, 128, 3, 1, 1, 0, 32),
    ("V 128>128 3x3 @256", 128, 0, 128, 256, 3, 1, 1, 0, 32),
    ("V up 512>512 3x3 ups @64>128", 512, 0, 512, 64, 3, 1, 1, 1, 32),
]
TILES = ["auto", "M128N32", "M128N64", "M64N32K2", "M32N32K4", "M64N64K2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    ap.add_argument("--tiles", default=",".join(TILES))
    ap.add_argument("--repeats", type=int, default=50)
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--precision", default="fp32", help="fp32 | bf16x3 (the split-bf16 form, codec tiles only)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tiles = args.tiles.split(",")
    print(f"{'layer':28s} " + " ".join(f"{t:>16s}" for t in tiles) + "   (us | TFLOP/s)")
    for name, c0, c1, co, H, ks, pad, stride, ups, G in (VAE_LAYERS if args.vae else LAYERS):
        if args.only and args.only not in name:
            continue
        B = args.batch
        x0 = torch.randn(B, c0, H, H, device=dev)
        x1 = torch.randn(B, c1, H, H, device=dev) if c1 else None
        w = torch.randn(co, c0 + c1, ks, ks, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        Ho = H * 2 if ups else (H + 2 * pad - ks) // stride + 1
        flops = 2.0 * B * Ho * Ho * ks * ks * (c0 + c1) * co
        cells = []
        for t in tiles:
            try:
                ms = conv_debug(x0, w, b, x1=x1, pad=pad, stride=stride, upsample=bool(ups), groups_out=G, tile=t, repeats=args.repeats, precision=args.precision)
                cells.append(f"{ms * 1e3:7.1f} |{flops / ms / 1e9:7.1f}")
            except ValueError:
                cells.append(f"{'n/a':>16s}")
        print(f"{name:28s} " + " ".join(f"{c:>16s}" for c in cells), flush=True)


if __name__ == "__main__":
    main()
