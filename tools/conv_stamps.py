#!/usr/bin/env python3
"""Phase timeline of the pipelined conv kernel from in-kernel s_memtime stamps (diagnostic; GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flocoder_amd import _binding as B
from flocoder_amd._ops import conv_debug
from tools.conv_microbench import LAYERS, VAE_LAYERS

NAMES = ["start", "descr", "issue", "stored0", "barrier0", "mfma_done", "kreduce", "stores", "end"]
dev = torch.device("cuda:0")
only = sys.argv[1] if len(sys.argv) > 1 else ""
tile = sys.argv[2] if len(sys.argv) > 2 else "auto"
vae = len(sys.argv) > 3 and sys.argv[3] == "vae"            # SD-VAE decoder shapes at batch 16
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"          # "bf16x3": the split-bf16 form
for name, c0, c1, co, H, ks, pad, stride, ups, G in (VAE_LAYERS if vae else LAYERS):
    if only and only not in name:
        continue
    Bn = 16 if vae else 64
    x0 = torch.randn(Bn, c0, H, H, device=dev); x1 = torch.randn(Bn, c1, H, H, device=dev) if c1 else None
    w = torch.randn(co, c0 + c1, ks, ks, device=dev) * 0.05; b = torch.randn(co, device=dev)
    conv_debug(x0, w, b, x1=x1, pad=pad, stride=stride, upsample=bool(ups), groups_out=G, tile=tile, precision=prec)   # warm
    buf = torch.zeros(8192 * 8 * 16, dtype=torch.int64, device=dev)
    B.check(B.lib().fc_debug_set_conv_stamps(buf.data_ptr()))
    conv_debug(x0, w, b, x1=x1, pad=pad, stride=stride, upsample=bool(ups), groups_out=G, tile=tile, precision=prec)
    B.check(B.lib().fc_debug_set_conv_stamps(None))
    torch.cuda.synchronize()
    st = buf.view(8192, 8, 16).cpu()
    nb = int((st[:, 0, 0] != 0).sum())
    st = st[:nb].double()
    print(f"{name}: {nb} blocks")
    rel = (st[:, :, :9] - st[:, :, 0:1])      # per wave, relative to its own start (shader cycles)
    for role, sl in (("consumer", slice(0, 4)), ("loader", slice(4, 8))):
        med = rel[:, sl].median(dim=0).values.median(dim=0).values
        print(f"   {role:8s} median timeline (cycles since wave start): " + ", ".join(f"{n}={int(v)}" for n, v in zip(NAMES, med.tolist())))
    print(f"   consumer cycles at chunk barriers (sum): {int(st[:, 0:4, 11].median())};  loader: vmcnt-wait {int(st[:, 4:8, 9].median())}, "
          f"LDS stores {int(st[:, 4:8, 10].median())}, barrier {int(st[:, 4:8, 11].median())}, patch issue {int(st[:, 4:8, 12].median())}, dma issue {int(st[:, 4:8, 13].median())}")
    print(f"   consumers: stats barrier + emit {int((st[:, 0:4, 14] - st[:, 0:4, 7]).median())} cycles, output stores {int((st[:, 0:4, 8] - st[:, 0:4, 14]).median())}"
          f" (activation / residual + LDS image {int((st[:, 0:4, 9] - st[:, 0:4, 14]).median())}, barrier {int((st[:, 0:4, 10] - st[:, 0:4, 9]).median())}, "
          f"LDS read + global stores {int((st[:, 0:4, 8] - st[:, 0:4, 10]).median())})")
    span = (st[:, :, 8].max(dim=1).values - st[:, :, 0].min(dim=1).values)
    print(f"   workgroup lifetime: median {int(span.median())} cycles, max {int(span.max())}")
    # the launch as a whole on the device-wide 100 MHz clock (slot 15 = that clock at each wave's start): when workgroups start relative
    # to the first one, and first start -> last end taking each workgroup's own lifetime at 2.4 GHz
    rt = st[:, :, 15].min(dim=1).values * 10.0                       # ns
    t0 = rt.min()
    ends = rt + span / 2.4
    print(f"   launch: workgroup starts spread over {float(rt.max() - t0) / 1e3:.2f} us (90 % started after {float((rt - t0).quantile(0.9)) / 1e3:.2f} us); "
          f"first start -> last end {float(ends.max() - t0) / 1e3:.2f} us; median workgroup lifetime {float(span.median()) / 2.4e3:.2f} us")
