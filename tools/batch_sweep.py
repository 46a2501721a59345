#!/usr/bin/env python3
"""The sampler's time per forward against the batch size (bench model, 64-step Euler): the intercept is the fixed cost of the launch
chain, the slope the work.  Measured r02: 1.17 ms per forward at B = 8, 1.19 at 16, 1.25 at 32, 1.51 at 64, 2.34 at 128 before the lean kernel
flavours; 1.04 / 1.07 / 1.12 / 1.37 / 2.16 with them.

    python tools/batch_sweep.py"""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from flocoder_amd.sampling import euler_sampler
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
g = torch.Generator().manual_seed(1234)
for B in (8, 16, 32, 64, 96, 128):
    noise = torch.randn(B, 4, 32, 32, generator=g).to(dev)
    ids = torch.randint(102, (B,), generator=g).to(dev)
    f = lambda: euler_sampler(model, (B, 4, 32, 32), 64, cond=ids, source=noise)[0]
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f"B={B:4d}: {1e3*t:7.2f} ms per 64-step call = {1e3*t/64:6.3f} ms per forward, {B/t:7.1f} samples/s", flush=True)
