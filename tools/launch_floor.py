#!/usr/bin/env python3
"""What one dependent launch costs on this machine with nothing to do: a captured graph of N dependent one-element kernels, and the
same with a 32 MB tensor written by each kernel (dirty data for the end-of-kernel write-back).

    python tools/launch_floor.py"""
import time
import torch

dev = torch.device("cuda", 0)
N = 400
for label, numel in (("1 element", 1), ("1 M floats (4 MB)", 1 << 20), ("8 M floats (32 MB)", 8 << 20)):
    x = torch.zeros(numel, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                x.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / (10 * N)
    print(f"{label:22s}: {1e6 * t:6.2f} us per dependent launch in a replayed graph")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10 * N):
        x.add_(1.0)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / (10 * N)
    print(f"{label:22s}: {1e6 * t:6.2f} us per launch from the host (stream order)")
