#!/usr/bin/env python3
"""Experiment: forward + loss + backward of one batch as ONE chain of B rows against TWO concurrent chains of B/2 rows (two library
objects with the same weights on two streams).  GroupNorm is per sample, so the two half-batch gradients add up to the full-batch one.

    python tools/exp_two_chains_train.py [--batch B --dim D --hw S --iters K]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    from flocoder_amd import _binding as B
    from flocoder_amd.unet import Unet
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    nets = [Unet(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=4, n_classes=args.classes).to(dev) for _ in range(3)]
    for n in nets[1:]:
        n.load_state_dict(nets[0].state_dict())
    for n in nets:
        n.train()
    bs = args.batch
    x = torch.randn(bs, 4, args.hw, args.hw, device=dev)
    vt = torch.randn_like(x)
    t = torch.rand(bs, device=dev) * 999
    cls = torch.randint(args.classes, (bs,), device=dev)
    lib = B.lib()
    scal = torch.zeros(8, device=dev)
    ws = [torch.zeros(256, device=dev) for _ in range(3)]
    grads = [torch.zeros(nets[0]._flat_numel, device=dev) for _ in range(3)]

    def fb(net, g, w, xs, ts, cs, vs):
        v = net._forward_native(xs, ts, cs, None, train=True)
        dv = torch.empty_like(v)
        B.check(lib.fc_mse_loss_grad(B.ptr(v), B.ptr(vs), B.ptr(dv), scal.data_ptr(), w.data_ptr(), v.numel(), B.current_stream(dev)))
        net.backward_native(xs, ts, cs, dv, g)

    def one():
        fb(nets[0], grads[0], ws[0], x, t, cls, vt)

    h = bs // 2
    parts = [(x[:h].contiguous(), t[:h].contiguous(), cls[:h].contiguous(), vt[:h].contiguous()),
             (x[h:].contiguous(), t[h:].contiguous(), cls[h:].contiguous(), vt[h:].contiguous())]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def two():
        main_s = torch.cuda.current_stream(dev)
        for k in range(2):
            streams[k].wait_stream(main_s)
            with torch.cuda.stream(streams[k]):
                fb(nets[1 + k], grads[1 + k], ws[1 + k], *parts[k])
        for k in range(2):
            main_s.wait_stream(streams[k])
        grads[1].add_(grads[2])

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(2)

    def chain(k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[k]):
            fb(nets[1 + k], grads[1 + k], ws[1 + k], *parts[k])

    def two_threads():
        main_s = torch.cuda.current_stream(dev)
        for k in range(2):
            streams[k].wait_stream(main_s)
        futs = [pool.submit(chain, k) for k in range(2)]
        for f in futs:
            f.result()
        for k in range(2):
            main_s.wait_stream(streams[k])
        grads[1].add_(grads[2])

    for name, fn in (("one chain", one), ("two chains", two), ("two threads", two_threads), ("one chain", one), ("two chains", two), ("two threads", two_threads)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{name:10s}: {1e3 * t_all / args.iters:7.3f} ms per forward+loss+backward (host enqueue {1e3 * t_host / args.iters:7.3f} ms)", flush=True)
    for name, fn in (("one chain", one), ("two chains", two), ("two threads", two_threads)):
        th = tg = 0.0
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            th += t1 - t0; tg += t2 - t0
        print(f"{name:10s} isolated: host call {1e3 * th / 20:7.3f} ms, until the device is done {1e3 * tg / 20:7.3f} ms", flush=True)
    one(); two()
    torch.cuda.synchronize()
    a, b = grads[0], grads[1] * 0.5
    print("relative difference of the two gradients:", float((a - b).norm() / a.norm()))


if __name__ == "__main__":
    main()
