#!/usr/bin/env python3
"""Where one flow training step spends its time on the device: HIP events around the phases of tools/bench_train.py's step
(pairing | step prologue in torch | interpolation + forward | loss + backward | clip + Adam + EMA), averaged over the timed steps,
next to the wall clock per step.  The event times include any idle time the stream spent waiting for the host.

    python tools/train_phases.py [--steps K --warmup W --batch B --dim D --hw S --classes C]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--classes", type=int, default=10)
    args = ap.parse_args()
    from flocoder_amd.ot import compute_ot_pairing
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = Unet(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=4, n_classes=args.classes).to(device)
    tr = FlowTrainer(model, lr=1e-4)
    B = args.batch
    target = torch.randn(B, 4, args.hw, args.hw, device=device)
    cls = torch.randint(args.classes, (B,), device=device)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    orig_prep, orig_lg, orig_opt = tr.prepare, tr.loss_and_grads, tr.optimizer_step

    def prep(*a, **k):                       # FlowTrainer.step's prologue is ONE library call since round 2 (fc_flow_prepare)
        mark("host: tensor checks / casts")
        r = orig_prep(*a, **k)
        mark("step prologue (fc_flow_prepare)")
        return r

    def lg(*a, **k):
        r = orig_lg(*a, **k)
        mark("forward + loss + backward")
        return r

    def opt(*a, **k):
        mark("flags / gradient averaging")
        r = orig_opt(*a, **k)
        mark("clip + Adam + EMA + repack")
        return r

    tr.prepare, tr.loss_and_grads, tr.optimizer_step = prep, lg, opt
    acc = {}
    wall = 0.0
    for it in range(args.warmup + args.steps):
        marks.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mark("start")
        source = torch.randn_like(target)
        tgt = target[compute_ot_pairing(source, target)]
        mark("noise + OT pairing + gather")
        tr.step(source, tgt, {"class_cond": cls, "mask_cond": None})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if it >= args.warmup:
            wall += t1 - t0
            for (_, a), (n, b) in zip(marks[:-1], marks[1:]):
                acc[n] = acc.get(n, 0.0) + a.elapsed_time(b)
    print(f"batch {B} dim {args.dim} latents 4x{args.hw}x{args.hw}: one step at a time (host synchronised between steps)")
    for n, v in acc.items():
        print(f"  {n:32s} {v / args.steps:7.3f} ms")
    print(f"  {'sum of phases':32s} {sum(acc.values()) / args.steps:7.3f} ms    wall per isolated step {1e3 * wall / args.steps:7.3f} ms")


if __name__ == "__main__":
    main()
