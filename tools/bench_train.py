#!/usr/bin/env python3
"""Training-step benchmark (BASELINE.json configs[3]): stl_sd.yaml class-conditional flow, one train_flow.py step =
OT pairing + interpolation + U-Net forward/backward + clip + Adam + EMA, batch 32 per GPU (global 256 on 8 GPUs),
latents 4x16x16, U-Net dim=16 dim_mults [1,2,4,8] n_classes=10.  `--dim 32 --hw 32` times the flowers_sd-sized model.

    python tools/bench_train.py [--steps K --warmup W --batch B --dim D --hw S]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_train.py --gpus N

Rank 0 prints one JSON line (same conventions as bench.py; value = whole-job samples/s, max-over-ranks time)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--no-ot", action="store_true")
    args = ap.parse_args()
    from flocoder_amd import dist as fdist
    from flocoder_amd.ot import compute_ot_pairing
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    rank, local_rank, world = fdist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", 0 if os.environ.get("FLOCODER_AMD_SINGLE_GPU") else local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    model = Unet(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=4, n_classes=args.classes).to(device)
    fdist.broadcast_weights(model, src=0)
    tr = FlowTrainer(model, lr=1e-4)
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    target = torch.randn(B, 4, args.hw, args.hw, generator=g).to(device)
    cls = torch.randint(args.classes, (B,), generator=g).to(device)

    def step():
        source = torch.randn_like(target)
        return tr.step(source, target, {"class_cond": cls, "mask_cond": None}, pairing=None if args.no_ot else compute_ot_pairing(source, target))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss)
    fwd = model.flops_per_sample
    line = {"metric": "train samples/sec (flow step: OT pairing + U-Net fwd/bwd + clip + Adam + EMA)",
            "value": round(B * world * args.steps / elapsed, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"train_flow step, batch {B} per GPU, latents 4x{args.hw}x{args.hw}, U-Net dim={args.dim} dim_mults [1,2,4,8] "
                                   f"n_classes={args.classes}, greedy OT pairing {'off' if args.no_ot else 'on'}, Adam lr 1e-4, EMA 0.999",
                       "global_batch": B * world, "parallelism": f"data-parallel x{world}, one flat-gradient all-reduce per step",
                       "fwd_gflop_per_sample": round(fwd / 1e9, 4)},
            "step_tflops_3x_fwd": round(3 * fwd * B * world * args.steps / elapsed / 1e12, 3), "final_loss": round(float(loss), 5)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
