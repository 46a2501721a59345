#!/usr/bin/env python3
"""Headline benchmark: latent samples/sec of the 64-step Euler sampler on SD-VAE-shaped latents (4x32x32),
flowers_sd U-Net (dim=32, dim_mults [1,2,4,8], n_classes=102), batch 64 per GPU  (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch: noise -> 64 Euler steps (64 U-Net evaluations) -> final latents,
inputs already resident in HBM.  Weak scaling: every rank integrates its own 64-sample shard, weights are broadcast
once from rank 0 over RCCL before the timed region, there is no per-step communication.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the dominant kernel (largest share of device time) priced against the exact-fp32 MFMA peak, with
                its per-launch duration measured live by HIP events on the launching stream (fc_unet_profile_ops)
  cpu_baseline  the CPU oracle (oracle/flow_oracle.py, a port of the reference's PyTorch path) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
N_EULER, BATCH, LATENT, DIM, NCLS = 64, 64, (4, 32, 32), 32, 102


def build_model(device):
    from flocoder_amd.unet import Unet
    torch.manual_seed(0)      # reference-identical default init (tests/test_abi_and_host.py)
    return Unet(dim=DIM, dim_mults=(1, 2, 4, 8), channels=LATENT[0], n_classes=NCLS).eval().to(device)


def synthetic_inputs(rank, world, device):
    """Global noise / class ids generated once from fixed seeds and sliced per rank (1-GPU and 8-GPU runs see the same
    samples in the same order)."""
    from flocoder_amd.dist import shard_range
    g = torch.Generator(device="cpu").manual_seed(1234)
    noise = torch.randn((BATCH * world,) + LATENT, generator=g)
    ids = torch.randint(NCLS, (BATCH * world,), generator=torch.Generator(device="cpu").manual_seed(1235))
    lo, hi = shard_range(BATCH * world, rank, world)
    return noise[lo:hi].to(device).contiguous(), ids[lo:hi].to(device)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    tools/pmc_forward.py, summarised by tools/pmc_summary.py with the guide's gfx950 correction).  A process cannot collect
    hardware counters on itself, so the figure comes from the newest profiles/*pmc_traffic.json; null when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d["per_kernel"][kernel]["traffic_bytes_per_launch"], os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


def roofline(model, batch):
    rows = model.profile_ops(batch, repeats=20)
    by = {}
    for r in rows:
        k = by.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, launches=0))
        k["ms"] += r["ms"]; k["flops"] += r["flops_per_sample"] * r["rows"]; k["launches"] += 1
    name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
    total_ms = sum(v["ms"] for v in by.values())
    rows_timed = rows[0]["rows"]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(name)
    return {
        "bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (avg over the kernel's launch sites)",
        "traffic_source": traffic_src,
        "launches_per_forward": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
        "flops_per_launch_avg": dom["flops"] / dom["launches"], "share_of_forward_time": round(dom["ms"] / total_ms, 3),
        "forward_sum_of_kernels_ms": round(total_ms, 4),
        "rows_per_launch": rows_timed, "chains": model.chains[0],
        "forward_tflops_all_kernels": round(model.flops_per_sample * rows_timed / (total_ms * 1e-3) / 1e12, 3),
        "per_kernel": {k: dict(ms=round(v["ms"], 4), launches=v["launches"], tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 3))
                       for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])},
    }


def cpu_baseline(model):
    """The CPU oracle on this box's host cores: 8 samples x 8 of the 64 Euler steps (cost per step is constant), scaled to
    a full trajectory.  Oracle = test infrastructure used here only as the timed baseline."""
    from oracle import flow_oracle as fo
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    b, steps = 8, 8
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((b,) + LATENT, generator=g)
    ids = torch.randint(NCLS, (b,), generator=torch.Generator().manual_seed(1235))
    ts = fo.euler_time_grid(N_EULER)[:steps]
    with torch.no_grad():
        fo.unet_forward(sd, x, torch.full((b,), 500.0), {"class_cond": ids})     # warm-up
        t0 = time.perf_counter()
        for t in ts:
            x = x + fo.unet_forward(sd, x, torch.ones(b) * t * 999, {"class_cond": ids}) * (1.0 / N_EULER)
        dt = time.perf_counter() - t0
    return {"value": round(b / (dt * N_EULER / steps), 4), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle U-Net, batch {b}, {steps} of {N_EULER} Euler steps timed ({dt:.2f} s), scaled x{N_EULER // steps}",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from flocoder_amd import dist as fdist
    from flocoder_amd.sampling import euler_sampler
    rank, local_rank, world = fdist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to time (the CPU oracle is only the baseline leg)")
    # FLOCODER_AMD_SINGLE_GPU=1 (with FLOCODER_AMD_DIST_BACKEND=gloo): every rank on cuda:0, a rehearsal of the N>1 path on a 1-GPU box
    device = torch.device("cuda", 0 if os.environ.get("FLOCODER_AMD_SINGLE_GPU") else local_rank)
    torch.cuda.set_device(device)

    model = build_model(device)
    fdist.broadcast_weights(model, src=0)            # the one collective: frozen weights over xGMI
    noise, ids = synthetic_inputs(rank, world, device)
    shape = (BATCH,) + LATENT

    def step():
        return euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    assert model.fused_tail_errors() == 0, "a fused Block tail timed out waiting for its sample group: results invalid"

    line = {
        "metric": "latent samples/sec (64-step Euler, SD-VAE 4x32x32)", "value": round(BATCH * world * args.steps / elapsed, 3),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "flowers_sd.yaml shape: 64-step legacy Euler, batch 64 per GPU, latents 4x32x32, U-Net dim=32 "
                               "dim_mults [1,2,4,8] n_classes=102, class-conditional, no CFG, ODE loop only (no VAE decode)",
                   "global_batch": BATCH * world, "nfe_per_sample": N_EULER, "parallelism": f"sample-shard x{world}, weights broadcast once",
                   "gflop_per_sample": round(model.flops_per_sample * N_EULER / 1e9, 3),
                   "launches_per_forward": model.launches_per_forward},
    }
    line["ode_tflops"] = round(line["value"] * model.flops_per_sample * N_EULER / 1e12, 3)
    if rank == 0:
        if not args.no_roofline:
            line["roofline"] = roofline(model, BATCH)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model)
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
