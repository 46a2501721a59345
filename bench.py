#!/usr/bin/env python3
"""Headline benchmark: latent samples/sec of the 64-step Euler sampler on SD-VAE-shaped latents (4x32x32),
flowers_sd U-Net (dim=32, dim_mults [1,2,4,8], n_classes=102), batch 64 per GPU  (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch: noise -> 64 Euler steps (64 U-Net evaluations) -> final latents,
inputs already resident in HBM.  Weak scaling: every rank integrates its own 64-sample shard, weights are broadcast
once from rank 0 over RCCL before the timed region, there is no per-step communication.  Rank 0 prints ONE JSON line.

Objects in the line besides the contract's fields:
  roofline      the dominant kernel (largest share of device time) of the U-Net forward priced against the exact-fp32 MFMA peak;
                its per-launch duration is measured live by HIP events on the launching stream (fc_unet_profile_ops)
  cpu_baseline  the CPU oracle (oracle/flow_oracle.py + oracle/sdvae_oracle.py, a port of the reference's PyTorch path) timed on this
                box's host cores on BASELINE config 1 (B=4, 16-step Euler), rank 0 at N=1 only
  secondary     the other BASELINE numbers on the same box: Euler with CFG, "100-step" RK4 with CFG at config 3's per-GPU share,
                SD-VAE decode (its own roofline entry) / encode, decoded images/s  (N=1: all; N>1: the RK4 share only)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
N_EULER, BATCH, LATENT, DIM, NCLS = 64, 64, (4, 32, 32), 32, 102
RK4_STEPS, RK4_SHARE, CFG = 100, 32, 3.0          # BASELINE configs[2]: 100-"step" RK4, B=256 over 8 GPUs = 32 per GPU, cfg_strength 3.0
DECODE_CHUNK = 16


def build_model(device):
    from flocoder_amd.unet import Unet
    torch.manual_seed(0)      # reference-identical default init (tests/test_abi_and_host.py)
    return Unet(dim=DIM, dim_mults=(1, 2, 4, 8), channels=LATENT[0], n_classes=NCLS).eval().to(device)


def synthetic_inputs(rank, world, device, per_rank=BATCH):
    """Global noise / class ids generated once from fixed seeds and sliced per rank (1-GPU and 8-GPU runs see the same
    samples in the same order)."""
    from flocoder_amd.dist import shard_range
    g = torch.Generator(device="cpu").manual_seed(1234)
    noise = torch.randn((per_rank * world,) + LATENT, generator=g)
    ids = torch.randint(NCLS, (per_rank * world,), generator=torch.Generator(device="cpu").manual_seed(1235))
    lo, hi = shard_range(per_rank * world, rank, world)
    return noise[lo:hi].to(device).contiguous(), ids[lo:hi].to(device)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    tools/pmc_forward.py, summarised by tools/pmc_summary.py with the guide's gfx950 correction).  A process cannot collect
    hardware counters on itself, so the figure comes from the latest profiles/*pmc_traffic.json (the files are named per round and build, so the
    last one by name; a checkout does not preserve modification times); null when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), key=os.path.basename)
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d["per_kernel"][kernel]["traffic_bytes_per_launch"], os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


def _by_kernel(rows):
    by = {}
    for r in rows:
        k = by.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, launches=0))
        k["ms"] += r["ms"]; k["flops"] += r["flops_per_sample"] * r["rows"]; k["launches"] += 1
    return by


def roofline(model, batch):
    rows = model.profile_ops(batch, repeats=20)
    by = _by_kernel(rows)
    name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
    total_ms = sum(v["ms"] for v in by.values())
    rows_timed = rows[0]["rows"]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(name)
    return {
        "bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (avg over the kernel's launch sites)",
        "traffic_source": traffic_src,
        "traffic_note": "HBM bytes from the committed rocprofv3 PMC passes named in traffic_source (a process cannot read its own counters); "
                        "duration / achieved are measured live in this run",
        "launches_per_forward": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
        "flops_per_launch_avg": dom["flops"] / dom["launches"], "share_of_forward_time": round(dom["ms"] / total_ms, 3),
        "forward_sum_of_kernels_ms": round(total_ms, 4),
        "rows_per_launch": rows_timed, "chains": model.chains[0],
        "forward_tflops_all_kernels": round(model.flops_per_sample * rows_timed / (total_ms * 1e-3) / 1e12, 3),
        "per_kernel": {k: dict(ms=round(v["ms"], 4), launches=v["launches"], tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 3))
                       for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])},
    }


# --------------------------------------------------------------------------------------------------- CPU baseline
def _median_time(fn, repeats, warmup=1):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline(model, mode="full"):
    """BASELINE config 1 on this box's host cores: flowers_sd shape, 16-step legacy Euler, B=4, seeded noise,
    the CPU oracle (a port of the reference's PyTorch path, pinned to reference-generated goldens) -- median of 5 after 1 warm-up at
    all threads and at the best thread count found, median of 3 at one thread; ODE only and ODE + SD-VAE decode (decode timed once
    per setting; at one thread on ONE image, scaled to 4).  `value` is the best ODE-only rate converted to the headline metric's
    64-step trajectories (x 16/64); the oracle is test infrastructure used here only as the timed baseline."""
    from oracle import flow_oracle as fo
    from oracle import sdvae_oracle as vo
    from oracle.synth import synth_state_dict
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    b, steps = 4, 16
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn((b,) + LATENT, generator=g)
    ids = torch.randint(NCLS, (b,), generator=torch.Generator().manual_seed(1235))
    all_threads = torch.get_num_threads()

    def ode():
        with torch.no_grad():
            return fo.euler_sampler(sd, x0, steps, ids)[0]

    def one_forward():
        with torch.no_grad():
            fo.unet_forward(sd, x0, torch.full((b,), 500.0), {"class_cond": ids})

    out = {"unit": "samples/s", "kind": "port", "host_cpus": os.cpu_count(), "torch_threads_default": all_threads,
           "config": "BASELINE config 1: flowers_sd shape, 16-step legacy Euler, B=4, no CFG"}
    if mode == "quick":
        t, _ = _median_time(ode, 1, warmup=1)
        out.update(value=round(b / t * steps / N_EULER, 4), cores=all_threads,
                   sample=f"oracle U-Net, B={b}, {steps}-step Euler once at {all_threads} threads ({t:.2f} s); value = rate x {steps}/{N_EULER}")
        return out
    try:
        # best thread count: tiny convolutions oversubscribe a 128-thread pool; probe a few sizes with single forwards
        probe = {}
        for n in sorted({1, 8, 16, 32, 64, all_threads}):
            if n > all_threads:
                continue
            torch.set_num_threads(n)
            one_forward()
            probe[n] = min(_median_time(one_forward, 2, warmup=0)[1])
        best_n = min((n for n in probe if n > 1), key=lambda n: probe[n], default=all_threads)
        res = {}
        for tag, n, reps in (("all_threads", all_threads, 5), ("best_threads", best_n, 5), ("one_thread", 1, 3)):
            torch.set_num_threads(n)
            med, ts = _median_time(ode, reps, warmup=1)
            res[tag] = {"threads": n, "ode_s_median": round(med, 3), "repeats": reps, "ode_samples_per_s_16step": round(b / med, 4)}
        # ODE + decode (seeded random VAE weights; parity-unpinned restatement of AutoencoderKL)
        vsd = synth_state_dict(vo.shapes(), 7)
        lat = ode() * 4.5
        for tag, n in (("all_threads", all_threads), ("best_threads", best_n), ("one_thread", 1)):
            torch.set_num_threads(n)
            nb = 1 if n == 1 else b
            with torch.no_grad():
                t0 = time.perf_counter()
                vo.decode(vsd, lat[:nb])
                td = (time.perf_counter() - t0) * (b / nb)
            r = res[tag]
            r["decode_s_b4"] = round(td, 3)
            r["decode_images_timed"] = nb
            r["ode_plus_decode_images_per_s"] = round(b / (r["ode_s_median"] + td), 4)
    finally:
        torch.set_num_threads(all_threads)
    top = max(res.values(), key=lambda r: r["ode_samples_per_s_16step"])
    out.update(res)
    out.update(value=round(top["ode_samples_per_s_16step"] * steps / N_EULER, 4), cores=top["threads"],
               thread_probe_forward_s={str(k): round(v, 4) for k, v in probe.items()},
               sample=f"oracle U-Net, B={b}, {steps}-step Euler, median of 5 after 1 warm-up at {all_threads} threads and at the best thread "
                      f"count probed ({best_n}), median of 3 at 1 thread; ODE + SD-VAE decode timed once per setting (1 image x4 at one "
                      f"thread); value = best ODE-only rate ({top['ode_samples_per_s_16step']} samples/s at {steps} steps, {top['threads']} "
                      f"threads) x {steps}/{N_EULER} = 64-step trajectories/s")
    return out


# --------------------------------------------------------------------------------------------------- secondary numbers
def _gpu_time(fn, device, reps, warmup=1):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / reps, out


def rk4_share(model, rank, world, device):
    """BASELINE configs[2] per-GPU share: "100-step" RK4 (99 intervals x 4 stages) with CFG 3.0, 32 samples per GPU (64 U-Net rows)."""
    from flocoder_amd.sampling import generate_latents_rk4
    noise, ids = synthetic_inputs(rank, world, device, per_rank=RK4_SHARE)
    t, lat = _gpu_time(lambda: generate_latents_rk4(model, (RK4_SHARE,) + LATENT, RK4_STEPS, {"class_cond": ids}, CFG, source=noise)[0], device, 1)
    assert torch.isfinite(lat).all()
    if world > 1:
        tt = torch.tensor([t], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        t = float(tt.item())
    evals = (RK4_STEPS - 1) * 4
    return {"workload": f"{RK4_STEPS}-step RK4 (warp_time grid, {evals} evaluations), CFG {CFG} as one 2B-row pass, {RK4_SHARE} samples per GPU",
            "ms": round(t * 1e3, 1), "samples_per_s": round(RK4_SHARE * world / t, 2),
            "tflops": round(RK4_SHARE * world * evals * 2 * model.flops_per_sample / t / 1e12, 2)}


def two_in_flight(model, noise, ids, device, steps=6):
    """Throughput mode for callers that generate many batches (e.g. 50 k samples for FID): two independent 64-sample trajectories
    in flight on two streams, each on its own model replica (own activation arena and captured graphs).  One batch of 64 keeps
    only ~2 workgroups per CU in lockstep phases; a second batch fills the matrix pipe while the first loads or stores.  NOT the
    headline (that is one batch at a time): reported beside it."""
    from flocoder_amd.sampling import euler_sampler
    twin = build_model(device)
    twin.load_state_dict(model.state_dict())
    shape = (BATCH,) + LATENT
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    models = [model, twin]
    cur = torch.cuda.current_stream(device)

    def run(k):
        outs = []
        for st in streams:
            st.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i & 1]):
                outs.append(euler_sampler(models[i & 1], shape, N_EULER, cond=ids, source=noise)[0])
        for st in streams:
            cur.wait_stream(st)
        return outs

    run(2)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    outs = run(steps)
    torch.cuda.synchronize(device)
    t = time.perf_counter() - t0
    assert all(torch.isfinite(o).all() for o in outs) and torch.equal(outs[0], outs[1])     # both replicas integrate the same samples
    del twin
    return {"workload": f"64-step Euler, B={BATCH} per call, TWO calls in flight (two streams, two model replicas), {steps} calls timed",
            "samples_per_s": round(BATCH * steps / t, 1), "ms_per_call_amortised": round(1e3 * t / steps, 2),
            "tflops": round(BATCH * steps * N_EULER * model.flops_per_sample / t / 1e12, 2)}


def train_step_secondary(device, dim, hw, batch, classes, steps=40, warmup=5):
    """BASELINE.json configs[3] on this rank alone: one train_flow.py step (OT pairing + prologue + U-Net forward / backward + clip +
    Adam + EMA) on synthetic latents, timed like tools/bench_train.py."""
    from flocoder_amd.ot import compute_ot_pairing
    from flocoder_amd.train import FlowTrainer
    from flocoder_amd.unet import Unet
    torch.manual_seed(0)
    net = Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=4, n_classes=classes).to(device)
    tr = FlowTrainer(net, lr=1e-4, distributed=False)
    g = torch.Generator().manual_seed(99)
    target = torch.randn(batch, 4, hw, hw, generator=g).to(device)
    cls = torch.randint(classes, (batch,), generator=g).to(device)

    def one():
        source = torch.randn_like(target)
        return tr.step(source, target, {"class_cond": cls, "mask_cond": None}, pairing=compute_ot_pairing(source, target))

    for _ in range(warmup):
        loss = one()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize(device)
    t = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss)
    out = {"workload": f"train_flow step: batch {batch}, latents 4x{hw}x{hw}, U-Net dim={dim} dim_mults [1,2,4,8] n_classes={classes}, greedy OT pairing, "
                       f"Adam lr 1e-4, EMA 0.999; {steps} steps timed on one GPU",
           "ms_per_step": round(1e3 * t, 3), "samples_per_s": round(batch / t, 1), "step_tflops_3x_fwd": round(3 * net.flops_per_sample * batch / t / 1e12, 2)}
    del tr, net
    return out


def secondary(model, noise, ids, device):
    from flocoder_amd.codecs import SD_VAE_Wrapper
    from flocoder_amd.sampling import decode_latents, euler_sampler
    out = {}
    shape = (BATCH,) + LATENT
    out["euler64_two_in_flight"] = two_in_flight(model, noise, ids, device)     # first: both replicas still hold their 64-row plans
    t_ode, lat = _gpu_time(lambda: euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0], device, 2)
    t_cfg, _ = _gpu_time(lambda: euler_sampler(model, shape, N_EULER, cond=ids, source=noise, cfg_strength=CFG)[0], device, 2)
    out["euler64_cfg"] = {"workload": f"64-step Euler with CFG {CFG} (128 U-Net rows per evaluation), B={BATCH}", "ms": round(t_cfg * 1e3, 1),
                          "samples_per_s": round(BATCH / t_cfg, 1), "tflops": round(BATCH * N_EULER * 2 * model.flops_per_sample / t_cfg / 1e12, 2)}
    vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(device)       # seeded random weights: no real checkpoint offline
    z = lat * (4.5 / float(lat.std()))                                       # unscaled SD latents have std ~4.5
    t_dec, img = _gpu_time(lambda: decode_latents(vae, z, chunk_size=DECODE_CHUNK), device, 2)
    t_enc, _ = _gpu_time(lambda: torch.cat([vae.encode(img[i:i + DECODE_CHUNK]) for i in range(0, BATCH, DECODE_CHUNK)]), device, 2)
    assert torch.isfinite(img).all()
    gf_dec, gf_enc = vae.flops_per_sample(True) / 1e9, vae.flops_per_sample(False) / 1e9
    rows = vae.profile_ops(z[:DECODE_CHUNK].contiguous(), img[:DECODE_CHUNK].contiguous(), decode=True, repeats=3)
    by = _by_kernel(rows)
    name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
    tot = sum(v["ms"] for v in by.values())
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    out["sdvae_decode"] = {
        "workload": f"SD-VAE decode 4x32x32 -> 3x256x256, B={BATCH} in chunks of {DECODE_CHUNK}, seeded random weights", "ms": round(t_dec * 1e3, 1),
        "images_per_s": round(BATCH / t_dec, 1), "gflop_per_image": round(gf_dec, 1), "tflops": round(BATCH * gf_dec / t_dec / 1e3, 1),
        "frac_of_fp32_mfma_peak": round(BATCH * gf_dec / t_dec / 1e3 / PEAK_FP32_MFMA_TFLOPS, 3),
        "roofline": {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None, "launches_per_decode": dom["launches"],
                     "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 1), "share_of_decode_time": round(dom["ms"] / tot, 3),
                     "decode_sum_of_kernels_ms": round(tot, 3), "rows_per_launch": DECODE_CHUNK}}
    out["sdvae_encode"] = {"workload": f"SD-VAE encode 3x256x256 -> 4x32x32, B={BATCH} in chunks of {DECODE_CHUNK}", "ms": round(t_enc * 1e3, 1),
                           "images_per_s": round(BATCH / t_enc, 1), "gflop_per_image": round(gf_enc, 1),
                           "tflops": round(BATCH * gf_enc / t_enc / 1e3, 1)}
    out["euler64_plus_decode"] = {"workload": "64-step Euler + SD-VAE decode (decoded images/s, SURVEY 8d secondary metric)",
                                  "images_per_s": round(BATCH / (t_ode + t_dec), 1), "ode_ms": round(t_ode * 1e3, 1), "decode_ms": round(t_dec * 1e3, 1)}
    del vae
    out["train_step_stl_sd"] = train_step_secondary(device, dim=16, hw=16, batch=32, classes=10)
    out["train_step_flowers_sized"] = train_step_secondary(device, dim=32, hw=32, batch=64, classes=102)
    return out


# --------------------------------------------------------------------------------------------------- launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` called plainly: this parent never touches the GPU; it starts the N rank processes (one per GPU)
    through torch.distributed.run on 127.0.0.1 and passes their output and exit code on.  Rank 0 prints the one JSON line."""
    have = torch.cuda.device_count()            # counts devices without initialising HIP
    if have < n and not os.environ.get("FLOCODER_AMD_SINGLE_GPU"):
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-baseline", choices=("full", "quick", "none"), default="full")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    from flocoder_amd import dist as fdist
    from flocoder_amd.sampling import euler_sampler
    rank, local_rank, world = fdist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to time (the CPU oracle is only the baseline leg)")
    # FLOCODER_AMD_SINGLE_GPU=1 (with FLOCODER_AMD_DIST_BACKEND=gloo): every rank on cuda:0, a rehearsal of the N>1 path on a 1-GPU box
    device = torch.device("cuda", 0 if os.environ.get("FLOCODER_AMD_SINGLE_GPU") else local_rank)
    torch.cuda.set_device(device)

    model = build_model(device)
    moved = fdist.broadcast_weights(model, src=0)    # the one collective: frozen weights over xGMI
    comm = None
    if world > 1:
        ones = torch.ones(1, device=device)
        torch.distributed.all_reduce(ones)           # every rank adds 1: what the collective library itself counts
        comm = {"backend": torch.distributed.get_backend(), "ranks_counted_by_allreduce": int(ones.item()),
                "world_size": torch.distributed.get_world_size(), "weights_broadcast_bytes": moved}
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
                   # plan entries of one forward; the nine attention entries are two kernels each, and inside the integrator the two
                   # conditioning entries are replaced by a table computed once per call
                   "plan_entries_per_forward": model.launches_per_forward,
                   "kernel_launches_per_forward": model.launches_per_forward + 9,
                   "kernel_launches_per_euler_step": model.launches_per_forward + 9 - 2},
    }
    line["ode_tflops"] = round(line["value"] * model.flops_per_sample * N_EULER / 1e12, 3)
    line["frac_of_fp32_mfma_peak_end_to_end"] = round(line["ode_tflops"] / (PEAK_FP32_MFMA_TFLOPS * world), 4)
    if comm:
        line["rccl_ranks"] = comm["ranks_counted_by_allreduce"]
        line["comm"] = comm
    sec = {}
    if not args.no_secondary:
        sec["rk4_100_cfg"] = rk4_share(model, rank, world, device)         # every rank takes part (max over ranks)
    if rank == 0:
        if not args.no_roofline:
            line["roofline"] = roofline(model, BATCH)
        if world == 1 and not args.no_secondary:
            sec.update(secondary(model, noise, ids, device))
        if sec:
            line["secondary"] = sec
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            line["cpu_baseline"] = cpu_baseline(model, mode)
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
