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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime reads its switches once, when it is first loaded: the package's process-level defaults (AMD_DIRECT_DISPATCH=0, see
# flocoder_amd.runtime_defaults) go into the environment before torch is imported.  FLOCODER_AMD_KEEP_ENV=1 leaves the environment alone.
if not os.environ.get("FLOCODER_AMD_KEEP_ENV"):
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

import torch  # noqa: E402

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


PEAK_HBM_GBS = 8000.0             # same guide, HBM3E


TWO_KERNEL_ENTRIES = 4 + (0 if os.environ.get("FLOCODER_AMD_LA_JOIN") == "one" else 5)   # attention plan entries that are two kernels


def _phase(name):
    """Phase marker on stderr (stdout carries the one JSON line): a fault or a hang in a long bench run is then attributable to its leg."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {name}", file=sys.stderr, flush=True)


def pmc_traffic(kernel, pattern="*pmc_traffic.json"):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    tools/pmc_forward.py or tools/pmc_codec.py, summarised by tools/pmc_summary.py with the guide's gfx950 correction).  A process
    cannot collect hardware counters on itself, so the figure comes from the latest profiles/<pattern> (named per round and build, so
    the last one by name; a checkout does not preserve modification times); null when there is none.  The summary is keyed exactly like
    this file's per-kernel table: the tile family (all its launch sites, Block-closing ones included), "<family>+fin" and
    "<family> (plain)" for the two slices of a family."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=os.path.basename)
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d["per_kernel"][kernel]["traffic_bytes_per_launch"], os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, os.path.basename(files[-1])


def _family(kernel):
    """Plan-entry kernel name -> tile family: a convolution that also closes its Block ('+fin') is the same kernel family."""
    return kernel[:-4] if kernel.endswith("+fin") else kernel


def _acc(by, key, r):
    k = by.setdefault(key, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0))
    k["ms"] += r["ms"]; k["flops"] += r["flops_per_sample"] * r["rows"]; k["launches"] += 1; k["bytes"] += r.get("bytes", 0.0)


def _by_kernel(rows, families=True):
    by = {}
    for r in rows:
        _acc(by, _family(r["kernel"]) if families else r["kernel"], r)
    return by


def _slices(rows, fam):
    """The family's launches split into the ones that close their Block and the plain ones (same keys as tools/pmc_summary.py)."""
    by = {}
    for r in rows:
        if _family(r["kernel"]) == fam:
            _acc(by, fam + ("+fin" if r["kernel"].endswith("+fin") else " (plain)"), r)
    return by


def _entry(v, traffic_key, pattern="*pmc_traffic.json"):
    ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
    traffic, src = pmc_traffic(traffic_key, pattern)
    alg = v["bytes"] / v["launches"] if v["bytes"] else None
    us = 1e3 * v["ms"] / v["launches"]
    return {"achieved": round(ach, 3), "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "launches": v["launches"], "avg_launch_us": round(us, 2),
            "flops_per_launch_avg": v["flops"] / v["launches"], "algorithmic_bytes_per_launch": round(alg) if alg else None,
            "traffic": traffic, "traffic_source": src,
            "hbm_gbs_algorithmic": round(alg / us / 1e3, 1) if alg else None}


def roofline(model, batch):
    """The dominant kernel FAMILY of the forward (largest share of device time over all its launch sites -- a convolution that closes
    its Block is the same kernel, so '+fin' launches count in) priced against the exact-fp32 MFMA peak.  `achieved`, `frac` and
    `traffic` all cover the same launch set; `slices` splits it into Block-closing and plain launches."""
    rows = model.profile_ops(batch, repeats=20)
    by = _by_kernel(rows)
    name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
    total_ms = sum(v["ms"] for v in by.values())
    rows_timed = rows[0]["rows"]
    e = _entry(dom, name)
    out = {
        "bound": "mfma", "kernel": name, "achieved": e["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": e["frac"],
        "traffic": e["traffic"], "traffic_unit": "HBM bytes/launch, averaged over the same launch sites as achieved / frac",
        "traffic_source": e["traffic_source"],
        "traffic_note": "HBM bytes from the committed rocprofv3 PMC passes named in traffic_source (a process cannot read its own counters); "
                        "duration / achieved are measured live in this run by HIP events on the launching stream",
        "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
        "launches_per_forward": dom["launches"], "avg_launch_us": e["avg_launch_us"], "flops_per_launch_avg": e["flops_per_launch_avg"],
        "share_of_forward_time": round(dom["ms"] / total_ms, 3), "forward_sum_of_kernels_ms": round(total_ms, 4),
        "rows_per_launch": rows_timed, "chains": model.chains[0],
        "forward_tflops_all_kernels": round(model.flops_per_sample * rows_timed / (total_ms * 1e-3) / 1e12, 3),
        # flops_per_sample is the reference's arithmetic; the launches execute less since nn.Upsample + conv3x3 runs as four 2x2
        # kernels -- the per-kernel figures here and `achieved` count what is executed
        "gflop_per_sample_reference": round(model.flops_per_sample / 1e9, 4),
        "gflop_per_sample_executed": round(sum(r["flops_per_sample"] for r in rows) / 1e9, 4),
        "slices": {k: _entry(v, k) for k, v in _slices(rows, name).items()},
        "per_kernel": {k: dict(ms=round(v["ms"], 4), launches=v["launches"], tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 3))
                       for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])},
    }
    return out


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


PARITY_ROWS, PARITY_GATE = 4, 2e-4


def headline_parity(model, out, noise, ids):
    """The bench verifies what it timed: rows 0..3 of the LAST timed step's output (the timed configuration, plan and runtime
    environment, nothing re-run) against the CPU oracle's 64-step Euler trajectories of the same noise / class ids (oracle/flow_oracle.py,
    pinned to the reference's goldens; trajectories are independent, so four rows are four trajectories).  ~1-2 s of host time."""
    from oracle import flow_oracle as fo
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    n = torch.get_num_threads()
    try:
        torch.set_num_threads(min(n, 16))
        with torch.no_grad():
            want = fo.euler_sampler(sd, noise[:PARITY_ROWS].cpu(), N_EULER, ids[:PARITY_ROWS].cpu())[0]
    finally:
        torch.set_num_threads(n)
    got = out[:PARITY_ROWS].cpu()
    rel = float((got.double() - want.double()).norm() / want.double().norm())
    return {"rel_l2": float(f"{rel:.3e}"), "gate": PARITY_GATE, "ok": bool(rel < PARITY_GATE), "rows": PARITY_ROWS,
            "against": f"oracle.flow_oracle.euler_sampler, {N_EULER} steps, rows 0..{PARITY_ROWS - 1} of the last timed step",
            "max_abs_diff": float((got.double() - want.double()).abs().max())}


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


def describe_diff(out, ref):
    """What separates `out` from `ref` (both [B, C, H, W]): finite?, bit-equal?, and if not how many rows differ, the first of them, the
    largest |difference| -- a failed comparison then says whether values were poisoned (NaN: a hand-off that timed out) or merely
    different (a race), and where."""
    fin = torch.isfinite(out)
    d = {"finite": bool(fin.all())}
    if not d["finite"]:
        bad = (~fin).flatten(1).any(1).nonzero().flatten().tolist()
        d.update(nonfinite_rows=len(bad), first_nonfinite_rows=bad[:8], nonfinite_values=int((~fin).sum()))
    d["equal"] = bool(torch.equal(out, ref))
    if not d["equal"]:
        diff = (out.double() - ref.double()).abs()
        diff = torch.where(torch.isfinite(diff), diff, torch.full_like(diff, float("inf")))
        rows = (out != ref).flatten(1).any(1).nonzero().flatten().tolist()
        r0 = rows[0]
        d.update(rows_differing=len(rows), first_rows_differing=rows[:8], max_abs_diff=float(diff.max()),
                 first_row_channels_differing=(out[r0] != ref[r0]).flatten(1).any(1).nonzero().flatten().tolist(),
                 first_row_values_differing=int((out[r0] != ref[r0]).sum()))
    return d


def two_in_flight(model, noise, ids, device, steps=6):
    """Throughput mode for callers that generate many batches (e.g. 50 k samples for FID): ``flocoder_amd.sampling.sample_many`` -- two
    independent 64-sample trajectories in flight on two streams, each on its own model replica (own activation arena and captured graphs),
    on the plan without cross-workgroup waits.  One batch of 64 is a chain of dependent launches that leaves the chip waiting on
    launch-to-launch latency; a second chain fills the gaps.  NOT the headline (that is one batch at a time): reported beside it.
    Self-verifying: every call integrates the same samples, so every output must be bit-equal to the one-at-a-time result of the same
    plan; what differs is reported per call (never a bare assert), with each replica's device error word, and marks the leg failed."""
    from flocoder_amd.sampling import euler_sampler, sample_many
    shape = (BATCH,) + LATENT
    excl = euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0]               # the headline's plan, for the distance between the two plans
    model.set_shared_device(True)
    ref = euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0]                # one at a time on the plan sample_many runs
    model.set_shared_device(None)
    batches = [({"class_cond": ids}, noise)] * steps
    sample_many(model, shape, batches[:2], method="euler", n_steps=N_EULER, in_flight=2)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    outs = sample_many(model, shape, batches, method="euler", n_steps=N_EULER, in_flight=2)
    torch.cuda.synchronize(device)
    t = time.perf_counter() - t0
    errs = []
    for m in [model] + list(getattr(model, "_replicas", [])):
        try:
            m.check_errors()
            errs.append("ok")
        except Exception as e:      # noqa: BLE001 -- reported, not raised: the comparison below says what the outputs look like
            errs.append(repr(e)[:200])
    calls = [dict(describe_diff(o, ref), call=i, replica=i % 2) for i, o in enumerate(outs)]
    bad = [c for c in calls if not (c["finite"] and c["equal"])]
    plans_rel = float((ref.double() - excl.double()).norm() / excl.double().norm())
    model._replicas = []
    res = {"workload": f"64-step Euler, B={BATCH} per call, TWO calls in flight (sample_many: two streams, two model replicas), {steps} calls timed",
           "samples_per_s": round(BATCH * steps / t, 1), "ms_per_call_amortised": round(1e3 * t / steps, 2),
           "tflops": round(BATCH * steps * N_EULER * model.flops_per_sample / t / 1e12, 2),
           "verified": f"{len(calls) - len(bad)}/{len(calls)} calls bit-equal to the one-at-a-time result of the same plan",
           "shared_vs_exclusive_plan_rel_l2": float(f"{plans_rel:.3e}"), "replica_error_words": errs}
    if bad or any(e != "ok" for e in errs) or not plans_rel < 1e-4:
        res["failed"] = True
        res["calls_that_differ"] = bad
    return res


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
           "ms_per_step": round(1e3 * t, 3), "samples_per_s": round(batch / t, 1), "step_tflops_3x_fwd": round(3 * net.flops_per_sample * batch / t / 1e12, 2),
           "runtime_env": "this (sampler) process: AMD_DIRECT_DISPATCH=" + os.environ.get("AMD_DIRECT_DISPATCH", "unset")}
    del tr, net
    # the training step is host-paced plain launches and wants the runtime's DEFAULT dispatch mode (flocoder_amd.runtime_defaults("training")):
    # the same step in a process of its own without the sampler's setting (tools/bench_train.py) is the figure a training job sees
    try:
        env = {k: v for k, v in os.environ.items() if k != "AMD_DIRECT_DISPATCH"}
        env["FLOCODER_AMD_KEEP_ENV"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py"), "--dim", str(dim), "--hw", str(hw), "--batch", str(batch),
                            "--classes", str(classes), "--steps", str(steps), "--warmup", str(warmup)], env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out["default_dispatch_process"] = {"ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "step_tflops_3x_fwd": d.get("step_tflops_3x_fwd")}
    except Exception as e:       # the leg is informative: never fail the bench line over it
        out["default_dispatch_process"] = {"error": repr(e)[:200]}
    return out


def secondary(model, noise, ids, device):
    """The other BASELINE numbers of this box, one leg at a time.  Every leg runs in its own try / except: a failure is recorded under the
    leg's own key ({"error": ...}) and counted in `secondary_failed`, and the remaining legs still run.  The experimental leg (two
    trajectories in flight) runs last."""
    import traceback
    from flocoder_amd.codecs import SD_VAE_Wrapper
    from flocoder_amd.sampling import decode_latents, euler_sampler
    out, failed, st = {}, [], {}
    shape = (BATCH,) + LATENT

    def leg(name, fn):
        _phase("secondary: " + name)
        try:
            res = fn()
            out.update(res)
            failed.extend(k for k, v in res.items() if isinstance(v, dict) and v.get("failed"))
        except Exception as e:      # noqa: BLE001 -- recorded under the leg's key; the other legs still run
            out[name] = {"error": repr(e)[:400], "where": traceback.format_exc().strip().splitlines()[-3:]}
            failed.append(name)
            try:
                torch.cuda.synchronize(device)
            except Exception:       # noqa: BLE001
                pass

    def euler_legs():
        st["t_ode"], st["lat"] = _gpu_time(lambda: euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0], device, 2)
        t_cfg, _ = _gpu_time(lambda: euler_sampler(model, shape, N_EULER, cond=ids, source=noise, cfg_strength=CFG)[0], device, 2)
        return {"euler64_cfg": {"workload": f"64-step Euler with CFG {CFG} (128 U-Net rows per evaluation), B={BATCH}", "ms": round(t_cfg * 1e3, 1),
                                "samples_per_s": round(BATCH / t_cfg, 1), "tflops": round(BATCH * N_EULER * 2 * model.flops_per_sample / t_cfg / 1e12, 2)}}

    def vae_inputs():
        if "lat" not in st:
            st["t_ode"], st["lat"] = _gpu_time(lambda: euler_sampler(model, shape, N_EULER, cond=ids, source=noise)[0], device, 1)
        if "vae" not in st:
            st["vae"] = SD_VAE_Wrapper(weights="random", seed=0).eval().to(device)       # seeded random weights: no real checkpoint offline
            st["z"] = st["lat"] * (4.5 / float(st["lat"].std()))                            # unscaled SD latents have std ~4.5
        return st["vae"], st["z"]

    def sdvae_fp32():
        vae, z = vae_inputs()
        t_dec, img = _gpu_time(lambda: decode_latents(vae, z, chunk_size=DECODE_CHUNK), device, 2)
        t_enc, _ = _gpu_time(lambda: torch.cat([vae.encode(img[i:i + DECODE_CHUNK]) for i in range(0, BATCH, DECODE_CHUNK)]), device, 2)
        assert torch.isfinite(img).all()
        st["img"], st["t_dec"] = img, t_dec
        gf_dec, gf_enc = vae.flops_per_sample(True) / 1e9, vae.flops_per_sample(False) / 1e9
        # (a scratch output: the per-launch timing repeats every launch in place, over pooled buffers -- what it leaves in `out` is not a decode)
        rows = vae.profile_ops(z[:DECODE_CHUNK].contiguous(), torch.empty_like(img[:DECODE_CHUNK]), decode=True, repeats=3)
        by = _by_kernel(rows)
        name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
        tot = sum(v["ms"] for v in by.values())
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        dec_traffic, dec_src = pmc_traffic(name, "*pmc_traffic_sdvae_decode.json")
        res = {"sdvae_decode": {
            "workload": f"SD-VAE decode 4x32x32 -> 3x256x256, B={BATCH} in chunks of {DECODE_CHUNK}, seeded random weights", "ms": round(t_dec * 1e3, 1),
            "images_per_s": round(BATCH / t_dec, 1), "gflop_per_image": round(gf_dec, 1), "tflops": round(BATCH * gf_dec / t_dec / 1e3, 1),
            "frac_of_fp32_mfma_peak": round(BATCH * gf_dec / t_dec / 1e3 / PEAK_FP32_MFMA_TFLOPS, 3),
            "frac_note": "whole decode in the REFERENCE's FLOPs (the folded upsampling executes fewer); roofline.frac below is the dominant kernel in EXECUTED FLOPs",
            "roofline": {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": dec_traffic, "traffic_source": dec_src,
                         "traffic_unit": "HBM bytes/launch (rocprofv3 PMC passes over tools/pmc_codec.py, same launch set)",
                         "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]) if dom["bytes"] else None,
                         "launches_per_decode": dom["launches"],
                         "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 1), "share_of_decode_time": round(dom["ms"] / tot, 3),
                         "decode_sum_of_kernels_ms": round(tot, 3), "rows_per_launch": DECODE_CHUNK}},
            "sdvae_encode": {"workload": f"SD-VAE encode 3x256x256 -> 4x32x32, B={BATCH} in chunks of {DECODE_CHUNK}", "ms": round(t_enc * 1e3, 1),
                             "images_per_s": round(BATCH / t_enc, 1), "gflop_per_image": round(gf_enc, 1),
                             "tflops": round(BATCH * gf_enc / t_enc / 1e3, 1)},
            "euler64_plus_decode": {"workload": "64-step Euler + SD-VAE decode (decoded images/s, SURVEY 8d secondary metric)",
                                    "images_per_s": round(BATCH / (st["t_ode"] + t_dec), 1), "ode_ms": round(st["t_ode"] * 1e3, 1),
                                    "decode_ms": round(t_dec * 1e3, 1)}}
        return res

    def sdvae_bf16():
        # opt-in split-bf16 arithmetic of the codec (fc_vae_set_precision): reported BESIDE the fp32 numbers with its measured error, never instead
        vae, z = vae_inputs()
        if "img" not in st:
            st["t_dec"], st["img"] = _gpu_time(lambda: decode_latents(vae, z, chunk_size=DECODE_CHUNK), device, 1)
        img, t_dec = st["img"], st["t_dec"]
        vae.set_precision("bf16x3")
        try:
            t_dec3, img3 = _gpu_time(lambda: decode_latents(vae, z, chunk_size=DECODE_CHUNK), device, 2)
            t_enc3, _ = _gpu_time(lambda: torch.cat([vae.encode(img[i:i + DECODE_CHUNK]) for i in range(0, BATCH, DECODE_CHUNK)]), device, 2)
        finally:
            vae.set_precision("fp32")
        err3 = float((img3.double() - img.double()).norm() / img.double().norm())
        res = {"workload": f"SD-VAE decode 4x32x32 -> 3x256x256, B={BATCH} in chunks of {DECODE_CHUNK}, seeded random weights; OPT-IN arithmetic: operands as "
                           "bf16 hi + lo, hi*hi + hi*lo + lo*hi on the bf16 matrix pipe, fp32 accumulation",
               "ms": round(t_dec3 * 1e3, 1), "images_per_s": round(BATCH / t_dec3, 1), "speedup_vs_fp32": round(t_dec / t_dec3, 2),
               "rel_l2_vs_fp32_decode": float(f"{err3:.3e}"), "gate": 1e-3, "encode_images_per_s": round(BATCH / t_enc3, 1),
               "euler64_plus_decode_images_per_s": round(BATCH / (st["t_ode"] + t_dec3), 1)}
        if not err3 < 1e-3:
            res["failed"] = True
        return {"sdvae_decode_split_bf16": res}

    def drop_vae():
        st.pop("vae", None); st.pop("img", None); st.pop("z", None)
        return {}

    leg("euler64_cfg", euler_legs)
    leg("sdvae_decode", sdvae_fp32)
    leg("sdvae_decode_split_bf16", sdvae_bf16)
    drop_vae()
    leg("train_step_stl_sd", lambda: {"train_step_stl_sd": train_step_secondary(device, dim=16, hw=16, batch=32, classes=10)})
    leg("train_step_flowers_sized", lambda: {"train_step_flowers_sized": train_step_secondary(device, dim=32, hw=32, batch=64, classes=102)})
    leg("config5_midi", lambda: {"config5_midi": config5(device)})
    leg("euler64_two_in_flight", lambda: {"euler64_two_in_flight": two_in_flight(model, noise, ids, device)})     # experimental: last
    out["secondary_failed"] = len(failed)
    if failed:
        out["secondary_failed_legs"] = failed
    return out


def config5(device):
    """BASELINE.json configs[4]: the VQGAN codec's encode / decode at the midi_vqgan.yaml shape (in=3, hidden 256,
    3 downsamples, internal 128, 4x16x16 latents), B=64 at 128x128, and the mask-conditioned RK4 inpainting sampler at the
    midi_inpainting.yaml flow shape (dim 8, latents 4x8x8, masks through the MaskEncoder).  Seeded default-initialised weights."""
    from flocoder_amd.codecs import VQVAE
    from flocoder_amd.inpainting import MaskEncoder, mask_blending
    from flocoder_amd.sampling import generate_latents_rk4
    from flocoder_amd.unet import Unet
    out = {}
    torch.manual_seed(5)
    vq = VQVAE(in_channels=3, hidden_channels=256, num_downsamples=3, internal_dim=128, vq_embedding_dim=4, codebook_levels=4,
               vq_num_embeddings=96).eval().to(device)
    g = torch.Generator().manual_seed(55)
    x = torch.rand(BATCH, 3, 128, 128, generator=g).to(device)
    t_enc, z = _gpu_time(lambda: vq.encode(x), device, 3)
    t_dec, y = _gpu_time(lambda: vq.decode(z), device, 3)
    assert torch.isfinite(z).all() and torch.isfinite(y).all() and tuple(z.shape) == (BATCH, 4, 16, 16) and y.shape == x.shape
    for tag, t, dec, inp, res in (("vqvae_encode", t_enc, False, x, z), ("vqvae_decode", t_dec, True, z, y)):
        gf = vq.flops_per_sample(dec) / 1e9
        rows = vq.profile_ops(inp, torch.empty_like(res), decode=dec, repeats=3)
        by = _by_kernel(rows)
        name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
        tot = sum(v["ms"] for v in by.values())
        e = _entry(dom, name, "*pmc_traffic_" + tag + ".json")
        out[tag] = {"workload": f"VQVAE {'decode 4x16x16 -> 3x128x128' if dec else 'encode 3x128x128 -> 4x16x16'} (midi_vqgan.yaml shape), B={BATCH} in one batch",
                    "ms": round(t * 1e3, 2), "images_per_s": round(BATCH / t, 1), "gflop_per_image": round(gf, 1),
                    "tflops": round(BATCH * gf / t / 1e3, 1), "frac_of_fp32_mfma_peak": round(BATCH * gf / t / 1e3 / PEAK_FP32_MFMA_TFLOPS, 3),
                    "launches": len(rows), "sum_of_kernels_ms": round(tot, 3),
                    "roofline": dict(bound="mfma", kernel=name, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", share_of_time=round(dom["ms"] / tot, 3), **e),
                    "per_kernel": {k: dict(ms=round(v["ms"], 4), launches=v["launches"], tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2))
                                   for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])}}
    del vq
    # inpainting sampler: flow dim 8 on 4x8x8 latents (midi_inpainting.yaml: 128x128 gray rolls, 4 downsamples), mask-conditioned
    torch.manual_seed(6)
    unet = Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=4, n_classes=0, mask_cond=True).eval().to(device)
    me = MaskEncoder().eval().to(device)
    pix = torch.zeros(BATCH, 1, 128, 128)
    for b in range(BATCH):                                  # rectangular holes of varying size / position (inpainting.py's simplest generator)
        h0, w0 = 8 + (b * 7) % 64, 8 + (b * 13) % 64
        pix[b, :, h0:h0 + 24 + b % 32, w0:w0 + 24 + (3 * b) % 32] = 1.0
    pix = pix.to(device)
    src_lat = torch.randn(BATCH, 4, 8, 8, generator=g).to(device)
    noise = torch.randn(BATCH, 4, 8, 8, generator=g).to(device)
    n_steps = RK4_STEPS

    def run():
        with torch.no_grad():
            mask = me(pix)
            source = mask_blending(src_lat, mask, noise)
            return generate_latents_rk4(unet, (BATCH, 4, 8, 8), n_steps, {"mask_cond": mask}, 0.0, source=source)[0]
    t_inp, lat = _gpu_time(run, device, 2)
    assert torch.isfinite(lat).all()
    evals = (n_steps - 1) * 4
    out["inpaint_rk4"] = {"workload": f"mask-conditioned inpainting sampler: MaskEncoder + blend + {n_steps}-step RK4 ({evals} evaluations), U-Net dim 8 "
                                      f"mask_cond, latents 4x8x8, B={BATCH}", "ms": round(t_inp * 1e3, 1), "samples_per_s": round(BATCH / t_inp, 1),
                          "us_per_evaluation": round(1e6 * t_inp / evals, 1), "gflop_per_sample_per_evaluation": round(unet.flops_per_sample / 1e9, 4),
                          "tflops": round(BATCH * evals * unet.flops_per_sample / t_inp / 1e12, 3), "plan_entries_per_forward": unet.launches_per_forward,
                          "note": f"{unet.flops_per_sample / 1e6:.1f} MFLOP per sample per evaluation over {unet.launches_per_forward} plan entries: "
                                  "launch-latency bound by construction, not an MFMA or HBM roofline case"}
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
    ap.add_argument("--secondary-only", action="store_true", help="(internal) run the one-GPU secondary legs and print their JSON: the main run "
                    "starts this in a child process, so that nothing in them can cost the headline line")
    args = ap.parse_args()

    if args.secondary_only:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        model = build_model(device)
        noise, ids = synthetic_inputs(0, 1, device)
        print(json.dumps(secondary(model, noise, ids, device)), flush=True)
        return

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
    if os.environ.get("FLOCODER_AMD_SINGLE_GPU") and world > 1:
        model.set_shared_device(True)                # the rehearsal puts every rank's process on one GPU: not an exclusive device
    moved = fdist.broadcast_weights(model, src=0)    # the one collective: frozen weights over xGMI
    comm = None
    if world > 1:
        ones = torch.ones(1, device=device)
        torch.distributed.all_reduce(ones)           # every rank adds 1: what the collective library itself counts
        comm = {"backend": torch.distributed.get_backend(), "ranks_counted_by_allreduce": int(ones.item()),
                "world_size": torch.distributed.get_world_size(), "weights_broadcast_bytes": moved}
    noise, ids = synthetic_inputs(rank, world, device)
    shape = (BATCH,) + LATENT

    # Every step integrates a DIFFERENT batch (four seeded variants of the rank's shard, resident in HBM before the timed region, taken in
    # turn): identical work per step, but a step that ran on anything left over from the step before it -- conditioning, class ids, state --
    # no longer produces the right answer by accident, and the parity check below looks at the last step.  (Round 4: with the same
    # samples in every step the bench could not see a graph replay overtaking its own prologue, see fc_unet_integrate.)
    variants = [(noise.roll(k, 0).contiguous(), ids.roll(k, 0).contiguous()) for k in range(4)]
    calls = {"n": 0}

    def step():
        nz, cl = variants[calls["n"] % len(variants)]
        calls["n"] += 1
        return euler_sampler(model, shape, N_EULER, cond=cl, source=nz)[0]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    _phase("warm-up")
    for _ in range(args.warmup):
        out = step()
    barrier()
    _phase("timed steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)                  # each rank's own clock around the same K steps
        per_rank = [round(BATCH * args.steps / float(e.item()), 1) for e in every]
        t = mine.clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    model.check_errors()                             # (every euler_sampler call above checked already: Unet.integrate(check=True))
    last_noise, last_ids = variants[(calls["n"] - 1) % len(variants)]
    parity = headline_parity(model, out, last_noise, last_ids) if rank == 0 else None     # the timed configuration's own output against the CPU oracle
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
                   # plan entries of one forward; the nine attention entries are two kernels each (the five low-resolution ones
                   # are one under FLOCODER_AMD_LA_JOIN=one), and inside the integrator the two conditioning
                   # entries are replaced by a table computed once per call
                   "plan_entries_per_forward": model.launches_per_forward,
                   "kernel_launches_per_forward": model.launches_per_forward + TWO_KERNEL_ENTRIES,
                   "kernel_launches_per_euler_step": model.launches_per_forward + TWO_KERNEL_ENTRIES - 2,
                   "plan": "exclusive device (cross-workgroup Block tails)" if model.meeting_launches else "shared device (no cross-workgroup waits)",
                   "runtime_env": {k: os.environ.get(k) for k in ("AMD_DIRECT_DISPATCH", "FLOCODER_AMD_KEEP_ENV") if os.environ.get(k) is not None}},
    }
    if parity:
        line["parity_rel_l2"] = parity["rel_l2"]
        line["parity"] = parity
    line["ode_tflops"] = round(line["value"] * model.flops_per_sample * N_EULER / 1e12, 3)
    line["frac_of_fp32_mfma_peak_end_to_end"] = round(line["ode_tflops"] / (PEAK_FP32_MFMA_TFLOPS * world), 4)
    if comm:
        # the count is what the collective library itself summed; it is RCCL's only when the backend is "nccl" (= RCCL on ROCm)
        line["rccl_ranks" if comm["backend"] == "nccl" else comm["backend"] + "_ranks"] = comm["ranks_counted_by_allreduce"]
        comm["per_rank_samples_per_s"] = per_rank          # value = global batch x steps / the SLOWEST rank's time; these are each rank's own
        comm["per_step_collectives"] = 0                   # the sampling path has none: one weight broadcast before the loop
        line["comm"] = comm
    sec = {}
    _phase("rk4 share")
    if not args.no_secondary:
        sec["rk4_100_cfg"] = rk4_share(model, rank, world, device)         # every rank takes part (max over ranks)
    if rank == 0:
        if not args.no_roofline:
            _phase("roofline (per-launch timing)")
            line["roofline"] = roofline(model, BATCH)
        if world == 1 and not args.no_secondary:
            # the secondary legs (codecs, two trajectories in flight, training steps, config 5) run in a process of their own: they are
            # informative, a failure in one of them must not cost the headline line above
            _phase("secondary legs (child process)")
            r = None
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary-only"], capture_output=True, text=True, timeout=900)
                sys.stderr.write(r.stderr[-4000:])
                lines = r.stdout.strip().splitlines()
                if lines:
                    sec.update(json.loads(lines[-1]))
                if r.returncode != 0 or not lines:      # the child died (a fault, a kill): say so, with what it last wrote
                    sec["secondary_error"] = f"secondary child process exited with code {r.returncode}; stderr tail: {r.stderr[-600:]!r}"
            except Exception as e:      # noqa: BLE001
                sec["secondary_error"] = repr(e)[:300] + (f"; child stderr tail: {r.stderr[-400:]!r}" if r is not None else "")
            if "secondary_error" in sec:
                sec["secondary_failed"] = sec.get("secondary_failed", 0) + 1
        if sec:
            line["secondary"] = sec
            line["secondary_failed"] = int(sec.get("secondary_failed", 0))
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            _phase("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(model, mode)
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if parity and not parity["ok"]:
        raise SystemExit(f"bench.py: the timed configuration's output is {parity['rel_l2']:.3e} rel-L2 away from the CPU oracle (gate {parity['gate']}): the line above is INVALID")


if __name__ == "__main__":
    main()
