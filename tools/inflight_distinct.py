#!/usr/bin/env python3
"""`sampling.sample_many` with DIFFERENT inputs per batch (the bench's two-in-flight leg integrates the same samples in every call, which hides
a mix-up between calls): every in-flight output is held against the one-at-a-time results of ALL batches, so a call that picked up another
call's noise / class ids / conditioning shows as "closest to batch j", and against the CPU oracle for its own batch.

    python tools/inflight_distinct.py [--batch 8] [--steps 3] [--calls 5] [--in-flight 2]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("FLOCODER_AMD_KEEP_ENV"):
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

import torch  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--method", choices=("euler", "rk4"), default="euler", help="rk4: generate_latents_rk4 with CFG 3.0 on `steps` grid points (more than 22 points = several graph replays per call)")
    args = ap.parse_args()
    import bench
    from flocoder_amd.sampling import euler_sampler, generate_latents_rk4, sample_many
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    g = torch.Generator().manual_seed(5)
    B = args.batch
    srcs = [torch.randn(B, 4, 32, 32, generator=g).to(dev) for _ in range(args.calls)]
    cls = [torch.randint(102, (B,), generator=g).to(dev) for _ in range(args.calls)]
    shape = (B, 4, 32, 32)
    rk4 = args.method == "rk4"

    def one(c, s):
        if rk4:
            return generate_latents_rk4(model, shape, args.steps, {"class_cond": c}, 3.0, source=s)[0]
        return euler_sampler(model, shape, args.steps, cond=c, source=s)[0]
    excl = [one(c, s) for c, s in zip(cls, srcs)]
    model.set_shared_device(True)
    shared = [one(c, s) for c, s in zip(cls, srcs)]
    model.set_shared_device(None)
    torch.cuda.synchronize()
    head = {"what": "one at a time", "method": args.method, "env": {k: os.environ.get(k) for k in ("AMD_DIRECT_DISPATCH", "FLOCODER_AMD_NO_GRAPH", "FLOCODER_AMD_NO_PRECOND")},
            "shared_vs_exclusive": [rel(a, b) for a, b in zip(shared, excl)]}
    if args.oracle:
        from oracle import flow_oracle as fo
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        if rk4:
            head["exclusive_vs_oracle"] = [rel(e.cpu(), fo.generate_latents_rk4(sd, s.cpu().clone(), args.steps, {"class_cond": c.cpu()}, 3.0)[0]) for e, s, c in zip(excl, srcs, cls)]
        else:
            head["exclusive_vs_oracle"] = [rel(e.cpu(), fo.euler_sampler(sd, s.cpu(), args.steps, c.cpu())[0]) for e, s, c in zip(excl, srcs, cls)]
    print(json.dumps(head), flush=True)
    bad = False
    for rnd in range(2):
        outs = sample_many(model, shape, [({"class_cond": c}, s) for c, s in zip(cls, srcs)], method=args.method, n_steps=args.steps, cfg_strength=3.0 if rk4 else 0.0,
                           in_flight=args.in_flight)
        torch.cuda.synchronize()
        rec = {"what": f"sample_many(in_flight={args.in_flight})", "round": rnd, "calls": []}
        for i, o in enumerate(outs):
            d = [rel(o, s) for s in shared]
            j = min(range(len(d)), key=lambda k: d[k])
            rec["calls"].append({"call": i, "replica": i % args.in_flight, "rel_to_own": d[i], "closest_batch": j, "rel_to_closest": d[j],
                                 "finite": bool(torch.isfinite(o).all())})
            bad |= not d[i] < 1e-6
        after = [one(c, s) for c, s in zip(cls, srcs)]      # the test's own comparison target
        rec["exclusive_after_vs_before"] = [rel(a, b) for a, b in zip(after, excl)]
        bad |= any(not r < 1e-6 for r in rec["exclusive_after_vs_before"])
        print(json.dumps(rec), flush=True)
    print(json.dumps({"what": "verdict", "failed": bad}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
