#!/usr/bin/env python3
"""Diagnosis of `sampling.sample_many(in_flight=2)` at the size and in the runtime environment that failed in round 3 (B=64, 64 Euler steps,
six calls, AMD_DIRECT_DISPATCH=0): every call of every round is compared with the one-at-a-time result of the same plan and reported on its
own -- finite?, max |delta|, how many rows differ and which is the first -- together with each replica's device error word and, under
FLOCODER_AMD_POISON=1, the fence check of every library buffer.  Four rows are also held against the CPU oracle.  One JSON line per
round on stdout; exit code 1 if anything differed.

    python tools/inflight_diag.py [--rounds 3] [--calls 6] [--in-flight 2] [--batch 64] [--steps 64] [--no-oracle]
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


def describe(out, ref):
    """What separates `out` from `ref` (both [B, C, H, W])."""
    fin = torch.isfinite(out)
    d = {"finite": bool(fin.all())}
    if not d["finite"]:
        bad_rows = (~fin).flatten(1).any(1).nonzero().flatten().tolist()
        d["nonfinite_rows"] = len(bad_rows)
        d["first_nonfinite_rows"] = bad_rows[:8]
        d["nonfinite_values"] = int((~fin).sum())
    if torch.equal(out, ref):
        d["equal"] = True
        return d
    d["equal"] = False
    diff = (out.double() - ref.double()).abs()
    diff = torch.where(torch.isfinite(diff), diff, torch.full_like(diff, float("inf")))
    rows = (out != ref).flatten(1).any(1).nonzero().flatten().tolist()
    d["rows_differing"] = len(rows)
    d["first_rows_differing"] = rows[:8]
    d["max_abs_diff"] = float(diff.max())
    r0 = rows[0]
    ch = (out[r0] != ref[r0]).flatten(1).any(1).nonzero().flatten().tolist()
    d["first_row_channels_differing"] = ch
    d["first_row_values_differing"] = int((out[r0] != ref[r0]).sum())
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--calls", type=int, default=6)
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()

    import bench
    from flocoder_amd import _binding as B
    from flocoder_amd.sampling import euler_sampler, sample_many
    import ctypes as C

    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    noise, ids = bench.synthetic_inputs(0, 1, dev, per_rank=args.batch)
    shape = (args.batch,) + bench.LATENT
    env = {k: os.environ.get(k) for k in ("AMD_DIRECT_DISPATCH", "FLOCODER_AMD_POISON", "FLOCODER_AMD_NO_GRAPH", "FLOCODER_AMD_UPS_FOLD",
                                           "FLOCODER_AMD_NO_PRECOND", "FLOCODER_AMD_NO_W4", "FLOCODER_AMD_LEAN_KERNELS")}
    failed = False

    def poison():
        bad, live = C.c_int(0), C.c_int(0)
        B.check(B.lib().fc_debug_poison_check(C.byref(bad), C.byref(live)))
        return {"buffers": live.value, "written_out_of_bounds": bad.value,
                **({"report": B.lib().fc_last_error().decode(errors="replace")} if bad.value else {})}

    # one at a time, exclusive plan (the headline configuration), then the same model on the plan without cross-workgroup waits
    excl = euler_sampler(model, shape, args.steps, cond=ids, source=noise)[0]
    excl2 = euler_sampler(model, shape, args.steps, cond=ids, source=noise)[0]
    model.set_shared_device(True)
    shared = [euler_sampler(model, shape, args.steps, cond=ids, source=noise)[0] for _ in range(3)]
    model.set_shared_device(None)
    torch.cuda.synchronize(dev)
    head = {"what": "one at a time", "env": env, "exclusive_repeat": describe(excl2, excl),
            "shared_vs_exclusive_rel_l2": float((shared[0].double() - excl.double()).norm() / excl.double().norm()),
            "shared_repeats": [describe(s, shared[0]) for s in shared[1:]], "poison": poison()}
    if not args.no_oracle:
        from oracle import flow_oracle as fo
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        want = fo.euler_sampler(sd, noise[:4].cpu(), args.steps, ids[:4].cpu())[0]
        head["oracle_rel_l2_rows0_3"] = {"exclusive": float((excl[:4].cpu().double() - want.double()).norm() / want.double().norm()),
                                         "shared": float((shared[0][:4].cpu().double() - want.double()).norm() / want.double().norm())}
    failed |= not head["exclusive_repeat"]["equal"] or any(not d["equal"] for d in head["shared_repeats"])
    print(json.dumps(head), flush=True)

    batches = [({"class_cond": ids}, noise)] * args.calls
    for rnd in range(args.rounds):
        outs = sample_many(model, shape, batches, method="euler", n_steps=args.steps, in_flight=args.in_flight)
        torch.cuda.synchronize(dev)
        rec = {"what": f"sample_many(in_flight={args.in_flight})", "round": rnd, "calls": []}
        for i, o in enumerate(outs):
            d = describe(o, shared[0])
            d["call"], d["replica"] = i, i % args.in_flight
            rec["calls"].append(d)
            failed |= not (d["finite"] and d["equal"])
        errs = []
        for m in [model] + list(getattr(model, "_replicas", [])):
            try:
                m.check_errors()
                errs.append("ok")
            except Exception as e:       # noqa: BLE001 -- the report is the point
                errs.append(repr(e)[:200])
        rec["replica_error_words"] = errs
        rec["poison"] = poison()
        failed |= rec["poison"]["written_out_of_bounds"] > 0
        print(json.dumps(rec), flush=True)
    print(json.dumps({"what": "verdict", "failed": bool(failed)}), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
