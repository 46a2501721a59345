#!/usr/bin/env python3
"""Localise a result that differs when two model replicas run side by side: two host threads, each driving its own replica on its own stream
with plain forwards at the bench shape (B=64, dim 32, shared plan), compare every output with the solo result on the device and, at the
first mismatch, copy every named activation of that replica's arena and hold it against the solo run's -- the first tap (in plan order)
that differs names the launch, and the rows / pixels / channels that differ name the tile.

    python tools/race_hunt.py [--iters 4000] [--batch 64] [--threads 2]

One JSON line per event on stdout."""
import argparse
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("FLOCODER_AMD_KEEP_ENV"):
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

import torch  # noqa: E402


def tap_names(levels=4):
    names = ["init"]
    for i in range(levels):
        p = f"downs.{i}"
        names += [f"{p}.0.h1", f"{p}.0.h2", f"{p}.0", f"{p}.1.h1", f"{p}.1.h2", f"{p}.1", f"{p}.2.y", f"{p}.2", f"{p}.3"]
    names += ["mid_block1.h1", "mid_block1.h2", "mid_block1", "mid_attn", "mid_block2.h1", "mid_block2.h2", "mid_block2"]
    for i in range(levels):
        p = f"ups.{i}"
        names += [f"{p}.0.h1", f"{p}.0.h2", f"{p}.0", f"{p}.1.h1", f"{p}.1.h2", f"{p}.1", f"{p}.2.y", f"{p}.2", f"{p}.3"]
    names += ["final_res_block.h1", "final_res_block.h2", "final_res_block"]
    return names


def grab(model, batch, stream):
    from flocoder_amd import _binding as B
    out = {}
    for n in tap_names():
        try:
            p, c, h, w = model.debug_tensor(n)
        except Exception:       # noqa: BLE001 -- not every tap exists in every plan
            continue
        t = torch.empty((batch, h, w, c), dtype=torch.float32, device="cuda:0")
        B.check(B.lib().fc_debug_copy(t.data_ptr(), p, t.numel() * 4, stream.cuda_stream))
        out[n] = t
    stream.synchronize()
    return out


def where(a, b):
    ne = a != b
    rows = ne.flatten(1).any(1).nonzero().flatten().tolist()
    r0 = rows[0]
    pix = ne[r0].any(-1).nonzero().tolist()
    ch = ne[r0].any(0).any(0).nonzero().flatten().tolist()
    d = (a.double() - b.double()).abs()
    d = torch.where(torch.isfinite(d), d, torch.full_like(d, float("inf")))
    return {"rows": rows[:16], "n_rows": len(rows), "first_row_pixels_yx": pix[:24], "n_pixels_first_row": len(pix),
            "first_row_channels": ch[:40], "n_channels_first_row": len(ch), "values": int(ne.sum()), "max_abs_diff": float(d.max()),
            "nonfinite": int((~torch.isfinite(a)).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--max-events", type=int, default=3)
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    base = bench.build_model(dev)
    noise, ids = bench.synthetic_inputs(0, 1, dev, per_rank=args.batch)
    tvec = torch.linspace(5.0, 990.0, args.batch, device=dev)
    models = [base] + [base.replica() for _ in range(args.threads - 1)]
    streams = [torch.cuda.Stream(dev) for _ in models]
    refs, taps = [], []
    for m, st in zip(models, streams):          # solo results, one replica at a time (nothing else on the GPU)
        m.set_shared_device(True)
        with torch.cuda.stream(st), torch.no_grad():
            o1 = m(noise, tvec, {"class_cond": ids})
            o2 = m(noise, tvec, {"class_cond": ids})
            st.synchronize()
            assert torch.equal(o1, o2)
            refs.append(o1)
            taps.append(grab(m, args.batch, st))
    print(json.dumps({"what": "solo", "replicas_equal": all(torch.equal(r, refs[0]) for r in refs), "taps": len(taps[0]),
                      "launches": models[0].launches_per_forward, "meeting_launches": models[0].meeting_launches}), flush=True)
    events, lock = [], threading.Lock()
    stop = threading.Event()

    def drive(k):
        m, st = models[k], streams[k]
        with torch.cuda.stream(st), torch.no_grad():
            for it in range(args.iters):
                if stop.is_set():
                    return
                o = m(noise, tvec, {"class_cond": ids})
                if bool((o != refs[k]).any().item()):
                    got = grab(m, args.batch, st)
                    ev = {"what": "mismatch", "replica": k, "iteration": it, "output": where(o, refs[k]), "taps": []}
                    for n in tap_names():
                        if n in got and not torch.equal(got[n], taps[k][n]):
                            ev["taps"].append({"tap": n, "shape": list(got[n].shape), **where(got[n], taps[k][n])})
                    with lock:
                        events.append(ev)
                        print(json.dumps(ev), flush=True)
                        if len(events) >= args.max_events:
                            stop.set()

    th = [threading.Thread(target=drive, args=(k,)) for k in range(len(models))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize(dev)
    print(json.dumps({"what": "done", "events": len(events), "iterations_per_replica": args.iters}), flush=True)


if __name__ == "__main__":
    main()
