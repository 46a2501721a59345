#!/usr/bin/env python3
"""How much do independent trajectories on separate streams overlap?  (round 3, under AMD_DIRECT_DISPATCH=0)

The 64 samples of the bench batch are independent, so one call could run as N groups of 64 / N samples on N streams (N model replicas:
own plan, arena and captured graphs).  This measures samples/s for per-stream batches 64, 32, 16 with 1-4 streams in flight; the line
"1 x 64" is the headline configuration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.environ.get("FLOCODER_AMD_KEEP_ENV"):
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
import torch  # noqa: E402

import bench  # noqa: E402
from flocoder_amd.sampling import euler_sampler  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    base = bench.build_model(dev)
    sd = base.state_dict()
    noise, ids = bench.synthetic_inputs(0, 1, dev)
    print(f"AMD_DIRECT_DISPATCH={os.environ.get('AMD_DIRECT_DISPATCH')}")
    cur = torch.cuda.current_stream(dev)
    for per in (64, 32, 16):
        for nstreams in (1, 2, 3, 4):
            if per * nstreams > 256:
                continue
            models = []
            for _ in range(nstreams):
                m = bench.build_model(dev)
                m.load_state_dict(sd)
                m.set_shared_device(nstreams > 1)
                models.append(m)
            streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
            x, c = noise[:per].contiguous(), ids[:per].contiguous()

            def run(calls):
                for st in streams:
                    st.wait_stream(cur)
                outs = []
                for i in range(calls):
                    with torch.cuda.stream(streams[i % nstreams]):
                        outs.append(euler_sampler(models[i % nstreams], (per, 4, 32, 32), bench.N_EULER, cond=c, source=x)[0])
                for st in streams:
                    cur.wait_stream(st)
                return outs
            run(nstreams)
            torch.cuda.synchronize()
            calls = 4 * nstreams
            t0 = time.perf_counter()
            outs = run(calls)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            assert all(torch.isfinite(o).all() for o in outs)
            print(f"{nstreams} x {per:3d} in flight: {per * calls / t:8.1f} samples/s   ({1e3 * t / calls * nstreams:7.2f} ms per round of {nstreams * per} samples)", flush=True)
            del models


if __name__ == "__main__":
    main()
