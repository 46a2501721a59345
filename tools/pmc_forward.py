#!/usr/bin/env python3
"""Smallest run that exercises every kernel of the headline workload once per launch site: a few plain (un-graphed) U-Net forwards
at the bench configuration (B=64, 4x32x32, dim=32, 102 classes).  Meant to sit under `rocprofv3 --pmc <counter> --kernel-trace`
(one counter group per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); tools/pmc_summary.py turns the passes into per-kernel
HBM traffic per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    noise, ids = bench.synthetic_inputs(0, 1, dev)
    t = torch.full((bench.BATCH,), 500.0, device=dev)
    with torch.no_grad():
        for _ in range(n):
            v = model(noise, t, {"class_cond": ids})
    torch.cuda.synchronize()
    print("ok", float(v.abs().mean()))


if __name__ == "__main__":
    main()
