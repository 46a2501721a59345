#!/usr/bin/env python3
"""MFMA-pipe occupancy per kernel family from one rocprofv3 PMC pass over tools/pmc_forward.py:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace ...
    python tools/pmc_mfma_summary.py <counter_collection.csv> > profiles/<name>_pmc_mfma.json

Per family (summed over its launches): matrix-pipe busy cycles, the kernel's active cycles, and
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)        (rocprofv3's own MfmaUtil formula; the csv
  reports GRBM_GUI_ACTIVE summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back")
  mfma_flops = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512                                        (cross-check against the plan's algorithmic FLOPs)
An fp32 32x32x2 MFMA keeps its SIMD's matrix pipe busy for 64 cycles per 4096 FLOP, so util x 157.3 TFLOP/s (at the clock the chip holds)
is what the arithmetic alone accounts for; the rest of a launch is its load burst, its store burst and the waits between them."""
import csv
import json
import sys
from collections import defaultdict

from pmc_summary import family

N_XCD, N_SIMD = 8, 1024


def main():
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        if "fc::" not in r["Kernel_Name"]:
            continue
        fam = family(r["Kernel_Name"])
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[fam].add(r["Dispatch_Id"])
    out = {}
    tot = defaultdict(float)
    for fam, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        if fam.startswith("pack"):
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[fam] = {"launches": len(launches[fam]), "gpu_active_cycles": round(gui), "mfma_busy_cycles_all_simds": round(busy),
                    "mfma_util": round(busy / (gui * N_SIMD), 4) if gui else None,
                    # counter collection serialises and slows the dispatches (GRBM_GUI_ACTIVE per launch reads about twice the un-profiled
                    # duration), so the busy cycles are also given per SIMD per launch: divide by (un-profiled duration x clock) for the
                    # utilisation inside a normal launch; at the 2.4 GHz peak clock they are the microseconds of pure matrix work per launch
                    "mfma_busy_cycles_per_simd_per_launch": round(busy / N_SIMD / max(len(launches[fam]), 1)),
                    "mfma_us_per_launch_at_2p4_ghz": round(busy / N_SIMD / max(len(launches[fam]), 1) / 2400.0, 2),
                    "mfma_gflop": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / 1e9, 3),
                    "sq_busy_cycles": round(c.get("SQ_BUSY_CYCLES", 0.0)), "sq_wave_cycles": round(c.get("SQ_WAVE_CYCLES", 0.0))}
        for k in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            tot[k] += c.get(k, 0.0)
    gui = tot["GRBM_GUI_ACTIVE"] / N_XCD
    print(json.dumps({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE "
                                "--kernel-trace over tools/pmc_forward.py (3 plain forwards, B=64 4x32x32 dim=32)",
                      "all_kernels": {"gpu_active_cycles": round(gui), "mfma_util": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * N_SIMD), 4) if gui else None,
                                      "mfma_gflop": round(tot["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / 1e9, 3)},
                      "per_kernel": out}, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
