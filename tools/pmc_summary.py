#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter group per pass) of
tools/pmc_forward.py, corrected as MI355X_MICROARCH.md "HBM" prescribes for gfx950: FETCH_SIZE (KB) x 2 for wide coalesced
reads (every global read of these kernels is 16 B per lane), WRITE_SIZE (KB) as is.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [n_forwards] > profiles/<name>.json
"""
import csv
import json
import re
import sys
from collections import defaultdict

TILES = {("4", "1", "1", "1", "1"): "conv_igemm<M128,N32>", ("4", "1", "1", "1", "2"): "conv_igemm<M128,N64>",
         ("2", "1", "2", "1", "1"): "conv_igemm<M64,N32,K2>", ("1", "1", "4", "1", "1"): "conv_igemm<M32,N32,K4>",
         ("2", "1", "2", "1", "2"): "conv_igemm<M64,N64,K2>"}


def family(name):
    m = re.search(r"conv_(?:pipe|igemm)_kernel<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return TILES.get(tuple(a[:5]), name)
    m = re.search(r"(\w+?)_kernel", name)
    fam = m.group(1) if m else name
    return {"la_ctx_fast": "linattn_fused", "la_apply_fast": "linattn_fused", "la_ctx": "linattn_fused", "la_apply": "linattn_fused",
            "la_head": "attn_heads", "la_join": "attn_heads", "cond_hidden": "temb", "cond_out": "temb"}.get(fam, fam)


def load(path, counter):
    by = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "fc::" in r["Kernel_Name"]:
            by[family(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return by


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    n_fwd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    skip = {"pack_conv", "pack_transpose", "pack_s2d", "pack_conv_pad", "pack_table"}
    out, tot_r, tot_w = {}, 0.0, 0.0
    for k in sorted(set(fetch) | set(write)):
        if k in skip:
            continue
        f, w = fetch.get(k, []), write.get(k, [])
        rd = 2.0 * sum(f) / max(len(f), 1)
        wr = sum(w) / max(len(w), 1)
        out[k] = {"launches_per_forward": len(f) // n_fwd, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "traffic_bytes_per_launch": round(rd + wr)}
        tot_r += 2.0 * sum(f) / n_fwd
        tot_w += sum(w) / n_fwd
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_forward.py, B=64 4x32x32 dim=32; "
                                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide-read correction)",
                      "forward_read_bytes": round(tot_r), "forward_write_bytes": round(tot_w), "per_kernel": out}, indent=1))


if __name__ == "__main__":
    main()
