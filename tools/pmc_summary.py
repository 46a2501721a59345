#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter group per pass) of
tools/pmc_forward.py, corrected as MI355X_MICROARCH.md "HBM" prescribes for gfx950: FETCH_SIZE (KB) x 2 for wide coalesced
reads (every global read of these kernels is 16 B per lane), WRITE_SIZE (KB) as is.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [n_forwards] [workload text] > profiles/<name>.json
"""
import csv
import json
import re
import sys
from collections import defaultdict

TILES = {("4", "1", "1", "2", "2"): "conv_igemm<M256,N64>",
         ("4", "1", "1", "1", "1"): "conv_igemm<M128,N32>", ("4", "1", "1", "1", "2"): "conv_igemm<M128,N64>",
         ("2", "1", "2", "1", "1"): "conv_igemm<M64,N32,K2>", ("1", "1", "4", "1", "1"): "conv_igemm<M32,N32,K4>",
         ("2", "1", "2", "1", "2"): "conv_igemm<M64,N64,K2>"}


FL_FIN, FL_ALL = 1, 4095          # conv_dev.h: bit 0 of a lean flavour's mask = the launch closes its Block


def conv_slice(name):
    """'+fin' / ' (plain)' for a convolution launch, from the flavour mask in the kernel name (10th template argument).  The all-in-one
    instantiation (mask 4095) does not say; the U-Net's Block-closing launches all go to lean flavours, so it counts as plain."""
    m = re.search(r"conv_pipe_kernel<([^>]*)>", name)
    if not m:
        return " (plain)"
    a = [x.strip() for x in m.group(1).split(",")]
    try:
        fl = int(a[9]) if len(a) > 9 else FL_ALL
    except ValueError:
        fl = FL_ALL
    return "+fin" if (fl != FL_ALL and fl & FL_FIN) else " (plain)"


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
            fam, v = family(r["Kernel_Name"]), float(r["Counter_Value"]) * 1024.0
            by[fam].append(v)
            if fam.startswith("conv_igemm<"):       # the same launches again under their slice: bench.py's roofline.slices keys
                by[fam + conv_slice(r["Kernel_Name"])].append(v)
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
        if not (k.endswith("+fin") or k.endswith(" (plain)")):     # slices repeat their family's launches
            tot_r += 2.0 * sum(f) / n_fwd
            tot_w += sum(w) / n_fwd
    what = sys.argv[4] if len(sys.argv) > 4 else "tools/pmc_forward.py, B=64 4x32x32 dim=32"
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over " + what + "; "
                                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide-read correction); keys = bench.py's kernel families, "
                                "'<family>+fin' / '<family> (plain)' = the family's Block-closing / other launches",
                      "forward_read_bytes": round(tot_r), "forward_write_bytes": round(tot_w), "per_kernel": out}, indent=1))


if __name__ == "__main__":
    main()
