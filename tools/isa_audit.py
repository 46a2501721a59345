#!/usr/bin/env python3
"""Find dependent global-load round trips in the compiled kernels (build container, no GPU needed).

Inside a sampler step every launch finds its operands cold, so a `global_load ... s_waitcnt vmcnt(0)` pair that repeats -- a loop
whose body is a load with a run-time trip count, or a straight-line run of load / wait / load / wait -- costs one memory round
trip (1-2 us) per repetition on the launch's critical path.  Warm per-kernel timings do not show it; the ISA does.

    python tools/isa_audit.py [file.hip ...]          # default: every csrc/*.hip

Prints, per kernel, (a) loops of < 400 instructions that contain both a global load and a full vmcnt(0) wait and no MFMA-only body,
(b) straight-line runs of >= 3 waits with <= 3 loads between consecutive waits.  Intentional cases (batched 32-row rounds of the
conditioning kernels, spin loops of the optional fused tail, tails of guarded loops) are for the reader to rule out.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flocoder_amd", "csrc")


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


def audit(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC, path, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    kern, labels = None, {}
    last_wait, loads, run, run_start = None, 0, 0, None
    findings = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, labels = m.group(1), {}
            last_wait, loads, run = None, 0, 0
            continue
        if kern is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            a = labels[m.group(1)]
            body = lines[a:i]
            nload = sum("global_load" in x for x in body)
            nwait = sum("s_waitcnt vmcnt(0)" in x for x in body)
            if nload and nwait and len(body) < 400:
                findings.append((kern, f"loop  lines {a}-{i}: {nload} loads, {nwait} full waits, {sum('v_mfma' in x for x in body)} mfma"))
        if "global_load" in l:
            loads += 1
        if "s_waitcnt vmcnt(0)" in l:
            if last_wait is not None and 1 <= loads <= 3 and i - last_wait < 60:
                if run == 0:
                    run_start = last_wait
                run += 1
            else:
                if run >= 3:
                    findings.append((kern, f"run   lines {run_start}-{last_wait}: {run} load/wait pairs in a row"))
                run = 0
            last_wait, loads = i, 0
    return findings


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for f in files:
        found = audit(f)
        print(f"== {os.path.basename(f)}: {len(found)} finding(s)")
        for kern, what in found:
            print(f"   {demangle(kern)[:110]:110s} {what}")


if __name__ == "__main__":
    main()
