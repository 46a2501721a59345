"""Builds the gfx950 shared library in-tree: ``python -m flocoder_amd.build``.

One ``hipcc --offload-arch=gfx950`` compile per ``csrc/*.hip`` (objects cached by mtime under ``csrc/_obj``),
linked into ``flocoder_amd/_lib/libflocoder_amd.so``.  No torch, no cmake: the library is plain C ABI.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libflocoder_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "flocoder_amd.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, built: list) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), _headers_mtime())
    if stale:
        built.append(src)
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


LAST_BUILD = {"rebuilt": 0, "units": 0, "linked": False}


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what is stale (everything with ``force``) and link.  ``LAST_BUILD`` records how many units were really compiled, so a
    caller can tell a build from a cache hit."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    built: list = []
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, built), srcs))
    relink = force or bool(built) or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs)
    LAST_BUILD.update(rebuilt=len(built), units=len(srcs), linked=relink)
    if relink:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[flocoder_amd.build] rebuilt {len(built)}/{len(srcs)} units, {'linked' if relink else 'link up to date'}: "
              f"{LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
