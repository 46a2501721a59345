"""flocoder_amd -- MI355X (gfx950) implementation of flocoder's latent flow-matching hot path.

Host side mirrors the reference's Python surfaces (``Unet``, ``sampler``/``generate_latents``, ``setup_codec``,
``ldcfg``); the arithmetic lives in hand-written HIP kernels behind the C ABI of ``include/flocoder_amd.h``.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def runtime_defaults(workload: str = "sampling") -> dict:
    """Process-level settings of the HIP runtime, per workload.  The runtime reads them once, when it is first loaded, so they have to be
    in the environment before torch is imported: ``bench.py`` does that for itself, ``apply_runtime_defaults()`` does it for a caller
    that imports this package first, INTEGRATION.md lists them for everyone else.  Never overrides a value the caller set.

    "sampling": AMD_DIRECT_DISPATCH=0 -- command submission goes through the runtime's own thread instead of the calling thread.  The
    sampler is a chain of ~4500 dependent launches per trajectory replayed from hipGraphs; measured on MI355X / ROCm 7.2
    (profiles/r03_env_sweep.txt, same box, alternating runs): 756-761 -> 780-790 samples/s at the bench configuration, the only one of
    thirty-odd runtime switches tried that moved the number up (and two trajectories in flight on two streams: 708 -> 955).
    "training": nothing -- the training step is ~340 plain launches paced by the host thread, which direct dispatch serves better
    (stl_sd step 2.66 ms against 2.76 under AMD_DIRECT_DISPATCH=0)."""
    return {"AMD_DIRECT_DISPATCH": "0"} if workload == "sampling" else {}


def apply_runtime_defaults(workload: str = "sampling") -> bool:
    """Put ``runtime_defaults(workload)`` into os.environ -- call it before anything imports torch.  Returns True when the settings
    are in effect for this process, False when the runtime was already loaded (nothing is changed then) or FLOCODER_AMD_KEEP_ENV is set."""
    want = runtime_defaults(workload)
    if _os.environ.get("FLOCODER_AMD_KEEP_ENV") or "torch" in _sys.modules:
        return all(_os.environ.get(k) == v for k, v in want.items())
    for k, v in want.items():
        _os.environ.setdefault(k, v)
    return all(_os.environ.get(k) == v for k, v in want.items())


from .unet import Unet  # noqa: F401,E402
