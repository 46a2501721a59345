"""flocoder_amd -- MI355X (gfx950) implementation of flocoder's latent flow-matching hot path.

Host side mirrors the reference's Python surfaces (``Unet``, ``sampler``/``generate_latents``, ``setup_codec``,
``ldcfg``); the arithmetic lives in hand-written HIP kernels behind the C ABI of ``include/flocoder_amd.h``.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def runtime_defaults() -> dict:
    """Process-level settings of the HIP runtime this package's launch pattern wants; applied by ``import flocoder_amd`` when nothing
    has loaded the runtime yet (importing this package BEFORE torch), by ``bench.py`` explicitly, and listed in INTEGRATION.md for
    callers that start torch first.  Never overrides a value the caller set.

    AMD_DIRECT_DISPATCH=0: hand command submission to the runtime's own thread instead of submitting from the calling thread.  The
    sampler is a chain of ~4500 dependent launches per trajectory replayed from hipGraphs; measured on MI355X / ROCm 7.2
    (profiles/r03_env_sweep.txt, same box, alternating runs): 756-761 -> 780-790 samples/s at the bench configuration, the only one of
    thirty-odd runtime switches tried that moved the number up."""
    return {"AMD_DIRECT_DISPATCH": "0"}


def apply_runtime_defaults() -> bool:
    """Set ``runtime_defaults()`` in os.environ unless torch (and with it the HIP runtime's flag table) is already loaded or
    FLOCODER_AMD_KEEP_ENV is set.  Returns True when the defaults are in effect for this process."""
    if _os.environ.get("FLOCODER_AMD_KEEP_ENV"):
        return all(_os.environ.get(k) == v for k, v in runtime_defaults().items())
    late = "torch" in _sys.modules
    for k, v in runtime_defaults().items():
        if not late:
            _os.environ.setdefault(k, v)
    return all(_os.environ.get(k) == v for k, v in runtime_defaults().items()) and not late


RUNTIME_DEFAULTS_APPLIED = apply_runtime_defaults()

from .unet import Unet  # noqa: F401,E402
