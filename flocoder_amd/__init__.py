"""flocoder_amd -- MI355X (gfx950) implementation of flocoder's latent flow-matching hot path.

Host side mirrors the reference's Python surfaces (``Unet``, ``sampler``/``generate_latents``, ``setup_codec``,
``ldcfg``); the arithmetic lives in hand-written HIP kernels behind the C ABI of ``include/flocoder_amd.h``.
"""
__version__ = "0.1.0"

from .unet import Unet  # noqa: F401
