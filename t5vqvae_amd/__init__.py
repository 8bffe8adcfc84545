"""MI355X-native VQ bottleneck: drop-in VectorQuantizer / VectorQuantizerEMA over hand-written HIP kernels."""
from .vqvae import VectorQuantizer, VectorQuantizerEMA  # noqa: F401

__all__ = ["VectorQuantizer", "VectorQuantizerEMA"]
