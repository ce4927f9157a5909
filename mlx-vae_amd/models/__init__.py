"""Model classes of the AR-CVAE SELFIES path, under the reference's names (models/__init__.py:6-11)."""
from .encoder import MLXEncoder
from .decoder import MLXAutoregressiveDecoder
from .decoder_sampling import MLXAutoregressiveDecoderSampling
from .vae import ARCVAE

__all__ = ["MLXEncoder", "MLXAutoregressiveDecoder", "MLXAutoregressiveDecoderSampling", "ARCVAE"]
