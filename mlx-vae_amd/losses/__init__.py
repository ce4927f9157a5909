"""Loss functions under the reference's names (losses/__init__.py:6-21), computed by the HIP kernels."""
from .recon import reconstruction_loss
from .kl import kl_divergence
from .enc import encoder_loss
from .dec import decoder_loss
from .info import mutual_information, posterior_collapse
from .prop import property_prediction_loss

__all__ = ["reconstruction_loss", "kl_divergence", "encoder_loss", "decoder_loss", "mutual_information",
           "posterior_collapse", "property_prediction_loss"]
