"""complete_vae_loss on MI355X: same signature, defaults and 12-key result as the reference
(complete_vae_loss.py:7-99); the values come from the fused HIP forward of arcvae_hip."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from arcvae_hip import api


def complete_vae_loss(encoder, decoder, property_predictor, x, conditions, beta: float = 0.4,
                      lambda_prop: float = 0.1, lambda_collapse: float = 0.01, teacher_forcing_ratio: float = 0.9,
                      free_bits: float = 0.5, lambda_mi: float = 0.0, target_mi: float = 4.85,
                      eps: Optional[torch.Tensor] = None, coins: Optional[Sequence[bool]] = None) -> dict:
    """total = recon + beta*kl + collapse + lambda_prop*prop + mi_penalty  (complete_vae_loss.py:76-82).

    `property_predictor` must be None: the reference's predictor branch cannot run (it calls
    property_prediction_loss with the wrong arity), so prop_loss is identically 0.
    `eps` / `coins` are additive hooks to inject the reparameterisation noise and the per-step
    teacher-forcing decisions; by default they are drawn as the reference draws them."""
    if property_predictor is not None:
        raise NotImplementedError("property_predictor is not supported (unreachable in the reference, Q10)")
    out = api.loss_forward(encoder, decoder, x, conditions, eps=eps, coins=coins,
                           teacher_forcing_ratio=teacher_forcing_ratio, beta=beta, lambda_collapse=lambda_collapse,
                           lambda_mi=lambda_mi, target_mi=target_mi, free_bits=free_bits)
    # lambda_prop * 0 == 0: weighted_prop_loss stays the zero the kernel wrote
    return out
