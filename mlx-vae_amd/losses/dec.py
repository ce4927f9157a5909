"""decoder_loss (reference losses/dec.py:6-35): reconstruction loss of a decoder pass; unused by the trainer."""
from __future__ import annotations

from .recon import reconstruction_loss


def decoder_loss(model, z, conditions, target_seq, teacher_forcing_ratio: float = 0.9, coins=None):
    logits = model(z, conditions, target_seq=target_seq, teacher_forcing_ratio=teacher_forcing_ratio, coins=coins)
    return reconstruction_loss(logits, target_seq, reduction="mean")
