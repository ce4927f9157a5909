"""kl_divergence (reference losses/kl.py:5-66): clip, per-dimension >= 0 and free-bits floors, sum over Z."""
from __future__ import annotations

import torch

from ._dev import device_sum, latent_stats


def kl_divergence(mu, logvar, reduction: str = "mean", free_bits: float = 0.0) -> torch.Tensor:
    stats, krow, B, Z = latent_stats(mu, logvar, free_bits, want_rows=True)
    if reduction == "mean":
        return device_sum(krow, 1.0 / B)
    if reduction == "sum":
        return device_sum(krow)
    return krow
