"""mutual_information / posterior_collapse (reference losses/info.py:3-78), batch-statistic estimator."""
from __future__ import annotations

import torch

from arcvae_hip._lib import call, ptr, stream_ptr

from ._dev import latent_stats


def _scalars(mu, logvar, target_mi: float, weight: float) -> torch.Tensor:
    stats, _, B, Z = latent_stats(mu, logvar, 0.0)
    dev = stats.device
    hyper = torch.tensor([0.0, weight, 0.0, target_mi, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32).to(dev)
    scalars = torch.zeros(16, dtype=torch.float32, device=dev)
    dummy = torch.zeros(1, dtype=torch.float32, device=dev)
    call("arcvae_latent_loss", ptr(stats), ptr(hyper), ptr(dummy), ptr(dummy), ptr(scalars), None, None, B, Z, 1,
         0.0, stream_ptr())
    return scalars


def mutual_information(mu, logvar) -> torch.Tensor:
    """max(mean_b KL_b - KL(aggregate), 0)  (losses/info.py:23-50)."""
    return _scalars(mu, logvar, 4.85, 0.0)[7]


def posterior_collapse(mu, logvar, target_mi: float = 4.85, weight: float = 0.1) -> torch.Tensor:
    """weight * max(0, target_mi - MI)  (losses/info.py:53-78)."""
    return _scalars(mu, logvar, target_mi, weight)[4]
