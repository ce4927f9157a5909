"""encoder_loss (reference losses/enc.py:6-42): (beta * KL, mu, logvar, z); unused by the trainer."""
from __future__ import annotations

from .kl import kl_divergence


def encoder_loss(model, x, conditions, beta: float = 0.4, eps=None):
    mu, logvar = model(x, conditions)
    z = model.reparameterize(mu, logvar, eps=eps)
    return beta * kl_divergence(mu, logvar, reduction="mean"), mu, logvar, z
