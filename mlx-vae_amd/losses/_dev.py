"""Shared device helpers for the stand-alone loss functions."""
from __future__ import annotations

import numpy as np
import torch

from arcvae_hip._lib import call, ptr, stream_ptr
from arcvae_hip.module import as_f32, resolve_device


def dev_of(*ts):
    for t in ts:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return resolve_device()


def latent_stats(mu, logvar, free_bits: float = 0.0, want_rows: bool = False):
    """(stats [2Z+4] device, krow [B] or None, B, Z): partial sums of losses/kl.py and losses/info.py."""
    dev = dev_of(mu, logvar)
    mu, logvar = as_f32(mu, dev), as_f32(logvar, dev)
    B, Z = mu.shape
    stats = torch.empty(2 * Z + 4, dtype=torch.float32, device=dev)
    krow = torch.empty(B, dtype=torch.float32, device=dev) if want_rows else None
    call("arcvae_latent_stats", ptr(mu), ptr(logvar), ptr(stats), ptr(krow), B, Z, float(free_bits), stream_ptr())
    return stats, krow, B, Z


def device_sum(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    out = torch.empty((), dtype=torch.float32, device=x.device)
    call("arcvae_sum", ptr(x), x.numel(), ptr(out), float(scale), stream_ptr())
    return out
