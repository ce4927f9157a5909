"""reconstruction_loss (reference losses/recon.py:6-64): mean CE over ALL B*T positions, pads included."""
from __future__ import annotations

import torch

from arcvae_hip._lib import call, ptr, stream_ptr
from arcvae_hip.module import as_f32, as_tokens

from ._dev import dev_of, device_sum


def reconstruction_loss(logits, targets, reduction: str = "mean") -> torch.Tensor:
    dev = dev_of(logits)
    logits = as_f32(logits, dev)
    V = logits.shape[-1]
    lf = logits.reshape(-1, V)
    tf = as_tokens(targets, dev).reshape(-1)
    R = lf.shape[0]
    ce = torch.empty(R, dtype=torch.float32, device=dev)
    call("arcvae_ce_rows", ptr(lf), ptr(tf), ptr(ce), R, V, stream_ptr())
    if reduction == "mean":
        return device_sum(ce, 1.0 / R)
    if reduction == "sum":
        return device_sum(ce)
    return ce
