"""property_prediction_loss (reference losses/prop.py:5-40).  Unreachable on the training path
(property_predictor is None and the reference's call site is broken); kept for the API."""
from __future__ import annotations

import torch

from arcvae_hip.module import as_f32

from ._dev import dev_of, device_sum


def property_prediction_loss(z, predicted_properties, target_properties, property_scales=None,
                             reduction: str = "mean") -> torch.Tensor:
    dev = dev_of(predicted_properties, target_properties)
    d = as_f32(predicted_properties, dev) - as_f32(target_properties, dev)
    mse = (d * d).contiguous()
    if property_scales is not None:
        s = as_f32(property_scales, dev)
        mse = (mse / (s * s + 1e-8)).contiguous()
    if reduction == "mean":
        return device_sum(mse, 1.0 / mse.numel())
    if reduction == "sum":
        return device_sum(mse)
    return mse
