"""Shared test helpers: oracle <-> engine plumbing and error metrics."""
from __future__ import annotations

import numpy as np
import torch

import arcvae_oracle as O


def rel_err(a, b) -> float:
    """max|a-b| / max|b| (norm-wise relative error; element-wise is meaningless near zero)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.abs(b).max()
    if denom == 0.0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / denom)


TINY = O.Config(vocab_size=80, embedding_dim=16, hidden_dim=64, latent_dim=8, num_conditions=1, num_layers=2)
SMALL = O.Config(vocab_size=40, embedding_dim=32, hidden_dim=64, latent_dim=64, num_conditions=3, num_layers=3)
DEFAULT = O.Config()

HYPER = dict(beta=0.05, lambda_collapse=0.001, free_bits=1.0, lambda_mi=0.01, target_mi=4.85)


def make_case(cfg: O.Config, B: int, T: int, tf_ratio: float = 0.7, seed: int = 67):
    params = O.init_params(cfg, 1234)
    x, cond = O.synthetic_batch(cfg, B, T, seed)
    eps = np.random.RandomState(4321).standard_normal((B, cfg.Z)).astype(np.float32)
    coins = O.draw_coins(np.random.RandomState(seed + 1), T, tf_ratio)
    return params, x, cond, eps, coins


def build_engine(cfg: O.Config, params, device="cuda"):
    from arcvae_hip.engine import ModelDims, StepEngine
    from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes

    dims = ModelDims(cfg.V, cfg.E, cfg.H, cfg.Z, cfg.C, cfg.L)
    enc = ParamStore(encoder_shapes(cfg.V, cfg.E, cfg.H, cfg.Z, cfg.C, cfg.L), device)
    dec = ParamStore(decoder_shapes(cfg.V, cfg.E, cfg.H, cfg.Z, cfg.C, cfg.L), device)
    enc.load_state_dict(params, prefix="encoder.")
    dec.load_state_dict(params, prefix="decoder.")
    return StepEngine(enc, dec, dims), enc, dec
