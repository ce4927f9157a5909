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


def elem_err(a, b, rtol: float = 1e-4, atol_frac: float = 1e-6):
    """Element-wise criterion next to the norm-wise one: the worst element of
    |a-b| / (rtol*|b| + atol_frac*max|b|)  (<= 1 passes), with its flat index.  A tensor whose small entries are all
    wrong passes rel_err as long as its largest entry is right; it does not pass this.  The absolute term is a
    fraction of the tensor's largest magnitude (an element far below the tensor's scale is a cancelled sum: its own
    magnitude says nothing about the fp32 rounding noise of the terms that formed it)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max()
    if scale == 0.0:
        worst = float(np.abs(a).max())
        return (0.0 if worst == 0.0 else float("inf")), 0
    ratio = np.abs(a - b) / (rtol * np.abs(b) + atol_frac * scale)
    i = int(np.argmax(ratio))
    return float(ratio.reshape(-1)[i]), i


# Element-wise tolerance = the judge's / north_star's figure taken literally: |a-b| <= 1e-4*|b| + 1e-6*max|b| for logits,
# mu, logvar, z and EVERY parameter gradient.  Headroom measured on MI355X with tools/elementwise_report.py over the
# parity cases (profiles/r02_elementwise_report.txt): the worst element of the HIP step sits at 0.36 of this bound
# (decoder.lstm_layer_0.Wx at H 512 / L 4), the fp32 ORACLE itself at 0.19 against the fp64 oracle, forward values
# below 0.1 -- so a violation is a bug, not fp32 noise.
ELEM_RTOL = 1e-4
ELEM_ATOL_FWD = 1e-6
ELEM_ATOL_GRAD = 1e-6


def assert_elem(a, b, name: str, atol_frac: float, rtol: float = ELEM_RTOL) -> float:
    worst, i = elem_err(a, b, rtol, atol_frac)
    if not worst <= 1.0:
        af, bf = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
        raise AssertionError(f"{name}: element {i} differs: got {af[i]!r}, oracle {bf[i]!r} "
                             f"(|d| = {abs(af[i] - bf[i]):.3e} = {worst:.2f} x the allowed "
                             f"{rtol:g}*|b| + {atol_frac:g}*max|b|, max|b| = {np.abs(bf).max():.3e})")
    return worst


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
