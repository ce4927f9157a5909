"""Extra pins for the oracle (the reference holds no vectors, PARITY UNPINNED):

  * a second, torch-free NumPy-fp64 restatement (oracle/arcvae_numpy.py) must give the torch oracle's forward
    values -- two independent readings of the reference's call sites agree;
  * central finite differences of the NumPy forward must give the torch oracle's autograd gradients -- the
    gradient oracle every hand-written HIP backward is checked against is itself checked;
  * hypothesis-driven shapes (ragged batches, T = 1, L = 1..4, C = 1..6, odd vocabularies).
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import arcvae_numpy as N
import arcvae_oracle as O
from helpers import HYPER, SMALL, TINY, make_case

SCALARS = ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "mutual_info", "mi_penalty")


def _both(cfg, B, T, tf, seed=67, hyper=HYPER):
    params, x, cond, eps, coins = make_case(cfg, B, T, tf, seed)
    ref = O.complete_vae_loss(O.to_torch(params, torch.float64), cfg, torch.tensor(x),
                              torch.tensor(cond, dtype=torch.float64), torch.tensor(eps, dtype=torch.float64), coins,
                              **hyper)
    got = N.complete_vae_loss(params, cfg.L, x, cond, eps, coins, **hyper)
    return params, (x, cond, eps, coins), ref, got


def _compare(ref, got):
    assert np.array_equal(got["fed_tokens"], ref["fed_tokens"].numpy())
    for k in SCALARS:
        assert abs(got[k] - float(ref[k])) <= 1e-11 * max(1.0, abs(float(ref[k]))), k
    for k in ("mu", "logvar", "z", "logits"):
        r = ref[k].numpy()
        assert np.abs(got[k] - r).max() <= 1e-11 * max(1.0, np.abs(r).max()), k


@pytest.mark.parametrize("cfg,B,T,tf", [(TINY, 4, 12, 0.7), (TINY, 7, 1, 0.5), (SMALL, 21, 17, 0.5), (SMALL, 5, 9, 0.0)])
def test_numpy_restatement_equals_torch_oracle(cfg, B, T, tf):
    _, _, ref, got = _both(cfg, B, T, tf)
    _compare(ref, got)


@settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck))
@given(V=st.integers(5, 97), E=st.integers(1, 24), Hq=st.integers(1, 3), Z=st.integers(1, 20), C=st.integers(1, 6),
       L=st.integers(1, 4), B=st.integers(1, 19), T=st.integers(1, 14), tf=st.sampled_from([0.0, 0.5, 1.0]),
       beta=st.sampled_from([0.0, 0.05, 0.4]), fb=st.sampled_from([0.0, 0.5, 1.0]))
def test_numpy_restatement_equals_torch_oracle_on_random_shapes(V, E, Hq, Z, C, L, B, T, tf, beta, fb):
    cfg = O.Config(vocab_size=V, embedding_dim=E, hidden_dim=8 * Hq, latent_dim=Z, num_conditions=C, num_layers=L)
    hyper = dict(beta=beta, lambda_collapse=0.01, free_bits=fb, lambda_mi=0.01, target_mi=4.85)
    _, _, ref, got = _both(cfg, B, T, tf, seed=V + 7 * B, hyper=hyper)
    _compare(ref, got)


def test_finite_differences_match_the_autograd_oracle():
    """~50 random entries of every parameter tensor (all entries of small ones): central differences of the NumPy
    forward against torch autograd on the oracle, at TINY in fp64.  Dead parameters (Q1/Q2) have exactly zero
    derivative both ways."""
    cfg, B, T = TINY, 4, 12
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.7)
    _, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}

    def loss(q):
        return N.complete_vae_loss(q, cfg.L, x, cond, eps, coins, **HYPER)["total_loss"]

    rs = np.random.RandomState(5)
    worst = 0.0
    for name, g in grads.items():
        n = g.size
        picks = np.arange(n) if n <= 50 else rs.choice(n, size=50, replace=False)
        scale = max(np.abs(g).max(), 1e-12)
        for flat in picks:
            idx = np.unravel_index(int(flat), g.shape)
            fd = N.finite_difference(p64, name, idx, 1e-4, loss)
            if np.abs(g).max() == 0.0:
                assert fd == 0.0, f"dead parameter {name}{idx} moves the loss"
                continue
            err = abs(fd - g[idx]) / scale
            worst = max(worst, err)
            assert err < 2e-6, (name, idx, fd, g[idx])
    assert worst < 2e-6


def test_finite_differences_three_layers_kl_active():
    """Same check on SMALL (L = 3, C = 3) with a KL weight that makes every latent term active."""
    cfg, B, T = SMALL, 5, 6
    hyper = dict(beta=0.4, lambda_collapse=0.01, free_bits=0.0, lambda_mi=0.05, target_mi=4.85)
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.5)
    _, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **hyper)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}

    def loss(q):
        return N.complete_vae_loss(q, cfg.L, x, cond, eps, coins, **hyper)["total_loss"]

    rs = np.random.RandomState(6)
    for name, g in grads.items():
        if np.abs(g).max() == 0.0:
            continue
        scale = np.abs(g).max()
        for flat in rs.choice(g.size, size=min(g.size, 12), replace=False):
            idx = np.unravel_index(int(flat), g.shape)
            fd = N.finite_difference(p64, name, idx, 1e-4, loss)
            assert abs(fd - g[idx]) / scale < 2e-6, (name, idx, fd, g[idx])
