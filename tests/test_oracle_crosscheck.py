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


# ---- a third pin: library implementations nobody in this repository wrote -------------------------------------------
# Where the reference's semantics coincide with a PyTorch library op, the oracle's hand-written restatement is checked
# against that op (fp64, CPU): the LSTM recurrence (torch.nn.LSTM: gate order i,f,g,o like MLX's nn.LSTM, M1; MLX's
# "hidden=None / cell=None" first step equals a zero initial state), the token cross-entropy (F.cross_entropy), the
# Gaussian KL (torch.distributions) and the argmax / softmax conventions.  What stays unpinned is unchanged: the MLX
# runtime's own numerics (not installable offline).
def test_oracle_lstm_stack_equals_torch_nn_lstm():
    rs = np.random.RandomState(11)
    B, T, E, H, L = 5, 9, 7, 12, 3
    x = torch.tensor(rs.standard_normal((B, T, E)))
    lib = torch.nn.LSTM(E, H, num_layers=L, batch_first=True).double()
    out = x
    with torch.no_grad():
        for l in range(L):
            Wx = getattr(lib, f"weight_ih_l{l}").detach()
            Wh = getattr(lib, f"weight_hh_l{l}").detach()
            bias = (getattr(lib, f"bias_ih_l{l}") + getattr(lib, f"bias_hh_l{l}")).detach()   # MLX keeps ONE bias (M1)
            out, cell = O.mlx_lstm(out, Wx, Wh, bias)
        ref, (hn, cn) = lib(x)
    assert torch.allclose(out, ref, rtol=0, atol=1e-12)
    assert torch.allclose(out[:, -1], hn[-1], rtol=0, atol=1e-12) and torch.allclose(cell[:, -1], cn[-1], rtol=0, atol=1e-12)
    # and the torch-free restatement's cell agrees with the same library op on one step
    pre = rs.standard_normal((B, 4 * H))
    c0 = rs.standard_normal((B, H))
    h1, c1 = N.lstm_cell(pre, c0) if hasattr(N, "lstm_cell") else (None, None)
    if h1 is not None:
        i, f, g, o = np.split(pre, 4, axis=1)
        sig = lambda v: 1.0 / (1.0 + np.exp(-v))
        c_ref = sig(f) * c0 + sig(i) * np.tanh(g)
        assert np.allclose(c1, c_ref, atol=1e-13) and np.allclose(h1, sig(o) * np.tanh(c_ref), atol=1e-13)


def test_oracle_losses_equal_torch_library_forms():
    rs = np.random.RandomState(12)
    B, T, V, Z = 6, 11, 23, 9
    logits = torch.tensor(rs.standard_normal((B, T, V)) * 3.0)
    tgt = torch.tensor(rs.randint(0, V, size=(B, T)))
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), reduction="mean")   # mean over ALL positions (Q3)
    assert abs(float(O.reconstruction_loss(logits, tgt)) - float(ce)) < 1e-12
    assert abs(float(O.reconstruction_loss(logits, tgt, "sum")) -
               float(torch.nn.functional.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), reduction="sum"))) < 1e-10
    # Gaussian KL(N(mu, sigma^2) || N(0, 1)) inside the clip range, no free bits: losses/kl.py:39-56
    mu = torch.tensor(rs.uniform(-2.5, 2.5, size=(B, Z)))
    logvar = torch.tensor(rs.uniform(-5.0, 2.5, size=(B, Z)))
    q = torch.distributions.Normal(mu, torch.exp(0.5 * logvar))
    p = torch.distributions.Normal(torch.zeros_like(mu), torch.ones_like(mu))
    kl = torch.distributions.kl_divergence(q, p).sum(dim=1)
    assert torch.allclose(O.kl_divergence(mu, logvar, reduction="none"), kl, rtol=0, atol=1e-12)
    assert abs(float(O.kl_divergence(mu, logvar)) - float(kl.mean())) < 1e-12
    # softmax / first-argmax conventions (models/decoder.py:185, decoder_sampling.py:110-117)
    assert torch.allclose(O.mlx_softmax_lastdim(logits), torch.softmax(logits, dim=-1), rtol=0, atol=1e-14)
    ties = torch.tensor([[1.0, 3.0, 3.0, 2.0], [5.0, 5.0, 5.0, 5.0]])
    assert O.mlx_argmax_lastdim(ties).tolist() == [1, 0]                 # the FIRST maximum, like mx.argmax


def test_oracle_adam_equals_torch_adam_without_bias_correction():
    """trainer.py:75-76 uses MLX optim.Adam, which applies NO bias correction (Q7).  torch.optim.Adam does; dividing its
    correction back out must reproduce the oracle's update exactly: p -= lr * m / (sqrt(v) + eps)."""
    rs = np.random.RandomState(13)
    p0 = rs.standard_normal(50)
    g1, g2 = rs.standard_normal(50), rs.standard_normal(50)
    lr, b1, b2, eps = 2e-4, 0.9, 0.999, 1e-8
    params = {"w": p0.copy()}
    m = {"w": np.zeros(50)}
    v = {"w": np.zeros(50)}
    for g in (g1, g2):
        O.adam_update(params, {"w": g}, m, v, lr, b1, b2, eps)      # in place
    # closed form of two uncorrected steps
    m1 = (1 - b1) * g1; v1 = (1 - b2) * g1 * g1
    p1 = p0 - lr * m1 / (np.sqrt(v1) + eps)
    m2 = b1 * m1 + (1 - b1) * g2; v2 = b2 * v1 + (1 - b2) * g2 * g2
    p2 = p1 - lr * m2 / (np.sqrt(v2) + eps)
    assert np.allclose(params["w"], p2, rtol=0, atol=1e-15)
    # torch's Adam keeps the SAME moments; only its step differs by the correction factors
    t = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([t], lr=lr, betas=(b1, b2), eps=eps)
    for g in (g1, g2):
        t.grad = torch.tensor(g)
        opt.step()
    st = opt.state[t]
    assert np.allclose(st["exp_avg"].numpy(), m["w"], atol=1e-15) and np.allclose(st["exp_avg_sq"].numpy(), v["w"], atol=1e-15)
