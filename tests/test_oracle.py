"""CPU tests of the oracle: golden fixtures, the reference's own sign/inequality checks
(test_loss_signs.py:18-86, the only thing the reference pins), and the behavioural quirks
Q1-Q13 of SURVEY.md section 0 that the HIP path is built to reproduce."""
import os

import numpy as np
import pytest
import torch

import arcvae_oracle as O
from helpers import HYPER, SMALL, TINY, make_case, rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg", [("tiny_step.npz", TINY), ("small_step.npz", SMALL)])
def test_fp32_oracle_matches_fp64_golden(name, cfg):
    g = np.load(os.path.join(GOLD, name))
    B, T, tf = (int(v) for v in g["meta"])
    params, x, cond, eps, coins = make_case(cfg, B, T, tf / 1000.0)
    vals, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float32, **HYPER)
    for k in ("total_loss", "recon_loss", "kl_loss", "mutual_info", "collapse_penalty", "mi_penalty"):
        assert abs(float(vals[k]) - float(g[f"val.{k}"])) < 2e-5 * max(1.0, abs(float(g[f"val.{k}"]))), k
    assert np.array_equal(vals["fed_tokens"], g["val.fed_tokens"])
    for k in ("mu", "logvar", "z", "logits"):
        assert rel_err(vals[k], g[f"val.{k}"]) < 1e-5, k
    for k, v in grads.items():
        ref = g[f"grad.{k}"]
        if np.abs(ref).max() == 0:
            assert np.abs(v).max() == 0, k
        else:
            assert rel_err(v, ref) < 2e-5, k


# ---- the reference's own checks (test_loss_signs.py), same shapes and input distributions -----
def _loss_sign_inputs():
    rs = np.random.RandomState(0)
    B, T, V, Z = 32, 120, 95, 128  # test_loss_signs.py:19-23
    logits = torch.tensor(rs.standard_normal((B, T, V)), dtype=torch.float32)
    targets = torch.tensor(rs.randint(0, V, (B, T)))
    mu = torch.tensor(rs.standard_normal((B, Z)) * 0.1, dtype=torch.float32)
    logvar = torch.tensor(rs.standard_normal((B, Z)) * 0.1 - 1.0, dtype=torch.float32)
    return logits, targets, mu, logvar


def test_reference_sign_checks():
    logits, targets, mu, logvar = _loss_sign_inputs()
    assert float(O.reconstruction_loss(logits, targets)) >= 0                       # :34-38
    assert float(O.kl_divergence(mu, logvar, free_bits=0.0)) >= 0                   # :49-53
    mi = float(O.mutual_information(mu, logvar))                                    # :56-61
    assert np.isfinite(mi) and mi >= 0
    assert float(O.posterior_collapse(mu, logvar, target_mi=4.85, weight=0.1)) >= 0  # :69-72


def test_recon_reductions_and_free_bits_floor():
    logits, targets, mu, logvar = _loss_sign_inputs()
    ce = O.reconstruction_loss(logits, targets, reduction="none")
    assert ce.shape == (32 * 120,)
    assert torch.allclose(ce.mean(), O.reconstruction_loss(logits, targets, "mean"))
    assert torch.allclose(ce.sum(), O.reconstruction_loss(logits, targets, "sum"), rtol=1e-5)
    per = O.kl_divergence(mu, logvar, reduction="none", free_bits=1.0)
    assert float(per.min()) >= 1.0 - 1e-6  # per-sample KL >= free_bits (Q12)


# ---- quirks --------------------------------------------------------------------------------------
def test_q1_q2_logits_ignore_z_and_decoder_wh():
    cfg = TINY
    params, x, cond, eps, coins = make_case(cfg, 4, 12, 0.7)
    a = O.complete_vae_loss(O.to_torch(params), cfg, torch.tensor(x), torch.tensor(cond), torch.tensor(eps), coins, **HYPER)
    p2 = {k: v.copy() for k, v in params.items()}
    for l in range(cfg.L):
        p2[f"decoder.lstm_layer_{l}.Wh"] += 1.0
    p2["decoder.z_to_hidden.weight"] += 1.0
    b = O.complete_vae_loss(O.to_torch(p2), cfg, torch.tensor(x), torch.tensor(cond), torch.tensor(eps * 5 + 1), coins, **HYPER)
    assert torch.equal(a["logits"], b["logits"])
    assert float(a["total_loss"]) == float(b["total_loss"])
    assert not torch.equal(a["z"], b["z"])


def test_dead_parameters_have_zero_grad_and_survive_adam():
    cfg = TINY
    params, x, cond, eps, coins = make_case(cfg, 4, 12, 0.7)
    p = {k: v.copy() for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(vv) for k, vv in p.items()}
    _, grads = O.train_step(p, m, v, cfg, x, cond, eps, coins, 2e-4, **HYPER)
    dead = ["decoder.z_to_hidden.weight", "decoder.z_to_hidden.bias", "decoder.condition_to_hidden.weight",
            "decoder.condition_to_hidden.bias"] + [f"decoder.lstm_layer_{l}.Wh" for l in range(cfg.L)]
    for k in dead:
        assert np.abs(grads[k]).max() == 0.0
        assert np.array_equal(p[k], params[k])
    # forget-gate rows of the decoder's Wx/bias are dead too (zero-state cell)
    H = cfg.H
    assert np.abs(grads["decoder.lstm_layer_1.Wx"][H:2 * H]).max() == 0.0
    assert np.abs(grads["decoder.lstm_layer_0.bias"][H:2 * H]).max() == 0.0
    assert np.abs(grads["decoder.fc_out.weight"]).max() > 0.0


def test_q5_coin_stream_is_the_global_legacy_stream():
    np.random.seed(67)
    a = O.draw_coins(np.random, 16, 0.9)
    rs = np.random.RandomState(67)
    b = np.array([rs.rand() < 0.9 for _ in range(16)])
    assert np.array_equal(a, b)
    np.random.seed(67)
    O.draw_coins(np.random, 16, 0.0)  # ratio 0 still consumes the stream (decoder.py:180)
    assert np.random.rand() == np.random.RandomState(67).rand(17)[-1]


def test_q7_adam_first_step_identity():
    g = np.array([0.3, -2.0, 1e-3, 0.0], dtype=np.float32)
    p = {"w": np.zeros(4, np.float32)}
    m = {"w": np.zeros(4, np.float32)}
    v = {"w": np.zeros(4, np.float32)}
    O.adam_update(p, {"w": g}, m, v, lr=2e-4)
    upd = 2e-4 * 0.1 * g / (np.sqrt(0.001) * np.abs(g) + 1e-8)
    assert np.allclose(-p["w"], upd, rtol=1e-5, atol=1e-12)
    assert p["w"][3] == 0.0  # zero-gradient parameter unchanged exactly


def test_q13_schedules():
    assert O.compute_beta(0, 0.0, 0.05, 20) == 0.0
    assert abs(O.compute_beta(1, 0.0, 0.05, 20) - 0.0025) < 1e-12
    assert O.compute_beta(25, 0.0, 0.05, 20) == 0.05
    assert O.compute_teacher_forcing_ratio(0, 30) == 0.9
    assert O.compute_teacher_forcing_ratio(30, 30) == 0.5


def test_m5_m6_tie_rules():
    a = torch.tensor([0.0, 1.0, -1.0], requires_grad=True)
    O.mlx_maximum(a, 0.0).sum().backward()
    assert a.grad.tolist() == [0.0, 1.0, 0.0]  # tie -> second operand
    x = torch.tensor([[1.0, 3.0, 3.0, 2.0]])
    assert int(O.mlx_argmax_lastdim(x)) == 1     # first maximal index


def test_sampler_is_greedy_and_temperature_invariant():
    cfg = TINY
    params = O.init_params(cfg, 1234)
    pd = {k[len("decoder."):]: torch.tensor(v) for k, v in params.items() if k.startswith("decoder.")}
    cond = torch.tensor(np.random.RandomState(3).standard_normal((6, cfg.C)).astype(np.float32))
    a = O.generate_with_temperature(pd, cond, cfg.L, max_length=20, temperature=1.0, early_stopping=False)
    b = O.generate_with_temperature(pd, cond, cfg.L, max_length=20, temperature=0.7, early_stopping=False)
    assert a.shape == (6, 20) and torch.equal(a, b)
