"""Second, torch-free restatement of the reference's loss path in NumPy float64.

*** TEST INFRASTRUCTURE ONLY *** (same rule as arcvae_oracle.py: only tests/ may import it).
*** PARITY UNPINNED *** against a real MLX run, for the reasons given in arcvae_oracle.py.

Why a second restatement: the reference holds no golden vectors and MLX cannot run offline, so the
only pins available are independent ones.  This file shares NO
code with arcvae_oracle.py (no torch, no autograd): it is written from the reference's call sites
again, and tests/test_oracle_crosscheck.py requires
  * its forward values to equal the torch oracle's (fp64) to ~1e-12, and
  * central finite differences of ITS total loss to equal the torch oracle's autograd gradients
    (the gradient oracle of every hand-written HIP backward).

Reference lines followed: models/encoder.py:76-153, models/decoder.py:113-190, losses/recon.py:29-64,
losses/kl.py:35-66, losses/info.py:23-50 and :73-78, complete_vae_loss.py:37-99; MLX layer semantics
M1 (nn.LSTM: gate order i,f,g,o; hidden=None skips h.Wh^T; cell=None gives c = i*g), M2 (Linear:
x.W^T + b), M5 (maximum ties), M6 (argmax = first maximal index) as listed in SURVEY.md section 8c.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

F = np.float64


def _sig(a):
    return 1.0 / (1.0 + np.exp(-a))


def _lstm_cell(pre, c_prev):
    """One MLX nn.LSTM time step from the pre-activation [B,4H] (M1).  c_prev None -> c = i*g."""
    H = pre.shape[1] // 4
    i, f, g, o = _sig(pre[:, :H]), _sig(pre[:, H:2 * H]), np.tanh(pre[:, 2 * H:3 * H]), _sig(pre[:, 3 * H:])
    c = i * g if c_prev is None else f * c_prev + i * g
    return o * np.tanh(c), c


def encoder(p: Dict[str, np.ndarray], x: np.ndarray, cond: np.ndarray, L: int):
    """models/encoder.py:76-132 -> (mu, logvar)."""
    B, T = x.shape
    seq = p["encoder.embedding.weight"][x]                                     # [B,T,E]   :93
    for l in range(L):                                                          # :98-101
        Wx, Wh, b = (p[f"encoder.lstm_layer_{l}.{n}"] for n in ("Wx", "Wh", "bias"))
        h = c = None
        outs = []
        for t in range(T):
            pre = seq[:, t, :] @ Wx.T + b
            if h is not None:
                pre = pre + h @ Wh.T
            h, c = _lstm_cell(pre, c)
            outs.append(h)
        seq = np.stack(outs, axis=1)
    hT = seq[:, -1, :]                                                          # :106
    cr = cond @ p["encoder.condition_fc.weight"].T + p["encoder.condition_fc.bias"]
    comb = np.concatenate([hT, cr], axis=1)
    mu_raw = comb @ p["encoder.fc_mu.weight"].T + p["encoder.fc_mu.bias"]
    lh = np.tanh(comb @ p["encoder.fc_logvar_hidden.weight"].T + p["encoder.fc_logvar_hidden.bias"])
    lv_raw = lh @ p["encoder.fc_logvar.weight"].T + p["encoder.fc_logvar.bias"]
    return 2.0 * np.tanh(mu_raw / 2.0), np.tanh(lv_raw / 2.0) - 1.0            # :126, :130


def decoder(p: Dict[str, np.ndarray], cond: np.ndarray, L: int, target: Optional[np.ndarray],
            coins: Optional[Sequence[bool]], max_length: int = 80):
    """models/decoder.py:134-188 -> (logits [B,T,V], fed tokens [B,T]).  Stateless cells (Q1), z unused (Q2)."""
    B = cond.shape[0]
    T = target.shape[1] if target is not None else max_length
    cur = np.zeros(B, dtype=np.int64)                                           # :146
    logits, fed = [], []
    for t in range(T):
        fed.append(cur)
        h = np.concatenate([p["decoder.embedding.weight"][cur], cond], axis=1)  # :154-157
        for l in range(L):                                                      # :165-168, hidden=None, cell=None
            pre = h @ p[f"decoder.lstm_layer_{l}.Wx"].T + p[f"decoder.lstm_layer_{l}.bias"]
            h, _ = _lstm_cell(pre, None)
        lg = h @ p["decoder.fc_out.weight"].T + p["decoder.fc_out.bias"]        # :175
        logits.append(lg)
        if target is not None and bool(coins[t]):                               # :180
            cur = target[:, t]
        else:
            cur = np.argmax(lg, axis=1)                                         # :185 (np.argmax = first maximum, M6)
    return np.stack(logits, axis=1), np.stack(fed, axis=1)


def recon_mean(logits: np.ndarray, targets: np.ndarray) -> float:
    """losses/recon.py:29-60, mean over all B*T positions (Q3)."""
    lf = logits.reshape(-1, logits.shape[-1])
    st = lf - lf.max(axis=1, keepdims=True)
    lsm = st - np.log(np.exp(st).sum(axis=1, keepdims=True))
    return float(-lsm[np.arange(lf.shape[0]), targets.reshape(-1)].mean())


def _clipped(mu, logvar):
    return np.minimum(np.maximum(mu, -3.0), 3.0), np.minimum(np.maximum(logvar, -6.0), 3.0)


def kl_mean(mu, logvar, free_bits: float) -> float:
    """losses/kl.py:39-62."""
    mu, logvar = _clipped(mu, logvar)
    k = np.maximum(-0.5 * (1.0 + logvar - mu * mu - np.exp(logvar)), 0.0)
    if free_bits > 0.0:
        k = np.maximum(k, free_bits / mu.shape[1])
    return float(k.sum(axis=1).mean())


def mutual_info(mu, logvar, log_eps: float = 0.0) -> float:
    """losses/info.py:31-48 (log_eps = 0) / trainer.py:549-575 (log_eps = 1e-8, Q20)."""
    mu, logvar = _clipped(mu, logvar)
    var = np.exp(logvar)
    mean_kl = (-0.5 * (1.0 + logvar - mu * mu - var).sum(axis=1)).mean()
    m, v = mu.mean(axis=0), var.mean(axis=0)
    agg = -0.5 * (1.0 + np.log(v + log_eps) - m * m - v).sum()
    return float(max(mean_kl - agg, 0.0))


def complete_vae_loss(params: Dict[str, np.ndarray], L: int, x, cond, eps, coins, beta=0.4, lambda_prop=0.1,
                      lambda_collapse=0.01, free_bits=0.5, lambda_mi=0.0, target_mi=4.85) -> Dict[str, object]:
    """complete_vae_loss.py:37-99 with property_predictor=None (Q10)."""
    p = {k: np.asarray(v, dtype=F) for k, v in params.items()}
    x = np.asarray(x, dtype=np.int64)
    cond = np.asarray(cond, dtype=F)
    mu, logvar = encoder(p, x, cond, L)
    z = mu + np.asarray(eps, dtype=F) * np.exp(0.5 * logvar)                    # encoder.py:147-153
    logits, fed = decoder(p, cond, L, x, coins)
    recon = recon_mean(logits, x)
    kl = kl_mean(mu, logvar, free_bits)
    mi = mutual_info(mu, logvar)
    collapse = lambda_collapse * max(0.0, target_mi - mi)                       # info.py:73-78
    mi_pen = lambda_mi * max(0.0, target_mi - mi)
    total = recon + beta * kl + collapse + lambda_prop * 0.0 + mi_pen           # :76-82
    return {"total_loss": total, "recon_loss": recon, "kl_loss": kl, "weighted_kl": beta * kl,
            "collapse_penalty": collapse, "prop_loss": 0.0, "weighted_prop_loss": 0.0, "mutual_info": mi,
            "mi_penalty": mi_pen, "mu": mu, "logvar": logvar, "z": z, "logits": logits, "fed_tokens": fed}


def finite_difference(params: Dict[str, np.ndarray], name: str, index, h: float, loss_fn) -> float:
    """Central difference of loss_fn(params) with respect to params[name][index]."""
    base = params[name]
    w = np.array(base, dtype=F, copy=True)
    q = dict(params)
    q[name] = w
    w[index] = base[index] + h
    up = loss_fn(q)
    w[index] = base[index] - h
    dn = loss_fn(q)
    return (up - dn) / (2.0 * h)
