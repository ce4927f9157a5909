"""CPU restatement of the Raiden-Makoto/MLX-VAE SELFIES training path.

*** TEST INFRASTRUCTURE ONLY ***  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.  The product path
(``mlx-vae_amd/``) never imports it and never falls back to it.

*** PARITY UNPINNED ***  The reference holds no golden vectors for this path
(``test_loss_signs.py`` only checks signs on unseeded
inputs) and its arithmetic lives in the third-party ``mlx>=0.11.0`` package
(``requirements.txt:193``, unpinned, not importable offline: ``ModuleNotFoundError``).
This file therefore restates the algorithm literally, op for op, from the
reference's call sites plus the published MLX layer semantics M1-M8 (kept
together in the ``MLX semantics`` block below so a session with MLX access can
re-check them).  What it *is* pinned against: the reference's own sign /
inequality properties (tests/test_oracle.py) and an fp64-vs-fp32 self check.

It is written with eager torch-CPU tensors (``dtype`` selectable: float32 is the
reference's arithmetic, float64 is the "truth" used to separate reduction-order
noise from bugs) and uses torch autograd for the backward, which replaces
``mx.value_and_grad`` (reference trainer.py:292).

Everything is deliberately literal: Python-unrolled time loops, one decoder
step per timestep, no batching tricks.  The HIP engine restructures the same
mathematics (vocabulary-dense decoder evaluation, wavefront LSTM sweep); this
file is what it is checked against.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# configuration / parameter naming (reference: models/encoder.py:46-69,
# models/decoder.py:51-73; MLX nn.LSTM parameter names Wx, Wh, bias)
# --------------------------------------------------------------------------- #
@dataclasses.dataclass(frozen=True)
class Config:
    vocab_size: int = 80       # train.py:25
    embedding_dim: int = 128   # train.py:26
    hidden_dim: int = 256      # train.py:27
    latent_dim: int = 128      # train.py:28
    num_conditions: int = 1    # train.py:29
    num_layers: int = 2        # train.py:30

    @property
    def V(self): return self.vocab_size
    @property
    def E(self): return self.embedding_dim
    @property
    def H(self): return self.hidden_dim
    @property
    def Z(self): return self.latent_dim
    @property
    def C(self): return self.num_conditions
    @property
    def L(self): return self.num_layers


def param_shapes(cfg: Config) -> Dict[str, Tuple[int, ...]]:
    """Ordered name -> shape map for encoder.* and decoder.* parameters."""
    V, E, H, Z, C, L = cfg.V, cfg.E, cfg.H, cfg.Z, cfg.C, cfg.L
    s: Dict[str, Tuple[int, ...]] = {}
    s["encoder.embedding.weight"] = (V, E)
    for l in range(L):
        s[f"encoder.lstm_layer_{l}.Wx"] = (4 * H, E if l == 0 else H)
        s[f"encoder.lstm_layer_{l}.Wh"] = (4 * H, H)
        s[f"encoder.lstm_layer_{l}.bias"] = (4 * H,)
    s["encoder.condition_fc.weight"] = (H, C)
    s["encoder.condition_fc.bias"] = (H,)
    s["encoder.fc_mu.weight"] = (Z, 2 * H)
    s["encoder.fc_mu.bias"] = (Z,)
    s["encoder.fc_logvar_hidden.weight"] = (2 * H, 2 * H)
    s["encoder.fc_logvar_hidden.bias"] = (2 * H,)
    s["encoder.fc_logvar.weight"] = (Z, 2 * H)
    s["encoder.fc_logvar.bias"] = (Z,)
    s["decoder.z_to_hidden.weight"] = (H, Z)
    s["decoder.z_to_hidden.bias"] = (H,)
    s["decoder.condition_to_hidden.weight"] = (H, C)
    s["decoder.condition_to_hidden.bias"] = (H,)
    s["decoder.embedding.weight"] = (V, E)
    for l in range(L):
        s[f"decoder.lstm_layer_{l}.Wx"] = (4 * H, E + C if l == 0 else H)
        s[f"decoder.lstm_layer_{l}.Wh"] = (4 * H, H)
        s[f"decoder.lstm_layer_{l}.bias"] = (4 * H,)
    s["decoder.fc_out.weight"] = (V, H)
    s["decoder.fc_out.bias"] = (V,)
    return s


def init_params(cfg: Config, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Random-init weights with the MLX layer distributions (M1-M3), drawn from a
    legacy ``np.random.RandomState`` so fixtures are stable across NumPy versions.

    M1 nn.LSTM: Wx, Wh, bias ~ U(-1/sqrt(H), 1/sqrt(H)).
    M2 nn.Linear(in,out): weight, bias ~ U(-1/sqrt(in), 1/sqrt(in)).
    M3 nn.Embedding(n,d): weight ~ N(0, 1/d).
    encoder.fc_logvar.bias := 0.35 (reference models/encoder.py:71-74).
    """
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shp in param_shapes(cfg).items():
        leaf = name.split(".")[-1]
        mod = name.split(".")[-2]
        if mod == "embedding":
            w = rs.standard_normal(shp) * np.sqrt(1.0 / shp[1])
        elif mod.startswith("lstm_layer_"):
            k = 1.0 / np.sqrt(cfg.H)
            w = rs.uniform(-k, k, size=shp)
        else:  # Linear
            fan_in = param_shapes(cfg)[name.rsplit(".", 1)[0] + ".weight"][1]
            k = 1.0 / np.sqrt(fan_in)
            w = rs.uniform(-k, k, size=shp)
        out[name] = w.astype(np.float32)
    out["encoder.fc_logvar.bias"] = np.full((cfg.Z,), 0.35, dtype=np.float32)
    return out


def synthetic_batch(cfg: Config, B: int, T: int, seed: int = 67):
    """SURVEY.md section 8(d) synthetic SELFIES-shaped batch.

    tokens: start/pad=0 (decoder.py:146, dataloader.py:18), EOS=2 (decoder.py:26),
    body tokens in [3,V).  conditions ~ N(0,1) (z-scored TPSA stand-in,
    dataloader.py:63-65).
    """
    rs = np.random.RandomState(seed)
    lo = min(20, max(1, T // 2))
    hi = max(lo + 1, T - 1)
    lengths = rs.randint(lo, hi, size=B)
    x = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        n = min(int(lengths[b]), T - 1)          # (T == 1: the row is just the EOS token)
        x[b, :n] = rs.randint(3, cfg.V, size=n)
        x[b, n] = 2
    cond = rs.standard_normal((B, cfg.C)).astype(np.float32)
    return x, cond


def draw_coins(rs, T: int, ratio: float) -> np.ndarray:
    """Teacher-forcing coins in the reference's stream order (decoder.py:180):
    one ``rand()`` per timestep, drawn even when ratio == 0.0.  ``rs`` is
    ``np.random`` (global legacy stream, as the reference) or a RandomState."""
    return np.array([rs.rand() < ratio for _ in range(T)], dtype=bool)


# --------------------------------------------------------------------------- #
# MLX semantics (M1-M8) -- all assumptions about the
# third-party runtime live in this block.
# --------------------------------------------------------------------------- #
def mlx_maximum(a: Tensor, b) -> Tensor:
    """M5: mx.maximum VJP sends the cotangent to ``a`` where a > b, else to ``b``."""
    b = torch.as_tensor(b, dtype=a.dtype)
    return torch.where(a > b, a, b)


def mlx_minimum(a: Tensor, b) -> Tensor:
    b = torch.as_tensor(b, dtype=a.dtype)
    return torch.where(a < b, a, b)


def mlx_clip(a: Tensor, lo: float, hi: float) -> Tensor:
    """mx.clip = minimum(maximum(a, lo), hi)."""
    return mlx_minimum(mlx_maximum(a, lo), hi)


def mlx_argmax_lastdim(x: Tensor) -> Tensor:
    """M6: first maximal index."""
    m = x.max(dim=-1, keepdim=True).values
    return (x == m).to(torch.int64).argmax(dim=-1)  # argmax of 0/1 -> first 1


def mlx_linear(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """M2: y = addmm(bias, x, weight^T); weight is [out, in]."""
    return torch.addmm(b, x, w.t())


def mlx_lstm(x: Tensor, Wx: Tensor, Wh: Tensor, bias: Tensor,
             hidden: Optional[Tensor] = None, cell: Optional[Tensor] = None):
    """M1: MLX nn.LSTM.__call__(x, hidden=None, cell=None), x is [B, T, in].

    Gate order i, f, g, o.  With hidden None the h.Wh^T term is skipped, with
    cell None the cell is i*g.  Returns (all_hidden [B,T,H], all_cell [B,T,H]).
    """
    B, T, _ = x.shape
    xw = torch.addmm(bias, x.reshape(B * T, -1), Wx.t()).reshape(B, T, -1)
    all_h: List[Tensor] = []
    all_c: List[Tensor] = []
    for t in range(T):
        ifgo = xw[:, t, :]
        if hidden is not None:
            ifgo = ifgo + hidden @ Wh.t()
        i, f, g, o = torch.split(ifgo, ifgo.shape[-1] // 4, dim=-1)
        i = torch.sigmoid(i)
        f = torch.sigmoid(f)
        g = torch.tanh(g)
        o = torch.sigmoid(o)
        cell = f * cell + i * g if cell is not None else i * g
        hidden = o * torch.tanh(cell)
        all_h.append(hidden)
        all_c.append(cell)
    return torch.stack(all_h, dim=-2), torch.stack(all_c, dim=-2)


def mlx_softmax_lastdim(x: Tensor) -> Tensor:
    m = x.max(dim=-1, keepdim=True).values
    e = torch.exp(x - m)
    return e / e.sum(dim=-1, keepdim=True)


# --------------------------------------------------------------------------- #
# model forward (reference models/encoder.py, models/decoder.py, models/vae.py)
# --------------------------------------------------------------------------- #
def _sub(p: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    n = len(prefix) + 1
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix + ".")}


def encoder_forward(pe: Dict[str, Tensor], x: Tensor, cond: Tensor, L: int):
    """models/encoder.py:76-132.  ``pe`` holds encoder params without prefix."""
    out = pe["embedding.weight"][x]                                   # :93
    for l in range(L):                                                 # :98-101
        out, _ = mlx_lstm(out, pe[f"lstm_layer_{l}.Wx"], pe[f"lstm_layer_{l}.Wh"],
                          pe[f"lstm_layer_{l}.bias"])
    final_hidden = out[:, -1, :]                                       # :106 (last padded position, Q3)
    cr = mlx_linear(cond, pe["condition_fc.weight"], pe["condition_fc.bias"])  # :109
    comb = torch.cat([final_hidden, cr], dim=1)                        # :112
    mu_raw = mlx_linear(comb, pe["fc_mu.weight"], pe["fc_mu.bias"])    # :115
    lh = torch.tanh(mlx_linear(comb, pe["fc_logvar_hidden.weight"], pe["fc_logvar_hidden.bias"]))  # :117
    lv_raw = mlx_linear(lh, pe["fc_logvar.weight"], pe["fc_logvar.bias"])  # :118
    mu = torch.tanh(mu_raw / 2.0) * 2.0                                # :126
    logvar = torch.tanh(lv_raw / 2.0) * 1.0 - 1.0                      # :130
    return mu, logvar


def reparameterize(mu: Tensor, logvar: Tensor, eps: Tensor) -> Tensor:
    """models/encoder.py:147-153 with eps injected (Q18)."""
    std = torch.exp(0.5 * logvar)
    return mu + eps * std


def decoder_forward(pd: Dict[str, Tensor], z: Tensor, cond: Tensor, L: int,
                    target_seq: Optional[Tensor], coins: Optional[Sequence[bool]],
                    max_length: int = 80):
    """models/decoder.py:113-190, literal per-timestep loop.

    Q1: every LSTM call has hidden=None/cell=None on a length-1 sequence, so no
    state is carried.  Q2: ``initialize_hidden_state(z, cond)`` is computed and
    dropped; z never reaches the logits.  ``coins[t]`` is the outcome of
    ``np.random.rand() < teacher_forcing_ratio`` at step t (Q5).
    Returns (logits [B,T,V], input tokens actually fed [B,T]).
    """
    B = z.shape[0]
    T = target_seq.shape[1] if target_seq is not None else max_length
    cur = torch.zeros(B, dtype=torch.int64)                            # :146 start token 0
    logits_list, fed = [], []
    for t in range(T):
        fed.append(cur)
        emb = pd["embedding.weight"][cur]                              # :154
        inp = torch.cat([emb, cond], dim=1)[:, None, :]                # :157-160
        out = inp
        for l in range(L):                                             # :165-168
            out, _ = mlx_lstm(out, pd[f"lstm_layer_{l}.Wx"], pd[f"lstm_layer_{l}.Wh"],
                              pd[f"lstm_layer_{l}.bias"])
        h = out[:, 0, :]
        logits = mlx_linear(h, pd["fc_out.weight"], pd["fc_out.bias"])  # :175
        logits_list.append(logits)
        if target_seq is not None and bool(coins[t]):                  # :180
            cur = target_seq[:, t]
        else:
            cur = mlx_argmax_lastdim(logits.detach())                  # :185 (non-differentiable)
    return torch.stack(logits_list, dim=1), torch.stack(fed, dim=1)


def initialize_hidden_state(pd, z, cond, L):
    """models/decoder.py:76-111 (dead on the loss path, kept for the API)."""
    hz = mlx_linear(z, pd["z_to_hidden.weight"], pd["z_to_hidden.bias"])
    hc = mlx_linear(cond, pd["condition_to_hidden.weight"], pd["condition_to_hidden.bias"])
    h0 = (hz + hc) / 2.0
    hidden = h0[None].repeat(L, 1, 1)
    return hidden, torch.zeros_like(hidden)


# --------------------------------------------------------------------------- #
# losses (reference losses/recon.py, losses/kl.py, losses/info.py)
# --------------------------------------------------------------------------- #
def reconstruction_loss(logits: Tensor, targets: Tensor, reduction: str = "mean") -> Tensor:
    """losses/recon.py:29-64.  Mean over ALL B*T positions, pads included (Q3)."""
    V = logits.shape[-1]
    lf = logits.reshape(-1, V)
    tf = targets.reshape(-1)
    m = lf.max(dim=1, keepdim=True).values
    st = lf - m
    lsm = st - torch.log(torch.exp(st).sum(dim=1, keepdim=True))
    ce = -lsm.gather(1, tf[:, None]).reshape(-1)
    if reduction == "mean":
        return ce.mean()
    if reduction == "sum":
        return ce.sum()
    return ce


def kl_divergence(mu: Tensor, logvar: Tensor, reduction: str = "mean", free_bits: float = 0.0) -> Tensor:
    """losses/kl.py:35-66."""
    Z = mu.shape[1]
    mu = mlx_clip(mu, -3.0, 3.0)
    logvar = mlx_clip(logvar, -6.0, 3.0)
    var = torch.exp(logvar)
    k = -0.5 * (1.0 + logvar - mu * mu - var)
    k = mlx_maximum(k, 0.0)
    if free_bits > 0.0:
        k = mlx_maximum(k, free_bits / Z)
    per = k.sum(dim=1)
    if reduction == "mean":
        return per.mean()
    if reduction == "sum":
        return per.sum()
    return per


def mutual_information(mu: Tensor, logvar: Tensor, log_eps: float = 0.0) -> Tensor:
    """losses/info.py:23-50 (log_eps=0) and trainer.py:549-575 (log_eps=1e-8, Q20)."""
    mu = mlx_clip(mu, -3.0, 3.0)
    logvar = mlx_clip(logvar, -6.0, 3.0)
    var = torch.exp(logvar)
    kls = -0.5 * (1.0 + logvar - mu * mu - var).sum(dim=1)
    mean_kl = kls.mean()
    mean_mu = mu.mean(dim=0)
    mean_var = var.mean(dim=0)
    mean_logvar = torch.log(mean_var + log_eps) if log_eps else torch.log(mean_var)
    agg = -0.5 * (1.0 + mean_logvar - mean_mu * mean_mu - mean_var).sum()
    return mlx_maximum(mean_kl - agg, 0.0)


def posterior_collapse(mu: Tensor, logvar: Tensor, target_mi: float = 4.85, weight: float = 0.1) -> Tensor:
    """losses/info.py:73-78:  weight * maximum(0.0, target - mi)."""
    mi = mutual_information(mu, logvar)
    d = target_mi - mi
    return weight * torch.where(torch.zeros_like(d) > d, torch.zeros_like(d), d)


def complete_vae_loss(p: Dict[str, Tensor], cfg: Config, x: Tensor, cond: Tensor, eps: Tensor,
                      coins: Sequence[bool], beta: float = 0.4, lambda_prop: float = 0.1,
                      lambda_collapse: float = 0.01, free_bits: float = 0.5,
                      lambda_mi: float = 0.0, target_mi: float = 4.85) -> Dict[str, Tensor]:
    """complete_vae_loss.py:37-99 (property_predictor=None, so prop_loss == 0; Q10)."""
    pe, pd = _sub(p, "encoder"), _sub(p, "decoder")
    mu, logvar = encoder_forward(pe, x, cond, cfg.L)
    z = reparameterize(mu, logvar, eps)
    logits, fed = decoder_forward(pd, z, cond, cfg.L, x, coins)
    recon = reconstruction_loss(logits, x)
    kl = kl_divergence(mu, logvar, free_bits=free_bits)
    collapse = posterior_collapse(mu, logvar, weight=lambda_collapse)
    mi = mutual_information(mu, logvar)
    d = target_mi - mi
    mi_pen = lambda_mi * torch.where(torch.zeros_like(d) > d, torch.zeros_like(d), d)
    prop = torch.zeros((), dtype=mu.dtype)
    total = recon + beta * kl + collapse + lambda_prop * prop + mi_pen
    return {
        "total_loss": total, "recon_loss": recon, "kl_loss": kl, "weighted_kl": beta * kl,
        "collapse_penalty": collapse, "prop_loss": prop, "weighted_prop_loss": lambda_prop * prop,
        "mutual_info": mi, "mi_penalty": mi_pen, "mu": mu, "logvar": logvar, "z": z,
        "logits": logits, "fed_tokens": fed,
    }


# --------------------------------------------------------------------------- #
# training step (reference trainer.py:267-333)
# --------------------------------------------------------------------------- #
def to_torch(params: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=False) -> Dict[str, Tensor]:
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}


def loss_and_grads(params: Dict[str, np.ndarray], cfg: Config, x: np.ndarray, cond: np.ndarray,
                   eps: np.ndarray, coins: Sequence[bool], dtype=torch.float32, **hyper):
    """mx.value_and_grad(model_loss_fn, argnums=[0,1]) (trainer.py:292,305).
    Unused (dead) parameters get zero gradients (M8)."""
    p = to_torch(params, dtype, requires_grad=True)
    out = complete_vae_loss(p, cfg, torch.as_tensor(x, dtype=torch.int64),
                            torch.as_tensor(cond, dtype=dtype), torch.as_tensor(eps, dtype=dtype),
                            coins, **hyper)
    out["total_loss"].backward()
    grads = {k: (v.grad.detach().numpy().copy() if v.grad is not None
                 else np.zeros(tuple(v.shape), dtype=v.detach().numpy().dtype))
             for k, v in p.items()}
    vals = {k: v.detach().numpy().copy() for k, v in out.items()}
    return vals, grads


def adam_update(params, grads, m, v, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """M4 / Q7: MLX optim.Adam without bias correction, evaluated in the arrays' dtype:
    m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g^2;  p = p - lr*m/(sqrt(v)+eps).
    Updates the dicts in place (numpy arrays)."""
    for k in params:
        dt = params[k].dtype.type
        g = grads[k].astype(params[k].dtype)
        m[k] = dt(b1) * m[k] + dt(1 - b1) * g
        v[k] = dt(b2) * v[k] + dt(1 - b2) * np.square(g)
        params[k] = params[k] - dt(lr) * m[k] / (np.sqrt(v[k]) + dt(eps))


def train_step(params, m, v, cfg, x, cond, eps, coins, lr, dtype=torch.float32, **hyper):
    """One optimizer step; clipping is a no-op in the reference (Q6)."""
    vals, grads = loss_and_grads(params, cfg, x, cond, eps, coins, dtype=dtype, **hyper)
    adam_update(params, grads, m, v, lr)
    return vals, grads


# --------------------------------------------------------------------------- #
# sampling (reference models/decoder_sampling.py:48-128, Q9)
# --------------------------------------------------------------------------- #
def generate_with_temperature(pd: Dict[str, Tensor], cond: Tensor, L: int, max_length: int = 80,
                              temperature: float = 1.0, early_stopping: bool = True,
                              end_token: int = 2) -> Tensor:
    B = cond.shape[0]
    cur = torch.zeros(B, dtype=torch.int64)
    ended = torch.zeros(B, dtype=torch.bool)
    toks = []
    for _ in range(max_length):
        if early_stopping and bool(ended.all()):
            break
        emb = pd["embedding.weight"][cur]
        out = torch.cat([emb, cond], dim=1)[:, None, :]
        for l in range(L):
            out, _ = mlx_lstm(out, pd[f"lstm_layer_{l}.Wx"], pd[f"lstm_layer_{l}.Wh"],
                              pd[f"lstm_layer_{l}.bias"])
        logits = mlx_linear(out[:, 0, :], pd["fc_out.weight"], pd["fc_out.bias"])
        probs = mlx_softmax_lastdim(logits / temperature)
        cur = mlx_argmax_lastdim(probs)
        toks.append(cur)
        ended = ended | (cur == end_token)
    return torch.stack(toks, dim=1)


# --------------------------------------------------------------------------- #
# schedules (reference trainer.py:102-114)
# --------------------------------------------------------------------------- #
def compute_beta(epoch: int, beta_start: float, beta_end: float, warmup: int) -> float:
    if epoch < warmup:
        return float(beta_start + (beta_end - beta_start) * (epoch / warmup))
    return float(beta_end)


def compute_teacher_forcing_ratio(epoch: int, total_epochs: int) -> float:
    return float(max(0.5, 0.9 - 0.4 * (epoch / total_epochs)))
