"""Throughput mode (StepEngine(precision="bf16"): SURVEY.md section 8(d) Config 2, "bf16-in/fp32-acc").

NOT a parity path: every matrix product off the latency-bound chain takes operands rounded to bf16 (8 significant bits)
with f32 accumulation.  What is asserted here is the mode's own stated tolerance against the fp64 oracle on the same
seeded inputs -- loss scalars within 2e-2 relative, mu / logvar within 5e-2 of their largest element, every parameter
gradient within 8e-2 in relative L2 norm and > 0.995 cosine -- and that the mode really ran (it differs from the fp32
engine by more than fp32 noise).  Teacher forcing is 1.0 in these cases: with greedy feedback a near-tie in the logits may
flip a fed-back token, which is a discrete change no tolerance describes."""
import numpy as np
import pytest
import torch

from helpers import HYPER, O, build_engine, make_case, rel_err

pytestmark = pytest.mark.gpu

LOSS_RTOL, ACT_TOL, GRAD_RTOL, GRAD_COS = 2e-2, 5e-2, 8e-2, 0.995


def _engine(cfg, params, precision):
    from arcvae_hip.engine import StepEngine
    eng, enc, dec = build_engine(cfg, params)
    if precision != "fp32":
        eng = StepEngine(enc, dec, eng.d, precision=precision)
    return eng, enc, dec


def _grads(enc, dec):
    out = {}
    for prefix, store in (("encoder.", enc), ("decoder.", dec)):
        for name in store.names():
            out[prefix + name] = store.g(name).detach().cpu().numpy().astype(np.float64)
    return out


CASES = [  # (cfg, B, T, env, forward sweep in bf16?)
    # tiled LSTM step kernels (forced by ARCVAE_STEP_TILE where the grid alone would not choose them)
    (O.Config(vocab_size=24, embedding_dim=32, hidden_dim=128, latent_dim=16, num_conditions=2, num_layers=2), 192, 7,
     {"ARCVAE_STEP_TILE": "2", "ARCVAE_PERSIST": "0"}, True),
    (O.Config(vocab_size=20, embedding_dim=16, hidden_dim=64, latent_dim=8, num_conditions=1, num_layers=3), 130, 5,
     {"ARCVAE_STEP_TILE": "4", "ARCVAE_PERSIST": "0"}, True),
    (O.Config(vocab_size=30, embedding_dim=32, hidden_dim=256, latent_dim=32, num_conditions=1, num_layers=1), 100, 6,
     {"ARCVAE_STEP_TILE": "1", "ARCVAE_PERSIST": "0"}, True),
    (O.Config(vocab_size=16, embedding_dim=16, hidden_dim=128, latent_dim=8, num_conditions=1, num_layers=4), 512, 4, {}, True),
    # persistent sweeps on the 4x4x4 bf16 blocks (H 256: forward up to 64 rows, reduce-scatter BPTT up to 128)
    (O.Config(vocab_size=30, embedding_dim=32, hidden_dim=256, latent_dim=32, num_conditions=1, num_layers=2), 64, 12, {}, True),
    (O.Config(vocab_size=30, embedding_dim=32, hidden_dim=256, latent_dim=32, num_conditions=1, num_layers=1), 37, 9, {}, True),
    (O.Config(vocab_size=30, embedding_dim=32, hidden_dim=256, latent_dim=16, num_conditions=1, num_layers=2), 100, 10, {}, False),
    (O.Config(), 64, 128, {}, True),     # BASELINE.json configs[1]
    # persistent FORWARD sweep + register-tiled bf16 BPTT (forced): the persistent forward writes the BPTT weight layouts
    # in the precision the backward will read them in (they were always f32 -- silently wrong gradients)
    (O.Config(vocab_size=24, embedding_dim=32, hidden_dim=128, latent_dim=16, num_conditions=1, num_layers=2), 64, 8,
     {"ARCVAE_STEP_TILE": "2"}, False),
    # the dense decoder's layers 1 .. L-1 on the bf16 tile kernels + octet weight-gradient kernel (forced; B*V a multiple of 32)
    (O.Config(vocab_size=24, embedding_dim=16, hidden_dim=128, latent_dim=8, num_conditions=2, num_layers=3), 64, 5,
     {"ARCVAE_STEP_TILE": "4", "ARCVAE_PERSIST": "0", "ARCVAE_DENSE_TILED": "2"}, True),
    (O.Config(vocab_size=20, embedding_dim=16, hidden_dim=64, latent_dim=8, num_conditions=1, num_layers=2), 96, 4,
     {"ARCVAE_STEP_TILE": "4", "ARCVAE_PERSIST": "0", "ARCVAE_DENSE_TILED": "2"}, True),
]


@pytest.mark.parametrize("cfg,B,T,env,fwd_bf16", CASES)
def test_bf16_mode_step_is_within_its_stated_tolerance(cfg, B, T, env, fwd_bf16, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    params, x, cond, eps, coins = make_case(cfg, B, T, tf_ratio=1.0)
    vals, grads = O.loss_and_grads(params, cfg, x, cond, eps, coins, dtype=torch.float64, **HYPER)
    res = {}
    for precision in ("fp32", "bf16"):
        eng, enc, dec = _engine(cfg, params, precision)
        assert eng.precision == precision
        out = eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
        torch.cuda.synchronize()
        assert float(out["step_status"]) == 0.0
        res[precision] = ({k: (v.detach().cpu().numpy().astype(np.float64) if v.dim() else float(v)) for k, v in out.items()},
                          _grads(enc, dec), eng.workspace(B, T).hseq.detach().cpu().numpy().astype(np.float64))
    out, got, hseq = res["bf16"]
    # the encoder's hidden states are produced by the sweeps alone (no GEMM upstream of them but the f32 token table): they
    # must carry bf16-sized differences, i.e. the bf16 step kernels ran, and stay within the mode's tolerance
    dh = rel_err(hseq, res["fp32"][2])
    assert (1e-5 if fwd_bf16 else -1.0) < dh < 3e-2, dh
    for k in ("total_loss", "recon_loss", "kl_loss", "mutual_info"):
        assert abs(out[k] - float(vals[k])) <= LOSS_RTOL * max(1.0, abs(float(vals[k]))), (k, out[k], float(vals[k]))
    for k in ("mu", "logvar"):
        ref = np.asarray(vals[k], dtype=np.float64)
        assert np.abs(out[k] - ref).max() <= ACT_TOL * max(np.abs(ref).max(), 1e-3), k
    worst = 0.0
    for name, g in grads.items():
        g = np.asarray(g, dtype=np.float64)
        a = got[name]
        if not np.any(g):
            assert not np.any(a), name            # dead parameters stay exactly zero in every mode
            continue
        err = rel_err(a, g)
        cos = float((a * g).sum() / (np.linalg.norm(a) * np.linalg.norm(g)))
        worst = max(worst, err)
        assert err < GRAD_RTOL and cos > GRAD_COS, (name, err, cos)
    # the mode really ran: it is not the fp32 engine's result
    diff = max(rel_err(got[n], res["fp32"][1][n]) for n in grads if np.any(grads[n]))
    assert diff > 1e-4, diff
    print(f"bf16 mode B={B} T={T} H={cfg.H} L={cfg.L}: worst gradient rel-L2 error {worst:.2e}; distance from the fp32 engine "
          f"{diff:.2e} (hidden states {dh:.2e})")


def test_precision_argument_is_validated():
    from arcvae_hip.engine import StepEngine
    cfg = O.Config(vocab_size=16, embedding_dim=16, hidden_dim=64, latent_dim=8, num_conditions=1, num_layers=1)
    eng, enc, dec = build_engine(cfg, O.init_params(cfg, 1))
    with pytest.raises(ValueError):
        StepEngine(enc, dec, eng.d, precision="fp8")
