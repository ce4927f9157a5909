"""HIP path against the committed golden fixtures (tests/golden/*.npz; nothing under oracle/ or
/root/reference is read here).  Tolerance 1e-4 relative (north_star), norm-wise and element-wise
(tests/helpers.py: |a-b| <= 1e-4*|b| + 1e-6*max|b| per element)."""
import os

import numpy as np
import pytest
import torch

import arcvae_oracle as O  # only for init_params / synthetic_batch (seeded input regeneration)
from helpers import DEFAULT, ELEM_ATOL_FWD, ELEM_ATOL_GRAD, HYPER, SMALL, TINY, assert_elem, build_engine, make_case, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _run(cfg, name):
    g = np.load(os.path.join(GOLD, name))
    B, T, tf = (int(v) for v in g["meta"])
    params, x, cond, eps, coins = make_case(cfg, B, T, tf / 1000.0)
    eng, enc, dec = build_engine(cfg, params)
    out = eng.train_step(x, cond, eps, coins, lr=2e-4, update=False, **HYPER)
    torch.cuda.synchronize()
    return g, eng, enc, dec, out, (params, x, cond, eps, coins, B, T)


@pytest.mark.parametrize("name,cfg", [("tiny_step.npz", TINY), ("small_step.npz", SMALL)])
def test_full_fixture(name, cfg):
    g, eng, enc, dec, out, (params, x, cond, eps, coins, B, T) = _run(cfg, name)
    for k in ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "mutual_info", "mi_penalty"):
        ref = float(g[f"val.{k}"])
        assert abs(float(out[k]) - ref) <= TOL * max(1.0, abs(ref)), k
    ws = eng.workspace(B, T)
    assert np.array_equal(ws.fed.cpu().numpy(), g["val.fed_tokens"])
    for k in ("mu", "logvar", "z"):
        assert rel_err(out[k].cpu().numpy(), g[f"val.{k}"]) < TOL, k
        assert_elem(out[k].cpu().numpy(), g[f"val.{k}"], k, ELEM_ATOL_FWD)
    assert rel_err(eng.gather_logits(ws).cpu().numpy(), g["val.logits"]) < TOL
    assert_elem(eng.gather_logits(ws).cpu().numpy(), g["val.logits"], "logits", ELEM_ATOL_FWD)
    for key in g.files:
        if not key.startswith("grad."):
            continue
        mod, pname = key[5:].split(".", 1)
        got = (enc if mod == "encoder" else dec).g(pname).cpu().numpy()
        ref = g[key]
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, key
        else:
            assert rel_err(got, ref) < TOL, key
            assert_elem(got, ref, key, ELEM_ATOL_GRAD)
    # one Adam step from zero state
    eng.train_step(x, cond, eps, coins, lr=2e-4, update=True, **HYPER)
    torch.cuda.synchronize()
    for key in g.files:
        if key.startswith("adam1."):
            mod, pname = key[6:].split(".", 1)
            got = (enc if mod == "encoder" else dec).p(pname).cpu().numpy()
            assert rel_err(got, g[key]) < 1e-5, key


def test_default_shape_digest():
    g, eng, enc, dec, out, (params, x, cond, eps, coins, B, T) = _run(DEFAULT, "default_digest.npz")
    for k in ("total_loss", "recon_loss", "kl_loss", "mutual_info"):
        ref = float(g[f"val.{k}"])
        assert abs(float(out[k]) - ref) <= TOL * max(1.0, abs(ref)), k
    ws = eng.workspace(B, T)
    assert np.array_equal(ws.fed.cpu().numpy(), g["val.fed_tokens"])
    assert rel_err(out["mu"].cpu().numpy(), g["val.mu"]) < TOL
    assert rel_err(out["logvar"].cpu().numpy(), g["val.logvar"]) < TOL
    logits = eng.gather_logits(ws).cpu().numpy().astype(np.float64)
    assert np.abs(logits.sum(-1) - g["val.logits_rowsum"]).max() <= TOL * float(g["val.logits_absmax"]) * logits.shape[-1] ** 0.5
    for key in g.files:
        if not key.startswith("grow."):
            continue
        mod, pname = key[5:].split(".", 1)
        got = (enc if mod == "encoder" else dec).g(pname).cpu().numpy().astype(np.float64)
        rows = got.reshape(got.shape[0], -1).sum(1)
        s = g["gsum." + key[5:]]
        if s[2] == 0:
            assert np.abs(got).max() == 0, key
            continue
        width = got.size // got.shape[0]
        assert np.abs(rows - g[key]).max() <= TOL * s[2] * max(1.0, width ** 0.5), key
        assert abs(np.abs(got).sum() - s[1]) <= TOL * s[1], key


# ---- full-size digests (round 4): the AUTO-selected kernel families at the sizes of BASELINE.json configs[2] and
# configs[3] against fp64 oracle digests (tests/golden/make_full_size_digests.py; a few CPU minutes each, generated once in the
# build container).  Before these, the tile-regime kernels met the oracle only at forced small shapes and at full size only each other.
CONFIGS2 = O.Config(vocab_size=80, embedding_dim=128, hidden_dim=512, latent_dim=256, num_conditions=1, num_layers=4)


def _check_digest(g, eng, enc, dec, out, B, T):
    for k in ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "mutual_info", "mi_penalty"):
        ref = float(g[f"val.{k}"])
        assert abs(float(out[k]) - ref) <= TOL * max(1.0, abs(ref)), (k, float(out[k]), ref)
    ws = eng.workspace(B, T)
    assert np.array_equal(ws.fed.cpu().numpy(), g["val.fed_tokens"].astype(np.int32)), "fed-back tokens differ"
    for k in ("mu", "logvar"):
        got = out[k].cpu().numpy().astype(np.float64)
        n = g[f"val.{k}"].shape[0]                                   # all rows up to 512, else the first 64 + every row's sum
        assert rel_err(got[:n], g[f"val.{k}"]) < TOL, k
        assert_elem(got[:n], g[f"val.{k}"], k, ELEM_ATOL_FWD)
        assert np.abs(got.sum(1) - g[f"val.{k}_rowsum"]).max() <= TOL * float(g[f"val.{k}_absmax"]) * got.shape[1] ** 0.5, k
    logits = eng.gather_logits(ws).cpu().numpy().astype(np.float64)
    assert np.abs(logits.sum(-1) - g["val.logits_rowsum"]).max() <= TOL * float(g["val.logits_absmax"]) * logits.shape[-1] ** 0.5
    for key in g.files:
        if not key.startswith("grow."):
            continue
        mod, pname = key[5:].split(".", 1)
        got = (enc if mod == "encoder" else dec).g(pname).cpu().numpy().astype(np.float64)
        rows = got.reshape(got.shape[0], -1).sum(1)
        s = g["gsum." + key[5:]]
        if s[2] == 0:
            assert np.abs(got).max() == 0, key                       # dead parameters (Q1/Q2): exactly zero
            continue
        width = got.size // got.shape[0]
        assert np.abs(rows - g[key]).max() <= TOL * s[2] * max(1.0, width ** 0.5), key     # per-row sums
        assert abs(np.abs(got).sum() - s[1]) <= TOL * s[1], key                           # abs-sum
        assert abs(np.abs(got).max() - s[2]) <= TOL * s[2], key                           # abs-max


def test_configs2_full_size_digest():
    """BASELINE.json configs[2] (H512 Z256 L4, bs 512, T 128) with whatever the engine selects at that size: three-piece forward
    tile, K-split BPTT tile, operand-plane weight gradients with the bias rider, dense decoder stack on the tile kernels."""
    g, eng, enc, dec, out, (params, x, cond, eps, coins, B, T) = _run(CONFIGS2, "configs2_digest.npz")
    assert (B, T) == (512, 128)
    eng.check_gates()
    ws = eng.workspace(B, T)
    assert ws.planes and ws.dense_ws is not None and ws.dense_fwd and ws.hseq_t.shape[1] == T, "the tile-regime families were not selected"
    _check_digest(g, eng, enc, dec, out, B, T)


@pytest.mark.parametrize("name,B", [("default_b256_digest.npz", 256), ("default_b2048_digest.npz", 2048)])
def test_default_model_large_batch_digests(name, B):
    """BASELINE.json configs[3] shapes of the default model: the 256-row shard one GPU of eight steps (persistent two-group forward
    sweep + whichever BPTT / weight-gradient family is the default at 129..256 rows) and the whole 2048-row global batch on one GPU
    (tile regime: planes, K-split tile, dense decoder stack)."""
    g, eng, enc, dec, out, (params, x, cond, eps, coins, B_, T) = _run(DEFAULT, name)
    assert (B_, T) == (B, 128)
    eng.check_gates()
    if B == 2048:
        ws = eng.workspace(B, T)
        assert ws.planes and ws.dense_ws is not None, "the tile-regime families were not selected at 2048 rows"
    _check_digest(g, eng, enc, dec, out, B, T)
