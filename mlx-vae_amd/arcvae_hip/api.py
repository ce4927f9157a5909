"""Glue between the reference-named modules and the step engine: engine cache, loss forward,
value_and_grad (the mx.value_and_grad of trainer.py:292) and RNG conventions (Q5/Q18)."""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import SCALAR_KEYS, StepEngine

_ENGINES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def engine_for(encoder, decoder) -> StepEngine:
    """One StepEngine (workspaces, side stream, captured graphs) per (encoder, decoder) pair."""
    per_enc = _ENGINES.setdefault(encoder, {})
    key = id(decoder)
    if key not in per_enc:
        if encoder.dims != decoder.dims:
            raise ValueError("encoder and decoder were built with different dimensions")
        per_enc[key] = StepEngine(encoder.store, decoder.store, encoder.dims)
    return per_enc[key]


def enable_data_parallel(encoder, decoder, group=None):
    """Route loss_forward / value_and_grad of this (encoder, decoder) pair through arcvae_hip.dp.EngineDataParallel: every
    call then takes the GLOBAL batch (identical on all ranks), works on this rank's rows and returns the global batch's
    loss scalars; mu / logvar / z in the result are the LOCAL rows'.  torch.distributed must be initialised (one process
    per GPU; backend nccl = RCCL).  Returns the driver (rank, world)."""
    from .dp import EngineDataParallel
    eng = engine_for(encoder, decoder)
    if getattr(eng, "dp", None) is None:
        eng.dp = EngineDataParallel(eng, group)
    return eng.dp


def data_parallel_of(encoder, decoder):
    """The pair's data-parallel driver (enable_data_parallel), or None: single process."""
    return getattr(engine_for(encoder, decoder), "dp", None)


def draw_coins(T: int, ratio: float) -> np.ndarray:
    """models/decoder.py:180: one np.random.rand() per timestep from the GLOBAL legacy stream, drawn even
    when ratio == 0.0 (validation)."""
    return np.array([np.random.rand() < ratio for _ in range(T)], dtype=np.uint8)


def draw_eps(B: int, Z: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The reference draws eps from MLX's unseeded global RNG (Q18): any N(0,1) draw is 'the same'."""
    return torch.randn(B, Z, device=device, dtype=torch.float32, generator=generator)


def _as_dict(eng: StepEngine, ws, clone: bool, status: bool = False) -> Dict[str, torch.Tensor]:
    sc = ws.scalars.clone()  # 16 floats: always detach from the static buffer the next call overwrites
    out = {k: sc[i] for i, k in enumerate(SCALAR_KEYS)}
    if status:   # [total_loss, step status]: ONE D2H read gives the trainer the loss and "stream order was lost"
        out["loss_and_status"] = sc[::15]     # elements 0 and 15 of the 16
    for k in ("mu", "logvar", "z"):
        t = getattr(ws, k)
        out[k] = t.clone() if clone else t
    return out


def loss_forward(encoder, decoder, x, conditions, eps=None, coins=None, teacher_forcing_ratio: float = 0.9,
                 **hyper) -> Dict[str, torch.Tensor]:
    eng = engine_for(encoder, decoder)
    B, T = int(x.shape[0]), int(x.shape[1])
    if coins is None:
        coins = draw_coins(T, teacher_forcing_ratio)
    if eps is None:
        eps = draw_eps(B, encoder.latent_dim, encoder.store.device)
    dp = getattr(eng, "dp", None)
    if dp is not None:       # N ranks: this rank's rows, one all-reduce of the partial sums, the global batch's scalars
        x, conditions = _dev_batch(eng, x, conditions)
        return _as_dict(eng, dp.forward_loss(x, conditions, _local_eps(dp, eps, B), coins, **hyper), clone=True)
    eng.forward_loss(x, conditions, eps, coins, **hyper)
    return _as_dict(eng, eng.workspace(B, T, train=False), clone=True)


def value_and_grad(encoder, decoder, x, conditions, eps=None, coins=None, teacher_forcing_ratio: float = 0.9,
                   lr: Optional[float] = None, **hyper):
    """(loss dict, (encoder grad tree, decoder grad tree)); with `lr` given the two Adam updates are applied
    in the same captured step (trainer.py:305-333)."""
    eng = engine_for(encoder, decoder)
    B, T = int(x.shape[0]), int(x.shape[1])
    if coins is None:
        coins = draw_coins(T, teacher_forcing_ratio)
    if eps is None:
        eps = draw_eps(B, encoder.latent_dim, encoder.store.device)
    dp = getattr(eng, "dp", None)
    if dp is not None:
        if lr is None:
            raise ValueError("the data-parallel step applies both Adam updates: pass lr")
        x, conditions = _dev_batch(eng, x, conditions)
        ws = dp.train_step(x, conditions, _local_eps(dp, eps, B), coins, lr, **hyper)
        return _as_dict(eng, ws, clone=False, status=True), (encoder.gradients(), decoder.gradients())
    eng.train_step(x, conditions, eps, coins, lr=lr if lr is not None else 0.0, update=lr is not None, **hyper)
    ws = eng.workspace(B, T, train=True)
    return _as_dict(eng, ws, clone=False, status=True), (encoder.gradients(), decoder.gradients())


def _dev_batch(eng: StepEngine, x, conditions):
    """Device-resident, contiguous global batch in the workspace's dtypes (what EngineDataParallel slices rows from)."""
    from .module import as_f32, as_tokens
    xt = as_tokens(x, eng.device)
    return xt, as_f32(conditions, eng.device).reshape(xt.shape[0], eng.d.C)


def _local_eps(dp, eps, B: int):
    """eps of the GLOBAL batch ([B, Z], sliced by the driver) -- an own draw per rank when none was injected."""
    if eps is None:
        return draw_eps(B, dp.eng.d.Z, dp.eng.device)
    from .module import as_f32
    return as_f32(eps, dp.eng.device)
