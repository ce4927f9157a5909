"""Minimal module base for the reference-named model classes.

The reference's modules are MLX `nn.Module`s whose parameters are nested dicts
(`module.parameters()['fc_mu']['weight']`).  Here a module owns one
flat device buffer (store.ParamStore) and exposes the same names as views.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .engine import ModelDims, Workspace
from .store import ParamStore


def resolve_device(device=None) -> torch.device:
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise _lib.ArcvaeHipError("arcvae_hip needs an MI355X (HIP) device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class HipModule:
    """Holds `store` (ParamStore) and `dims` (ModelDims); sub-modules appear as attributes."""

    store: ParamStore
    dims: ModelDims

    def _bind_views(self) -> None:
        mods: Dict[str, Dict[str, torch.Tensor]] = {}
        for name in self.store.names():
            mod, leaf = name.rsplit(".", 1)
            mods.setdefault(mod, {})[leaf] = self.store.p(name)
        for mod, leaves in mods.items():
            setattr(self, mod, SimpleNamespace(**leaves))
        self._ws: Dict = {}

    # MLX-style trees -------------------------------------------------------------------------
    def parameters(self) -> Dict[str, Dict[str, torch.Tensor]]:
        return self.store.tree("flat")

    def gradients(self) -> Dict[str, Dict[str, torch.Tensor]]:
        return self.store.tree("grad")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.state_dict()

    def load_state_dict(self, sd, prefix: str = "") -> None:
        self.store.load_state_dict(sd, prefix)

    def workspace(self, B: int, T: int, train: bool = False) -> Workspace:
        key = (B, T, train)
        if key not in self._ws:
            self._ws[key] = Workspace(self.dims, B, T, self.store.device, train)
        return self._ws[key]


def as_tokens(x, device) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=device, dtype=torch.int32).contiguous()


def as_f32(x, device) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=device, dtype=torch.float32).contiguous()
