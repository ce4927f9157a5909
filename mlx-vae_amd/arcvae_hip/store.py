"""Flat, 256-byte-aligned float32 parameter storage with named views.

One flat buffer per module (encoder / decoder) mirrors the reference's two parameter trees
(models/encoder.py:46-69, models/decoder.py:51-73) and its two Adam optimizers
(trainer.py:75-76): a module's gradients, Adam m and v are flat buffers with the same
offsets, so the optimizer is one HBM-bound launch per module and a data-parallel gradient
all-reduce is one bucket per module.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

ALIGN = 64  # floats (256 B): keeps every tensor 16-byte aligned for dwordx4 access


def encoder_shapes(V: int, E: int, H: int, Z: int, C: int, L: int) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["embedding.weight"] = (V, E)
    for l in range(L):
        s[f"lstm_layer_{l}.Wx"] = (4 * H, E if l == 0 else H)
        s[f"lstm_layer_{l}.Wh"] = (4 * H, H)
        s[f"lstm_layer_{l}.bias"] = (4 * H,)
    s["condition_fc.weight"] = (H, C)
    s["condition_fc.bias"] = (H,)
    s["fc_mu.weight"] = (Z, 2 * H)
    s["fc_mu.bias"] = (Z,)
    s["fc_logvar_hidden.weight"] = (2 * H, 2 * H)
    s["fc_logvar_hidden.bias"] = (2 * H,)
    s["fc_logvar.weight"] = (Z, 2 * H)
    s["fc_logvar.bias"] = (Z,)
    return s


def decoder_shapes(V: int, E: int, H: int, Z: int, C: int, L: int) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["z_to_hidden.weight"] = (H, Z)
    s["z_to_hidden.bias"] = (H,)
    s["condition_to_hidden.weight"] = (H, C)
    s["condition_to_hidden.bias"] = (H,)
    s["embedding.weight"] = (V, E)
    for l in range(L):
        s[f"lstm_layer_{l}.Wx"] = (4 * H, E + C if l == 0 else H)
        s[f"lstm_layer_{l}.Wh"] = (4 * H, H)
        s[f"lstm_layer_{l}.bias"] = (4 * H,)
    s["fc_out.weight"] = (V, H)
    s["fc_out.bias"] = (V,)
    return s


class ParamStore:
    """Named views into one flat device buffer (plus same-layout grad / Adam-state buffers)."""

    def __init__(self, shapes: "OrderedDict[str, Tuple[int, ...]]", device):
        self.shapes = shapes
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shp in shapes.items():
            self.offsets[name] = off
            n = int(np.prod(shp))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel_padded = off
        self.numel = sum(int(np.prod(s)) for s in shapes.values())
        self.device = torch.device(device)
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)

    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        shp = self.shapes[name]
        o = self.offsets[name]
        return buf[o:o + int(np.prod(shp))].view(*shp)

    def p(self, name: str) -> torch.Tensor:
        return self._view(self.flat, name)

    def g(self, name: str) -> torch.Tensor:
        return self._view(self.grad, name)

    def names(self) -> Iterable[str]:
        return self.shapes.keys()

    # ---- MLX-style initialisation (M1-M3): only used when no weights are injected ----------
    def init_mlx_like(self, H: int, generator: torch.Generator) -> None:
        for name, shp in self.shapes.items():
            mod = name.split(".")[-2]
            t = self.p(name)
            if mod == "embedding":
                w = torch.randn(shp, generator=generator) * math.sqrt(1.0 / shp[1])
            elif mod.startswith("lstm_layer_"):
                k = 1.0 / math.sqrt(H)
                w = (torch.rand(shp, generator=generator) * 2 - 1) * k
            else:
                fan_in = self.shapes[name.rsplit(".", 1)[0] + ".weight"][1]
                k = 1.0 / math.sqrt(fan_in)
                w = (torch.rand(shp, generator=generator) * 2 - 1) * k
            t.copy_(w.to(torch.float32))

    # ---- state dict ------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: self.p(n).detach().clone() for n in self.shapes}

    def load_state_dict(self, sd, prefix: str = "") -> None:
        for n in self.shapes:
            v = sd[prefix + n]
            v = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v)
            if tuple(v.shape) != tuple(self.shapes[n]):
                raise ValueError(f"shape mismatch for {prefix + n}: {tuple(v.shape)} vs {self.shapes[n]}")
            self.p(n).copy_(v.to(torch.float32))

    def tree(self, buf: str = "flat") -> Dict[str, Dict[str, torch.Tensor]]:
        """Nested dict like an MLX module's .parameters() (M8)."""
        src = getattr(self, buf)
        out: Dict[str, Dict[str, torch.Tensor]] = {}
        for n in self.shapes:
            mod, leaf = n.rsplit(".", 1)
            out.setdefault(mod, {})[leaf] = self._view(src, n)
        return out
