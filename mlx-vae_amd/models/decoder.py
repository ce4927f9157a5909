"""MLXAutoregressiveDecoder on MI355X (reference models/decoder.py:7-190).

Same constructor / `initialize_hidden_state` / `__call__` surface and parameter names
(models/decoder.py:51-73).  The reference's step loop never carries LSTM state and never feeds z
to the LSTM, so logits_t = F(token_t, conditions_b): the kernels evaluate F for all
B*V (row, token) pairs at once and turn the teacher-forcing / argmax-feedback loop into a table
walk (arcvae_hip/csrc/decoder.hip)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from arcvae_hip import engine as E
from arcvae_hip import _lib
from arcvae_hip._lib import call, ptr, stream_ptr
from arcvae_hip.module import HipModule, as_f32, as_tokens, resolve_device
from arcvae_hip.store import ParamStore, decoder_shapes


def draw_coins(T: int, ratio: float, has_target: bool = True) -> np.ndarray:
    """One `np.random.rand() < ratio` per timestep from NumPy's global legacy stream, in the
    reference's order (models/decoder.py:180); drawn even when ratio == 0.0, not drawn without a target."""
    if not has_target:
        return np.zeros(T, dtype=np.uint8)
    return np.array([np.random.rand() < ratio for _ in range(T)], dtype=np.uint8)


class MLXAutoregressiveDecoder(HipModule):
    def __init__(self, vocab_size: int, embedding_dim: int = 256, hidden_dim: int = 512, latent_dim: int = 200,
                 num_conditions: int = 6, num_layers: int = 3, pad_token: int = 0, end_token: int = 2, device=None,
                 generator: Optional[torch.Generator] = None):
        self.vocab_size, self.embedding_dim, self.hidden_dim = vocab_size, embedding_dim, hidden_dim
        self.latent_dim, self.num_conditions, self.num_layers = latent_dim, num_conditions, num_layers
        self.pad_token, self.end_token = pad_token, end_token
        self.dims = E.ModelDims(vocab_size, embedding_dim, hidden_dim, latent_dim, num_conditions, num_layers)
        self.dims.validate()
        self.store = ParamStore(decoder_shapes(vocab_size, embedding_dim, hidden_dim, latent_dim, num_conditions,
                                               num_layers), resolve_device(device))
        self.store.init_mlx_like(hidden_dim, generator or torch.Generator().manual_seed(torch.seed() % (2 ** 31)))
        self._bind_views()

    def initialize_hidden_state(self, z, conditions) -> Tuple[torch.Tensor, torch.Tensor]:
        """(hidden [L,B,H], cell [L,B,H] = 0), hidden = (z_to_hidden(z) + condition_to_hidden(c)) / 2 repeated L
        times (models/decoder.py:76-111).  Dead on the loss path (Q2); kept for the API."""
        dev = self.store.device
        z = as_f32(z, dev)
        c = as_f32(conditions, dev).reshape(z.shape[0], self.num_conditions)
        B, H, Z, Cc = z.shape[0], self.hidden_dim, self.latent_dim, self.num_conditions
        hz = torch.empty(B, H, device=dev)
        _lib.gemm(False, True, B, H, Z, z, Z, self.store.p("z_to_hidden.weight"), Z, hz, H,
                  self.store.p("z_to_hidden.bias"))
        _lib.gemm(False, True, B, H, Cc, c, Cc, self.store.p("condition_to_hidden.weight"), Cc, hz, H,
                  self.store.p("condition_to_hidden.bias"), flags=_lib.GEMM_ACCUMULATE)  # hz += cond.Wc^T + b
        call("arcvae_scale_inplace", ptr(hz), hz.numel(), 0.5, stream_ptr())
        hidden = hz.unsqueeze(0).repeat(self.num_layers, 1, 1).contiguous()
        return hidden, torch.zeros_like(hidden)

    def __call__(self, z, conditions, target_seq=None, max_length: int = 80, teacher_forcing_ratio: float = 0.5,
                 coins: Optional[Sequence[bool]] = None) -> torch.Tensor:
        """logits [B,T,V] (models/decoder.py:113-190).  `coins` (optional) injects the per-step
        teacher-forcing decisions; by default they are drawn from np.random exactly as the reference does."""
        dev = self.store.device
        cond = as_f32(conditions, dev)
        B = cond.shape[0]
        if target_seq is not None:
            tgt = as_tokens(target_seq, dev)
            T = tgt.shape[1]
        else:
            tgt, T = None, max_length
        ws = self.workspace(B, T)
        ws.cond.copy_(cond.reshape(B, self.num_conditions))
        if tgt is not None:
            ws.x.copy_(tgt)
        else:
            ws.x.zero_()
        if coins is None:
            coins = draw_coins(T, teacher_forcing_ratio, tgt is not None)
        ws.coins.copy_(torch.as_tensor(np.asarray(coins).astype(np.uint8)).to(dev))
        E.decoder_forward_dense(self.store, ws, self.dims)
        E.decoder_chain(ws, self.dims)
        out = torch.empty(B, T, self.vocab_size, dtype=torch.float32, device=dev)
        call("arcvae_dec_gather_logits", ptr(ws.logits), ptr(ws.fed), ptr(out), B, T, self.vocab_size, stream_ptr())
        return out
