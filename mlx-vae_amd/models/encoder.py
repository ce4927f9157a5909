"""MLXEncoder on MI355X: same constructor/`__call__`/`reparameterize` surface as the reference
(models/encoder.py:5-155), parameters named as in models/encoder.py:46-69, compute in the HIP
kernels of arcvae_hip (table-lookup layer-0 projection, wavefront LSTM sweep, fused heads)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from arcvae_hip import engine as E
from arcvae_hip._lib import call, ptr, stream_ptr
from arcvae_hip.module import HipModule, as_f32, as_tokens, resolve_device
from arcvae_hip.store import ParamStore, encoder_shapes


class MLXEncoder(HipModule):
    def __init__(self, vocab_size: int, embedding_dim: int = 256, hidden_dim: int = 512, latent_dim: int = 200,
                 num_conditions: int = 6, num_layers: int = 3, dropout: float = 0.2, device=None,
                 generator: Optional[torch.Generator] = None):
        self.vocab_size, self.embedding_dim, self.hidden_dim = vocab_size, embedding_dim, hidden_dim
        self.latent_dim, self.num_conditions, self.num_layers = latent_dim, num_conditions, num_layers
        # `dropout` is accepted and ignored, as in the reference (no dropout layer exists)
        self.dims = E.ModelDims(vocab_size, embedding_dim, hidden_dim, latent_dim, num_conditions, num_layers)
        self.dims.validate()
        self.store = ParamStore(encoder_shapes(vocab_size, embedding_dim, hidden_dim, latent_dim, num_conditions,
                                               num_layers), resolve_device(device))
        self.store.init_mlx_like(hidden_dim, generator or torch.Generator().manual_seed(torch.seed() % (2 ** 31)))
        self.store.p("fc_logvar.bias").fill_(0.35)  # models/encoder.py:71-74
        self._bind_views()

    def __call__(self, x, conditions) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [B,T] tokens, conditions [B,C] -> (mu [B,Z], logvar [B,Z])  (models/encoder.py:76-132)."""
        dev = self.store.device
        xt = as_tokens(x, dev)
        B, T = xt.shape
        ws = self.workspace(B, T)
        ws.x.copy_(xt)
        ws.cond.copy_(as_f32(conditions, dev).reshape(B, self.num_conditions))
        E.encoder_forward(self.store, ws, self.dims, 0.0)
        return ws.mu.clone(), ws.logvar.clone()

    @staticmethod
    def reparameterize(mu: torch.Tensor, logvar: torch.Tensor, eps: Optional[torch.Tensor] = None,
                       generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """z = mu + eps * exp(0.5 logvar)  (models/encoder.py:134-155).  The reference draws eps from MLX's
        unseeded global RNG (Q18); here eps may be injected, else it is drawn on the device."""
        mu = mu.contiguous()
        logvar = logvar.contiguous()
        if eps is None:
            eps = torch.randn(mu.shape, device=mu.device, dtype=torch.float32, generator=generator)
        eps = as_f32(eps, mu.device)
        z = torch.empty_like(mu)
        call("arcvae_reparameterize", ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(), stream_ptr())
        return z
