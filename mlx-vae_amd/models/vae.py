"""ARCVAE on MI355X: encoder + decoder + separate sampling decoder (reference models/vae.py:8-131)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from .decoder import MLXAutoregressiveDecoder
from .decoder_sampling import MLXAutoregressiveDecoderSampling
from .encoder import MLXEncoder


class ARCVAE:
    def __init__(self, vocab_size: int, embedding_dim: int = 256, hidden_dim: int = 512, latent_dim: int = 200,
                 num_conditions: int = 6, num_layers: int = 3, dropout: float = 0.2, device=None, generator=None):
        kw = dict(vocab_size=vocab_size, embedding_dim=embedding_dim, hidden_dim=hidden_dim, latent_dim=latent_dim,
                  num_conditions=num_conditions, num_layers=num_layers, device=device, generator=generator)
        self.encoder = MLXEncoder(dropout=dropout, **kw)
        self.decoder = MLXAutoregressiveDecoder(**kw)
        self.decoder_sampling = MLXAutoregressiveDecoderSampling(**kw)  # own weights (Q9)
        self.latent_dim = latent_dim

    def __call__(self, x, conditions, target_seq=None, teacher_forcing_ratio: float = 0.5,
                 eps: Optional[torch.Tensor] = None, coins: Optional[Sequence[bool]] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(recon_logits [B,T,V], mu, logvar, z)  (models/vae.py:63-99).  With target_seq=None the decoder
        free-runs for its default max_length=80 steps, as in the reference."""
        mu, logvar = self.encoder(x, conditions)
        z = self.encoder.reparameterize(mu, logvar, eps=eps)
        logits = self.decoder(z, conditions, target_seq=target_seq, teacher_forcing_ratio=teacher_forcing_ratio,
                              coins=coins)
        return logits, mu, logvar, z

    def generate(self, batch_size: int, conditions, max_length: int = 80, temperature: float = 1.0, *, sample: bool = False,
                 seed: int = 0) -> torch.Tensor:
        """models/vae.py:101-131: z ~ N(0,I) (unused downstream, Q2) -> greedy sampler.  sample / seed (keyword-only extension):
        true categorical sampling instead of the reference's argmax, see MLXAutoregressiveDecoderSampling."""
        dev = self.decoder_sampling.decoder.store.device
        z = torch.randn(batch_size, self.latent_dim, device=dev)
        return self.decoder_sampling.generate_with_temperature(z, conditions, max_length=max_length,
                                                               temperature=temperature, sample=sample, seed=seed)

    def parameters(self):
        return {"encoder": self.encoder.parameters(), "decoder": self.decoder.parameters(),
                "decoder_sampling": self.decoder_sampling.parameters()}
