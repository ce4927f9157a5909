"""MLXAutoregressiveDecoderSampling on MI355X (reference models/decoder_sampling.py:6-129).

Owns its OWN, never-trained decoder (Q9: not weight-shared with ARCVAE.decoder); "temperature
sampling" is argmax(softmax(logits / T)) = greedy; tokens after EOS are still generated; with
early stopping the output is cut where every row has emitted EOS.  One dense decoder pass
(B*V rows) + one table walk replaces the reference's max_length dependent steps and its per-step
host sync; `load_from_decoder` is an explicit extension to sample from trained weights, and `sample=True` the true categorical
sampling the reference marks TODO (decoder_sampling.py:115-116): an extension with no reference behaviour to match."""
from __future__ import annotations

import torch

from arcvae_hip import engine as E
from arcvae_hip._lib import call, ptr, stream_ptr
from arcvae_hip.module import as_f32

from .decoder import MLXAutoregressiveDecoder


class MLXAutoregressiveDecoderSampling:
    def __init__(self, vocab_size: int, embedding_dim: int = 256, hidden_dim: int = 512, latent_dim: int = 200,
                 num_conditions: int = 6, num_layers: int = 3, pad_token: int = 0, end_token: int = 2, device=None,
                 generator=None):
        self.decoder = MLXAutoregressiveDecoder(vocab_size, embedding_dim, hidden_dim, latent_dim, num_conditions,
                                                num_layers, pad_token, end_token, device=device, generator=generator)
        self.vocab_size, self.embedding_dim, self.hidden_dim = vocab_size, embedding_dim, hidden_dim
        self.latent_dim, self.num_conditions = latent_dim, num_conditions
        self.pad_token, self.end_token = pad_token, end_token
        self._graphs = {}

    def load_from_decoder(self, other: MLXAutoregressiveDecoder) -> None:
        """Extension (absent in the reference): copy trained decoder weights into the sampler."""
        self.decoder.store.flat.copy_(other.store.flat)

    def parameters(self):
        return {"decoder": self.decoder.parameters()}

    def generate_with_temperature(self, z, conditions, max_length: int = 80, temperature: float = 1.0,
                                  early_stopping: bool = True, use_graph: bool = True, *, sample: bool = False,
                                  seed: int = 0) -> torch.Tensor:
        """[B, t_stop] int32 tokens (models/decoder_sampling.py:48-128).  z is accepted and unused (Q2).
        sample=True (keyword-only extension; the default reproduces the reference's greedy "temperature sampling", Q9): every token is
        DRAWN from softmax(logits / temperature) -- what decoder_sampling.py:115-116 leaves as a TODO -- with a counter-based
        generator keyed by (seed, row, step): the same seed returns the same molecules."""
        if sample:
            return self._generate_categorical(conditions, max_length, temperature, early_stopping, int(seed))
        dec = self.decoder
        dev = dec.store.device
        cond = as_f32(conditions, dev)
        B = cond.shape[0]
        ws = dec.workspace(B, max_length)
        ws.cond.copy_(cond.reshape(B, dec.num_conditions))
        key = (B, max_length, float(temperature))
        if key not in self._graphs:
            self._graphs[key] = dict(tokens=torch.zeros(B, max_length, dtype=torch.int32, device=dev),
                                     first_end=torch.zeros(B, dtype=torch.int32, device=dev), graph=None)
        st = self._graphs[key]

        def enqueue():
            E.decoder_forward_dense(dec.store, ws, dec.dims, mode=1, temperature=temperature, keep_gpre=False, alone=True)
            call("arcvae_dec_sample_chain", ptr(ws.nxt), ptr(st["tokens"]), ptr(st["first_end"]), B, dec.vocab_size,
                 max_length, dec.end_token, stream_ptr())

        if not use_graph:
            enqueue()
        elif st["graph"] is None:
            enqueue()  # first call runs eagerly, then the decode pass is captured for replay
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue()
            st["graph"] = g
        else:
            st["graph"].replay()
        tokens = st["tokens"]
        if early_stopping:
            # reference: the loop breaks at the first step t where every row has already ended, i.e. after
            # max_b(first EOS index) + 1 tokens; one host read replaces its per-step mx.all() sync
            t_stop = int(st["first_end"].max().item()) + 1
            tokens = tokens[:, :min(t_stop, max_length)]
        return tokens.clone()

    def _generate_categorical(self, conditions, max_length: int, temperature: float, early_stopping: bool, seed: int) -> torch.Tensor:
        """One dense decoder pass (raw logits of all B*V (row, token) pairs), then arcvae_dec_sample_chain_categorical: a wave per
        row walks start token -> sampled token -> ... through the table of distributions.  Eager launches (the seed is a launch
        argument; the greedy path's captured pass is untouched)."""
        import ctypes as C
        if not temperature > 0.0:
            raise ValueError("temperature must be > 0 for categorical sampling")
        dec = self.decoder
        dev = dec.store.device
        cond = as_f32(conditions, dev)
        B = cond.shape[0]
        ws = dec.workspace(B, max_length)
        ws.cond.copy_(cond.reshape(B, dec.num_conditions))
        tokens = torch.zeros(B, max_length, dtype=torch.int32, device=dev)
        first_end = torch.zeros(B, dtype=torch.int32, device=dev)
        E.decoder_forward_dense(dec.store, ws, dec.dims, mode=0, keep_gpre=False, alone=True)
        call("arcvae_dec_sample_chain_categorical", ptr(ws.logits), ptr(tokens), ptr(first_end), B, dec.vocab_size, max_length,
             dec.end_token, float(temperature), C.c_ulonglong(seed & 0xFFFFFFFFFFFFFFFF), stream_ptr())
        if early_stopping:
            tokens = tokens[:, :min(int(first_end.max().item()) + 1, max_length)]
        return tokens.clone()
