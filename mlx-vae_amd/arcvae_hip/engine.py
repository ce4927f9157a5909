"""Host-side driver of the HIP kernels: one training step of the AR-CVAE SELFIES path.

Reference path: trainer.py:303-333 -> complete_vae_loss.py:37-99 ->
models/encoder.py:76-155, models/decoder.py:113-190, losses/{recon,kl,info}.py, then two
MLX Adam updates.  Here the same step is a fixed sequence of launches on three HIP streams:

  main stream : tokens^T -> table0 GEMM -> LSTM wavefront sweep -> heads -> [stats seam]
                -> latent loss -> heads dcomb chain -> BPTT wavefront -> join -> Adam      (the dependent chain)
  side stream : dense decoder fwd (B*V rows) -> TF walk + CE -> dlogits -> dense decoder bwd -> its Adam;
                at the end the token-table half of the last weight-gradient chunk
  aux stream  : parameter-gradient GEMMs of the encoder, chunk by chunk behind the BPTT sweep

The decoder never reads z, so the streams only meet at the loss scalars.  Each
stream's launch sequence is captured once per (B,T) shape into linear hipGraph segments and
replayed; the streams are ordered among each other by device-side gates (class Gates), not by
events; inputs, teacher-forcing coins and the epoch-scheduled hyper-parameters live in static
device buffers.

Nothing here computes on the CPU and nothing falls back to PyTorch ops for the math: torch is
used for device memory, streams, graphs and (in dp.py) torch.distributed/RCCL.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, ptr_array, stream_ptr
from .store import ParamStore

SCALAR_KEYS = ("total_loss", "recon_loss", "kl_loss", "weighted_kl", "collapse_penalty", "prop_loss",
               "weighted_prop_loss", "mutual_info", "mi_penalty")


@dataclasses.dataclass(frozen=True)
class ModelDims:
    V: int
    E: int
    H: int
    Z: int
    C: int
    L: int

    def validate(self) -> None:
        if self.H % 64 != 0 or not (64 <= self.H <= 512):
            raise ValueError("hidden_dim must be a multiple of 64 in [64, 512] for the gfx950 LSTM kernels")
        if not (1 <= self.L <= 8):
            raise ValueError("num_layers must be in [1, 8]")
        if not (1 <= self.C <= 8):
            raise ValueError("num_conditions must be in [1, 8]")
        if not (2 <= self.V <= 255):
            # the decoder is evaluated over all B*V (row, token) pairs and its cross-entropy backward keeps a V x V count
            # histogram per batch row in LDS (16-bit counts from V = 128 on: 131 KB at 255) -- INTEGRATION.md section 5
            raise ValueError("vocab_size must be in [2, 255] (the vocabulary-dense decoder's per-row V x V histogram lives in LDS)")


class Workspace:
    """Static device buffers for one (B, T) shape."""

    def __init__(self, d: ModelDims, B: int, T: int, device, train: bool = True):
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        V, E, H, Z, Cc, L = d.V, d.E, d.H, d.Z, d.C, d.L
        G, BV = 4 * H, B * V
        self.B, self.T = B, T
        self.bf16 = False   # throughput mode (StepEngine(precision="bf16")): see StepEngine
        self.bf16_parts = 0
        self.planes = False  # the weight-gradient GEMMs read the sweeps' operand planes (StepEngine.workspace)
        # inputs
        self.x = torch.zeros(B, T, **i32)
        self.x_tb = torch.zeros(T, B, **i32)
        self.coins = torch.ones(T, dtype=torch.uint8, device=device)
        self.cond = torch.zeros(B, Cc, **f32)
        self._hyper_vals = None
        self.eps = torch.zeros(B, Z, **f32)
        self.hyper = torch.zeros(8, **f32)
        # encoder forward
        self.table0 = torch.empty(V, G, **f32)
        self.hseq = torch.empty(L, T, B, H, **f32)
        RS = int(os.environ.get("ARCVAE_RING", "16"))  # ring slots over t (csrc/common.h: arcvae_ring_slots)
        RS = T if (RS <= 0 or RS > T) else RS
        # (x 3/2: in their three-piece form the register-tiled sweeps keep these copies as three bf16 planes, 6 bytes per value)
        self.hseq_t = torch.empty(L, RS, B * H * 3 // 2, **f32)     # k-chunk-major copy of h_t in slot t % RS (next launch's operand)
        self.wt = torch.empty(2 * L - 1, G * H * 3 // 2, **f32)        # k-chunk-major weights (forward layout)
        self.cseq = torch.empty(L, T, B, H, **f32)
        self.gseq = torch.empty(L, T, B, G, **f32)
        self.comb = torch.empty(B, 2 * H, **f32)
        self.lh = torch.empty(B, 2 * H, **f32)
        self.mu_raw = torch.empty(B, Z, **f32)
        self.lv_raw = torch.empty(B, Z, **f32)
        self.mu = torch.empty(B, Z, **f32)
        self.logvar = torch.empty(B, Z, **f32)
        self.z = torch.empty(B, Z, **f32)
        self.stats = torch.zeros(2 * Z + 4, **f32)
        self.psync = torch.zeros(8192, **i32)    # scratch of the persistent sweeps: forward flags / role counters at [0, 272), the
        #                                          BPTT sweeps' at [512, 848), [500] = the sticky error word of both; the
        #                                          two-group forms' (129..256 rows per GPU) at [1024, 4352)
        # words a step's prologue re-arms (arcvae_enc_lstm_persist_groups: two groups of 16 rows per XCD, two blocks per CU)
        self.psync_words = 5696   # (one-group sweeps: [0, 848); two-group forms: [1024, 4352); the reduce-scatter sweep's
        #                           "gathered" words of its single-buffered exchange: [4352, 4864); the fused seam's: [4864, 5696))
        self.seam_bwd_done = False               # this step's fused seam launch has already written the loss scalars, d(mu_raw),
        #                                          d(lv_raw), dlh and dcomb (EncoderBackwardPlan.heads skips its dcomb chain once)
        self.bptt_rearmed = False                # the prologue of this step's forward has zeroed the BPTT sweep's words too
        # optional diagnostic stamps of the sweep launches / ticks (StepEngine.enable_trace), passed per call
        self.trace_fwd: Optional[torch.Tensor] = None
        self.trace_bwd: Optional[torch.Tensor] = None
        self.scalars = torch.zeros(16, **f32)
        # decoder forward (dense over B*V rows)
        self.tableD = torch.empty(V, G, **f32)
        self.hact = torch.empty(L, BV, H, **f32)
        self.gpre = torch.empty(max(L - 1, 1), BV, G, **f32)
        self.logits = torch.empty(BV, V, **f32)
        self.lse = torch.empty(BV, **f32)
        self.nxt = torch.zeros(BV, **i32)
        self.fed = torch.zeros(B, T, **i32)
        self.rowloss = torch.zeros(B, **f32)
        if train:
            self.dmu_raw = torch.empty(B, Z, **f32)
            self.dlv_raw = torch.empty(B, Z, **f32)
            self.dlh = torch.empty(B, 2 * H, **f32)
            self.dcomb = torch.empty(B, 2 * H, **f32)
            # Working set of a step: 390 -> 215 MB, i.e. inside the 256 MB Infinity Cache.  The gate gradients are
            # written IN PLACE over the saved gates (same thread, after it has read them), and everything only the
            # next BPTT launch consumes lives in short rings over t (csrc/lstm.hip, common.h: arcvae_ring_slots).
            self.dG = self.gseq if os.environ.get("ARCVAE_INPLACE_DG", "1") != "0" else torch.empty(L, T, B, G, **f32)
            self.dG_t = torch.empty(L, RS, B * G * 3 // 2, **f32)      # k-chunk-major copy of dG_t in slot t % RS
            self.dcs = torch.empty(L, RS, B, H, **f32)
            self.dxs = torch.empty(L, RS, B, H, **f32)
            self.wT = torch.empty(2 * L - 1, H * G * 3 // 2, **f32)
            self.dtables = torch.zeros(2, V, G, **f32)        # both token tables: one zero fill (arcvae_enc_prologue)
            self.dtable0 = self.dtables[0]
            self.dtable1 = self.dtables[1]                    # the last chunk's token table, folded by the main stream itself
            self.tables_zeroed = False                        # this step's prologue has zeroed them (persistent forward path)
            self.onehot = torch.empty(T * B, (V + 3) // 4 * 4, **f32)   # one-hot token rows (token-table gradient)
            npart = _lib.load().arcvae_enc_lstm_bwd_rs_part_floats(B, T, H, L)   # partial sums in flight of the reduce-scatter BPTT sweep
            if npart > 0:                                   # (12.6 MB per group of 8 rows per XCD; the size is the library's to say)
                self.ppart = torch.empty(npart, **f32)
            self.dlogits = torch.empty(BV, V, **f32)
            self.ddh = torch.empty(2, BV, H, **f32)
            self.ddG = torch.empty(BV, G, **f32)
            self.dtableD = torch.empty(V, G, **f32)
            self.wcpart = torch.empty(V, G, max(Cc, 1), **f32)


def _lstm_flags(ws) -> int:
    """Precision of the launch-based sweeps where they run on the register-tiled kernels (the MFMA-bound regime): throughput
    mode -> bf16 operands; else the three-piece form (three bf16 pieces per operand, six products: fp32-class accuracy at
    6/16 of the exact-f32 matrix time; ARCVAE_LSTM_SPLIT3=0: exact-f32 MFMA)."""
    if ws.bf16_parts & 1:
        return _lib.LSTM_BF16
    return _lib.LSTM_SPLIT3 if os.environ.get("ARCVAE_LSTM_SPLIT3", "1") != "0" else 0


def _caps(ws, h=None, g=None):
    """ws_floats argument of the launch-based sweeps / the plane weight gradients (include/arcvae_hip.h): what THIS workspace
    allocated, in floats, for {hseq_t, dG_t, wt, wT} -- the library refuses (ARCVAE_ERR_ARG) a call whose kernel family, decided
    at that call from the shape and the ARCVAE_* knobs, would need more.  h / g: other buffers standing in for the two rings."""
    def n(t):
        return 0 if t is None else t.numel() * t.element_size() // 4
    h = ws.hseq_t if h is None else h
    g = getattr(ws, "dG_t", None) if g is None else g
    return (C.c_long * 4)(n(h), n(g), n(ws.wt), n(getattr(ws, "wT", None)))


def _oct(ws, name: str) -> C.c_void_p:
    """Throughput mode: the octet-major bf16 operand copies of the tiled sweeps (Workspace.h_oct / dG_oct), or null."""
    t = getattr(ws, name, None)
    return ptr(t) if t is not None else C.c_void_p(0)


def _layer_ptrs(store: ParamStore, L: int, leaf: str, grad: bool = False, skip0: bool = False):
    get = store.g if grad else store.p
    ts = [None if (skip0 and l == 0) else get(f"lstm_layer_{l}.{leaf}") for l in range(L)]
    return ptr_array(ts)


# --------------------------------------------------------------------------------------------
# op-level drivers (each is a handful of C-ABI calls on the current stream)
# --------------------------------------------------------------------------------------------
def bptt_reduce_scatter_ok(ws, d: ModelDims) -> bool:
    """The BPTT sweep of this shape runs as the persistent reduce-scatter kernel (lstm_bwd_persist_rs_kernel).
    ARCVAE_PERSIST_BWD: unset / "3" = where the shape allows (H 256, L <= 2, B <= 64); "0" = per-step launches;
    "1" = the output-split persistent kernel (slower, kept for reference)."""
    if os.environ.get("ARCVAE_PERSIST_BWD", "3") != "3":
        return False
    return _lib.load().arcvae_enc_lstm_bwd_rs_ok(ws.B, ws.T, d.H, d.L) == 1


def fused_wgrad_ok(ws, d: ModelDims) -> bool:
    """ARCVAE_FUSED_WGRAD=1 (opt-in): the BPTT sweep of this shape forms the stack's weight gradients itself (FW variant of
    lstm_bwd_persist_rs_kernel: no weight-gradient GEMMs, no chunks).  Parity-green, but measured SLOWER than the sweep +
    GEMMs on aux / side (the tick grows from 2.8 to 4.7 us -- nothing added to the chain's wave
    hides -- which costs 240 us per step against the 220 us the separate GEMMs cost), so it is not the default."""
    return os.environ.get("ARCVAE_FUSED_WGRAD", "0") == "1" and d.V <= 128 and bptt_reduce_scatter_ok(ws, d)


def persistent_forward_ok(ws: Workspace, d: ModelDims) -> bool:
    """The forward sweep of this shape can run as one persistent launch (csrc/lstm.hip: lstm_fwd_persist_kernel)."""
    return _lib.load().arcvae_enc_lstm_persistent_ok(ws.B, ws.T, d.H, d.L) == 1


def seam_fused_ok(ws: Workspace, d: ModelDims) -> bool:
    """The chain between the two sweeps of this shape runs as the fused per-XCD seam kernel (csrc/latent.hip: enc_seam_kernel):
    behind a persistent forward sweep (comb and the zeroed stats are ready, the sync words re-armed by the prologue)."""
    return persistent_forward_ok(ws, d) and _lib.load().arcvae_enc_seam_ok(ws.B, d.H, d.Z) == 1


def encoder_seam(enc: ParamStore, ws: Workspace, d: ModelDims, free_bits: float, phases: int) -> None:
    """arcvae_enc_seam: phases 1 = heads forward + statistics, 2 = loss scalars + latent gradients + dlh + dcomb from GLOBAL
    stats, 3 = both in one launch (single process)."""
    train = (phases & 2) != 0
    call("arcvae_enc_seam", ptr(ws.comb), ptr(enc.p("fc_mu.weight")), ptr(enc.p("fc_mu.bias")),
         ptr(enc.p("fc_logvar_hidden.weight")), ptr(enc.p("fc_logvar_hidden.bias")), ptr(enc.p("fc_logvar.weight")),
         ptr(enc.p("fc_logvar.bias")), ptr(ws.eps), ptr(ws.hyper), ptr(ws.lh), ptr(ws.mu_raw), ptr(ws.lv_raw), ptr(ws.mu),
         ptr(ws.logvar), ptr(ws.z), ptr(ws.stats), ptr(ws.scalars), ptr(ws.dmu_raw) if train else C.c_void_p(0),
         ptr(ws.dlv_raw) if train else C.c_void_p(0), ptr(ws.dlh) if train else C.c_void_p(0),
         ptr(ws.dcomb) if train else C.c_void_p(0), ptr(ws.psync), ws.B, d.H, d.Z, ws.T, float(free_bits), phases, 1, stream_ptr())
    if train:
        ws.seam_bwd_done = True


def encoder_forward(enc: ParamStore, ws: Workspace, d: ModelDims, free_bits: float,
                    start_signal: Optional[C.c_void_p] = None, zero_grad: bool = False, seam: int = 1) -> None:
    """models/encoder.py:76-153 + per-rank latent statistics.  start_signal (persistent sweep only): device word
    bumped when the sweep starts.  zero_grad: also clear the encoder's gradient buffer (start of a training step).
    seam = 3 (single-process training step): where the fused seam kernel applies, it also runs the loss + dcomb chain
    (sets ws.seam_bwd_done: the caller skips arcvae_latent_loss and the heads' phase 1)."""
    B, T = ws.B, ws.T
    G = 4 * d.H
    s = stream_ptr()
    wx, _k1 = _layer_ptrs(enc, d.L, "Wx", skip0=True)
    wh, _k2 = _layer_ptrs(enc, d.L, "Wh")
    bs, _k3 = _layer_ptrs(enc, d.L, "bias", skip0=True)
    if persistent_forward_ok(ws, d):
        # Two launches in front of the sweep instead of six: [tokens^T + the gradient buffer's zero fill + the sweep's
        # re-arm] and table0 = embedding . Wx_0^T + bias_0 ([V,4H]: the layer-0 input projection of every token).  The
        # sweep reads the row-major weights itself; k-chunk-major BPTT copies are only written for a launch-based backward.
        # The prologue also writes the condition half of the heads' input and clears `stats`; the sweep's last tick stores
        # h_{T-1} of the top layer into the other half: the heads start without a build launch.
        grad = enc.grad if zero_grad else None
        tabs = ws.dtables if (zero_grad and hasattr(ws, "dtables")) else None   # the token tables of the step's two chunks
        ws.tables_zeroed = tabs is not None
        ws.bptt_rearmed = True                   # consumed by the next EncoderBackwardPlan.sweep(chunk 0)
        call("arcvae_enc_prologue", ptr(ws.x), ptr(ws.x_tb), ptr(grad), C.c_long(grad.numel() if grad is not None else 0),
             ptr(tabs), C.c_long(tabs.numel() if tabs is not None else 0), ptr(ws.psync), ws.psync_words, 500, ptr(ws.cond), ptr(enc.p("condition_fc.weight")), ptr(enc.p("condition_fc.bias")),
             ptr(ws.comb), ptr(ws.stats), 2 * d.Z + 4, ptr(ws.onehot) if (zero_grad and hasattr(ws, "onehot")) else C.c_void_p(0),
             d.V, B, T, d.H, d.C, s)
        comb_ready = 1
        call("arcvae_gemm_f32", 0, 1, d.V, G, d.E, ptr(enc.p("embedding.weight")), d.E,
             ptr(enc.p("lstm_layer_0.Wx")), d.E, ptr(ws.table0), G, ptr(enc.p("lstm_layer_0.bias")), 0, s)
        need_wT = hasattr(ws, "wT") and not bptt_reduce_scatter_ok(ws, d)
        call("arcvae_enc_lstm_forward_persistent", ptr(ws.x_tb), ptr(ws.table0), wx, wh, bs, ptr(ws.hseq),
             ptr(ws.cseq), ptr(ws.gseq), ptr(ws.wT) if need_wT else C.c_void_p(0), C.c_long(ws.wT.numel() if need_wT else 0),
             ptr(ws.comb), ptr(ws.psync),
             start_signal if start_signal is not None else C.c_void_p(0), B, T, d.V, d.H, d.L,
             1 | (_lib.PERSIST_BF16 if ws.bf16_parts & 1 else 0) | (_lstm_flags(ws) & _lib.LSTM_SPLIT3), ptr(ws.trace_fwd), s)
    else:
        comb_ready = 0
        if zero_grad:
            enc.grad.zero_()
        call("arcvae_transpose_tokens", ptr(ws.x), ptr(ws.x_tb), B, T, s)
        # table0 = embedding . Wx_0^T + bias_0   ([V,4H]; the layer-0 input projection of every token)
        call("arcvae_gemm_f32", 0, 1, d.V, G, d.E, ptr(enc.p("embedding.weight")), d.E,
             ptr(enc.p("lstm_layer_0.Wx")), d.E, ptr(ws.table0), G, ptr(enc.p("lstm_layer_0.bias")), 0, s)
        wT = ptr(ws.wT) if hasattr(ws, "wT") else C.c_void_p(0)
        if start_signal is not None:
            call("arcvae_gate_set", start_signal, 1, 1, s)
        call("arcvae_enc_lstm_forward", ptr(ws.x_tb), ptr(ws.table0), wx, wh, bs, ptr(ws.hseq), ptr(ws.hseq_t),
             ptr(ws.cseq), ptr(ws.gseq), ptr(ws.wt), wT, B, T, d.V, d.H, d.L, _lstm_flags(ws), _caps(ws),
             _oct(ws, "h_oct"), ptr(ws.trace_fwd), s)
    if comb_ready and seam_fused_ok(ws, d):
        encoder_seam(enc, ws, d, free_bits, 3 if (seam == 3 and hasattr(ws, "dcomb")) else 1)
        return
    hT = ws.hseq[d.L - 1, T - 1]  # [B,H] contiguous slab: last padded position (Q3)
    call("arcvae_enc_heads_forward", ptr(hT), ptr(ws.cond), ptr(enc.p("condition_fc.weight")),
         ptr(enc.p("condition_fc.bias")), ptr(enc.p("fc_mu.weight")), ptr(enc.p("fc_mu.bias")),
         ptr(enc.p("fc_logvar_hidden.weight")), ptr(enc.p("fc_logvar_hidden.bias")),
         ptr(enc.p("fc_logvar.weight")), ptr(enc.p("fc_logvar.bias")), ptr(ws.eps), ptr(ws.comb), ptr(ws.lh),
         ptr(ws.mu_raw), ptr(ws.lv_raw), ptr(ws.mu), ptr(ws.logvar), ptr(ws.z), ptr(ws.stats), B, d.H, d.Z,
         d.C, float(free_bits), comb_ready, s)


def _dec_gemm_bits(ws) -> int:
    """How the decoder's B*V-row products run: throughput mode -> bf16 operands; beside the one-row-group persistent sweeps
    (Workspace.dec_split3, set by StepEngine.workspace) -> three bf16 pieces / six products: fp32-class accuracy at 2.7x less
    matrix-pipe time, in 32- instead of 64-cycle instructions, on the SIMDs the forward sweep's waves issue on."""
    if ws.bf16_parts & 2:
        return _lib.DEC_BF16
    return _lib.DEC_SPLIT3 if getattr(ws, "dec_split3", False) else 0


def decoder_forward_dense(dec: ParamStore, ws: Workspace, d: ModelDims, mode: int = 0,
                          temperature: float = 1.0, keep_gpre: bool = True, alone: bool = False) -> None:
    """models/decoder.py:152-175 for all B*V (row, token) pairs.  keep_gpre=False (sampler, loss-only forward): the layers'
    pre-activations are not kept, so a layer's GEMM and cell run as one kernel (csrc/gemm.hip gemm_cell_zero_kernel).
    alone=True (the sampler: no encoder sweep shares the chip) with ARCVAE_SAMPLER_TILED=1: layers 1 .. L-1 on the three-piece tile
    kernel where its grid fills the chip (arcvae_dense_stack_forward, forward-only form)."""
    if not keep_gpre:
        mode |= _lib.DEC_NO_GPRE
    wx, _k1 = _layer_ptrs(dec, d.L, "Wx")
    bs, _k2 = _layer_ptrs(dec, d.L, "bias")

    def part(bits):
        call("arcvae_dec_forward_dense", ptr(dec.p("embedding.weight")), wx, bs, ptr(dec.p("fc_out.weight")),
             ptr(dec.p("fc_out.bias")), ptr(ws.cond), ptr(ws.tableD), ptr(ws.hact), ptr(ws.gpre), ptr(ws.logits),
             ptr(ws.lse), ptr(ws.nxt), ws.B, d.V, d.E, d.C, d.H, d.L, mode | bits | _dec_gemm_bits(ws),
             float(temperature), stream_ptr())

    # (opt-in, ARCVAE_SAMPLER_TILED=1: parity-green, measured no faster than the fused exact-f32 GEMM + cell kernel -- 6.59 vs 6.63 ms
    # per 10 k molecules: the split pass over hact[0] and the tile launch cost what the matrix time saves)
    if alone and not keep_gpre and not (ws.bf16_parts & 2) and os.environ.get("ARCVAE_SAMPLER_TILED", "0") == "1" \
            and _lib.load().arcvae_dense_stack_ok(ws.B * d.V, d.H, d.L) == 1:
        if getattr(ws, "dense_ws_fwd", None) is None:
            n = C.c_long(0)
            _lib.load().arcvae_dense_stack_ws_floats(ws.B * d.V, d.H, d.L, C.byref(n))
            ws.dense_ws_fwd = torch.empty(n.value, dtype=torch.float32, device=ws.hact.device)
        mode &= ~_lib.DEC_NO_GPRE
        part(_lib.DEC_PART_HEAD)
        call("arcvae_dense_stack_forward", wx, bs, ptr(ws.hact), C.c_void_p(0), ptr(ws.dense_ws_fwd), C.c_long(ws.dense_ws_fwd.numel()),
             ws.B * d.V, d.H, d.L, 1, stream_ptr())
        part(_lib.DEC_PART_TAIL)
        return
    if keep_gpre and getattr(ws, "dense_ws", None) is not None:
        # MFMA-bound regime (StepEngine.workspace): layers 1 .. L-1 -- zero-state cells over B*V rows -- on the three-piece tile
        # kernels of the encoder's sweeps; `gpre` then holds the POST-activation gates (what their backward reads)
        part(_lib.DEC_PART_HEAD)
        call("arcvae_dense_stack_forward", wx, bs, ptr(ws.hact), ptr(ws.gpre), ptr(ws.dense_ws), C.c_long(ws.dense_ws.numel()),
             ws.B * d.V, d.H, d.L, _lib.LSTM_BF16 if (ws.bf16_parts & 2) else 0, stream_ptr())
        part(_lib.DEC_PART_TAIL)
        ws.dense_fwd = True
        return
    ws.dense_fwd = False
    part(0)


def decoder_chain(ws: Workspace, d: ModelDims) -> None:
    """models/decoder.py:146-185 walk + losses/recon.py row sums."""
    call("arcvae_dec_chain_ce", ptr(ws.x), ptr(ws.coins), ptr(ws.nxt), ptr(ws.lse), ptr(ws.logits), ptr(ws.fed),
         ptr(ws.rowloss), ws.B, ws.T, d.V, stream_ptr())


def decoder_backward(dec: ParamStore, ws: Workspace, d: ModelDims, inv_count: float) -> None:
    s = stream_ptr()
    call("arcvae_dec_ce_backward", ptr(ws.x), ptr(ws.fed), ptr(ws.logits), ptr(ws.lse), ptr(ws.dlogits), ws.B,
         ws.T, d.V, float(inv_count), s)
    wx, _k1 = _layer_ptrs(dec, d.L, "Wx")
    bs, _k2 = _layer_ptrs(dec, d.L, "bias")
    dwx, _k3 = _layer_ptrs(dec, d.L, "Wx", grad=True)
    dbs, _k4 = _layer_ptrs(dec, d.L, "bias", grad=True)

    def part(bits, dh):
        call("arcvae_dec_backward_dense", ptr(dec.p("embedding.weight")), wx, bs, ptr(dec.p("fc_out.weight")),
             ptr(ws.cond), ptr(ws.tableD), ptr(ws.hact), ptr(ws.gpre), ptr(ws.dlogits), dh, ptr(ws.ddG),
             ptr(ws.dtableD), ptr(ws.wcpart), ptr(dec.g("embedding.weight")), dwx, dbs, ptr(dec.g("fc_out.weight")),
             ptr(dec.g("fc_out.bias")), ws.B, d.V, d.E, d.C, d.H, d.L, bits | _dec_gemm_bits(ws), s)

    if getattr(ws, "dense_fwd", False):
        # the forward of this step ran layers 1 .. L-1 on the tile kernels (decoder_forward_dense): fc_out's part, the stack's
        # backward on the same kernels (dh_top in ddh[0] -> dh_0 in ddh[1]), then layer 0 and the token table
        part(_lib.DEC_PART_TAIL, ptr(ws.ddh))
        call("arcvae_dense_stack_backward", ptr(ws.gpre), ptr(ws.ddh[0]), ptr(ws.ddG), ptr(ws.ddh[1]), dwx, dbs,
             ptr(ws.dense_ws), C.c_long(ws.dense_ws.numel()), ws.B * d.V, d.H, d.L, _lib.LSTM_BF16 if (ws.bf16_parts & 2) else 0, s)
        part(_lib.DEC_PART_HEAD, ptr(ws.ddh[1]))
        return
    part(0, ptr(ws.ddh))


def latent_loss(ws: Workspace, d: ModelDims, free_bits: float, with_grads: bool) -> None:
    """complete_vae_loss.py:45-99 from the (global) stats; optionally d/d(mu_raw, lv_raw)."""
    call("arcvae_latent_loss", ptr(ws.stats), ptr(ws.hyper), ptr(ws.mu), ptr(ws.logvar), ptr(ws.scalars),
         ptr(ws.dmu_raw) if with_grads else C.c_void_p(0), ptr(ws.dlv_raw) if with_grads else C.c_void_p(0),
         ws.B, d.Z, ws.T, float(free_bits), stream_ptr())


def _inline(key, fn, stream) -> None:
    with torch.cuda.stream(stream):
        fn()


class Gates:
    """Device-side cross-stream ordering for the single-process step (csrc/misc.hip: gate_wait/gate_set).

    Why not events: a stream waiting on another stream's event parks a barrier packet at the head of its hardware
    queue, and on MI355X every such blocked queue adds ~1 us to EACH dependent dispatch of the running LSTM chain
    (tools/step_trace.py: 4.72 -> 5.68 us per launch; the host runs steps ahead, so `side` and `aux` were blocked
    almost all the time).  With gates the waiting stream runs a one-wave polling kernel instead.

    Words (one 128-B line each):  P main's signal count (exactly STRIDE per step), Q aux's (1 per step),
    R completions of aux + side (2 per step: main's join before the optimizer step), D decoder segments finished by side
    (1 per step: the data-parallel step reduces the decoder's gradients early), NS / NA / NM steps finished by
    side / aux / main (the waiter's own ticket counter), ERR expired gates, PROBE self-test, H "the BPTT sweep has started"
    (1 per step, raised by the sweep's first launch: the heads' dcomb chain in front of it is complete, so their
    parameter gradients may be formed -- on aux, while the first chunk is still being swept and aux has nothing else),
    HG "the heads' parameter gradients of this step are formed" (1 per step, raised by aux behind heads_wgrad: the
    data-parallel step reduces that 1.58 MB bucket early, on side -- dp.EngineOps.seam_buckets).
    """
    STRIDE = 8
    LONG, SHORT = 16_000_000, 3_000    # polls (~1.5 us each): ~25 s before a gate gives up (a first RCCL collective
                                       # or a peer still capturing its graphs may hold main up for seconds); ~4 ms probe
    P, Q, NS, NA, ERR, PROBE, R, NM, D, H, HG = range(11)

    def __init__(self, device):
        self.mem = torch.zeros(11 * 32, dtype=torch.int32, device=device)
        self._probed: Dict[Tuple[int, int], bool] = {}

    def word(self, i: int) -> C.c_void_p:
        return C.c_void_p(self.mem.data_ptr() + 128 * i)

    def signal(self, w: int, add: int = 1) -> None:
        call("arcvae_gate_set", self.word(w), add, 1, stream_ptr())

    def wait(self, w: int, steps: int, stride: int, offset: int, advance: bool = False) -> None:
        call("arcvae_gate_wait", self.word(w), self.word(steps), stride, offset, int(advance), self.LONG,
             self.word(self.ERR), stream_ptr())

    def join(self) -> None:
        """Main-stream gate at the end of a step: aux and side have reported their last piece on R."""
        self.wait(self.R, self.NM, 2, 2, advance=True)

    def errors(self) -> int:
        return int(self.mem[self.ERR * 32].item())

    def probe(self, waiter: torch.cuda.Stream, signaller: torch.cuda.Stream) -> bool:
        """True iff a gate on `waiter` can be released from `signaller`, i.e. the two streams do NOT share a
        hardware queue (HIP multiplexes streams onto a few queues; on a shared one the gate would sit in front of
        its own release).  Costs a few ms once per stream pair."""
        key = (waiter.cuda_stream, signaller.cuda_stream)
        if key not in self._probed:
            torch.cuda.synchronize()
            self.mem[self.PROBE * 32] = 0
            before = self.errors()
            with torch.cuda.stream(waiter):
                call("arcvae_gate_wait", self.word(self.PROBE), C.c_void_p(0), 0, 1, 0, self.SHORT,
                     self.word(self.ERR), stream_ptr())
            with torch.cuda.stream(signaller):
                call("arcvae_gate_set", self.word(self.PROBE), 1, 0, stream_ptr())
            torch.cuda.synchronize()
            ok = self.errors() == before
            self.mem[self.ERR * 32] = before
            torch.cuda.synchronize()
            self._probed[key] = ok
        return self._probed[key]


class SegmentRunner:
    """Runs named launch sequences on given streams -- eagerly, or captured ONCE each as a single-stream
    hipGraph and replayed.  Single-stream (linear) segments with eager events between them are used instead
    of one forked multi-stream graph: on ROCm 7.2 a forked hipGraph starts its second branch only when the
    first reaches the join (profiles/r01_graph_vs_segments), which serialises decoder and encoder."""

    def __init__(self, capture: bool):
        self.capture = capture
        self._g: Dict[str, object] = {}
        self._cap_stream: Optional[torch.cuda.Stream] = None  # capture may not happen on the default stream
        self.steps = 0          # single-process steps enqueued through this runner (StepEngine._enqueue_step)

    def __call__(self, key: str, fn, stream: torch.cuda.Stream, eager_first: bool = True) -> None:
        """eager_first=False: the segment's first use is capture + replay WITHOUT the eager run (and its stream.synchronize())
        -- for a segment that contains a gate other streams release only later in the same step (the merged main + finish
        segment: its join would hold the host in synchronize() until the gate expired); every launch in it must have run
        before (lazy kernel attributes are set)."""
        with torch.cuda.stream(stream):
            if not self.capture:
                fn()
                return
            g = self._g.get(key)
            if g is None and not eager_first:
                try:
                    if self._cap_stream is None:
                        self._cap_stream = torch.cuda.Stream(device=stream.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._cap_stream, capture_error_mode="thread_local"):
                        fn()
                    self._g[key] = g
                    g.replay()
                except Exception as exc:
                    print(f"[arcvae_hip] graph capture of segment {key!r} failed ({exc}); running it eagerly")
                    torch.cuda.synchronize()
                    self._g[key] = False
                    fn()
                return
            if g is None:
                fn()  # first step runs eagerly (its results are this step's results) ...
                stream.synchronize()
                try:  # ... and is then recorded (on a private stream) for replay from the next step on
                    if self._cap_stream is None:
                        self._cap_stream = torch.cuda.Stream(device=stream.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._cap_stream, capture_error_mode="thread_local"):
                        fn()
                    self._g[key] = g
                except Exception as exc:  # keep training eagerly if capture is refused
                    print(f"[arcvae_hip] graph capture of segment {key!r} failed ({exc}); running it eagerly")
                    torch.cuda.synchronize()
                    self._g[key] = False
            elif g is False:
                fn()
            else:
                g.replay()


class EncoderBackwardPlan:
    """The pieces of the encoder backward as closures over (enc, ws), plus the chunk schedule.

    Critical chain (main stream): heads(1) [dcomb] -> sweep chunks.  Off the chain (aux stream): heads(2)
    [the heads' parameter gradients] and wgrad chunks, each released by an event after the sweep chunk that
    completes its time range.  Chunks shrink geometrically so the part of the weight-gradient work that
    cannot overlap the sweep (the last chunk) is small."""

    # Chunk boundaries as fractions of the sweep (ARCVAE_BPTT_CHUNKS overrides both).  Launch-based sweep: four chunks.
    # Persistent sweep: two -- every chunk is a relaunch of a kernel that needs all 256 CUs at once, and whatever runs
    # beside the sweep slows its ticks (2.6 us alone, 3.6 with the weight-gradient GEMMs on the same CUs), so the
    # gradient GEMMs are better concentrated late and spread over two streams (measured on one box, steady state:
    # 4 chunks 1.073-1.078 ms, "0.5,0.85,1" 1.069, two chunks at 0.5 / 0.6 / 0.65 / 0.7 / 0.8 of the sweep with the
    # dWx GEMMs on side 1.060 / 1.040-1.044 / 1.037 / 1.045 / 1.051).
    FRACTIONS_LAUNCHES = "0.3,0.6,0.85,1.0"
    FRACTIONS_PERSISTENT = "0.63,1.0"
    FRACTIONS_FUSED = "1.0"      # weight gradients formed inside the sweep: one launch, nothing to hand over

    def __init__(self, enc: ParamStore, ws: Workspace, d: ModelDims):
        self.enc, self.ws, self.d = enc, ws, d
        L, T = d.L, ws.T
        self.persistent = bptt_reduce_scatter_ok(ws, d)
        self.fused = fused_wgrad_ok(ws, d)
        default = self.FRACTIONS_FUSED if self.fused else (self.FRACTIONS_PERSISTENT if self.persistent else self.FRACTIONS_LAUNCHES)
        self.FRACTIONS = tuple(float(f) for f in os.environ.get("ARCVAE_BPTT_CHUNKS", default).split(","))
        self.S = T + 2 * (L - 1)  # launches of the BPTT wavefront (csrc/lstm.hip)
        self._wx = _layer_ptrs(enc, L, "Wx", skip0=True)
        self._wh = _layer_ptrs(enc, L, "Wh")
        self._dwx = _layer_ptrs(enc, L, "Wx", grad=True)
        self._dwh = _layer_ptrs(enc, L, "Wh", grad=True)
        self._dbs = _layer_ptrs(enc, L, "bias", grad=True)
        self.chunks = self.chunk_schedule(T, L, self.FRACTIONS)

    @staticmethod
    def chunk_schedule(T: int, L: int, fractions) -> list:
        """[(s0, s1, t_lo, t_hi, first, last)]: tick ranges [s0, s1) of the T + 2(L-1)-tick BPTT wavefront and the
        time range [t_lo, t_hi) whose gate gradients are complete in EVERY layer once the ticks up to s1 are done
        (the weight-gradient GEMMs of a chunk read exactly that range).  The ranges tile [0, T) from the top."""
        S = T + 2 * (L - 1)
        bounds = sorted({0, S} | {min(S, max(1, round(f * S))) for f in fractions})
        chunks = []
        t_hi = T
        for s0, s1 in zip(bounds[:-1], bounds[1:]):
            # after launches [0, s1) every layer has finished all t >= T - s1 + 2(L-1)
            t_lo = 0 if s1 >= S else min(T, max(0, T - s1 + 2 * (L - 1)))
            chunks.append((s0, s1, t_lo, t_hi, t_hi == T, s1 >= S))
            t_hi = t_lo
        return chunks

    def heads(self, phase: int) -> None:
        enc, ws, d = self.enc, self.ws, self.d
        if phase in (0, 1) and getattr(ws, "seam_bwd_done", False):
            ws.seam_bwd_done = False           # the fused seam launch of this step has written dlh / dcomb already
            if phase == 1:
                return
            phase = 2
        call("arcvae_enc_heads_backward", ptr(ws.cond), ptr(enc.p("fc_mu.weight")),
             ptr(enc.p("fc_logvar_hidden.weight")), ptr(enc.p("fc_logvar.weight")), ptr(ws.comb), ptr(ws.lh),
             ptr(ws.dmu_raw), ptr(ws.dlv_raw), ptr(ws.dlh), ptr(ws.dcomb), ptr(enc.g("condition_fc.weight")),
             ptr(enc.g("condition_fc.bias")), ptr(enc.g("fc_mu.weight")), ptr(enc.g("fc_mu.bias")),
             ptr(enc.g("fc_logvar_hidden.weight")), ptr(enc.g("fc_logvar_hidden.bias")),
             ptr(enc.g("fc_logvar.weight")), ptr(enc.g("fc_logvar.bias")), ws.B, d.H, d.Z, d.C, phase, stream_ptr())

    def sweep(self, s0: int, s1: int, start_signal: Optional[C.c_void_p] = None, chunk_index: int = 0) -> None:
        """Ticks [s0, s1) of the BPTT wavefront; chunk_index = position of this call among the sweep's chunk calls
        (the persistent kernels draw their block roles from per-chunk counters)."""
        # d/d(hT) = dcomb[:, :H] (row stride 2H)
        ws, d = self.ws, self.d
        sig = start_signal if start_signal is not None else C.c_void_p(0)
        # flags bit 0 of the reduce-scatter entry point: its sync words are armed (the prologue of the forward that ran
        # last zeroed them and no BPTT sweep has used them since) -- no zero-fill launch between the seam and the sweep
        rearmed = 1 if (chunk_index == 0 and getattr(ws, "bptt_rearmed", False)) else 0
        if chunk_index == 0:
            ws.bptt_rearmed = False
        if self.fused:
            # default shape: the reduce-scatter sweep with the weight gradients formed inside it
            enc = self.enc
            call("arcvae_enc_lstm_backward_fused", self._wx[0], self._wh[0], ptr(ws.cseq), ptr(ws.gseq), ptr(ws.hseq),
                 ptr(ws.x_tb), ptr(ws.dcomb), 2 * d.H, ptr(ws.dG), ptr(ws.dcs), ptr(ws.dxs), ptr(ws.ppart), C.c_long(ws.ppart.numel()), ptr(ws.psync),
                 sig, self._dwx[0], self._dwh[0], self._dbs[0], ptr(ws.dtable0), ws.B, ws.T, d.V, d.H, d.L, s0, s1,
                 chunk_index, ptr(ws.trace_bwd), stream_ptr())
            return
        if self.persistent:
            # latency regime, default shape: persistent BPTT sweep in its reduce-scatter form, one launch per chunk
            # (csrc/lstm.hip: lstm_bwd_persist_rs_kernel)
            # (half-batch form, round 4: 129..256 rows as two sweeps of 16 rows per XCD per chunk, one behind the other)
            halves = _lib.load().arcvae_enc_lstm_bwd_rs_halves(ws.B, ws.T, d.H, d.L) == 1
            for hf in ((_lib.RS_HALF, _lib.RS_HALF | _lib.RS_HALF1) if halves else (0,)):
                call("arcvae_enc_lstm_backward_persistent_rs", self._wx[0], self._wh[0], ptr(ws.cseq), ptr(ws.gseq),
                     ptr(ws.dcomb), 2 * d.H, ptr(ws.dG), ptr(ws.dcs), ptr(ws.dxs), ptr(ws.ppart), C.c_long(ws.ppart.numel()), ptr(ws.psync),
                     sig if not (hf & _lib.RS_HALF1) else C.c_void_p(0),
                     ws.B, ws.T, d.H, d.L, s0, s1, chunk_index, (_lib.PERSIST_BF16 if ws.bf16_parts & 1 else 0) | rearmed | hf,
                     ptr(ws.trace_bwd) if not (hf & _lib.RS_HALF1) else C.c_void_p(0), stream_ptr())
            return
        if _lib.load().arcvae_enc_lstm_bwd_persistent_ok(ws.B, ws.T, d.H, d.L) == 1:
            # latency regime: one persistent launch per chunk (csrc/lstm.hip: lstm_bwd_persist_kernel)
            call("arcvae_enc_lstm_backward_persistent", ptr(ws.cseq), ptr(ws.gseq), ptr(ws.dcomb), 2 * d.H, ptr(ws.dG),
                 ptr(ws.dcs), ptr(ws.dxs), ptr(ws.wT), ptr(ws.psync), sig, ws.B, ws.T, d.H, d.L, s0, s1, chunk_index,
                 ptr(ws.trace_bwd), stream_ptr())
            return
        call("arcvae_enc_lstm_backward", self._wx[0], self._wh[0], ptr(ws.cseq), ptr(ws.gseq), ptr(ws.dcomb),
             2 * d.H, ptr(ws.dG), ptr(ws.dG_t), ptr(ws.dcs), ptr(ws.dxs), ptr(ws.wT), ws.B, ws.T, d.H, d.L, s0, s1,
             _lstm_flags(ws),  # no LSTM_RETILE: the forward of this step already wrote the BPTT weight layouts
             _caps(ws), _oct(ws, "dG_oct"), start_signal if start_signal is not None else C.c_void_p(0), ptr(ws.trace_bwd), stream_ptr())

    def wgrad(self, t_lo: int, t_hi: int, first: bool, last: bool, parts: int = 3, table=None) -> None:
        """Weight gradients of the time range.  `table`: token-table workspace of this call (default ws.dtable0); the
        table path is linear, so a range may own a workspace and be zeroed, accumulated and folded by itself
        (first = last = True)."""
        enc, ws, d = self.enc, self.ws, self.d
        if os.environ.get("ARCVAE_DEBUG_SKIP_WGRAD", "0") == "1":   # timing experiments only (gradients are WRONG)
            return
        if self.fused:
            # the sweep has formed dWh_l, dWx_l, dbias_l (l >= 1) and the token table itself; what is left is folding the
            # table into embedding.weight / lstm_layer_0.{Wx,bias} once the last chunk is done
            if last and (parts & 2):
                call("arcvae_table_finalize", ptr(ws.dtable0), ptr(enc.p("lstm_layer_0.Wx")), d.E,
                     ptr(enc.p("embedding.weight")), ptr(enc.g("embedding.weight")), ptr(enc.g("lstm_layer_0.Wx")),
                     ptr(enc.g("lstm_layer_0.bias")), d.V, d.E, 4 * d.H, stream_ptr())
            return
        if (ws.bf16_parts & 4) and not self.persistent:
            # throughput mode: one bf16 product per GEMM step.  Not beside a persistent sweep: there the matrix pipe is
            # idle anyway and what counts is what fits on the sweep's SIMDs (measured at bs 64: 1.077 vs 1.03 ms)
            parts |= _lib.WGRAD_BF16
        elif getattr(ws, "pl_g", None) is not None:
            # 129..512 rows per GPU (persistent or per-step-launch BPTT, f32 gate gradients): the call splits dG / h of its time
            # range into three-piece planes itself (one elementwise pass) and runs the plane GEMMs -- 2.7x less matrix-pipe time
            # than the exact-f32 tile GEMM, no re-splitting per tile
            parts |= 2048 | 4096 | 16
            call("arcvae_enc_lstm_wgrad", ptr(ws.x_tb), ptr(enc.p("embedding.weight")), ptr(enc.p("lstm_layer_0.Wx")),
                 ptr(ws.hseq), ptr(ws.dG), ptr(table if table is not None else ws.dtable0), ptr(ws.onehot),
                 ptr(enc.g("embedding.weight")), self._dwx[0], self._dwh[0], self._dbs[0], ws.B, ws.T, d.V, d.E, d.H, d.L,
                 t_lo, t_hi, int(first), int(last), parts, ptr(ws.pl_h), ptr(ws.pl_g), _caps(ws, ws.pl_h, ws.pl_g), stream_ptr())
            return
        elif getattr(ws, "planes", False) and not self.persistent:
            # MFMA-bound regime, three-piece sweeps: the GEMMs read the sweeps' operand planes (all T slots kept) -- see
            # StepEngine.workspace; (| 16: the token-table one-hot GEMM keeps its exact-f32 tile form)
            parts |= 2048 | 16
            call("arcvae_enc_lstm_wgrad", ptr(ws.x_tb), ptr(enc.p("embedding.weight")), ptr(enc.p("lstm_layer_0.Wx")),
                 ptr(ws.hseq), ptr(ws.dG), ptr(table if table is not None else ws.dtable0), ptr(ws.onehot),
                 ptr(enc.g("embedding.weight")), self._dwx[0], self._dwh[0], self._dbs[0], ws.B, ws.T, d.V, d.E, d.H, d.L,
                 t_lo, t_hi, int(first), int(last), parts, ptr(ws.hseq_t), ptr(ws.dG_t), _caps(ws), stream_ptr())
            return
        elif (not self.persistent and os.environ.get("ARCVAE_WGRAD_SPLIT3", "1") != "0"
              and (_lib.load().arcvae_enc_lstm_tiled_for(ws.B, d.H, d.L, _lstm_flags(ws)) & 2)):
            # MFMA-bound regime (the BPTT runs on the register-tiled kernels): the weight-gradient GEMMs as three-piece tile
            # GEMMs -- 2.7x less matrix-pipe time at fp32-class accuracy (bs 2048: 22.3 -> 21.3 ms, configs[2]: 42.0 -> 40.9;
            # at 128 / 256 rows, beside the persistent or 2x2 sweeps, the exact-f32 tile GEMM stays ahead: 1.705 vs 1.777, 3.24 vs 3.52)
            parts |= 1024 | 16      # (| 16: the token-table one-hot GEMM keeps its exact-f32 tile form)
        elif not (self.persistent and ws.B <= 64):
            # The split-bf16 GEMM pays beside the one-row-group persistent sweep only (1.061 vs 1.083 ms at bs 64: its
            # 208 registers fit on a SIMD next to the sweep's 296, and it leaves the matrix pipe to the chain).  Beside
            # the 2-group sweep (343 registers) it cannot be resident, and beside the per-step launches the exact-f32
            # tile GEMM is the faster one (bs 256: 3.254 vs 3.363 ms).
            parts |= 16
        elif last and os.environ.get("ARCVAE_WIDE_TAIL", "0") == "1":
            # opt-in: the last chunk's GEMMs run behind the sweep, so the 128-row split tile could be resident -- measured
            # slower (1.072 vs 1.028 ms: three concurrent launches of one-block-per-CU kernels serialise in the tail)
            parts |= 64
        call("arcvae_enc_lstm_wgrad", ptr(ws.x_tb), ptr(enc.p("embedding.weight")), ptr(enc.p("lstm_layer_0.Wx")),
             ptr(ws.hseq), ptr(ws.dG), ptr(table if table is not None else ws.dtable0), ptr(ws.onehot),
             ptr(enc.g("embedding.weight")), self._dwx[0],
             self._dwh[0],
             self._dbs[0], ws.B, ws.T, d.V, d.E, d.H, d.L, t_lo, t_hi, int(first), int(last), parts,
             _oct(ws, "h_oct"), _oct(ws, "dG_oct"), None, stream_ptr())


def encoder_backward(enc: ParamStore, ws: Workspace, d: ModelDims, aux: Optional[torch.cuda.Stream] = None,
                     run=_inline, prologue=None, after_first=None, aux2: Optional[torch.cuda.Stream] = None,
                     gates: Optional[Gates] = None, epilogue=None) -> None:
    """Backward of heads + LSTM stack on (current stream, aux).  `prologue` (optional) is enqueued at the head
    of the first main-stream segment (used to fuse encoder forward + loss into it); `after_first` (optional) is
    called on the host right after that first segment has been enqueued (the decoder is enqueued there: early
    enough to finish beside the sweep, but never ahead of the critical chain).  The current stream joins `aux`
    before returning."""
    plan = EncoderBackwardPlan(enc, ws, d)
    main = torch.cuda.current_stream()
    if aux is None:
        def everything():
            if prologue:
                prologue()
            plan.heads(0)
            plan.sweep(0, plan.S)
            plan.wgrad(0, ws.T, True, True)
        run("enc_all", everything, main)
        if after_first:
            after_first()
        return
    if gates is not None:
        _encoder_backward_gated(plan, ws, aux, aux2, run, prologue, after_first, gates, epilogue)
        return
    for c, (s0, s1, t_lo, t_hi, first, last) in enumerate(plan.chunks):
        def main_seg(c=c, s0=s0, s1=s1):
            if c == 0:
                if prologue:
                    prologue()
                plan.heads(1)
            plan.sweep(s0, s1, None, c)

        split_tail = last and aux2 is not None  # the last chunk is the exposed tail: run its two halves side by side

        def aux_seg(c=c, t_lo=t_lo, t_hi=t_hi, first=first, last=last, split_tail=split_tail):
            if c == 0:
                plan.heads(2)
            plan.wgrad(t_lo, t_hi, first, last, 1 if split_tail else 3)

        run(f"main{c}", main_seg, main)
        if c == 0 and after_first:
            after_first()
        ev = torch.cuda.Event()
        ev.record(main)
        aux.wait_event(ev)
        if split_tail:
            aux2.wait_stream(aux)  # the token table accumulates across chunks: follow the earlier chunks
            aux2.wait_event(ev)
            run(f"aux2_{c}", lambda a=t_lo, b=t_hi, f=first, l=last: plan.wgrad(a, b, f, l, 2), aux2)
        run(f"aux{c}", aux_seg, aux)
    main.wait_stream(aux)
    if aux2 is not None:
        main.wait_stream(aux2)


def _tables_on_main(plan, ws) -> bool:
    """Every BPTT chunk folds its own token table and main forms the last chunk's (see _encoder_backward_gated)."""
    # (beside the persistent reduce-scatter sweep only: with the per-step launches of larger batches the accumulate-on-side
    # form is the faster one -- bs 256: 3.321 vs 3.354 ms)
    return (ws is not None and hasattr(ws, "dtable1") and os.environ.get("ARCVAE_TABLE_ON_MAIN", "1") != "0"
            and bool(getattr(plan, "persistent", False)) and persistent_forward_ok(ws, plan.d))


def _encoder_backward_gated(plan: EncoderBackwardPlan, ws: Workspace, aux, aux2, run, prologue, after_first,
                            g: Gates, epilogue=None) -> None:
    """encoder_backward with device-side gates instead of event waits (class Gates).  Main's signals of a step:
    #1 inputs ready (enqueued by the caller), #(2+c) after sweep chunk c, the last chunk topping P up to a multiple
    of STRIDE.  aux: chunk c runs behind signal #(2+c); it signals Q once (after its last-but-one chunk).  side:
    the tail chunk's token-table half runs behind Q (the table accumulates across aux's earlier chunks) and the last
    signal.  aux and side report their last piece of the step on R; the current stream joins them with a gate
    on R at the very end (the decoder segment precedes side's tail piece in stream order, so R covers it too)."""
    main = torch.cuda.current_stream()
    nc = len(plan.chunks)
    if nc + 1 > g.STRIDE:
        raise ValueError("too many BPTT chunks for the gate stride")
    if getattr(plan, "fused", False) and nc == 1 and aux2 is not None:
        _encoder_backward_gated_fused(plan, ws, aux, aux2, run, prologue, after_first, g)
        return
    tail_on_side = aux2 is not None and nc >= 2
    # Since the persistent sweeps the weight-gradient stream, not the chain, ends the step (aux is busy without a gap
    # from the end of chunk 0 to the end of the step): the token-table half of EVERY chunk goes to side, which is idle
    # once the decoder is done (ARCVAE_TABLE_ON_SIDE=0: only the tail chunk's, as before).
    table_on_side = tail_on_side and os.environ.get("ARCVAE_TABLE_ON_SIDE", "1") != "0"
    # ... and, beside a persistent sweep, the dWx_l GEMMs and the bias sums too (a third of aux's GEMM work; disjoint
    # outputs): two grouped launches in flight.  With the four chunks of the launch-based sweep this was 1 % slower
    # (1.084-1.088 vs 1.073 ms); with the two late chunks of the persistent sweep it is what makes them pay
    # (ARCVAE_WX_ON_SIDE=0/1 overrides).
    # (round 3: also with the mid-batch step kernels from 256 rows per GPU on -- bs 256: 3.081 -> 3.042 ms, bs 512: 6.47 -> 6.27;
    # no difference at 160 / 192 rows)
    mid_batch = ws is not None and ws.B >= 256 and not (_lib.load().arcvae_enc_lstm_tiled_for(ws.B, plan.d.H, plan.d.L, _lstm_flags(ws)) & 2)
    wx_on_side = table_on_side and os.environ.get("ARCVAE_WX_ON_SIDE", "1" if (plan.persistent or mid_batch) else "0") != "0"
    # Round 2: the token-table path is linear, so every chunk folds its OWN table (zero, one-hot GEMM, fold: first = last
    # = True) -- and the LAST chunk's, the only one in the exposed tail, is formed by main itself right behind the sweep
    # (main is idle there and needs no gate for its own sweep), in parallel with the last dWx on side and dWh on aux:
    # side's tail was a chain of four kernels (~100 us).  Needs the one-hot rows from arcvae_enc_prologue (persistent
    # forward path).  ARCVAE_TABLE_ON_MAIN=0: round 1's accumulate-on-side form.
    own_tables = table_on_side and _tables_on_main(plan, ws)

    def pre_zeroed() -> int:
        # parts bit 8: this step's prologue has zeroed both token tables and each is used once (two chunks): no zero-fill
        # launch in front of the one-hot GEMM -- 5 us of main's exposed tail, and on side a dispatch that waited ~40 us
        # for CU resources beside the sweep (profiles/r02_tail_timeline.txt)
        return 256 if (nc == 2 and ws is not None and getattr(ws, "tables_zeroed", False)) else 0

    def main_seg():
        # the whole critical chain of the backward as ONE captured segment: the "chunk c done" signal is raised by
        # the first launch of chunk c+1 when it starts (no launch of its own, no segment seam); only the last
        # chunk's signal is a kernel, topping P up to a multiple of STRIDE
        if prologue:
            prologue()
        plan.heads(1)
        for c, (s0, s1, _t_lo, _t_hi, _first, _last) in enumerate(plan.chunks):
            plan.sweep(s0, s1, g.word(g.P) if c > 0 else g.word(g.H), c)   # chunk 0's first launch: "the sweep has started"
        g.signal(g.P, g.STRIDE - nc)
        if own_tables:
            _s0, _s1, t_lo_l, t_hi_l, _f, _l = plan.chunks[-1]
            plan.wgrad(t_lo_l, t_hi_l, True, True, 2 | 32 | pre_zeroed(), table=getattr(ws, "dtable1", None))
        if epilogue is not None:
            # the step's finish (join of aux / side, loss finalize, the encoder's Adam) as the END of this segment: one graph
            # launch less on main (~9 us of seam between two graph launches in the exposed tail: profiles/r04_timeline_*.txt)
            epilogue()

    if epilogue is not None:
        # (its join is released by segments the host enqueues below: never run eagerly + synchronised -- SegmentRunner.eager_first)
        run("main+finish", main_seg, main, eager_first=False)
    else:
        run("main", main_seg, main)
    if after_first:
        after_first()
    for c, (s0, s1, t_lo, t_hi, first, last) in enumerate(plan.chunks):
        def aux_seg(c=c, t_lo=t_lo, t_hi=t_hi, first=first, last=last):
            # ARCVAE_HEADS_EARLY=1 (opt-in): form the heads' parameter gradients at the sweep's START (gate word H) instead of
            # behind chunk 0.  It paid while they were eight launches that dribbled in front of chunk 0's GEMMs (1.024 ->
            # 1.017 ms at bs 64; but 1.78 -> 1.96 at bs 128); as ONE launch (heads_wgrad_kernel: 1.025 -> 1.006 ms) the
            # two orders measure the same at bs 64 and the old one is ahead at bs 128 (1.765 vs 1.795), so it stays.
            heads_early = os.environ.get("ARCVAE_HEADS_EARLY", "0") != "0"
            if c == 0 and heads_early:
                # the dcomb chain in front of the sweep is complete once the sweep has started
                g.wait(g.H, g.NA, 1, 1)
                plan.heads(2)
                g.signal(g.HG, 1)
            g.wait(g.P, g.NA, g.STRIDE, g.STRIDE if last else 2 + c, advance=last)
            if c == 0 and not heads_early:
                plan.heads(2)
                g.signal(g.HG, 1)   # (every gated step, single process too: HG stays in lockstep with side's ticket counter)
            plan.wgrad(t_lo, t_hi, first, last,
                       8 if wx_on_side else (1 if (table_on_side or (last and tail_on_side)) else 3))
            if c == max(nc - 2, 0):
                g.signal(g.Q, 1)
            if last:
                g.signal(g.R, 1 if tail_on_side else 2)

        def side_seg(c=c, t_lo=t_lo, t_hi=t_hi, first=first, last=last):
            if table_on_side:   # the table accumulates on side only: chunk c behind main's signal #(2+c), as aux
                g.wait(g.P, g.NS, g.STRIDE, g.STRIDE if last else 2 + c, advance=last)
            else:
                g.wait(g.Q, g.NS, 1, 1)
                g.wait(g.P, g.NS, g.STRIDE, g.STRIDE, advance=True)
            if own_tables:   # this chunk's own table (not the last chunk's: main forms that one)
                plan.wgrad(t_lo, t_hi, True, True, ((4 if wx_on_side else 0) | (0 if last else 2)) | 32 | pre_zeroed())
            else:
                plan.wgrad(t_lo, t_hi, first, last, 6 if wx_on_side else 2)
            if last:
                g.signal(g.R, 1)

        # Host order: aux's segment of a chunk BEFORE side's.  aux waits for main's signals only; side may also wait for aux's (Q; under
        # data parallelism HG, "the heads' gradients are formed", in front of the early heads-bucket reduce) -- and a segment's FIRST
        # use runs eagerly and then synchronises its stream for the capture (SegmentRunner), so whatever could release a gate of that
        # stream has to be enqueued by then (round 4: with side first, the first data-parallel step sat in side.synchronize() behind
        # the HG gate until its bounded spin expired).  tests/test_gate_protocol.py replays this first step too.
        run(f"aux{c}", aux_seg, aux)
        if table_on_side or (last and tail_on_side):
            run(f"aux2_{c}", side_seg, aux2)
    if aux2 is not None and not tail_on_side:
        main.wait_stream(aux2)  # single-chunk sweeps (T <= 3): side only ran the decoder and reported nothing
    # The caller's next main-stream segment must BEGIN with `Gates.join(g)`: main polls R there (a gate reacts within
    # ~2 us; an event wait on two other queues cost ~20 us of the exposed tail).


NO_GUARDS = (C.c_void_p(0), C.c_void_p(0))


def _encoder_backward_gated_fused(plan: EncoderBackwardPlan, ws: Workspace, aux, aux2, run, prologue, after_first,
                                  g: Gates) -> None:
    """The gated backward when the sweep forms the weight gradients itself: ONE sweep launch, no chunks.  main: loss +
    dcomb + sweep (raises signal #2 when it starts) + token-table fold + top-up of P.  aux: behind signal #2 (the heads'
    dcomb chain is done) the heads' parameter gradients -- small GEMMs beside the first ticks --, then R.  side: its decoder
    segment (enqueued by `after_first`, ticket advanced there), then R.  The caller's next main segment begins with
    Gates.join (R == 2 per step), exactly as in the chunked form."""
    (s0, s1, t_lo, t_hi, first, last), = plan.chunks

    def main_seg():
        if prologue:
            prologue()
        plan.heads(1)
        plan.sweep(s0, s1, g.word(g.P), 0)
        plan.wgrad(t_lo, t_hi, first, last, 2)        # token table -> embedding.weight, lstm_layer_0.{Wx, bias}
        g.signal(g.P, g.STRIDE - 2)

    def aux_seg():
        g.wait(g.P, g.NA, g.STRIDE, 2, advance=True)
        plan.heads(2)
        g.signal(g.HG, 1)
        g.signal(g.Q, 1)
        g.signal(g.R, 1)

    run("main", main_seg, torch.cuda.current_stream())
    if after_first:
        after_first()
    run("aux2_0", lambda: g.signal(g.R, 1), aux2)     # behind the decoder segment in side's stream order
    run("aux0", aux_seg, aux)


def adam_update(store: ParamStore, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8,
                guards=NO_GUARDS) -> None:
    """trainer.py:320,324 -> MLX optim.Adam (no bias correction, Q7) over the module's flat buffer.
    guards: two device error words (StepEngine.guards); the kernel skips the update when either is non-zero."""
    call("arcvae_adam_update", ptr(store.flat), ptr(store.grad), ptr(store.adam_m), ptr(store.adam_v),
         C.c_long(store.numel_padded), float(lr), float(b1), float(b2), float(eps), guards[0], guards[1], stream_ptr())


# --------------------------------------------------------------------------------------------
class StepEngine:
    """Owns workspaces, streams and captured launch segments for one (encoder, decoder) pair.

    Streams:  main (caller's current stream): encoder forward -> loss -> BPTT (the critical chain)
              side: dense decoder forward/backward (+ its Adam)        aux: parameter-gradient GEMMs
    `mode`:   "segments" (default) per-stream linear hipGraph segments with eager events between them;
              "eager" plain launches; "graph" the whole step as ONE forked hipGraph (kept for comparison).
    """

    def __init__(self, enc: ParamStore, dec: ParamStore, dims: ModelDims, precision: Optional[str] = None):
        dims.validate()
        _lib.load()  # fail loudly when the extension is missing
        self.enc, self.dec, self.d = enc, dec, dims
        # "fp32" (default): the parity path (1e-4 against the oracle).  "bf16": throughput mode (SURVEY.md section 8(d)
        # Config 2, "bf16-in/fp32-acc") -- every matrix product off the latency-bound chain takes bf16 operands with f32
        # accumulation: the decoder's B*V-row GEMMs, the weight-gradient GEMMs, and the LSTM sweeps where they run on the
        # register-tiled step kernels; parameters, optimizer state, gates, cell state, losses stay f32.  Not a parity path:
        # its tolerance is stated in tests/test_bf16_mode_gpu.py and DESIGN.md section 10b.
        self.precision = precision or os.environ.get("ARCVAE_PRECISION", "fp32")
        if self.precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {self.precision!r}")
        self.device = enc.device
        self._ws: Dict[Tuple[int, int, bool], Workspace] = {}
        self._graphs: Dict[Tuple, torch.cuda.CUDAGraph] = {}
        self._runners: Dict[Tuple, SegmentRunner] = {}
        # the critical chain runs on the caller's stream (stream priorities made no measurable difference)
        self.side = torch.cuda.Stream(device=self.device)
        self.aux = torch.cuda.Stream(device=self.device)
        self.hyper_host = dict(beta=0.4, lambda_collapse=0.01, lambda_mi=0.0, target_mi=4.85, free_bits=0.5)
        self.mode = "segments"
        self.ev_chain = torch.cuda.Event()
        self.ev_dec_bwd = torch.cuda.Event()
        self.ev_enc_fwd = torch.cuda.Event()
        # device-side gates instead of event waits in the single-process step (class Gates); ARCVAE_GATES=0: events
        self.gates: Optional[Gates] = Gates(self.device) if os.environ.get("ARCVAE_GATES", "1") != "0" else None
        self._gating: Dict[int, bool] = {}
        self.dp = None     # arcvae_hip.dp.EngineDataParallel once api.enable_data_parallel was called for this pair

    # `use_graph` is the older boolean switch: True -> captured segments, False -> eager launches
    @property
    def use_graph(self) -> bool:
        return self.mode != "eager"

    @use_graph.setter
    def use_graph(self, v: bool) -> None:
        self.mode = "segments" if v else "eager"

    def _gating_ok(self, main: torch.cuda.Stream) -> bool:
        """Gates need main, side and aux on three different hardware queues: probe once per main stream, replacing
        side / aux by other pool streams until the probe passes; otherwise keep the event waits."""
        if self.gates is None:
            return False
        key = main.cuda_stream
        if key not in self._gating:
            g, ok = self.gates, False
            for _ in range(8):
                if not g.probe(self.side, main):
                    self.side = torch.cuda.Stream(device=self.device)
                    continue
                if not (g.probe(self.aux, main) and g.probe(self.side, self.aux) and g.probe(self.aux, self.side)):
                    self.aux = torch.cuda.Stream(device=self.device)
                    continue
                ok = True
                break
            if not ok:
                print("[arcvae_hip] no three streams on distinct hardware queues found: keeping event waits")
            self._gating = {key: ok}  # side/aux may have changed: other main streams are re-probed
        return self._gating[key]

    def guards(self, ws: Workspace):
        """The two device error words of a step: the gates' ERR counter and the persistent sweeps' sticky error word.
        The loss-finalize and Adam kernels read them ON THE DEVICE: a step whose stream order was lost gets NaN loss
        scalars (+ scalars[15] = 1) and its ENCODER update is skipped.  The decoder's update rides at the end of the
        decoder's own segment, ~1 ms before the sweeps finish: it is skipped when the words are already set by then (a
        failure of an earlier step, a forward sweep that gave up), but a word raised later in the same step finds it
        applied -- the two modules are then one update apart, which is why the trainer stops at that batch and refuses
        to checkpoint (`step_status`, read with the per-batch loss; `check_gates`)."""
        ga = self.gates.word(Gates.ERR) if self.gates is not None else C.c_void_p(0)
        return ga, C.c_void_p(ws.psync.data_ptr() + 4 * 500)

    def enable_trace(self, ws: Workspace) -> None:
        """Diagnostic: allocate the per-launch / per-tick stamp buffers of the two sweeps ({start, end} in 100 MHz
        ticks at [2s], [2s+1]); must precede the first (captured) step of this workspace."""
        L = self.d.L
        ws.trace_fwd = torch.zeros(2 * (ws.T + L + 1), dtype=torch.int64, device=self.device)
        ws.trace_bwd = torch.zeros(2 * (ws.T + 2 * L + 2), dtype=torch.int64, device=self.device)

    def check_gates(self) -> None:
        """Raise if a gate ever gave up waiting (results after that point are not ordered).  Host sync."""
        if self.gates is not None and self.gates.errors() != 0:
            raise _lib.ArcvaeHipError("a device-side gate expired (stream ordering was lost); set ARCVAE_GATES=0")
        for ws in self._ws.values():
            if int(ws.psync[500].item()) != 0:
                raise _lib.ArcvaeHipError("the persistent forward sweep gave up waiting (no 32 resident blocks per XCD "
                                          "or a stalled block); set ARCVAE_PERSIST=0")

    # ---- buffers ----------------------------------------------------------------------------
    def workspace(self, B: int, T: int, train: bool = True) -> Workspace:
        key = (B, T, train)
        if key not in self._ws:
            if (B, T, True) in self._ws:  # a training workspace also serves forward-only calls
                return self._ws[(B, T, True)]
            ws = Workspace(self.d, B, T, self.device, train)
            ws.bf16 = self.precision == "bf16"
            # the decoder's GEMMs run beside the forward sweep: in their three-piece form where that sweep is the persistent
            # one-row-group kernel (0.991 -> 0.976 ms at bs 64; with more rows per GPU the exact-f32 tile GEMM is ahead:
            # bs 128 1.730 vs 1.758, bs 256 3.27 vs 3.38; ARCVAE_DEC_SPLIT3=0/1 overrides)
            ws.dec_split3 = os.environ.get("ARCVAE_DEC_SPLIT3", "1" if (B <= 64 and persistent_forward_ok(ws, self.d)) else "0") == "1"
            ws.bf16_parts = int(os.environ.get("ARCVAE_BF16_PARTS", "7")) if ws.bf16 else 0   # ablation: 1 sweeps, 2 decoder, 4 weight gradients
            if (train and (ws.bf16_parts & 5) == 5 and B % 16 == 0 and os.environ.get("ARCVAE_BF16_OCT", "1") != "0"
                    and _lib.load().arcvae_enc_lstm_tiled_for(B, self.d.H, self.d.L, _lstm_flags(ws)) == 3
                    and not persistent_forward_ok(ws, self.d) and not bptt_reduce_scatter_ok(ws, self.d)
                    and _lib.load().arcvae_enc_lstm_bwd_persistent_ok(B, T, self.d.H, self.d.L) != 1):
                # bf16 copies of hseq / dG in the weight-gradient kernel's operand layout, written by the tiled sweeps
                ws.h_oct = torch.zeros(self.d.L * T * B * self.d.H, dtype=torch.bfloat16, device=self.device)
                ws.dG_oct = torch.zeros(self.d.L * T * B * 4 * self.d.H, dtype=torch.bfloat16, device=self.device)
            # Operand rings of the launch-based sweeps: where both run on the three-piece tile kernels the library keeps ALL T time
            # slots of the operand planes (arcvae_enc_lstm_operand_slots), and the weight-gradient GEMMs read those planes
            # directly (arcvae_enc_lstm_wgrad parts bit 11: no f32 loads, no re-splitting) -- configs[2], the 2048-row leg
            lib = _lib.load()
            slots = lib.arcvae_enc_lstm_operand_slots(B, T, self.d.H, self.d.L, _lstm_flags(ws))
            # (sizes from the library -- arcvae_enc_lstm_ws_floats: ring slots x slab, three-piece planes at 3/2 -- and every sweep
            # call states these capacities back: a knob changed after this point is an ARCVAE_ERR_ARG, not an overrun)
            need = (C.c_long * 4)()
            _lib.check(lib.arcvae_enc_lstm_ws_floats(B, T, self.d.H, self.d.L, _lstm_flags(ws), need), "arcvae_enc_lstm_ws_floats")
            if need[0] > ws.hseq_t.numel():
                ws.hseq_t = torch.empty(self.d.L, slots, need[0] // (self.d.L * slots), dtype=torch.float32, device=self.device)
            if train and need[1] > ws.dG_t.numel():
                ws.dG_t = torch.empty(self.d.L, slots, need[1] // (self.d.L * slots), dtype=torch.float32, device=self.device)
            assert need[2] <= ws.wt.numel() and (not train or need[3] <= ws.wT.numel())
            # ... and the dense decoder's layers 1 .. L-1 (zero-state cells over B*V rows) on the same kernels
            # (only where the encoder's sweeps are launch-based tile kernels themselves: beside a PERSISTENT sweep the tile kernels'
            # blocks -- 370-420 registers, 64-70 KB of LDS -- cannot be resident and the two serialise: bs 128 1.63 -> 2.00 ms)
            ws.dense_ws = None
            dense_force = os.environ.get("ARCVAE_DENSE_TILED", "1") == "2"
            # (throughput mode: the same stack on the bf16 tile kernels and the octet weight-gradient kernel, where its decoder and
            # sweep parts are both on: ARCVAE_BF16_PARTS bits 0 and 1)
            if (train and (not ws.bf16 or (ws.bf16_parts & 3) == 3) and lib.arcvae_dense_stack_ok(B * self.d.V, self.d.H, self.d.L) == 1
                    and (dense_force or (lib.arcvae_enc_lstm_tiled_for(B, self.d.H, self.d.L, _lstm_flags(ws)) == 3
                                         and not persistent_forward_ok(ws, self.d) and not bptt_reduce_scatter_ok(ws, self.d)))):
                n = C.c_long(0)
                lib.arcvae_dense_stack_ws_floats(B * self.d.V, self.d.H, self.d.L, C.byref(n))
                ws.dense_ws = torch.empty(n.value, dtype=torch.float32, device=self.device)
            ws.planes = bool(train and slots == T and _lstm_flags(ws) == _lib.LSTM_SPLIT3 and B % 32 == 0
                             and lib.arcvae_enc_lstm_tiled_for(B, self.d.H, self.d.L, _lstm_flags(ws)) == 3
                             and os.environ.get("ARCVAE_WGRAD_PLANES", "1") != "0"
                             and not persistent_forward_ok(ws, self.d) and not bptt_reduce_scatter_ok(ws, self.d)
                             and lib.arcvae_enc_lstm_bwd_persistent_ok(B, T, self.d.H, self.d.L) != 1)
            # Mid-size batches (the BPTT on per-step launches, not tiled), opt-in (ARCVAE_WGRAD_CONVERT=1; 2: any batch, tests):
            # scratch plane rings for the weight-gradient calls, which then split dG / h themselves and run the plane GEMMs
            # (arcvae_enc_lstm_wgrad parts bits 11 | 12).  Parity-green, measured SLOWER than the exact-f32 tile GEMMs there:
            # bs 256 3.28 vs 3.04 ms, bs 512 6.46 vs 6.12 -- the plane GEMM's blocks (144 KB of LDS, two waves per SIMD) and
            # the split pass take more from the sweep beside them than the matrix time they save.
            ws.pl_h = ws.pl_g = None
            conv = os.environ.get("ARCVAE_WGRAD_CONVERT", "0")
            if (train and not ws.planes and not ws.bf16 and B % 32 == 0 and self.d.H % 64 == 0 and conv != "0"
                    and (conv == "2" or B >= 256) and not (lib.arcvae_enc_lstm_tiled_for(B, self.d.H, self.d.L, _lstm_flags(ws)) & 2)
                    and (B > 128 or not bptt_reduce_scatter_ok(ws, self.d))):
                ws.pl_h = torch.empty(self.d.L * T * B * self.d.H * 3 // 2, dtype=torch.float32, device=self.device)
                ws.pl_g = torch.empty(self.d.L * T * B * 4 * self.d.H * 3 // 2, dtype=torch.float32, device=self.device)
            self._ws[key] = ws
            self._probe_persistent(ws)
        return self._ws[key]

    def _probe_persistent(self, ws: Workspace) -> None:
        """The persistent forward sweep assumes 8 XCDs x 32 CUs with one resident block per CU.  Dry-run it once per
        process on the first eligible workspace; if any block gives up (a partitioned or CU-masked device, a different
        part), switch the whole process to the per-step launches."""
        if getattr(StepEngine, "_persist_probed", False) or not persistent_forward_ok(ws, self.d):
            return
        StepEngine._persist_probed = True
        torch.cuda.synchronize()
        ws.x_tb.zero_()
        encoder_forward(self.enc, ws, self.d, float(self.hyper_host["free_bits"]))
        torch.cuda.synchronize()
        if int(ws.psync[500].item()) != 0:
            print("[arcvae_hip] persistent forward sweep not usable on this device (blocks per XCD != 32?): "
                  "falling back to per-step launches")
            os.environ["ARCVAE_PERSIST"] = "0"
            ws.psync.zero_()

    def set_hyper(self, ws: Workspace, **kw) -> None:
        h = dict(self.hyper_host)
        h.update(kw)
        self.hyper_host = h
        vals = (h["beta"], h["lambda_collapse"], h["lambda_mi"], h["target_mi"], h["free_bits"])
        if ws._hyper_vals != vals:  # epoch-scheduled values change rarely: skip the H2D copy otherwise
            ws.hyper.copy_(torch.tensor(list(vals) + [0.0, 0.0, 0.0], dtype=torch.float32))
            ws._hyper_vals = vals

    def load_inputs(self, ws: Workspace, x, cond, eps=None, coins=None) -> None:
        dev = self.device
        # device-resident inputs of the workspace's own dtypes: ONE copy launch instead of four (csrc/misc.hip)
        pairs = [(x, ws.x), (cond, ws.cond)] + ([(eps, ws.eps)] if eps is not None else []) + \
                ([(coins, ws.coins)] if coins is not None else [])
        if all(isinstance(a, torch.Tensor) and a.device == b.device and a.dtype == b.dtype and a.is_contiguous()
               and a.numel() == b.numel() for a, b in pairs):
            src, _k1 = ptr_array([a for a, _ in pairs])
            dst, _k2 = ptr_array([b for _, b in pairs])
            nb = (C.c_long * len(pairs))(*[b.numel() * b.element_size() for _, b in pairs])
            call("arcvae_copy_buffers", src, dst, nb, len(pairs), stream_ptr())
            return
        xt = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
        ws.x.copy_(xt.to(device=dev, dtype=torch.int32))
        ct = torch.as_tensor(np.asarray(cond) if not isinstance(cond, torch.Tensor) else cond)
        ws.cond.copy_(ct.to(device=dev, dtype=torch.float32).reshape(ws.B, self.d.C))
        if eps is not None:
            et = torch.as_tensor(np.asarray(eps) if not isinstance(eps, torch.Tensor) else eps)
            ws.eps.copy_(et.to(device=dev, dtype=torch.float32))
        if coins is not None:
            ck = torch.as_tensor(np.asarray(coins).astype(np.uint8))
            ws.coins.copy_(ck.to(dev))

    def runner(self, ws: Workspace, lr: float, global_rows: int, capture: bool) -> SegmentRunner:
        key = (ws.B, ws.T, float(lr), float(self.hyper_host["free_bits"]), int(global_rows), bool(capture))
        if key not in self._runners:
            self._runners[key] = SegmentRunner(capture)
        return self._runners[key]

    # ---- the phases of a step (data-parallel collectives go between them, dp.py) ------------------------
    def enqueue_decoder(self, ws: Workspace, global_rows: int, run=_inline, backward: bool = True,
                        wait_current: bool = True, split_events: bool = True, gate=None,
                        adam_lr: Optional[float] = None) -> None:
        """Dense decoder on the side stream: forward + TF walk + CE row sums (ev_chain), then its whole backward
        (ev_dec_bwd).  Independent of the encoder (Q2); it only has to follow the input copies."""
        d = self.d
        if wait_current:
            self.side.wait_stream(torch.cuda.current_stream())

        def dec_fwd():
            if gate is not None:  # (gates, advance): wait for main's "inputs ready" signal of this step
                gate[0].wait(Gates.P, Gates.NS, Gates.STRIDE, 1, advance=gate[1])
            if backward:
                self.dec.grad.zero_()
            decoder_forward_dense(self.dec, ws, d, keep_gpre=backward)
            decoder_chain(ws, d)

        def dec_bwd():
            decoder_backward(self.dec, ws, d, 1.0 / (global_rows * ws.T))
            if adam_lr is not None:
                adam_update(self.dec, adam_lr, guards=self.guards(ws))
            if gate is not None:
                # decoder gradients and CE row sums of this step are complete (read by the data-parallel step, which
                # reduces them early; raised in every gated step so that D stays in lockstep with main's step count)
                gate[0].signal(Gates.D, 1)

        if backward and not split_events:  # single process: one side-stream segment (one graph launch less)
            run("dec_all", lambda: (dec_fwd(), dec_bwd()), self.side)
            self.ev_chain.record(self.side)
            self.ev_dec_bwd.record(self.side)
            return
        run("dec_fwd", dec_fwd, self.side)
        self.ev_chain.record(self.side)
        if backward:
            run("dec_bwd", dec_bwd, self.side)
            self.ev_dec_bwd.record(self.side)

    def _enc_fwd(self, ws: Workspace, backward: bool, start_signal=None, seam: int = 1) -> None:
        encoder_forward(self.enc, ws, self.d, float(self.hyper_host["free_bits"]), start_signal, zero_grad=backward, seam=seam)

    def enqueue_encoder_forward(self, ws: Workspace, run=_inline, backward: bool = True, start_signal=None) -> None:
        """Encoder forward on the current stream; leaves this process's partial latent `stats` (the CE slot
        stats[2Z+3] is filled later by enqueue_finish).  start_signal: gate word raised when the sweep starts."""
        run("enc_fwd", lambda: self._enc_fwd(ws, backward, start_signal), torch.cuda.current_stream())
        self.ev_enc_fwd.record(torch.cuda.current_stream())

    def enqueue_backward(self, ws: Workspace, run=_inline, fuse_forward: bool = False, after_first=None,
                         gates: Optional[Gates] = None, start_signal=None, epilogue=None) -> None:
        """`stats[:2Z+3]` holds GLOBAL sums (or, with fuse_forward, will: single process): latent loss scalars
        and gradients, then the encoder backward.  Does NOT wait for the decoder."""
        fb = float(self.hyper_host["free_bits"])

        def prologue():
            if fuse_forward:
                self._enc_fwd(ws, True, start_signal, seam=3)      # (fused seam: loss + dcomb chain in the same launch)
            elif seam_fused_ok(ws, self.d):
                encoder_seam(self.enc, ws, self.d, fb, 2)          # data-parallel step: `stats` is global by now
            if not ws.seam_bwd_done:
                latent_loss(ws, self.d, fb, True)

        # the tail chunk's token-table half goes to the SIDE stream: the decoder finished long ago and, unlike a
        # fourth stream, `side` owns a hardware queue of its own (HIP maps streams onto 4 queues), so the two
        # halves really run side by side
        encoder_backward(self.enc, ws, self.d, aux=self.aux, run=run, prologue=prologue, after_first=after_first,
                         aux2=self.side, gates=gates, epilogue=epilogue)

    def enqueue_recon(self, ws: Workspace, run=_inline) -> None:
        """stats[2Z+3] = sum of this process's CE row sums (after the decoder's TF walk)."""
        main = torch.cuda.current_stream()
        main.wait_event(self.ev_chain)
        run("recon", lambda: self._recon(ws), main)

    def _recon(self, ws: Workspace) -> None:
        call("arcvae_stats_set_recon", ptr(ws.rowloss), ws.B, ptr(ws.stats), self.d.Z, stream_ptr())

    def enqueue_finish(self, ws: Workspace, lr: float, update: bool, run=_inline, with_recon: bool = False,
                       dec_adam: bool = True, join_side: bool = True, gates: Optional[Gates] = None) -> None:
        """[join ->] [CE sum ->] recon/total scalars, both Adam updates: one segment."""
        main = torch.cuda.current_stream()
        if join_side:
            main.wait_stream(self.side)

        fin = self._finish_ops(ws, lr, update, with_recon, dec_adam, gates)
        run(("finish" if update else "finish_noupdate") + ("_r" if with_recon else ""), fin, main)

    def _finish_ops(self, ws: Workspace, lr: float, update: bool, with_recon: bool, dec_adam: bool, gates: Optional[Gates]):
        """The launches of a step's finish as a closure (its own segment, or the end of the main segment: _enqueue_step)."""
        def fin():
            if gates is not None:
                gates.join()
            ga, gb = self.guards(ws)
            if with_recon and update:
                # CE sum + recon / total scalars ride in block 0 of the encoder's Adam launch (round 4: one launch less in the tail)
                if dec_adam:
                    adam_update(self.dec, lr, guards=(ga, gb))
                st = self.enc
                call("arcvae_adam_update_finalize", ptr(st.flat), ptr(st.grad), ptr(st.adam_m), ptr(st.adam_v),
                     C.c_long(st.numel_padded), float(lr), 0.9, 0.999, 1e-8, ga, gb, ptr(ws.rowloss), ws.B, ptr(ws.stats),
                     ptr(ws.scalars), self.d.Z, ws.T, stream_ptr())
                return
            if with_recon:   # CE sum + recon/total scalars in one launch
                call("arcvae_recon_finalize", ptr(ws.rowloss), ws.B, ptr(ws.stats), ptr(ws.scalars), self.d.Z, ws.T,
                     ga, gb, stream_ptr())
            else:
                call("arcvae_loss_finalize", ptr(ws.stats), ptr(ws.scalars), self.d.Z, ws.T, ga, gb, stream_ptr())
            if update:
                if dec_adam:
                    adam_update(self.dec, lr, guards=(ga, gb))
                adam_update(self.enc, lr, guards=(ga, gb))
        return fin

    def _enqueue_step(self, ws: Workspace, lr: float, global_rows: int, update: bool, run=_inline) -> None:
        """Single-process step.  Host enqueue order: the decoder segment (one short graph launch; it overlaps the
        forward sweep, whose launches leave CUs idle at every seam), the encoder forward, then the fused loss +
        dcomb + BPTT segments with the weight-gradient chunks on aux; the decoder is only joined at the very end
        (nothing on the encoder's backward path needs the reconstruction term)."""
        main = torch.cuda.current_stream()
        if self.mode != "graph" and self._gating_ok(main):
            g = self.gates
            nc = len(EncoderBackwardPlan(self.enc, ws, self.d).chunks)

            def grun(key, fn, stream, _run=run):             # gated segments are recorded under their own names
                _run(f"gated:{key}", fn, stream)

            def grun_e(key, fn, stream, eager_first=True, _run=run):
                if eager_first:
                    _run(f"gated:{key}", fn, stream)
                else:
                    _run(f"gated:{key}", fn, stream, eager_first=False)

            # Signal #1 ("inputs ready": releases the decoder on side) is raised by the forward sweep when it starts.
            # The main segment -- encoder forward + loss + dcomb + the whole BPTT chain, no seam in it -- is enqueued
            # first, the decoder right behind it (in the first, eager step the decoder's gate could otherwise be
            # waited for before its signal exists; with the persistent sweep its blocks should also be resident before
            # the decoder's GEMMs fill the CUs).  The decoder's Adam update rides at the end of its own segment (its
            # gradients are complete ~1 ms before the encoder's): one launch less in the exposed tail of the step.
            def dec_after_main():
                self.enqueue_decoder(ws, global_rows, grun, wait_current=False, split_events=False, gate=(g, nc < 2),
                                     adam_lr=lr if update else None)

            # From a runner's second step on, the finish rides at the end of the main segment (ARCVAE_MERGE_FINISH=0: never).  Not
            # in its first step -- that one runs every segment eagerly and synchronises, and the finish begins with a join that aux
            # and side release only later -- and not for single-chunk sweeps (their join is an event wait between the segments).
            steps_done = getattr(run, "steps", 0)
            if hasattr(run, "steps"):
                run.steps += 1
            merged = (nc >= 2 and steps_done >= 1 and getattr(run, "capture", False)
                      and not fused_wgrad_ok(ws, self.d) and os.environ.get("ARCVAE_MERGE_FINISH", "1") != "0")
            if merged:
                fin = self._finish_ops(ws, lr, update, True, False, g)
                self.enqueue_backward(ws, grun_e, gates=g, fuse_forward=True, after_first=dec_after_main,
                                      start_signal=g.word(g.P), epilogue=fin)
                return
            self.enqueue_backward(ws, grun, gates=g, fuse_forward=True, after_first=dec_after_main,
                                  start_signal=g.word(g.P))
            self.enqueue_finish(ws, lr, update, grun, with_recon=True, dec_adam=False, join_side=False, gates=g)
            return
        self.side.wait_stream(main)                          # the decoder only has to follow the input copies
        self.enqueue_decoder(ws, global_rows, run, wait_current=False, split_events=False)
        self.enqueue_encoder_forward(ws, run)
        self.enqueue_backward(ws, run)
        self.enqueue_finish(ws, lr, update, run, with_recon=True)

    # ---- public API --------------------------------------------------------------------------------
    def forward_loss(self, x, cond, eps, coins, **hyper) -> Dict[str, torch.Tensor]:
        """complete_vae_loss forward only (validation / logging path)."""
        B, T = int(x.shape[0]), int(x.shape[1])
        ws = self.workspace(B, T, train=False)
        self.set_hyper(ws, **hyper)
        self.load_inputs(ws, x, cond, eps, coins)
        # encoder first: the blocks of a persistent sweep should be resident before the decoder's GEMMs fill the CUs
        self.side.wait_stream(torch.cuda.current_stream())
        self.enqueue_encoder_forward(ws, backward=False)
        self.enqueue_decoder(ws, B, backward=False, wait_current=False)
        latent_loss(ws, self.d, float(self.hyper_host["free_bits"]), False)
        self.enqueue_recon(ws)
        ga, gb = self.guards(ws)
        call("arcvae_loss_finalize", ptr(ws.stats), ptr(ws.scalars), self.d.Z, ws.T, ga, gb, stream_ptr())
        torch.cuda.current_stream().wait_stream(self.side)
        return self._results(ws)

    def train_step(self, x, cond, eps, coins, lr: float, update: bool = True, **hyper) -> Dict[str, torch.Tensor]:
        """loss + grads (+ Adam) for one minibatch; single process (see dp.py for N ranks)."""
        B, T = int(x.shape[0]), int(x.shape[1])
        ws = self.workspace(B, T, train=True)
        self.set_hyper(ws, **hyper)
        self.load_inputs(ws, x, cond, eps, coins)
        self.run_step(ws, lr, update)
        return self._results(ws)

    def run_step(self, ws: Workspace, lr: float, update: bool = True) -> None:
        """Enqueue (or replay) the step on the current stream; inputs/hyper already in `ws`."""
        if self.mode == "graph":
            key = (ws.B, ws.T, float(lr), float(self.hyper_host["free_bits"]), bool(update))
            g = self._graphs.get(key)
            if g is None:
                self._enqueue_step(ws, lr, ws.B, False)  # warm-up once eagerly, then capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue_step(ws, lr, ws.B, update)
                self._graphs[key] = g
            g.replay()
            return
        run = self.runner(ws, lr, ws.B, capture=(self.mode == "segments"))
        if not update:  # gradient-only steps (tests, value_and_grad without lr) use their own segment set
            run = self.runner(ws, -1.0, ws.B, capture=(self.mode == "segments"))
        self._enqueue_step(ws, lr, ws.B, update, run)

    def _results(self, ws: Workspace) -> Dict[str, torch.Tensor]:
        out = {k: ws.scalars[i] for i, k in enumerate(SCALAR_KEYS)}
        out["mu"], out["logvar"], out["z"] = ws.mu, ws.logvar, ws.z
        out["step_status"] = ws.scalars[15]   # 1.0: a gate expired / a persistent sweep gave up (values NaN, no update)
        return out

    def gather_logits(self, ws: Workspace) -> torch.Tensor:
        out = torch.empty(ws.B, ws.T, self.d.V, dtype=torch.float32, device=self.device)
        call("arcvae_dec_gather_logits", ptr(ws.logits), ptr(ws.fed), ptr(out), ws.B, ws.T, self.d.V, stream_ptr())
        return out
