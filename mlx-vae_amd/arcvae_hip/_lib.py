"""ctypes binding of libarcvae_hip.so (C ABI: include/arcvae_hip.h).

The library is the product path: there is NO CPU or eager-PyTorch fallback.  If the shared
object is missing, or a kernel entry point returns an error, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCVAE_HIP_LIB", os.path.join(_HERE, "libarcvae_hip.so"))  # override: A/B of two builds

GEMM_ACCUMULATE = 1
GEMM_TANH = 2
GEMM_SPLITK = 4
GEMM_NO_SKINNY = 8
GEMM_BF16 = 256      # throughput mode: bf16 operands, f32 accumulate (include/arcvae_hip.h ARCVAE_GEMM_BF16)
GEMM_SPLIT3 = 512    # three bf16 pieces per operand, six products: fp32-class accuracy (ARCVAE_GEMM_SPLIT3)
LSTM_RETILE = 1      # arcvae_enc_lstm_backward flags
LSTM_BF16 = 2        # arcvae_enc_lstm_forward / _backward flags: throughput mode (tiled regime)
LSTM_SPLIT3 = 4      # same places: three bf16 pieces per operand, six products -- a parity path (tiled regime)
RS_HALF, RS_HALF1 = 16, 32   # arcvae_enc_lstm_backward_persistent_rs flags: the half-batch form (129..256 rows as two 16-row-per-XCD sweeps)
PERSIST_BF16 = 2     # arcvae_enc_lstm_forward_persistent / _backward_persistent_rs flags: throughput mode (4x4x4 bf16 blocks)
DEC_BF16 = 256       # arcvae_dec_forward_dense mode bit / arcvae_dec_backward_dense flags bit
DEC_SPLIT3 = 512     # same places: three-piece (fp32-class) products
DEC_PART_HEAD = 2048  # arcvae_dec_forward_dense mode / _backward_dense flags bit: token table + layer 0 only
DEC_PART_TAIL = 4096  # ... fc_out only (layers 1 .. L-1 by arcvae_dense_stack_forward / _backward)
DEC_NO_GPRE = 1024   # arcvae_dec_forward_dense mode bit: forward only, pre-activations not kept (fused GEMM + cell)
WGRAD_BF16 = 128     # arcvae_enc_lstm_wgrad parts bit

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_l = C.c_long
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_long)

# name -> argtypes (restype is always int)
SIGNATURES = {
    "arcvae_abi_version": [_ip],
    "arcvae_gemm_f32": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp],
    "arcvae_transpose_tokens": [_vp, _vp, _i, _i, _vp],
    "arcvae_enc_lstm_ws_floats": [_i, _i, _i, _i, _i, _lp],
    "arcvae_enc_lstm_forward": [_vp, _vp, _pp, _pp, _pp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _lp, _vp, _vp, _vp],
    "arcvae_enc_lstm_tiled": [_i, _i, _i],
    "arcvae_enc_lstm_tiled_for": [_i, _i, _i, _i],
    "arcvae_enc_lstm_operand_slots": [_i, _i, _i, _i, _i],
    "arcvae_dense_stack_ok": [_l, _i, _i],
    "arcvae_dense_stack_ws_floats": [_l, _i, _i, _lp],
    "arcvae_dense_stack_forward": [_pp, _pp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _vp],
    "arcvae_dense_stack_backward": [_vp, _vp, _vp, _vp, _pp, _pp, _vp, _l, _l, _i, _i, _i, _vp],
    "arcvae_enc_lstm_persistent_ok": [_i, _i, _i, _i],
    "arcvae_enc_lstm_forward_persistent": [_vp, _vp, _pp, _pp, _pp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "arcvae_enc_prologue": [_vp, _vp, _vp, _l, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "arcvae_enc_lstm_bwd_persistent_ok": [_i, _i, _i, _i],
    "arcvae_enc_lstm_backward_persistent": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "arcvae_enc_lstm_bwd_rs_ok": [_i, _i, _i, _i],
    "arcvae_enc_lstm_bwd_rs_halves": [_i, _i, _i, _i],
    "arcvae_enc_lstm_persist_groups": [_i, _i, _i],
    "arcvae_enc_lstm_bwd_rs_part_floats": [_i, _i, _i, _i],
    "arcvae_enc_lstm_backward_persistent_rs": [_pp, _pp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "arcvae_enc_lstm_backward_fused": [_pp, _pp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _pp, _pp, _pp, _vp,
                                       _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "arcvae_enc_lstm_backward": [_pp, _pp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _lp, _vp, _vp, _vp, _vp],
    "arcvae_enc_lstm_wgrad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pp, _pp, _pp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _lp, _vp],
    "arcvae_enc_heads_forward": [_vp] * 19 + [_i, _i, _i, _i, _f, _i, _vp],
    "arcvae_enc_seam_ok": [_i, _i, _i],
    "arcvae_enc_seam": [_vp] * 22 + [_i, _i, _i, _i, _f, _i, _i, _vp],
    "arcvae_stats_set_recon": [_vp, _i, _vp, _i, _vp],
    "arcvae_latent_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "arcvae_loss_finalize": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "arcvae_recon_finalize": [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "arcvae_enc_heads_backward": [_vp] * 18 + [_i, _i, _i, _i, _i, _vp],
    "arcvae_dec_forward_dense": [_vp, _pp, _pp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "arcvae_dec_chain_ce": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "arcvae_dec_ce_backward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "arcvae_dec_gather_logits": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "arcvae_dec_sample_chain": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "arcvae_dec_sample_chain_categorical": [_vp, _vp, _vp, _i, _i, _i, _i, _f, C.c_ulonglong, _vp],
    "arcvae_dec_backward_dense": [_vp, _pp, _pp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _pp, _pp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "arcvae_reparameterize": [_vp, _vp, _vp, _vp, _l, _vp],
    "arcvae_latent_stats": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "arcvae_ce_rows": [_vp, _vp, _vp, _l, _i, _vp],
    "arcvae_sum": [_vp, _l, _vp, _f, _vp],
    "arcvae_adam_update": [_vp, _vp, _vp, _vp, _l, _d, _d, _d, _d, _vp, _vp, _vp],
    "arcvae_adam_update_finalize": [_vp, _vp, _vp, _vp, _l, _d, _d, _d, _d, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp],
    "arcvae_colsum_accum": [_vp, _i, _i, _i, _vp, _f, _vp],
    "arcvae_segsum_rows_accum": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "arcvae_transpose_batched": [_pp, _pp, _ip, _ip, _i, _vp],
    "arcvae_scale_inplace": [_vp, _l, _f, _vp],
    "arcvae_debug_occupy": [_i, _i, _i, _i, _vp],
    "arcvae_zero": [_vp, _i, _i, _i, _vp],
    "arcvae_table_finalize": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "arcvae_gate_wait": [_vp, _vp, C.c_uint, C.c_uint, _i, C.c_uint, _vp, _vp],
    "arcvae_gate_set": [_vp, C.c_uint, _i, _vp],
    "arcvae_tile_weights": [_pp, _pp, _ip, _ip, _i, _i, _vp],
    "arcvae_copy_buffers": [_pp, _pp, _lp, _i, _vp],
}

LONG_RESULTS = {"arcvae_enc_lstm_bwd_rs_part_floats"}      # size queries returning `long`; everything else returns an int code

_lib: Optional[C.CDLL] = None


class ArcvaeHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP extension; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArcvaeHipError(
            f"{LIB_PATH} not found: build it with `make -C mlx-vae_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_long if name in LONG_RESULTS else C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ArcvaeHipError(f"{what} failed with code {rc} "
                             "(-1 bad argument/shape, -2 launch error, -3 device)")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    """Raw device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise ArcvaeHipError("arcvae_hip kernels need device tensors (no CPU path)")
    if not t.is_contiguous():
        raise ArcvaeHipError("arcvae_hip kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else ptr(t).value
    return C.cast(arr, _pp), arr  # keep `arr` alive at the call site


def stream_ptr(stream: Optional[torch.cuda.Stream] = None) -> C.c_void_p:
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def gemm(transA: bool, transB: bool, M: int, N: int, K: int, A: torch.Tensor, lda: int, Bm: torch.Tensor,
         ldb: int, Cm: torch.Tensor, ldc: int, bias: Optional[torch.Tensor] = None, flags: int = 0,
         a_off: int = 0, b_off: int = 0, c_off: int = 0) -> None:
    """C[M,N] (+)= op(A) op(B) (+bias); *_off are element offsets into the tensors."""
    def p(t, off):
        return C.c_void_p(t.data_ptr() + 4 * off)
    call("arcvae_gemm_f32", int(transA), int(transB), M, N, K, p(A, a_off), lda, p(Bm, b_off), ldb,
         p(Cm, c_off), ldc, ptr(bias), flags, stream_ptr())
