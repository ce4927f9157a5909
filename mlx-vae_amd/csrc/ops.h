// Internal host-side declarations shared between the translation units of libarcvae_hip.so.
// The public C ABI is include/arcvae_hip.h; everything here is also extern "C" so the
// symbols are the same ones the header declares.
#pragma once
#include "common.h"

#define ARCVAE_GEMM_ACCUMULATE 1
#define ARCVAE_GEMM_TANH 2
#define ARCVAE_GEMM_SPLITK 4
#define ARCVAE_GEMM_NO_SKINNY 8
#define ARCVAE_GEMM_TILE64 16   /* force 64x64 tiles (tuning / tests) */
#define ARCVAE_GEMM_TILE128 32  /* force 128x128 tiles */
#define ARCVAE_GEMM_DTANH 64    /* C = (A.B) * (1 - T^2), T = `bias` read as an [M,ldc] matrix (tanh backward) */
#define ARCVAE_GEMM_TILE_WIDE 128 /* split-bf16 TN path: 128-row tile (the caller knows no persistent sweep is resident) */
#define ARCVAE_GEMM_BF16 256      /* throughput mode: operands rounded to bf16, f32 accumulate (not a parity path) */
#define ARCVAE_GEMM_SPLIT3 512    /* three bf16 pieces per operand, six products: fp32-class accuracy on the bf16 matrix pipe */

#define ARCVAE_LSTM_RETILE 1      /* arcvae_enc_lstm_backward flags: write the BPTT weight layouts first */
#define ARCVAE_LSTM_BF16 2        /* arcvae_enc_lstm_forward / _backward flags: throughput mode (tiled regime only) */
#define ARCVAE_LSTM_SPLIT3 4      /* same places: three bf16 pieces per operand, six products (parity path; tiled regime only) */

#define ARCVAE_PERSIST_BF16 2     /* arcvae_enc_lstm_forward_persistent / _backward_persistent_rs flags bit 1: throughput mode */
#define ARCVAE_DEC_SPLIT3 512      /* same places: the B*V-row products with ARCVAE_GEMM_SPLIT3 (a parity path) */
#define ARCVAE_DEC_BF16 256        /* arcvae_dec_forward_dense `mode` bit / arcvae_dec_backward_dense `flags` bit: throughput mode */

// internal (C++ linkage): grouped weight-gradient GEMMs, see gemm.hip
int arcvae_gemm_tn_group_accum(int n, int M, int N, const int* K, const float* const* A, int lda,
                               const float* const* B, int ldb, float* const* C, int ldc, int allow_split,
                               float* const* colsum /* optional: colsum_i[M] += column sums of A_i */, hipStream_t stream);

// internal (C++ linkage): throughput mode, weight gradients from octet-major bf16 operand copies, see gemm.hip
int arcvae_wgrad_octet_group(int n, int M, int N, const int* K, const void* const* A, const void* const* B,
                             float* const* C, int ldc, hipStream_t stream);

// internal (C++ linkage): weight gradients of the tiled three-piece sweeps from their operand planes, see gemm.hip
int arcvae_wgrad_planes_group(int n, int M, int N, int rows, const void* const* A, const int* tA0, const void* const* B,
                              const int* tB0, const int* nT, float* const* C, int ldc, float* const* colsum /* optional */,
                              hipStream_t stream);

// internal (C++ linkage): a forward-only decoder layer (GEMM + zero-state cell, no pre-activations kept), see gemm.hip
int arcvae_gemm_cell_zero(int M, int H, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                          float* Hout, hipStream_t stream);
#define ARCVAE_DEC_NO_GPRE 1024    /* arcvae_dec_forward_dense `mode` bit 10: forward only, the layers' pre-activations are not kept */
#define ARCVAE_DEC_PART_HEAD 2048  /* arcvae_dec_forward_dense `mode` / _backward_dense `flags` bit 11: only the token table and layer 0 */
#define ARCVAE_DEC_PART_TAIL 4096  /* bit 12: only fc_out (logits + row statistics / its gradients + dh_top): layers 1 .. L-1 by the caller */

// internal (C++ linkage): two skinny products in one launch, see gemm.hip
int arcvae_gemm_skinny_pair(int transB, const int* M, const int* N, const int* K, const float* const* A, const int* lda,
                            const float* const* B, const int* ldb, float* const* C, const int* ldc,
                            const float* const* bias, const int* flags, hipStream_t stream);

extern "C" {
int arcvae_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, int flags,
                    hipStream_t stream);
// out[c] += scale * sum_r X[r*ld + c]
int arcvae_colsum_accum(const float* X, int rows, int cols, int ld, float* out, float scale,
                        hipStream_t stream);
// dst_i[c*rows_i + r] = src_i[r*cols_i + c] for i < n (n <= 16); pointer arrays are HOST arrays.
int arcvae_transpose_batched(const float* const* src, float* const* dst, const int* rows,
                             const int* cols, int n, hipStream_t stream);
// out[seg[r], :] += X[r, :]   (out is [nseg, cols], pre-initialised by the caller)
int arcvae_segsum_rows_accum(const float* X, const int32_t* seg, int rows, int nseg, int cols,
                             float* out, hipStream_t stream);
// k-chunk-major copies of LSTM weights (mode 0: forward layout, 1: transposed/backward layout); HOST arrays
int arcvae_tile_weights(const float* const* src, float* const* dst, const int* cols, const int* mode, int n, int H,
                        hipStream_t stream);
// x[r*ld + c] = 0
int arcvae_zero(float* x, int rows, int cols, int ld, hipStream_t stream);
// dEmb += dT . Wx0[:, :E];  dWx0[:, :E] += dT^T . emb;  db0 += colsum(dT)   (dT [V,4H], one launch)
int arcvae_table_finalize(const float* dT, const float* Wx0, int ldw, const float* emb, float* dEmb, float* dWx0,
                          float* db0, int V, int E, int G, hipStream_t stream);
// dst[t*B + b] = src[b*T + t]
int arcvae_transpose_tokens(const int32_t* src, int32_t* dst, int B, int T, hipStream_t stream);
}
