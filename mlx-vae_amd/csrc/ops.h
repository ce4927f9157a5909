// Internal host-side declarations (C++ linkage) shared between the translation units of libarcvae_hip.so.
// The C ABI -- flag bits, error codes, every extern "C" prototype -- is include/arcvae_hip.h, which common.h includes: nothing
// of it is re-declared here.
#pragma once
#include "common.h"

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

// internal (C++ linkage): two skinny products in one launch, see gemm.hip
int arcvae_gemm_skinny_pair(int transB, const int* M, const int* N, const int* K, const float* const* A, const int* lda,
                            const float* const* B, const int* ldb, float* const* C, const int* ldc,
                            const float* const* bias, const int* flags, hipStream_t stream);

