// 16x16 output tile per 256-thread block, K split over the block's 4 waves, on
// v_mfma_f32_16x16x4_f32 (exact f32 FMA chain, 32-cycle issue per SIMD).
//
// This is the latency-oriented contraction used where M (batch rows) is small and the
// op sits on a dependent chain: the encoder LSTM step (fwd and BPTT) and the encoder
// heads.  A launch spreads (M/16)*(N/16) tiles over the chip; inside a tile each wave
// owns one quarter of K, so the MFMA chain per wave is K/16 instructions long.
//
// MFMA 16x16x4 f32 operand maps (cdna guide section 3): lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D: col = l&15, row = 4*(l>>4)+reg.
// The k index inside one MFMA is only summed over, so any bijection between (step, l>>4)
// and the wave's K-slice is valid as long as A and B use the same one.  We give each
// 16-lane group q a CONTIGUOUS run of the slice, so a lane's A and B operands are plain
// float4 loads.
#pragma once
#include "common.h"

// Accumulate sum_k A[arow][k] * W[wrow][k] over this wave's quarter of [0,K) into acc0/acc1.
// A and W are K-contiguous rows (16-byte aligned, K % 64 == 0).  Lane (j = l&15, q = l>>4):
// arow = the A row of tile row (l&15) (clamped by the caller), wrow = the W row feeding tile
// column (l&15).
__device__ __forceinline__ void skinny_accum_kk(f32x4& acc0, f32x4& acc1,
                                                const float* __restrict__ A, long arow_off,
                                                const float* __restrict__ W, long wrow_off,
                                                int K, int wave, int lane) {
    const int Kw = K >> 2;          // per wave
    const int Kq = Kw >> 2;         // per 16-lane group (multiple of 4)
    const int kbase = wave * Kw + (lane >> 4) * Kq;
    const float4* ap = reinterpret_cast<const float4*>(A + arow_off + kbase);
    const float4* wp = reinterpret_cast<const float4*>(W + wrow_off + kbase);
    const int nchunk = Kq >> 2;
    int c = 0;
    for (; c + 4 <= nchunk; c += 4) {
        float4 a0 = ap[c], a1 = ap[c + 1], a2 = ap[c + 2], a3 = ap[c + 3];
        float4 b0 = wp[c], b1 = wp[c + 1], b2 = wp[c + 2], b3 = wp[c + 3];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, b2.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, b2.w, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.x, b3.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.y, b3.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.z, b3.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.w, b3.w, acc1, 0, 0, 0);
    }
    for (; c < nchunk; ++c) {
        float4 a0 = ap[c];
        float4 b0 = wp[c];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc1, 0, 0, 0);
    }
}

// Same contraction with W stored [K][N] (N-contiguous): lane reads W[k][wcol], one dword per MFMA.
__device__ __forceinline__ void skinny_accum_kn(f32x4& acc0, f32x4& acc1,
                                                const float* __restrict__ A, long arow_off,
                                                const float* __restrict__ W, int ldw, int wcol,
                                                int K, int wave, int lane) {
    const int Kw = K >> 2;
    const int Kq = Kw >> 2;
    const int kbase = wave * Kw + (lane >> 4) * Kq;
    const float4* ap = reinterpret_cast<const float4*>(A + arow_off + kbase);
    const float* wp = W + (long)kbase * ldw + wcol;
    const int nchunk = Kq >> 2;
#pragma unroll 2
    for (int c = 0; c < nchunk; ++c) {
        float4 a0 = ap[c];
        float b0 = wp[(long)(4 * c + 0) * ldw];
        float b1 = wp[(long)(4 * c + 1) * ldw];
        float b2 = wp[(long)(4 * c + 2) * ldw];
        float b3 = wp[(long)(4 * c + 3) * ldw];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b3, acc1, 0, 0, 0);
    }
}

// Write this wave's 16x16 partial tile to LDS as red[wave][row][col] (row-major, 16 cols).
__device__ __forceinline__ void skinny_store_partial(float* red, const f32x4& acc0, const f32x4& acc1,
                                                     int wave, int lane) {
    float* p = red + wave * 256 + ((lane >> 4) * 4) * 16 + (lane & 15);
    p[0] = acc0[0] + acc1[0];
    p[16] = acc0[1] + acc1[1];
    p[32] = acc0[2] + acc1[2];
    p[48] = acc0[3] + acc1[3];
}

// After __syncthreads(): full sum of tile element (row, col).
__device__ __forceinline__ float skinny_reduced(const float* red, int row, int col) {
    const int o = row * 16 + col;
    return (red[o] + red[256 + o]) + (red[512 + o] + red[768 + o]);
}
