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
// and the wave's K-slice is valid as long as A and B use the same one.  We cut the slice
// into 16-float chunks and give 16-lane group q floats [4q, 4q+4) of every chunk: a lane's A
// and B operands are plain float4 loads AND the four q-lanes of a row read 64 contiguous
// bytes per instruction (two chunks = one 128-B line), i.e. 16 lines per wave-instruction
// instead of 64.  (The first version gave each q a contiguous run of the slice -- 64 distinct
// lines per instruction, TA-bound: 7.5 us / 13.6 us per fwd / bwd step launch, profiles/r01_first.)
#pragma once
#include "common.h"

#define SKINNY_MFMA4(a, b)                                                   \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((a).x, (b).x, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((a).y, (b).y, acc1, 0, 0, 0); \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((a).z, (b).z, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((a).w, (b).w, acc1, 0, 0, 0);

// Accumulate sum_k A[arow][k] * W[wrow][k] over this wave's quarter of [0,K) into acc0/acc1.
// A and W are K-contiguous rows (16-byte aligned, K % 64 == 0).  Lane (j = l&15, q = l>>4):
// arow_off = element offset of the A row of tile row (l&15) (clamped by the caller),
// wrow_off = element offset of the W row feeding tile column (l&15).
__device__ __forceinline__ void skinny_accum_kk(f32x4& acc0, f32x4& acc1,
                                                const float* __restrict__ A, long arow_off,
                                                const float* __restrict__ W, long wrow_off,
                                                int K, int wave, int lane) {
    const int Kw = K >> 2;  // per wave, a multiple of 16
    const int kbase = wave * Kw + (lane >> 4) * 4;
    const float4* ap = reinterpret_cast<const float4*>(A + arow_off + kbase);  // chunk c at ap[4c]
    const float4* wp = reinterpret_cast<const float4*>(W + wrow_off + kbase);
    const int nchunk = Kw >> 4;
    int c = 0;
    for (; c + 4 <= nchunk; c += 4) {
        const float4 a0 = ap[4 * c], a1 = ap[4 * c + 4], a2 = ap[4 * c + 8], a3 = ap[4 * c + 12];
        const float4 b0 = wp[4 * c], b1 = wp[4 * c + 4], b2 = wp[4 * c + 8], b3 = wp[4 * c + 12];
        SKINNY_MFMA4(a0, b0)
        SKINNY_MFMA4(a1, b1)
        SKINNY_MFMA4(a2, b2)
        SKINNY_MFMA4(a3, b3)
    }
    for (; c < nchunk; ++c) {
        const float4 a0 = ap[4 * c];
        const float4 b0 = wp[4 * c];
        SKINNY_MFMA4(a0, b0)
    }
}

// Same contraction with W stored [K][N] (N-contiguous): lane reads W[k][wcol], one dword per MFMA
// (16 consecutive lanes read 64 contiguous bytes of one k-row).
__device__ __forceinline__ void skinny_accum_kn(f32x4& acc0, f32x4& acc1,
                                                const float* __restrict__ A, long arow_off,
                                                const float* __restrict__ W, int ldw, int wcol,
                                                int K, int wave, int lane) {
    const int Kw = K >> 2;
    const int kbase = wave * Kw + (lane >> 4) * 4;
    const float4* ap = reinterpret_cast<const float4*>(A + arow_off + kbase);
    const float* wp = W + (long)kbase * ldw + wcol;
    const int nchunk = Kw >> 4;
#pragma unroll 4
    for (int c = 0; c < nchunk; ++c) {
        const float4 a0 = ap[4 * c];
        float4 b0;
        b0.x = wp[(long)(16 * c + 0) * ldw];
        b0.y = wp[(long)(16 * c + 1) * ldw];
        b0.z = wp[(long)(16 * c + 2) * ldw];
        b0.w = wp[(long)(16 * c + 3) * ldw];
        SKINNY_MFMA4(a0, b0)
    }
}

// ---- static (all loads first) form ---------------------------------------------------------
// The step kernels run one block per CU-ish with nothing else to hide memory latency behind, so a
// loop of {8 loads -> wait -> 16 MFMA} exposes one full L2/Infinity-Cache round trip per
// iteration (8 of them in the first BPTT step kernel: 10 us per launch).  SkinnyFrag holds a wave's
// whole K-slice of both operands in registers: every load of every source is issued before the
// first MFMA, so the round trip is paid once.  CH = 16-float chunks per wave and source.
template <int CH>
struct SkinnyFrag {
    float4 a[CH];
    float4 w[CH];
};

// Kw = K-slice per wave = 16*CH floats; wave w of NW covers [w*Kw, (w+1)*Kw).
template <int CH>
__device__ __forceinline__ void skinny_load(SkinnyFrag<CH>& f, const float* __restrict__ A, long arow_off,
                                            const float* __restrict__ W, long wrow_off, int wave, int lane) {
    const int kbase = wave * (16 * CH) + (lane >> 4) * 4;
    const float4* ap = reinterpret_cast<const float4*>(A + arow_off + kbase);
    const float4* wp = reinterpret_cast<const float4*>(W + wrow_off + kbase);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        f.a[c] = ap[4 * c];
        f.w[c] = wp[4 * c];
    }
}

// k-chunk-major ("tiled") operands X_t[(k/16)][row][k%16] (R rows): the 64 lanes of one load instruction
// (16 rows x 4 q-groups x 16 B) then read ONE contiguous 1 KB piece = 8 full 128-B lines, instead of 16 half
// lines of 16 different rows.  tools/probe_step.hip: -17 % per forward launch at B=64, -32 % at B=256.
template <int CH>
__device__ __forceinline__ void skinny_load_tiled(SkinnyFrag<CH>& f, const float* __restrict__ At, int arow, int RA,
                                                  const float* __restrict__ Wt, int wrow, int RW, int wave,
                                                  int lane) {
    const int q4 = (lane >> 4) * 4;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const long kc = wave * CH + c;
        f.a[c] = *reinterpret_cast<const float4*>(At + (kc * RA + arow) * 16 + q4);
        f.w[c] = *reinterpret_cast<const float4*>(Wt + (kc * RW + wrow) * 16 + q4);
    }
}

template <int CH>
__device__ __forceinline__ void skinny_mfma(const SkinnyFrag<CH>& f, f32x4& acc0, f32x4& acc1) {
#pragma unroll
    for (int c = 0; c < CH; ++c) { SKINNY_MFMA4(f.a[c], f.w[c]) }
}

// NW-wave variants of the partial-tile exchange: red[NW][16][16].
__device__ __forceinline__ void skinny_store_partial_n(float* red, const f32x4& acc0, const f32x4& acc1,
                                                       int wave, int lane) {
    float* p = red + wave * 256 + ((lane >> 4) * 4) * 16 + (lane & 15);
    p[0] = acc0[0] + acc1[0];
    p[16] = acc0[1] + acc1[1];
    p[32] = acc0[2] + acc1[2];
    p[48] = acc0[3] + acc1[3];
}
template <int NW>
__device__ __forceinline__ float skinny_reduced_n(const float* red, int row, int col) {
    const int o = row * 16 + col;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 256 + o];
    return s;
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
