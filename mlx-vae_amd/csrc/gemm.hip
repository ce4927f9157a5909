// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32).
//
// The reference computes everything in float32 (MLX default dtype) and
// parity is stated at 1e-4 relative, so the contractions run on the *exact-f32* MFMA
// forms (bitwise an fmaf chain; gfx950 has no xf32/TF32).  Their peak is the f32 vector
// peak (157 TFLOP/s) -- the roofline this file is measured against.
//
//   C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]) (tanh)
//   transA = 0: A stored [M,K] row-major (K contiguous)   transA = 1: A stored [K,M]
//   transB = 0: B stored [K,N] row-major (N contiguous)   transB = 1: B stored [N,K]
//
// Two kernels:
//   * tile kernel: BMxBN block tile (128x128 or 64x64), BK = 16, 4 waves in 2x2, register
//     prefetch of the next K-tile + double-buffered LDS (one barrier per K-tile); optional
//     split-K over gridDim.z with f32 atomics for the K-long / small-MN weight-gradient shapes.
//   * skinny kernel (skinny.h): 16x16 tile per block, K split over the 4 waves -- used when M is
//     a minibatch (<= 256 rows) and the op is on the step's critical path.
#include "ops.h"
#ifdef ARCVAE_PLANE_CLOCK
#include <algorithm>
#include <cstdio>
#include <vector>
#endif
#include "skinny.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned split_pk(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// (x0, x1) -> packed bf16 pairs hi, mid, lo with x = hi + mid + lo (the two subtractions are exact in f32)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = split_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = split_pk(r0, r1);
    lo = split_pk(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
}

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int accumulate;  // C += result (non-split path)
    int act;         // 0 none, 1 tanh, 2 d-tanh: C = acc * (1 - T^2) with T = `bias` read as an [M,ldc] matrix
    int kchunk;      // K range per blockIdx.z (multiple of 16)
};

constexpr int BK = 16;
constexpr int PAD = 4;

// Loader for one operand tile [ROWS x BK], staged in LDS k-major: S[k][ROWS+PAD].
// KCONTIG: element (r,k) at P[r*ld + k]; else at P[k*ld + r].
template <int ROWS, bool KCONTIG, int VEC>
struct TileLoader {
    static constexpr int NV = (ROWS * BK) / (256 * VEC);
    float4 v4[VEC == 4 ? NV : 1];
    float v1[VEC == 1 ? NV : 1];

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int rmax,
                                         int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if constexpr (VEC == 4) {
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (KCONTIG) {
                    const int r = r0 + (idx >> 2), k = k0 + (idx & 3) * 4;
                    if (r < rmax && k < kend) z = *reinterpret_cast<const float4*>(P + (long)r * ld + k);
                } else {
                    const int k = k0 + idx / (ROWS / 4), r = r0 + (idx % (ROWS / 4)) * 4;
                    if (r < rmax && k < kend) z = *reinterpret_cast<const float4*>(P + (long)k * ld + r);
                }
                v4[i] = z;
            } else {
                float z = 0.f;
                if constexpr (KCONTIG) {
                    const int r = r0 + (idx >> 4), k = k0 + (idx & 15);
                    if (r < rmax && k < kend) z = P[(long)r * ld + k];
                } else {
                    const int k = k0 + idx / ROWS, r = r0 + idx % ROWS;
                    if (r < rmax && k < kend) z = P[(long)k * ld + r];
                }
                v1[i] = z;
            }
        }
    }

    __device__ __forceinline__ void store(float* S, int tid) const {
        constexpr int LD = ROWS + PAD;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if constexpr (VEC == 4) {
                if constexpr (KCONTIG) {
                    const int r = idx >> 2, k = (idx & 3) * 4;
                    S[(k + 0) * LD + r] = v4[i].x;
                    S[(k + 1) * LD + r] = v4[i].y;
                    S[(k + 2) * LD + r] = v4[i].z;
                    S[(k + 3) * LD + r] = v4[i].w;
                } else {
                    const int k = idx / (ROWS / 4), r = (idx % (ROWS / 4)) * 4;
                    *reinterpret_cast<float4*>(S + k * LD + r) = v4[i];
                }
            } else {
                if constexpr (KCONTIG) {
                    const int r = idx >> 4, k = idx & 15;
                    S[k * LD + r] = v1[i];
                } else {
                    const int k = idx / ROWS, r = idx % ROWS;
                    S[k * LD + r] = v1[i];
                }
            }
        }
    }
};

template <int BM, int BN, bool AK, bool BKC, int VA, int VB>
__device__ __forceinline__ void gemm_tile_body(const GemmP& p, const int bx, const int by, const int bz,
                                               const bool split) {
    constexpr int WM = BM / 2, WN = BN / 2;   // wave tile (2x2 waves)
    constexpr int MT = WM / 32, NT = WN / 32;  // 32x32 MFMA tiles per wave
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA_S + LDB_S)];
    float* const As0 = smem;                   // As[buf] = As0 + buf * BK * LDA_S
    float* const Bs0 = smem + 2 * BK * LDA_S;  // Bs[buf] = Bs0 + buf * BK * LDB_S

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<BM, AK, VA> la;
    TileLoader<BN, BKC, VB> lb;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(p.A, p.lda, m0, p.M, kbeg, kend, tid);
        lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid);
        la.store(As0, tid);
        lb.store(Bs0, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
            lb.load(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
        }
        const float* as = As0 + cur * BK * LDA_S + (lane >> 5) * LDA_S + wm * WM + (lane & 31);
        const float* bs = Bs0 + cur * BK * LDB_S + (lane >> 5) * LDB_S + wn * WN + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = as[kk * LDA_S + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = bs[kk * LDB_S + j * 32];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.store(As0 + (cur ^ 1) * BK * LDA_S, tid);
            lb.store(Bs0 + (cur ^ 1) * BK * LDB_S, tid);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WN + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = (p.bias && p.act != 2 && bz == 0) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float* c = p.C + (long)row * p.ldc + col;
                float v = acc[i][j][r] + bv;
                if (split) {
                    atomicAdd(c, v);
                } else {
                    if (p.accumulate) v += *c;
                    if (p.act == 1) v = tanhf(v);
                    if (p.act == 2) { const float t = p.bias[(long)row * p.ldc + col]; v *= (1.0f - t * t); }
                    *c = v;
                }
            }
        }
}

template <int BM, int BN, bool AK, bool BKC, int VA, int VB>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmP p) {
    gemm_tile_body<BM, BN, AK, BKC, VA, VB>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z > 1);
}

// ---- bf16-in / f32-accumulate tile GEMM (throughput mode, ARCVAE_GEMM_BF16) ----------------------------------------------
// Same contract as gemm_tile_kernel (operands f32 in memory, any of the four layouts, bias / tanh / d-tanh / "+=" /
// split-K epilogues), but the operands are rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest even) on their way into
// LDS and the products run on v_mfma_f32_32x32x16_bf16 with f32 accumulators: 16x the f32 matrix rate, ~3 significant
// digits per operand.  NOT a parity path (DESIGN.md section 10b states its tolerance); chosen per call by the flag.
// LDS image: [row][BKT + 8] bf16, k contiguous (one ds_read_b128 = a lane's 8 k of one MFMA; the 16-byte row pad keeps
// the 8 lanes of a read group on different 16-byte slots).  A k-contiguous operand is staged by 2 x 16-byte loads per
// lane and one ds_write_b128; a k-strided one ([K, rows] storage) by 8 loads down its column(s) -- a wave covers
// 64 * VW consecutive rows per load -- and VW ds_write_b128.
template <int ROWS, bool KCONTIG, int BKT>
struct BfLoader {
    static constexpr int LDK = BKT + 8;
    static constexpr int VW = ROWS / 64;                    // k-strided: rows per lane
    static constexpr int NT = KCONTIG ? (ROWS * BKT / 8) / 256 : BKT / 32;
    float v[NT][KCONTIG ? 8 : 8 * VW];

    // vector loads allowed? (k-contiguous: 16-byte rows; k-strided: VW-float groups)
    static inline bool ok(const float* P, int ld, int rows, int K) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(P);
        if (KCONTIG) return (a & 15) == 0 && (ld % 4) == 0 && (K % 4) == 0;
        return (a & (4 * VW - 1)) == 0 && (ld % VW) == 0 && (rows % VW) == 0;
    }

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int rmax, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (KCONTIG) {
                const int idx = tid + i * 256, r = r0 + idx / (BKT / 8), k = k0 + (idx % (BKT / 8)) * 8;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (r < rmax) {
                    const float* q = P + (long)r * ld + k;
                    if (k < kend) a = *reinterpret_cast<const float4*>(q);
                    if (k + 4 < kend) b = *reinterpret_cast<const float4*>(q + 4);
                }
                v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
                v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
            } else {
                const int o = (tid >> 6) + 4 * i, r = r0 + VW * (tid & 63);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = k0 + 8 * o + kk;
                    const bool in = r < rmax && k < kend;
                    const float* q = P + (long)k * ld + r;
                    if constexpr (VW == 4) {
                        const float4 t = in ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
                        v[i][kk] = t.x; v[i][8 + kk] = t.y; v[i][(VW > 2 ? 16 : 0) + kk] = t.z; v[i][(VW > 2 ? 24 : 0) + kk] = t.w;
                    } else if constexpr (VW == 2) {
                        const float2 t = in ? *reinterpret_cast<const float2*>(q) : make_float2(0.f, 0.f);
                        v[i][kk] = t.x; v[i][(VW > 1 ? 8 : 0) + kk] = t.y;
                    } else {
                        v[i][kk] = in ? *q : 0.f;
                    }
                }
            }
        }
    }

    // 8 consecutive k of one row -> its LDS slot(s).  PIECES = 1: rounded to bf16.  PIECES = 3: the hi / mid / lo pieces
    // (x = hi + mid + lo) into three images `img` elements apart (the split form: fp32-class accuracy, six products).
    template <int PIECES>
    __device__ __forceinline__ static void put8(__bf16* dst, const float* x, int img) {
        if constexpr (PIECES == 1) {
            *reinterpret_cast<u32x4_g*>(dst) = u32x4_g{split_pk(x[0], x[1]), split_pk(x[2], x[3]), split_pk(x[4], x[5]), split_pk(x[6], x[7])};
        } else {
            u32x4_g h, m, l;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned a, b, c;
                split3(x[2 * d], x[2 * d + 1], a, b, c);
                h[d] = a; m[d] = b; l[d] = c;
            }
            *reinterpret_cast<u32x4_g*>(dst) = h;
            *reinterpret_cast<u32x4_g*>(dst + img) = m;
            *reinterpret_cast<u32x4_g*>(dst + 2 * img) = l;
        }
    }
    template <int PIECES = 1>
    __device__ __forceinline__ void store(__bf16* S, int tid, int img = 0) const {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (KCONTIG) {
                const int idx = tid + i * 256, r = idx / (BKT / 8), o = idx % (BKT / 8);
                put8<PIECES>(S + r * LDK + 8 * o, v[i], img);
            } else {
                const int o = (tid >> 6) + 4 * i, r = VW * (tid & 63);
#pragma unroll
                for (int j = 0; j < VW; ++j) put8<PIECES>(S + (r + j) * LDK + 8 * o, v[i] + 8 * j, img);
            }
        }
    }
};

template <int BM, int BN, bool AK, bool BKC, int BKT, int PIECES = 1>
__device__ __forceinline__ void gemm_bf16_body(const GemmP& p, const int bx, const int by, const int bz, const bool split,
                                               __bf16* smem) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32, LDK = BKT + 8;
    constexpr int IA = BM * LDK, IB = BN * LDK;     // one operand image; a buffer holds PIECES of them
    __bf16* const As0 = smem;                       // As[buf][piece] = As0 + (buf * PIECES + piece) * IA
    __bf16* const Bs0 = smem + 2 * PIECES * IA;     // Bs[buf][piece] = Bs0 + (buf * PIECES + piece) * IB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // One K-tile's MFMAs are 16 x shorter than in the f32 kernel, far less than a global-load round trip; what hides the
    // loads is occupancy (~100 registers: 4 blocks per CU), not a deeper register ring: a 4-tile ring (240 registers, 2 blocks
    // per CU) measured 216 instead of 278 TFLOP/s on [40960 x 2048] x K = 512 (profiles/r02_bf16_mode.txt).
    BfLoader<BM, AK, BKT> la;
    BfLoader<BN, BKC, BKT> lb;
    const int nk = (kend - kbeg + BKT - 1) / BKT;
    if (nk > 0) {
        la.load(p.A, p.lda, m0, p.M, kbeg, kend, tid);
        lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid);
        la.template store<PIECES>(As0, tid, IA);
        lb.template store<PIECES>(Bs0, tid, IB);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BKT, kend, tid);
            lb.load(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BKT, kend, tid);
        }
        const __bf16* as = As0 + cur * PIECES * IA + (wm * WM + (lane & 31)) * LDK + 8 * (lane >> 5);
        const __bf16* bs = Bs0 + cur * PIECES * IB + (wn * WN + (lane & 31)) * LDK + 8 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 16) {
            bf16x8_t a[PIECES][MT], b[PIECES][NT];
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[q][i] = *reinterpret_cast<const bf16x8_t*>(as + q * IA + i * 32 * LDK + kk);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[q][j] = *reinterpret_cast<const bf16x8_t*>(bs + q * IB + j * 32 * LDK + kk);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (PIECES == 3) {   // the six terms of weight >= 2^-16, small ones first (pieces: 0 hi, 1 mid, 2 lo)
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PIECES - 1][i], b[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[PIECES - 1][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PIECES > 1 ? 1 : 0][i], b[PIECES > 1 ? 1 : 0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PIECES > 1 ? 1 : 0][i], b[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[PIECES > 1 ? 1 : 0][j], c, 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                }
        }
        if (kt + 1 < nk) {
            la.template store<PIECES>(As0 + (cur ^ 1) * PIECES * IA, tid, IA);
            lb.template store<PIECES>(Bs0 + (cur ^ 1) * PIECES * IB, tid, IB);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WN + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = (p.bias && p.act != 2 && bz == 0) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float* c = p.C + (long)row * p.ldc + col;
                float v = acc[i][j][r] + bv;
                if (split) {
                    atomicAdd(c, v);
                } else {
                    if (p.accumulate) v += *c;
                    if (p.act == 1) v = tanhf(v);
                    if (p.act == 2) { const float t = p.bias[(long)row * p.ldc + col]; v *= (1.0f - t * t); }
                    *c = v;
                }
            }
        }
}

template <int BM, int BN, bool AK, bool BKC, int BKT, int PIECES = 1>
__global__ __launch_bounds__(256) void gemm_bf16_tile_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 bf_smem[];
    gemm_bf16_body<BM, BN, AK, BKC, BKT, PIECES>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z > 1, bf_smem);
}

template <int BM, int BN, bool AK, bool BKC, int BKT, int PIECES = 1>
void launch_bf16_tile(const GemmP& p, dim3 grid, hipStream_t s) {
    const size_t lds = sizeof(__bf16) * 2 * PIECES * (BM + BN) * (BKT + 8);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)gemm_bf16_tile_kernel<BM, BN, AK, BKC, BKT, PIECES>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm_bf16_tile_kernel<BM, BN, AK, BKC, BKT, PIECES>), grid, dim3(256), lds, s, p);
}
template <int BM, int BN, int BKT, int PIECES = 1>
bool launch_bf16_tile_t(const GemmP& p, dim3 grid, bool ak, bool bk, hipStream_t s) {
    const bool aok = ak ? BfLoader<BM, true, BKT>::ok(p.A, p.lda, p.M, p.K) : BfLoader<BM, false, BKT>::ok(p.A, p.lda, p.M, p.K);
    const bool bok = bk ? BfLoader<BN, true, BKT>::ok(p.B, p.ldb, p.N, p.K) : BfLoader<BN, false, BKT>::ok(p.B, p.ldb, p.N, p.K);
    if (!aok || !bok) return false;
    if (ak && bk) launch_bf16_tile<BM, BN, true, true, BKT, PIECES>(p, grid, s);
    else if (ak) launch_bf16_tile<BM, BN, true, false, BKT, PIECES>(p, grid, s);
    else if (bk) launch_bf16_tile<BM, BN, false, true, BKT, PIECES>(p, grid, s);
    else launch_bf16_tile<BM, BN, false, false, BKT, PIECES>(p, grid, s);
    return true;
}

// ---- forward-only decoder layer: G = A . W^T + b and the zero-state cell h = o * tanh(i * g) in ONE kernel ----------------
// The sampler (BASELINE.json configs[4]) and the loss-only forward do not need the pre-activations G [B*V, 4H] afterwards,
// yet wrote them (335 MB per bs-1024 batch) and read them back in cell_zero_fwd_kernel -- 130 us of a 740 us batch.  Same
// tile kernel as gemm_tile_body<128, 128, A k-contiguous, W [4H, K]> (same loads, same k order of the exact-f32 MFMAs,
// same "+ bias", same cell arithmetic as cell_zero_fwd_kernel: bit-identical h), except that the block's 128 output
// columns are 32 units x (i, f, g, o): logical column c = 4 * unit + gate reads weight row gate * H + unit, so the gates
// of a unit are four neighbouring floats of a row of the pre-activation tile, which goes through LDS once: every thread then
// evaluates 16 cells and h leaves in full 128-byte row segments.
struct CellGemmP {
    const float* A;      // [M, K] (lda)
    const float* W;      // [4H, K] (ldw), rows gate * H + unit
    const float* bias;   // [4H]
    float* Hout;         // [M, H]
    int M, H, K, lda, ldw;
};
__global__ __launch_bounds__(256) void gemm_cell_zero_kernel(CellGemmP p) {
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, MT = 2, NT = 2;
    constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // max(operand tiles: 33.8 KB, pre-activation tile [128][129]: 66 KB)
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * BK * LDA_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;      // n0: first LOGICAL column (4 * unit + gate)
    const int G = 4 * p.H;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TileLoader<BM, true, 4> la;
    float4 vb[2];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, c = n0 + (idx >> 2), k = k0 + (idx & 3) * 4;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < G && k < p.K) z = *reinterpret_cast<const float4*>(p.W + (long)((c & 3) * p.H + (c >> 2)) * p.ldw + k);
            vb[i] = z;
        }
    };
    auto store_b = [&](float* S) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, r = idx >> 2, k = (idx & 3) * 4;
            S[(k + 0) * LDB_S + r] = vb[i].x;
            S[(k + 1) * LDB_S + r] = vb[i].y;
            S[(k + 2) * LDB_S + r] = vb[i].z;
            S[(k + 3) * LDB_S + r] = vb[i].w;
        }
    };
    const int nk = (p.K + BK - 1) / BK;
    la.load(p.A, p.lda, m0, p.M, 0, p.K, tid);
    load_b(0);
    la.store(As0, tid);
    store_b(Bs0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid);
            load_b((kt + 1) * BK);
        }
        const float* as = As0 + cur * BK * LDA_S + (lane >> 5) * LDA_S + wm * WM + (lane & 31);
        const float* bs = Bs0 + cur * BK * LDB_S + (lane >> 5) * LDB_S + wn * WN + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = as[kk * LDA_S + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = bs[kk * LDB_S + j * 32];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.store(As0 + (cur ^ 1) * BK * LDA_S, tid);
            store_b(Bs0 + (cur ^ 1) * BK * LDB_S);
        }
        __syncthreads();
    }
    // ---- the pre-activation tile (+ bias) goes to LDS [128 rows][129]; then every thread evaluates 16 cells -- its unit's
    // gates i, g, o are three neighbouring floats of a row -- and writes h in full 128-byte row segments.  (Evaluating the
    // cell on the accumulators needs the gates of a unit in ONE lane: with lane shifts only a quarter of the lanes work.)
    float* gs = smem;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cl = wn * WN + j * 32 + (lane & 31);       // logical column within the block tile
            const int c = n0 + cl;
            const float bv = c < G ? p.bias[(c & 3) * p.H + (c >> 2)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                gs[row * 129 + cl] = acc[i][j][r] + bv;
            }
        }
    __syncthreads();
    const int u0 = n0 >> 2;                            // first unit of the block tile
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = q * 8 + (tid >> 5), ul = tid & 31;
        const float* g4 = gs + row * 129 + 4 * ul;
        const float ig = sigmoidf_acc(g4[0]), gg = tanhf(g4[2]), og = sigmoidf_acc(g4[3]);
        if (m0 + row < p.M && u0 + ul < p.H) p.Hout[(long)(m0 + row) * p.H + u0 + ul] = og * tanhf(ig * gg);
    }
}

// Grouped launch: up to 8 same-layout problems in ONE launch (blockIdx.z walks [problem][k-slice]); used for the
// per-layer weight-gradient GEMMs of a BPTT chunk, which are small, independent and otherwise each pay a launch.
#define ARCVAE_GEMM_GROUP_MAX 8
struct GemmGroup {
    GemmP p[ARCVAE_GEMM_GROUP_MAX];
    int zoff[ARCVAE_GEMM_GROUP_MAX + 1];
    int n;
};
template <int BM, int BN, bool AK, bool BKC, int VA, int VB>
__global__ __launch_bounds__(256) void gemm_tile_group_kernel(GemmGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.z >= g.zoff[i + 1]) ++i;
    const GemmP& p = g.p[i];
    if ((int)blockIdx.y * BM >= p.M || (int)blockIdx.x * BN >= p.N) return;
    gemm_tile_body<BM, BN, AK, BKC, VA, VB>(p, blockIdx.x, blockIdx.y, blockIdx.z - g.zoff[i], true);
}

// Skinny: 16x16 tile per block; A [M,K] K-contiguous; B either [N,K] (BKC) or [K,N].
template <bool BKC>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p) {
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int arow = min(m0 + (lane & 15), p.M - 1);
    const int bcol = min(n0 + (lane & 15), p.N - 1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BKC)
        skinny_accum_kk(acc0, acc1, p.A, (long)arow * p.lda, p.B, (long)bcol * p.ldb, p.K, wave, lane);
    else
        skinny_accum_kn(acc0, acc1, p.A, (long)arow * p.lda, p.B, p.ldb, bcol, p.K, wave, lane);
    skinny_store_partial(red, acc0, acc1, wave, lane);
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const int gr = m0 + row, gc = n0 + col;
    if (gr < p.M && gc < p.N) {
        float v = skinny_reduced(red, row, col);
        if (p.bias && p.act != 2) v += p.bias[gc];
        float* c = p.C + (long)gr * p.ldc + gc;
        if (p.accumulate) v += *c;
        if (p.act == 1) v = tanhf(v);
        if (p.act == 2) { const float t = p.bias[(long)gr * p.ldc + gc]; v *= (1.0f - t * t); }
        *c = v;
    }
}

// ---- split-bf16 ("bf16x3") TN GEMM: fp32 accuracy on the bf16 matrix cores --------------------------------------------
// C[M,N] += A^T . B with A stored [K,M], B stored [K,N] (the weight-gradient shape: K = time x batch, long; M, N =
// weight dims).  The exact-f32 MFMA runs at 1/16 of the bf16 rate and holds a SIMD's matrix pipe 64 cycles per K = 2; the
// weight-gradient GEMMs beside the BPTT sweep are the largest consumers of that pipe on the sweep's own CUs.  Here every
// f32 operand is split into three bf16 pieces, x = hi + mid + lo (8 + 8 + 8 significant bits: exact up to the last
// rounding), and a product is the six terms of weight >= 2^-16: hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid, all
// accumulated in f32 by v_mfma_f32_32x32x16_bf16 -- 6 x 32 cycles per K = 16 instead of 8 x 64: 2.7x less matrix-pipe time,
// error ~2^-22 of |a||b| per product (the dropped terms), i.e. the accuracy class of an f32 fmaf chain (tests compare both
// against fp64).
// No LDS, no barrier, one wave per 128 x 64 output tile and K slice: a lane loads its operands straight into the MFMA
// fragment layout.  A fragment needs 8 consecutive k of ONE row per lane, but k is the slow index of both operands, so a
// lane takes 8 loads (k = 8h .. 8h+7) of 4 consecutive m (one float4: lanes r = 0..31 cover 512 contiguous bytes) and the
// four m feed the four 32-row MFMA tiles of the wave: tile i holds rows m0 + 4r + i (an interleaved row order, undone
// in the epilogue's row index); B columns stay natural (n0 + 32j + c, dword loads: 128 contiguous bytes per half-wave),
// so the epilogue's float atomics cover two full 128-B row segments per wave-instruction (the full-rate shape).

struct SplitTN {
    const float* A;   // [K, M]
    const float* B;   // [K, N]
    float* C;         // [M, N], +=
    float* colsum;    // optional [M]: += the column sums of A over K (the bias gradient that goes with a weight-gradient GEMM:
                      // the first column of tiles sums the values it loads anyway -- a separate column-sum launch beside the
                      // persistent sweep waited ~60 us for CU resources, profiles/r02_tail_timeline.txt)
    int M, N, K, lda, ldb, ldc, kchunk;
};
#define ARCVAE_SPLIT_GROUP_MAX 8
struct SplitTNGroup {
    SplitTN p[ARCVAE_SPLIT_GROUP_MAX];
    int zoff[ARCVAE_SPLIT_GROUP_MAX + 1];
    int n;
};


// Block = 4 waves on ONE 128 x 32*NJ output tile: the waves take the K = 16 steps of the block's K chunk round-robin
// (the outputs are small and K is long, so parallelism has to come from K; split-K over blocks alone would multiply
// the float-atomic epilogue -- 128 wave-instructions per tile and slice), their accumulators are summed through LDS
// (two 32 KB regions: waves 1, 3 -> 0, 2, then 2 -> 0) and every wave commits a quarter of the tile with atomics.
// MI = 4: 128-row tiles, float4 loads (M % 4 == 0).  MI = 2: 64-row tiles, 8-byte loads -- the accumulators, raw operands
// and bf16 pieces of a 64 x 64 tile fit ~200 registers, which is what a wave may use BESIDE a persistent LSTM sweep wave
// (296 of a SIMD's 512): the 128-row form (322 registers) can only run where no sweep block is resident, i.e. it waits
// for the sweep's chunk to end (measured: step 1.147 vs 1.083 ms).
// ONE: plain bf16 operands (the hi piece only, one product): the throughput mode's weight-gradient GEMM.
template <int MI, int NJ, bool ONE = false>
__device__ __forceinline__ void split_tn_body(const SplitTN& p, const int bx, const int by, const int bz, float* red) {
    constexpr int TT = MI * NJ;                    // 32 x 32 MFMA tiles of the block tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = by * 32 * MI, n0 = bx * 32 * NJ;
    const int kbeg = bz * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int am = m0 + MI * r;                    // my MI rows (columns of the stored A): am .. am + MI - 1
    const bool aok = am < p.M;                     // M % MI == 0: all or none
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float ra[2][8][MI];
    float rb[2][8][NJ];
    const bool do_cs = p.colsum != nullptr && bx == 0;   // block-uniform
    float csum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) csum[i] = 0.f;
#define SPLIT_LOAD(BUF, K0)                                                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                                            \
        const int k = (K0) + 8 * h + kk;                                                                          \
        const bool kok = k < kend;                                                                                \
        if constexpr (MI == 4) {                                                                                  \
            const float4 v = (kok && aok) ? *reinterpret_cast<const float4*>(p.A + (long)k * p.lda + am)          \
                                          : make_float4(0.f, 0.f, 0.f, 0.f);                                      \
            ra[BUF][kk][0] = v.x; ra[BUF][kk][1] = v.y; ra[BUF][kk][MI > 2 ? 2 : 0] = v.z; ra[BUF][kk][MI > 2 ? 3 : 0] = v.w; \
        } else {                                                                                                  \
            const float2 v = (kok && aok) ? *reinterpret_cast<const float2*>(p.A + (long)k * p.lda + am)          \
                                          : make_float2(0.f, 0.f);                                                \
            ra[BUF][kk][0] = v.x; ra[BUF][kk][1] = v.y;                                                           \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            const int n = n0 + 32 * j + r;                                                                        \
            rb[BUF][kk][j] = (kok && n < p.N) ? p.B[(long)k * p.ldb + n] : 0.f;                                   \
        }                                                                                                         \
    }
    // one K = 16 step from buffer BUF: split into bf16 pieces, refill the buffer two steps ahead, 6 products per tile
#define SPLIT_STEP(BUF, K0)                                                                                       \
    {                                                                                                             \
        if (do_cs) {                                                                                              \
            _Pragma("unroll") for (int kk = 0; kk < 8; ++kk)                                                      \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) csum[i] += ra[BUF][kk][i];                             \
        }                                                                                                         \
        u32x4_g ah[MI], amid[MI], al[MI], bh[NJ], bm[NJ], bl[NJ];                                                 \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                           \
            unsigned x, y = 0, z = 0;                                                                             \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                      \
                if constexpr (ONE) x = split_pk(ra[BUF][2 * d][i], ra[BUF][2 * d + 1][i]);                        \
                else split3(ra[BUF][2 * d][i], ra[BUF][2 * d + 1][i], x, y, z);                                   \
                ah[i][d] = x; amid[i][d] = y; al[i][d] = z;                                                       \
            }                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                      \
                if constexpr (ONE) x = split_pk(rb[BUF][2 * d][j], rb[BUF][2 * d + 1][j]);                        \
                else split3(rb[BUF][2 * d][j], rb[BUF][2 * d + 1][j], x, y, z);                                   \
                bh[j][d] = x; bm[j][d] = y; bl[j][d] = z;                                                         \
            }                                                                                                     \
        }                                                                                                         \
        if ((K0) + 128 < kend) { SPLIT_LOAD(BUF, (K0) + 128) }                                                    \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            const bf16x8_t AH = __builtin_bit_cast(bf16x8_t, ah[i]), AM = __builtin_bit_cast(bf16x8_t, amid[i]),   \
                           AL = __builtin_bit_cast(bf16x8_t, al[i]);                                              \
            const bf16x8_t BH = __builtin_bit_cast(bf16x8_t, bh[j]), BM_ = __builtin_bit_cast(bf16x8_t, bm[j]),    \
                           BL = __builtin_bit_cast(bf16x8_t, bl[j]);                                              \
            f32x16 c = acc[i][j];                 /* small terms first */                                         \
            if constexpr (!ONE) {                                                                                 \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, c, 0, 0, 0);                                  \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, c, 0, 0, 0);                                  \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM_, c, 0, 0, 0);                                 \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, c, 0, 0, 0);                                  \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM_, c, 0, 0, 0);                                 \
            }                                                                                                     \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, c, 0, 0, 0);                                      \
            acc[i][j] = c;                                                                                        \
        }                                                                                                         \
    }
    int k0 = kbeg + 16 * wave;                     // my steps: k0, k0 + 64, k0 + 128, ...
    if (k0 < kend) { SPLIT_LOAD(0, k0) }
    if (k0 + 64 < kend) { SPLIT_LOAD(1, k0 + 64) }
    for (; k0 < kend; k0 += 128) {
        SPLIT_STEP(0, k0)
        if (k0 + 64 < kend) SPLIT_STEP(1, k0 + 64)
    }
#undef SPLIT_STEP
#undef SPLIT_LOAD
    if (do_cs) {   // my columns am .. am + MI - 1: the two k-halves of the wave, then one atomic per wave and column
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float v = csum[i] + __shfl_xor(csum[i], 32);
            if (h == 0 && aok) atomicAdd(p.colsum + am + i, v);
        }
    }
    // ---- sum the four waves' accumulators: red [2 regions][TT tiles][16][64 lanes]
    const int reg = wave >> 1;
    if (wave & 1) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[((reg * TT + i * NJ + j) * 16 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    if (!(wave & 1)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] += red[((reg * TT + i * NJ + j) * 16 + q) * 64 + lane];
    }
    __syncthreads();
    if (!(wave & 1)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[((reg * TT + i * NJ + j) * 16 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    // ---- every wave commits TT / 4 tiles: t = wave * TT/4 + u
    static_assert(TT % 4 == 0, "a quarter of the tiles per wave");
#pragma unroll
    for (int u = 0; u < TT / 4; ++u) {
        const int t = wave * (TT / 4) + u, i = t / NJ, j = t - i * NJ;
        const int n = n0 + 32 * j + r;
        if (n >= p.N) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + MI * ((q & 3) + 8 * (q >> 2) + 4 * h) + i;   // MFMA row -> interleaved row
            const float v = red[((0 * TT + t) * 16 + q) * 64 + lane] + red[((1 * TT + t) * 16 + q) * 64 + lane];
            if (m < p.M) atomicAdd(p.C + (long)m * p.ldc + n, v);
        }
    }
}

template <int MI, int NJ, bool ONE = false>
__global__ __launch_bounds__(256) void gemm_split_tn_group_kernel(SplitTNGroup g) {
    extern __shared__ __attribute__((aligned(16))) float split_red[];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.z >= g.zoff[i + 1]) ++i;
    const SplitTN& p = g.p[i];
    if ((int)blockIdx.y * 32 * MI >= p.M || (int)blockIdx.x * 32 * NJ >= p.N) return;
    split_tn_body<MI, NJ, ONE>(p, blockIdx.x, blockIdx.y, blockIdx.z - g.zoff[i], split_red);
}

// ARCVAE_GEMM_SPLIT (default 1): the TN "+=" GEMMs (weight gradients, one-hot token-table gradient) on the split-bf16
// kernel; 0 = the exact-f32 MFMA tile kernel of round 1.
inline bool split_tn_enabled() {
    static const int v = arcvae_env_int("ARCVAE_GEMM_SPLIT", 1);
    return v != 0;
}
inline bool split_tn_ok(int M, int N, const float* A, int lda, const float* B, int ldb) {
    return split_tn_enabled() && (M % 4) == 0 && (lda % 4) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && M >= 32 && N >= 16;
}
// n same-layout problems in one launch; K slices of `kslice` (multiple of 16) walk blockIdx.z
int launch_split_tn_group(int n, const SplitTN* probs, bool wide, bool one, hipStream_t stream) {
    SplitTNGroup g;
    g.n = n;
    int Mmax = 0, Nmax = 0, ztot = 0;
    static const int target = arcvae_env_int("ARCVAE_SPLIT_BLOCKS", 256);   // blocks (of 4 waves) wanted per launch
    for (int i = 0; i < n; ++i) {
        g.p[i] = probs[i];
        Mmax = max(Mmax, probs[i].M); Nmax = max(Nmax, probs[i].N);
    }
    // ARCVAE_SPLIT_TILE: 2 (default) = 64 x 64 tiles (fits beside a persistent sweep), 4 = 128 x 64 everywhere;
    // `wide`: the caller knows no sweep is resident (the last chunk's GEMMs run behind the sweep): 128 x 64
    static const int mi_env = arcvae_env_int("ARCVAE_SPLIT_TILE", 2) == 4 ? 4 : 2;
    const int mi = (wide || one) ? 4 : mi_env;
    const int tiles = ceil_div(Mmax, 32 * mi) * ceil_div(Nmax, 64);
    for (int i = 0; i < n; ++i) {
        SplitTN& p = g.p[i];
        int z = min(ceil_div(target, tiles * n), max(1, p.K / 256));           // at least 256 of K per block (4 steps a wave)
        z = max(1, z);
        p.kchunk = ceil_div(ceil_div(p.K, z), 64) * 64;
        z = ceil_div(p.K, p.kchunk);
        g.zoff[i] = ztot;
        ztot += z;
    }
    g.zoff[n] = ztot;
    for (int i = n; i < ARCVAE_SPLIT_GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.zoff[i + 1] = ztot; }
    dim3 grid(ceil_div(Nmax, 64), ceil_div(Mmax, 32 * mi), ztot);
    const size_t lds = sizeof(float) * 2 * (mi * 2) * 16 * 64;                   // two accumulator images: 32 / 64 KB
    if (one) {   // throughput mode: one bf16 product per step (its 128-row tile needs no operand pieces: 4 x 2 tiles always)
        (void)hipFuncSetAttribute((const void*)gemm_split_tn_group_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_split_tn_group_kernel<4, 2, true>), grid, dim3(256), lds, stream, g);
    } else if (mi == 4) {
        (void)hipFuncSetAttribute((const void*)gemm_split_tn_group_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_split_tn_group_kernel<4, 2>), grid, dim3(256), lds, stream, g);
    } else {
        hipLaunchKernelGGL((gemm_split_tn_group_kernel<2, 2>), grid, dim3(256), lds, stream, g);
    }
    return arcvae_launch_status();
}

// ---- throughput mode: weight gradients from OCTET-MAJOR bf16 copies ------------------------------------------------------
// C[M,N] += A^T . B where the sweeps' epilogues have left bf16 copies of both operands in the layout an MFMA fragment is
// loaded in: X_oct[k / 8][row][k % 8] (k = t * B + b, the contraction index; row = gate column or hidden unit), so a
// lane's 8 consecutive k of one row are ONE 16-byte load and the 32 rows of a half-wave are 512 contiguous bytes -- no
// LDS, no conversion, no transposition.  The f32-sourced forms (gemm_bf16_tile_kernel TN: 8 strided loads + cvt + LDS
// round trip per operand piece; the split kernel's one-product form: 48 bytes of f32 per MFMA cycle and wave) reach
// 80-210 TFLOP/s on the [4H x H] x K = T*B gradients of BASELINE.json configs[2]; here a wave owns a 128 x 128 tile
// (16 accumulators, 8 loads per 16 MFMAs = 16 bytes per MFMA cycle), the 2 x 2 waves of a block share their A / B rows
// through L1, and K is split over blockIdx.z with float atomics into the "+=" target.
struct OctTN {
    const __bf16* A;   // [K/8][M][8]
    const __bf16* B;   // [K/8][N][8]
    float* C;          // [M, N] row-major, +=
    int K;             // multiple of 16
    int kchunk;        // K per blockIdx.z slice (multiple of 16)
};
#define ARCVAE_OCT_GROUP_MAX 8
struct OctTNGroup {
    OctTN p[ARCVAE_OCT_GROUP_MAX];
    int zoff[ARCVAE_OCT_GROUP_MAX + 1];
    int n, M, N, ldc;
};
__global__ __launch_bounds__(256) void wgrad_octet_kernel(OctTNGroup g) {
    int pi = 0;
    while (pi + 1 < g.n && (int)blockIdx.z >= g.zoff[pi + 1]) ++pi;
    const OctTN& p = g.p[pi];
    const int bz = blockIdx.z - g.zoff[pi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 256 + (wave >> 1) * 128, n0 = blockIdx.x * 256 + (wave & 1) * 128;
    if (n0 >= g.N || m0 >= g.M) return;            // (no barrier in this kernel)
    const int kbeg = bz * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    int arow[4], brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        arow[i] = min(m0 + 32 * i + r, g.M - 1);
        brow[i] = min(n0 + 32 * i + r, g.N - 1);
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    constexpr int NS = 3;
    u32x4_g fa[NS][4], fb[NS][4];
    auto load = [&](int st, int k0) {             // the K = 16 step at k0: octets k0/8 + h
        const long o = (long)(k0 >> 3) + h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[st][i] = *reinterpret_cast<const u32x4_g*>(p.A + (o * g.M + arow[i]) * 8);
            fb[st][i] = *reinterpret_cast<const u32x4_g*>(p.B + (o * g.N + brow[i]) * 8);
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (kbeg + 16 * s < kend) load(s, kbeg + 16 * s);
    for (int k0 = kbeg; k0 < kend; k0 += 16 * NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = k0 + 16 * s;
            if (k < kend) {
                if (k + 16 * (NS - 1) < kend) load((s + NS - 1) % NS, k + 16 * (NS - 1));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[s][i]),
                                                                            __builtin_bit_cast(bf16x8_t, fb[s][j]),
                                                                            acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 32 * j + r;
            if (col >= g.N) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (row < g.M) atomicAdd(p.C + (long)row * g.ldc + col, acc[i][j][q]);
            }
        }
}

// Two independent skinny problems of the same layout in ONE launch (blockIdx.x walks the column tiles of the first, then
// of the second): the encoder heads' pairs [mu_raw | tanh(lh)] and [dcomb | dlh] sit on the step's critical chain, where
// every launch is a ~6 us seam.
struct GemmPair { GemmP p[2]; int nx0; };
template <bool BKC>
__global__ __launch_bounds__(256) void gemm_skinny_pair_kernel(GemmPair g) {
    __shared__ float red[4 * 256];
    const int which = (int)blockIdx.x >= g.nx0 ? 1 : 0;
    const GemmP& p = g.p[which];
    const int bx = (int)blockIdx.x - (which ? g.nx0 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 16, n0 = bx * 16;
    if (m0 >= p.M || n0 >= p.N) return;
    const int arow = min(m0 + (lane & 15), p.M - 1);
    const int bcol = min(n0 + (lane & 15), p.N - 1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BKC)
        skinny_accum_kk(acc0, acc1, p.A, (long)arow * p.lda, p.B, (long)bcol * p.ldb, p.K, wave, lane);
    else
        skinny_accum_kn(acc0, acc1, p.A, (long)arow * p.lda, p.B, p.ldb, bcol, p.K, wave, lane);
    skinny_store_partial(red, acc0, acc1, wave, lane);
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const int gr = m0 + row, gc = n0 + col;
    if (gr < p.M && gc < p.N) {
        float v = skinny_reduced(red, row, col);
        if (p.bias && p.act != 2) v += p.bias[gc];
        float* c = p.C + (long)gr * p.ldc + gc;
        if (p.accumulate) v += *c;
        if (p.act == 1) v = tanhf(v);
        if (p.act == 2) { const float t = p.bias[(long)gr * p.ldc + gc]; v *= (1.0f - t * t); }
        *c = v;
    }
}

template <int BM, int BN, bool AK, bool BKC>
void launch_tile(const GemmP& p, dim3 grid, bool va4, bool vb4, hipStream_t s) {
    const unsigned pad = arcvae_side_lds_pad(2 * BK * (BM + BN + 2 * PAD) * sizeof(float));
    if (va4 && vb4)
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AK, BKC, 4, 4>), grid, dim3(256), pad, s, p);
    else if (va4)
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AK, BKC, 4, 1>), grid, dim3(256), pad, s, p);
    else if (vb4)
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AK, BKC, 1, 4>), grid, dim3(256), pad, s, p);
    else
        hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, AK, BKC, 1, 1>), grid, dim3(256), pad, s, p);
}

template <int BM, int BN>
void launch_tile_t(const GemmP& p, dim3 grid, bool ak, bool bk, bool va4, bool vb4, hipStream_t s) {
    if (ak && bk) launch_tile<BM, BN, true, true>(p, grid, va4, vb4, s);
    else if (ak) launch_tile<BM, BN, true, false>(p, grid, va4, vb4, s);
    else if (bk) launch_tile<BM, BN, false, true>(p, grid, va4, vb4, s);
    else launch_tile<BM, BN, false, false>(p, grid, va4, vb4, s);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int arcvae_gemm_f32(int transA, int transB, int M, int N, int K,
                               const float* A, int lda, const float* B, int ldb,
                               float* C, int ldc, const float* bias, int flags, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return ARCVAE_ERR_ARG;
    if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return ARCVAE_ERR_ARG;
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.accumulate = (flags & ARCVAE_GEMM_ACCUMULATE) ? 1 : 0;
    p.act = (flags & ARCVAE_GEMM_TANH) ? 1 : ((flags & ARCVAE_GEMM_DTANH) ? 2 : 0);
    if (p.act == 2 && (!bias || p.accumulate)) return ARCVAE_ERR_ARG;
    p.kchunk = ((K + BK - 1) / BK) * BK;

    const bool ak = !transA, bk = transB != 0;
    // throughput mode, large TN shapes (K long AND a grid of 128 x 128 tiles worth staging through LDS): the bf16 tile kernel
    // below beats the LDS-free split kernel's one-product form, which is bound by its operand loads (207 vs 97 TFLOP/s on
    // the [2048 x 512] x K = 65536 weight gradients of BASELINE.json configs[2])
    const bool bf16_tile_tn = (flags & ARCVAE_GEMM_BF16) && M >= 256 && N >= 256 && K >= 2048;
    // TN "+=" with split-K allowed (weight-gradient / token-table shapes): split-bf16 kernel
    if (!bf16_tile_tn)
    if (transA && !transB && (flags & ARCVAE_GEMM_ACCUMULATE) && (flags & ARCVAE_GEMM_SPLITK) && !bias && p.act == 0 &&
        !(flags & (ARCVAE_GEMM_TILE64 | ARCVAE_GEMM_TILE128)) && split_tn_ok(M, N, A, lda, B, ldb)) {
        SplitTN q;
        q.A = A; q.B = B; q.C = C; q.colsum = nullptr; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.kchunk = 0;
        return launch_split_tn_group(1, &q, (flags & ARCVAE_GEMM_TILE_WIDE) != 0, (flags & ARCVAE_GEMM_BF16) != 0, stream);
    }
    // ARCVAE_GEMM_SPLIT3: any layout on the bf16 matrix pipe at fp32-class accuracy -- three bf16 pieces per operand, six
    // products, 64 x 64 tiles (61 KB of LDS).  For GEMMs that run BESIDE a persistent sweep: 2.7x less matrix-pipe time
    // than the exact-f32 form, in 32-cycle instead of 64-cycle instructions, on the SIMDs the chain's waves issue on.
    if ((flags & ARCVAE_GEMM_SPLIT3) && !(flags & ARCVAE_GEMM_BF16) && !(transA == 0 && M <= 256)) {
        dim3 grid(ceil_div(N, 64), ceil_div(M, 64), 1);
        if ((flags & ARCVAE_GEMM_SPLITK) && p.act == 0) {
            static const int target3 = arcvae_env_int("ARCVAE_SPLITK_BLOCKS", 512);
            int z = min(ceil_div(target3, (int)(grid.x * grid.y)), max(1, K / 256));
            if (z > 1) {
                p.kchunk = ceil_div(ceil_div(K, z), 64) * 64;
                grid.z = ceil_div(K, p.kchunk);
                if (!p.accumulate && arcvae_zero(C, M, N, ldc, stream) != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
            }
        }
        // (a 128 x 128 form -- 123 KB of LDS -- was built and dropped twice: no faster than the f32 128 x 128 tile where that is used
        // (round 2), 53 ms against 36 for the weight gradients of configs[2] (round 3: one block per CU, split and products in sequence),
        // and it cannot share a CU with a persistent sweep block, which is where this flag is for: the decoder's GEMMs waited
        // for the forward sweep to end, 0.98 -> 1.03 ms per step)
        if (launch_bf16_tile_t<64, 64, 32, 3>(p, grid, ak, bk, stream)) return arcvae_launch_status();
        p.kchunk = ((K + BK - 1) / BK) * BK;     // operands not vector-loadable: the f32 kernels below
    }
    // throughput mode: bf16 operands, f32 accumulate (everything but the minibatch-sized products on the chain, which
    // are latency-bound: they keep the skinny f32 path below)
    if ((flags & ARCVAE_GEMM_BF16) && !(transA == 0 && M <= 256) ) {
        // 128 x 128 tiles when they fill the chip
        // (for the K-long weight-gradient shapes the 64 x 64 tile with 2 K slices measured 207 TFLOP/s, the 128 x 128 tile
        // with 20 slices 81: the float-atomic epilogue is paid per slice)
        const bool big = M >= 128 && N >= 128 && ceil_div(M, 128) * ceil_div(N, 128) >= 256;
        const int bm = big ? 128 : 64;
        dim3 grid(ceil_div(N, bm), ceil_div(M, bm), 1);
        if ((flags & ARCVAE_GEMM_SPLITK) && p.act == 0) {
            static const int target = arcvae_env_int("ARCVAE_SPLITK_BLOCKS", 512);
            int z = min(ceil_div(target, (int)(grid.x * grid.y)), max(1, K / 256));
            if (z > 1) {
                p.kchunk = ceil_div(ceil_div(K, z), 64) * 64;
                grid.z = ceil_div(K, p.kchunk);
                if (!p.accumulate && arcvae_zero(C, M, N, ldc, stream) != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
            }
        }
        // K-tile 32 (a 64-deep tile measured no better: 251 vs 278 TFLOP/s on [40960 x 2048] x K = 512)
        const bool fits = big ? launch_bf16_tile_t<128, 128, 32>(p, grid, ak, bk, stream)
                              : launch_bf16_tile_t<64, 64, 32>(p, grid, ak, bk, stream);
        if (fits) return arcvae_launch_status();
        p.kchunk = ((K + BK - 1) / BK) * BK;     // operands not vector-loadable: the f32 kernels below
    }
    // Skinny path: minibatch-sized M on the critical path.
    if (!(flags & ARCVAE_GEMM_NO_SKINNY) && ak && M <= 256 && (K % 64) == 0 && (lda % 4) == 0 &&
        aligned16(A) && (!bk || ((ldb % 4) == 0 && aligned16(B)))) {
        dim3 grid(ceil_div(N, 16), ceil_div(M, 16), 1);
        if (bk) hipLaunchKernelGGL(gemm_skinny_kernel<true>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(gemm_skinny_kernel<false>, grid, dim3(256), 0, stream, p);
        return arcvae_launch_status();
    }

    // vector (16-byte) global loads need the contiguous extent and the leading dimension to be
    // multiples of 4 floats and an aligned base; otherwise scalar loads (e.g. ld = E+C = 129).
    const bool va4 = aligned16(A) && (lda % 4) == 0 && ((ak ? K : M) % 4) == 0;
    const bool vb4 = aligned16(B) && (ldb % 4) == 0 && ((bk ? K : N) % 4) == 0;

    // Tile choice (tools/bench_gemm.py on MI355X): the 64x64 tile (one 32x32 MFMA tile per wave, 4+ blocks per
    // CU hiding each other's barrier bubbles) beats the 128x128 tile on every shape of the training step --
    // 73 vs 50 TFLOP/s on the [5120,256]x[256,1024] projection, 44 vs 23 on [5120,1024]x[1024,256] -- so the
    // big tile is only chosen when even it oversubscribes the chip several times.
    const int blocks128 = ceil_div(M, 128) * ceil_div(N, 128);
    bool big = M >= 128 && N >= 128 && blocks128 >= 2048;
    if (flags & ARCVAE_GEMM_TILE128) big = true;
    if (flags & ARCVAE_GEMM_TILE64) big = false;
    const int bm = big ? 128 : 64, bn = big ? 128 : 64;
    dim3 grid(ceil_div(N, bn), ceil_div(M, bm), 1);
    if ((flags & ARCVAE_GEMM_SPLITK) && p.act == 0) {
        const int blocks = grid.x * grid.y;
        static const int target = arcvae_env_int("ARCVAE_SPLITK_BLOCKS", 512);
        int want = ceil_div(target, blocks);
        const int maxz = max(1, K / 128);  // at least 128 of K per slice
        int z = min(want, maxz);
        if (z > 1) {
            p.kchunk = ceil_div(ceil_div(K, z), BK) * BK;
            z = ceil_div(K, p.kchunk);
            grid.z = z;
            if (!p.accumulate) {
                if (arcvae_zero(C, M, N, ldc, stream) != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
            }
        }
    }
    if (big) launch_tile_t<128, 128>(p, grid, ak, bk, va4, vb4, stream);
    else launch_tile_t<64, 64>(p, grid, ak, bk, va4, vb4, stream);
    return arcvae_launch_status();
}

// Two skinny products in one launch (internal, ops.h): C_i[M_i,N_i] (+)= A_i[M_i,K_i] . op(B_i) (+ bias_i) (act_i), i = 0, 1;
// A K-contiguous rows; transB as arcvae_gemm_f32 (the same for both); flags_i: ARCVAE_GEMM_ACCUMULATE / _TANH / _DTANH.
// Same operand requirements as the skinny path of arcvae_gemm_f32 (K % 64 == 0, 16-byte aligned rows, M <= 256).
int arcvae_gemm_skinny_pair(int transB, const int* M, const int* N, const int* K, const float* const* A, const int* lda,
                            const float* const* B, const int* ldb, float* const* C, const int* ldc,
                            const float* const* bias, const int* flags, hipStream_t stream) {
    GemmPair g;
    for (int i = 0; i < 2; ++i) {
        GemmP& p = g.p[i];
        if (M[i] <= 0 || N[i] <= 0 || K[i] <= 0 || !A[i] || !B[i] || !C[i] || (K[i] % 64) != 0 || (lda[i] % 4) != 0 || M[i] > 256 ||
            !aligned16(A[i]) || (transB && ((ldb[i] % 4) != 0 || !aligned16(B[i]))))
            return ARCVAE_ERR_ARG;
        p.A = A[i]; p.B = B[i]; p.C = C[i]; p.bias = bias[i];
        p.M = M[i]; p.N = N[i]; p.K = K[i]; p.lda = lda[i]; p.ldb = ldb[i]; p.ldc = ldc[i];
        p.accumulate = (flags[i] & ARCVAE_GEMM_ACCUMULATE) ? 1 : 0;
        p.act = (flags[i] & ARCVAE_GEMM_TANH) ? 1 : ((flags[i] & ARCVAE_GEMM_DTANH) ? 2 : 0);
        if (p.act == 2 && (!bias[i] || p.accumulate)) return ARCVAE_ERR_ARG;
        p.kchunk = 0;
    }
    g.nx0 = ceil_div(N[0], 16);
    dim3 grid(g.nx0 + ceil_div(N[1], 16), ceil_div(max(M[0], M[1]), 16), 1);
    if (transB) hipLaunchKernelGGL(gemm_skinny_pair_kernel<true>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(gemm_skinny_pair_kernel<false>, grid, dim3(256), 0, stream, g);
    return arcvae_launch_status();
}

// C_i[M,N] += A_i^T . B_i for i < n, one launch: A_i stored [K_i, M] (lda), B_i stored [K_i, N] (ldb), all f32
// atomics (split-K inside each problem).  Internal (ops.h); falls back to n single launches when the operands do
// not allow the 16-byte path.
// colsum (optional, per problem, entries may be null): colsum_i[M] += column sums of A_i over K_i -- inside the split kernel
// where that runs, by arcvae_colsum_accum launches on every other path.
int arcvae_gemm_tn_group_accum(int n, int M, int N, const int* K, const float* const* A, int lda,
                               const float* const* B, int ldb, float* const* C, int ldc, int allow_split,
                               float* const* colsum, hipStream_t stream) {
    if (n <= 0 || n > ARCVAE_GEMM_GROUP_MAX || M <= 0 || N <= 0) return ARCVAE_ERR_ARG;
    auto colsums_by_launch = [&]() -> int {
        for (int i = 0; colsum && i < n; ++i)
            if (colsum[i] && K[i] > 0) {
                const int rc = arcvae_colsum_accum(A[i], K[i], M, lda, colsum[i], 1.0f, stream);
                if (rc) return rc;
            }
        return ARCVAE_OK;
    };
    if ((allow_split & 4) && M >= 256 && N >= 256) {   // throughput mode, large outputs: one bf16 tile-kernel launch per problem
        for (int i = 0; i < n; ++i) {
            if (K[i] <= 0) continue;
            const int rc = arcvae_gemm_f32(1, 0, M, N, K[i], A[i], lda, B[i], ldb, C[i], ldc, nullptr,
                                           ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK | ARCVAE_GEMM_BF16, stream);
            if (rc) return rc;
        }
        return colsums_by_launch();
    }
    if (allow_split & 7) {   // bit 0 = split-bf16 kernel; bit 1 = its 128-row tile (no sweep resident); bit 2 = one bf16 product (throughput mode)
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && K[i] > 0 && split_tn_ok(M, N, A[i], lda, B[i], ldb);
        if (ok) {
            const bool ride = arcvae_env_int("ARCVAE_COLSUM_FUSED", 1) != 0;   // 0: column sums by launch (A/B experiments)
            SplitTN q[ARCVAE_SPLIT_GROUP_MAX];
            for (int i = 0; i < n; ++i) {
                q[i].A = A[i]; q[i].B = B[i]; q[i].C = C[i]; q[i].colsum = (colsum && ride) ? colsum[i] : nullptr; q[i].M = M; q[i].N = N; q[i].K = K[i];
                q[i].lda = lda; q[i].ldb = ldb; q[i].ldc = ldc; q[i].kchunk = 0;
            }
            const int rc = launch_split_tn_group(n, q, (allow_split & 2) != 0, (allow_split & 4) != 0, stream);
            if (rc || ride) return rc;
            return colsums_by_launch();
        }
    }
    if (allow_split & 8) {   // the three-piece tile GEMM (LDS-staged, ~130 registers): the MFMA-bound regime's choice -- 2.7x less
        // matrix-pipe time than the exact-f32 tile GEMM beside the tiled sweeps (bs 2048: 22.3 -> 21.3 ms, configs[2]: 42.0 -> 40.9)
        for (int i = 0; i < n; ++i) {
            if (K[i] <= 0) continue;
            const int rc = arcvae_gemm_f32(1, 0, M, N, K[i], A[i], lda, B[i], ldb, C[i], ldc, nullptr,
                                           ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK | ARCVAE_GEMM_SPLIT3 | ARCVAE_GEMM_TILE64, stream);
            if (rc) return rc;
        }
        return colsums_by_launch();
    }
    bool vec = (lda % 4) == 0 && (ldb % 4) == 0 && (M % 4) == 0 && (N % 4) == 0;
    for (int i = 0; i < n; ++i) vec = vec && aligned16(A[i]) && aligned16(B[i]) && K[i] > 0;
    if (!vec) {
        for (int i = 0; i < n; ++i) {
            if (K[i] <= 0) continue;
            const int rc = arcvae_gemm_f32(1, 0, M, N, K[i], A[i], lda, B[i], ldb, C[i], ldc, nullptr,
                                           ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK, stream);
            if (rc) return rc;
        }
        return colsums_by_launch();
    }
    GemmGroup g;
    g.n = n;
    const int tiles = ceil_div(M, 64) * ceil_div(N, 64);
    int ztot = 0;
    for (int i = 0; i < n; ++i) {
        GemmP& p = g.p[i];
        p.A = A[i]; p.B = B[i]; p.C = C[i]; p.bias = nullptr;
        p.M = M; p.N = N; p.K = K[i]; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
        p.accumulate = 1; p.act = 0;
        static const int target = arcvae_env_int("ARCVAE_GROUP_BLOCKS", 512);
        int z = min(ceil_div(target, tiles * n), max(1, K[i] / 128));
        z = max(1, z);
        p.kchunk = ceil_div(ceil_div(K[i], z), BK) * BK;
        z = ceil_div(K[i], p.kchunk);
        g.zoff[i] = ztot;
        ztot += z;
    }
    g.zoff[n] = ztot;
    for (int i = n; i < ARCVAE_GEMM_GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.zoff[i + 1] = ztot; }
    dim3 grid(ceil_div(N, 64), ceil_div(M, 64), ztot);
    const unsigned pad = arcvae_side_lds_pad(2 * BK * (64 + 64 + 2 * PAD) * sizeof(float));
    hipLaunchKernelGGL((gemm_tile_group_kernel<64, 64, false, false, 4, 4>), grid, dim3(256), pad, stream, g);
    if (arcvae_launch_status() != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
    return colsums_by_launch();
}

// ---- weight gradients of the tiled three-piece sweeps, straight from the sweeps' operand planes (round 3) ------------------
// In the MFMA-bound regime every dG and h value already exists as hi / mid / lo bf16 planes: the step epilogues write them as
// the next launches' operands ([plane][col >> 5][row][32], csrc/lstm.hip).  Kept for ALL t (full-length rings) they are the
// weight-gradient GEMM's operands too: dW[m][n] += sum over rows k of dG[k][m] . h[k][n] contracts over ROWS, and a 32-column
// block of a plane is a row-major [row][32] image -- `ds_read_b64_tr_b16` hands a lane 4 consecutive rows of one column
// (tools/probe_tr.hip shows the lane map), two of them are the 8 k of a 16x16x32 operand.  So: no f32 loads, no split on the
// VALU (the three-piece tile GEMM splits every element once per tile that needs it: 8-32 times), no transposed copy; a K-step
// is 24 contiguous 2 KB pieces moved global -> LDS by LDS-DMA.  Block = 128 (m: gate columns) x 128 (n: units), four waves of
// 64 x 64, K-step = 32 rows of one time step, three LDS stages of 48 KB (two K-steps in flight).
struct PlaneTN {
    const __bf16* A;    // dG planes of the layer, slot of t = 0: [t][plane][M >> 5][rows][32]
    const __bf16* B;    // h planes of the source layer:            [t][plane][N >> 5][rows][32]
    float* C;           // [M, N] row-major, +=
    float* colsum;      // optional [M]: += sum over rows of dG (the bias gradient rides as a product with a column of ones)
    int tA0, tB0;       // first time slot of each operand
    int ksteps;         // (time steps) * (rows / 32)
};
#define ARCVAE_PLANE_GROUP_MAX 8
struct PlaneTNGroup {
    PlaneTN p[ARCVAE_PLANE_GROUP_MAX];
    int n, M, N, rows, ldc, z;     // z = K slices per problem (atomic accumulation when > 1)
#ifdef ARCVAE_PLANE_CLOCK
    unsigned long long* clk;       // diagnostic build (tools/r4_plane_clock.sh): per block {shader cycles, 100 MHz ticks} around the K loop
#endif
};
typedef short s16x4_g __attribute__((ext_vector_type(4)));
struct PlaneFrag { s16x4_g lo, hi; };
namespace {
// DBG: timing experiments only (ARCVAE_PLANE_DEBUG, tools/r4_planes_knobs.sh): 1 no products, 2 no fragment reads either, 4 no LDS-DMA
// in the loop.  A template parameter since round 4: as run-time branches they cut the K-step into one basic block per column tile,
// and the compiler could not put a tile's fragment reads under the previous tile's products.
// Where this loop stands (round 4): 1.27 us per K-step at the 1.77-1.84 GHz the chip holds in it (diagnostic build
// -DARCVAE_PLANE_CLOCK) = 1.02 PFLOP/s of bf16 products, against 1.27-1.37 PFLOP/s of the vendor library's large bf16 GEMM on random
// data on the same box (tools/r4_bf16_peak.py) and 0.59 on a GEMM of this call's shape.  Built, parity-green and dropped (commit
// history): two consuming waves per SIMD with 4 or 8 loader waves (1.46 / 1.57 ms per call against 1.42), and the ring handed over
// through LDS flags instead of the block barrier, the next K-step's first reads ahead of the last tile's products (1.53 ms) -- the
// loop is bound by the matrix pipe at the rate the chip sustains, not by its bubbles.
template <int DBG>
__global__ __launch_bounds__(512) void wgrad_planes_kernel(PlaneTNGroup g) {
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];     // [3 stages][24 pieces: (operand, plane, column block)][32 rows][64 B]
    constexpr int STAGE = 24 * 2048;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Blocks are dealt to the XCDs round-robin by linear id: hand XCD k a CONTIGUOUS range of the (problem, m tile, n tile) order,
    // so that the 32 blocks it runs at a time are 8 m tiles x 4 n tiles of one problem -- every dG piece is fetched by 4 blocks
    // and every h piece by 8 under ONE L2 (plain order: the blocks that share a dG piece sit on four different XCDs).
    int bx, by, bz;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
        const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = id & 7, slot = id >> 3, q = n >> 3, r = n & 7;
        const unsigned nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        bx = __builtin_amdgcn_readfirstlane((int)(nid % gx));
        by = __builtin_amdgcn_readfirstlane((int)((nid / gx) % gy));
        bz = __builtin_amdgcn_readfirstlane((int)(nid / (gx * gy)));
    }
    const int prob = __builtin_amdgcn_readfirstlane(bz / g.z), slice = bz - prob * g.z;
    // the problem's fields once, into scalar registers: an ordinary load inside the K loop would make the compiler drain
    // the LDS-DMA queue (vmcnt(0)) at its use
    PlaneTN p;
    {
        const PlaneTN& q = g.p[prob];
        // (readfirstlane returns a signed int: without the unsigned cast a low half with its top bit set sign-extends over the
        // high half -- seen as a memory fault at 0xffff....)
        auto uni64 = [](unsigned long long v) -> unsigned long long {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
            return ((unsigned long long)hi << 32) | (unsigned long long)lo;
        };
        p.A = reinterpret_cast<const __bf16*>(uni64(reinterpret_cast<unsigned long long>(q.A)));
        p.B = reinterpret_cast<const __bf16*>(uni64(reinterpret_cast<unsigned long long>(q.B)));
        p.C = reinterpret_cast<float*>(uni64(reinterpret_cast<unsigned long long>(q.C)));
        p.colsum = reinterpret_cast<float*>(uni64(reinterpret_cast<unsigned long long>(q.colsum)));
        p.tA0 = __builtin_amdgcn_readfirstlane(q.tA0); p.tB0 = __builtin_amdgcn_readfirstlane(q.tB0);
        p.ksteps = __builtin_amdgcn_readfirstlane(q.ksteps);
    }
    const int gM = g.M, gN = g.N, grows = g.rows;
    const int m0 = by * 128, n0 = bx * 128;
    const int rb = grows >> 5;
    const int per = (p.ksteps + g.z - 1) / g.z;
    const int kbeg = slice * per, kend = min(p.ksteps, kbeg + per);
    if (kbeg >= kend) return;                                           // block-uniform
    const long planeA = (long)grows * gM, planeB = (long)grows * gN;
    // loader: wave 4 + w moves pieces 6w .. 6w+5 of a stage, two 1 KB halves each (a lane: 16 bytes of row 16 hf + (lane >> 2)):
    // ALWAYS 12 LDS-DMA instructions per loader wave and stage (the waits below count them; a column block beyond N repeats the last
    // one, its image is never read).  The LDS image is lane-linear; the 16-byte octet a lane FETCHES is swizzled by its row's
    // bit 3 so that the four 16-lane groups of a transposing read (rows 8 kq ..) fall on both halves of the banks.
    auto issue = [&](int ks, int stage) {
        const int t = ks / rb, b0 = (ks - t * rb) << 5;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int pc = 6 * (wave & 3) + q, op = pc >= 12 ? 1 : 0, pl = (pc - 12 * op) >> 2, cb = pc & 3;
            const int colblock = min(((op ? n0 : m0) >> 5) + cb, ((op ? gN : gM) >> 5) - 1);
            const __bf16* src = op ? p.B + ((long)(p.tB0 + t) * 3 + pl) * planeB + ((long)colblock * grows + b0) * 32
                                   : p.A + ((long)(p.tA0 + t) * 3 + pl) * planeA + ((long)colblock * grows + b0) * 32;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int row = 16 * hf + (lane >> 2), oct = (lane & 3) ^ (((row >> 3) & 1) << 1);
                // (as assembly: for the builtin the compiler drains the whole LDS-DMA queue -- vmcnt(0) -- in front of the first LDS
                // read that might alias it, i.e. every K-step; the counted waits below are the ordering)
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) char*)(pl_smem + stage * STAGE + pc * 2048 + hf * 1024));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + row * 32 + oct * 8), "s"(dst) : "memory");
            }
        }
    };
    // Eight waves, two per SIMD: waves 0-3 CONSUME (a 64 x 64 output tile each: fragment reads + products), waves 4-7 LOAD
    // (the LDS-DMA of a stage).  An LDS-DMA instruction costs its wave 60-180 cycles of issue (MI355X_MICROARCH.md), 12 of them
    // per K-step as much as the K-step's products: issued by the consuming waves the two simply added up (measured: LDS-DMA
    // 0.52 us + reads and products 0.96 us = 1.41 us per K-step); on waves of their own they run under the products of the
    // consumer that shares the SIMD.
    const bool loader = wave >= 4;
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const bool active = n0 + 64 * wn < gN;                               // wave-uniform
    const int kq = lane >> 4, kp = (lane & 15) >> 2, a4 = lane & 3;
    // transposing read: lane 4 kp + a4 of a 16-lane group supplies row 8 kq + 4 hh + kp, columns 16 tt2 + 4 a4 .. + 3 of the block
    auto frag = [&](int stage, int op, int pl, int tt) -> bf16x8_t {
        const int cb = 2 * (op ? wn : wm) + (tt >> 1), tt2 = tt & 1;
        const char* base = pl_smem + stage * STAGE + ((op * 3 + pl) * 4 + cb) * 2048;
        const int oct = (2 * tt2 + (a4 >> 1)) ^ ((kq & 1) << 1);
        const char* q0 = base + (8 * kq + kp) * 64 + oct * 16 + (a4 & 1) * 8;
        PlaneFrag f;
        f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_g __attribute__((address_space(3)))*)(q0));
        f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_g __attribute__((address_space(3)))*)(q0 + 4 * 64));
        return __builtin_bit_cast(bf16x8_t, f);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias rider: colsum[m] += sum_k dG[k][m] as one more product per A fragment with a B operand of ones (every column of the
    // result is the sum; column 0 is kept) -- by the waves of the block column n0 = 0 that own the tile's first 64 columns.
    // Replaces a separate pass over the f32 dG (1.4 ms of kernel time per configs[2] step).
    const bool rider = !loader && p.colsum != nullptr && n0 == 0 && wn == 0;     // wave-uniform
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (__bf16)1.0f;
    // Three LDS stages, two K-steps in flight: a stage is read one barrier AFTER the counted wait that retires its LDS-DMA (each
    // wave waits for its own 12, the barrier makes that true for all four), and re-filled one barrier after its last read.
    if (loader) {
        issue(kbeg, 0);
        if (kbeg + 1 < kend) issue(kbeg + 1, 1);
    }
    int cur = 0;
#ifdef ARCVAE_PLANE_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ks = kbeg; ks < kend; ++ks) {
        if (loader) {
            if (ks + 1 < kend) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // Invariant of the ring: the stage the loaders refill right behind THIS barrier is the one the consumers
            // read at K-step ks - 1, so every fragment read of ks - 1 must have RETURNED before a consumer arrives here.  The MFMAs
            // that consumed them are no memory operations and gfx950's barrier carries no implicit waitcnt, so say it explicitly
            // (free: those products were issued already) instead of relying on where the scheduler puts the waitcnt.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");       // (compiler fence: no LDS read of this K-step may be scheduled above the barrier)
        if (loader && ks + 2 < kend && !(DBG & 4)) issue(ks + 2, cur == 0 ? 2 : cur - 1);         // (the stage read at ks - 1)
        if (!loader && active && !(DBG & 2)) {
            // A fragments of the K-step stay in registers (48); the B fragments of ONE 16-column tile (12) are read while the
            // products of the tile before it run (two sets).  Read order = use order: the K-step's first product waits for 10 reads,
            // not for 30.  (Round 4: reads and products of a K-step used to alternate -- reads, drain, 24 products, reads ... --
            // 1.29 us per K-step without any LDS-DMA against 0.64 us of products.)
            bf16x8_t fa[3][4], fb[2][3];
#define PLANE_LOADA(PL) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) fa[PL][tt] = frag(cur, 0, PL, tt);
            PLANE_LOADA(2) fb[0][0] = frag(cur, 1, 0, 0);
            PLANE_LOADA(0) fb[0][2] = frag(cur, 1, 2, 0);
            PLANE_LOADA(1) fb[0][1] = frag(cur, 1, 1, 0);
#undef PLANE_LOADA
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = j & 1;
                if (j < 3) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fb[b ^ 1][pl] = frag(cur, 1, pl, j + 1);
                }
                if constexpr ((DBG & 1) != 0) {      // (timing experiment: keep the reads alive without the products)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) { acc[pl][j][0] += (float)fa[pl][j][0]; acc[pl][j][1] += (float)fb[b][pl][0]; }
                } else {
                    // the six products of weight >= 2^-16, small ones first (pieces: 0 hi, 1 mid, 2 lo)
#define PLANE_S3(PA, PB)                                                                                         \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA][i], fb[b][PB], acc[i][j], 0, 0, 0);
                    PLANE_S3(2, 0) PLANE_S3(0, 2) PLANE_S3(1, 1) PLANE_S3(0, 1) PLANE_S3(1, 0) PLANE_S3(0, 0)
#undef PLANE_S3
                }
            }
            if constexpr ((DBG & 1) == 0) {
                // The order asked of the scheduler for this block (48 reads, 96 products): the 10 reads of the first product, then
                // the other 20 reads of tile 0 under its first 8 products, then the 6 reads of the next tile under the first 12-18 products of a tile.
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); }
#pragma unroll
                for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
                for (int k = 0; k < 6; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                for (int jj = 1; jj < 3; ++jj) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                if (rider) {
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)       // lo, mid, hi: small pieces first
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pl][i], ones, accb[i], 0, 0, 0);
                }
            }
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
#ifdef ARCVAE_PLANE_CLOCK
    if (g.clk && tid == 0) {
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        g.clk[2 * id] = __builtin_amdgcn_s_memtime() - clk_c0;
        g.clk[2 * id + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
    if (loader || !active) return;
    if (rider && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + m0 + 64 * wm + 16 * i + 4 * (lane >> 4) + r, accb[i][r]);
    }
    // D[m = 4 (lane >> 4) + r][n = lane & 15] of tile (i, j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 64 * wn + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 64 * wm + 16 * i + 4 * (lane >> 4) + r;
                float* c = p.C + (long)m * g.ldc + n;
                if (g.z > 1) atomicAdd(c, acc[i][j][r]);
                else *c += acc[i][j][r];
            }
        }
}

}  // namespace

// dW_i[M, N] += dG_i^T . h_i over the time steps [t, t + nT) of each problem, operands = the sweeps' three-plane copies (all
// time slots kept).  M % 128 == 0, N % 64 == 0, rows % 32 == 0.  Internal (ops.h).
int arcvae_wgrad_planes_group(int n, int M, int N, int rows, const void* const* A, const int* tA0, const void* const* B,
                              const int* tB0, const int* nT, float* const* C, int ldc, float* const* colsum /* optional */,
                              hipStream_t stream) {
    if (n <= 0 || n > ARCVAE_PLANE_GROUP_MAX || M <= 0 || N <= 0 || (M % 128) || (N % 64) || rows <= 0 || (rows % 32) || ldc < N)
        return ARCVAE_ERR_ARG;
    PlaneTNGroup g;
    g.n = n; g.M = M; g.N = N; g.rows = rows; g.ldc = ldc;
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!A[i] || !B[i] || !C[i] || nT[i] <= 0 || tA0[i] < 0 || tB0[i] < 0) return ARCVAE_ERR_ARG;
        PlaneTN& p = g.p[i];
        p.A = static_cast<const __bf16*>(A[i]); p.B = static_cast<const __bf16*>(B[i]); p.C = C[i];
        p.colsum = colsum ? colsum[i] : nullptr;
        p.tA0 = tA0[i]; p.tB0 = tB0[i]; p.ksteps = nT[i] * (rows / 32);
        kmax = max(kmax, p.ksteps);
    }
    for (int i = n; i < ARCVAE_PLANE_GROUP_MAX; ++i) g.p[i] = g.p[0];
    const int tiles = (M / 128) * ceil_div(N, 128);
    static const int target = arcvae_env_int("ARCVAE_PLANE_BLOCKS", 448);     // blocks wanted per launch
    g.z = max(1, min(ceil_div(target, tiles * n), kmax / 16));                // at least 16 K-steps per slice
    const dim3 grid(ceil_div(N, 128), M / 128, n * g.z);
    const size_t lds = 3 * 24 * 2048;
#ifdef ARCVAE_PLANE_CLOCK
    // (diagnostic build only: a buffer per call, synchronous -- no host state kept between calls)
    unsigned long long* clk_buf = nullptr;
    const size_t clk_n = 2 * (size_t)grid.x * grid.y * grid.z;
    if (arcvae_env_int("ARCVAE_PLANE_CLOCK_PRINT", 0)) (void)hipMalloc((void**)&clk_buf, clk_n * sizeof(unsigned long long));
    g.clk = clk_buf;
#endif
#define PLANE_LAUNCH(D)                                                                                              \
    { (void)hipFuncSetAttribute((const void*)wgrad_planes_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL(wgrad_planes_kernel<D>, grid, dim3(512), lds, stream, g); }
    switch (arcvae_env_int("ARCVAE_PLANE_DEBUG", 0)) {      // (timing experiments; results are wrong for anything but 0)
        case 1: PLANE_LAUNCH(1) break;
        case 2: PLANE_LAUNCH(2) break;
        case 4: PLANE_LAUNCH(4) break;
        case 5: PLANE_LAUNCH(5) break;
        case 6: PLANE_LAUNCH(6) break;
        default: PLANE_LAUNCH(0)
    }
#undef PLANE_LAUNCH
#ifdef ARCVAE_PLANE_CLOCK
    if (clk_buf) {       // in-kernel clock = shader cycles / (100 MHz ticks) x 100 MHz, over the blocks of this call
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> host(clk_n);
        (void)hipMemcpy(host.data(), clk_buf, clk_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(clk_buf);
        std::vector<double> ghz, us;
        for (size_t b = 0; b < clk_n / 2; ++b)
            if (host[2 * b + 1] > 0) { ghz.push_back((double)host[2 * b] / (double)host[2 * b + 1] * 0.1); us.push_back((double)host[2 * b + 1] * 0.01); }
        std::sort(ghz.begin(), ghz.end());
        std::sort(us.begin(), us.end());
        if (!ghz.empty())
            fprintf(stderr, "[plane clock] blocks %zu: in-kernel clock min %.3f median %.3f max %.3f GHz; K loop median %.1f us\n", ghz.size(),
                    ghz.front(), ghz[ghz.size() / 2], ghz.back(), us[us.size() / 2]);
    }
#endif
    return arcvae_launch_status();
}

// C_i[M,N] += A_i^T . B_i from octet-major bf16 operand copies (wgrad_octet_kernel), i < n <= 8, one launch.  Internal (ops.h).
int arcvae_wgrad_octet_group(int n, int M, int N, const int* K, const void* const* A, const void* const* B,
                             float* const* C, int ldc, hipStream_t stream) {
    if (n <= 0 || n > ARCVAE_OCT_GROUP_MAX || M <= 0 || N <= 0 || ldc < N) return ARCVAE_ERR_ARG;
    OctTNGroup g;
    g.n = n; g.M = M; g.N = N; g.ldc = ldc;
    const int tiles = ceil_div(M, 256) * ceil_div(N, 256);
    static const int target = arcvae_env_int("ARCVAE_OCT_BLOCKS", 512);      // blocks wanted per launch (two rounds of 256 CUs)
    int ztot = 0;
    for (int i = 0; i < n; ++i) {
        if (!A[i] || !B[i] || !C[i] || K[i] <= 0 || (K[i] % 16) != 0) return ARCVAE_ERR_ARG;
        OctTN& p = g.p[i];
        p.A = static_cast<const __bf16*>(A[i]); p.B = static_cast<const __bf16*>(B[i]); p.C = C[i]; p.K = K[i];
        int z = max(1, min(ceil_div(target, tiles * n), K[i] / 512));        // at least 32 K = 16 steps per slice
        p.kchunk = ceil_div(ceil_div(K[i], z), 16) * 16;
        z = ceil_div(K[i], p.kchunk);
        g.zoff[i] = ztot;
        ztot += z;
    }
    g.zoff[n] = ztot;
    for (int i = n; i < ARCVAE_OCT_GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.zoff[i + 1] = ztot; }
    hipLaunchKernelGGL(wgrad_octet_kernel, dim3(ceil_div(N, 256), ceil_div(M, 256), ztot), dim3(256), 0, stream, g);
    return arcvae_launch_status();
}

// h = zero_state_cell(A . W^T + bias) for a decoder layer whose pre-activations are not needed afterwards (gemm_cell_zero_kernel).
// Internal (ops.h).  Returns ARCVAE_ERR_ARG when the operands do not allow the 16-byte loads (the caller then takes the two-launch path).
int arcvae_gemm_cell_zero(int M, int H, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                          float* Hout, hipStream_t stream) {
    if (M <= 0 || H <= 0 || K <= 0 || !A || !W || !bias || !Hout) return ARCVAE_ERR_ARG;
    if ((K % 4) != 0 || (lda % 4) != 0 || (ldw % 4) != 0 || !aligned16(A) || !aligned16(W) || lda < K || ldw < K) return ARCVAE_ERR_ARG;
    CellGemmP p;
    p.A = A; p.W = W; p.bias = bias; p.Hout = Hout; p.M = M; p.H = H; p.K = K; p.lda = lda; p.ldw = ldw;
    const int lds = (int)sizeof(float) * 128 * 129;   // >= the operand tiles' 2 * 16 * (132 + 132) floats
    (void)hipFuncSetAttribute((const void*)gemm_cell_zero_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(gemm_cell_zero_kernel, dim3(ceil_div(4 * H, 128), ceil_div(M, 128)), dim3(256), lds, stream, p);
    return arcvae_launch_status();
}
