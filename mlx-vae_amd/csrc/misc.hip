// HBM-bound helper kernels: column sums, small transposes, segment (token) sums, and the
// fused un-bias-corrected Adam (reference trainer.py:75-76,320,324 -> MLX optim.Adam, Q7).
#include "ops.h"
#include <algorithm>

namespace {

// ---- colsum: out[c] += scale * sum_r X[r, c] ------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int rows, int cols,
                                                     int ld, float* out, float scale, int rows_per_blk) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int rbeg = blockIdx.y * rows_per_blk, rend = min(rows, rbeg + rows_per_blk);
    float s = 0.f;
    if (c < cols)
        for (int r = rbeg + rl; r < rend; r += 4) s += X[(long)r * ld + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        atomicAdd(out + c, s * scale);
    }
}

// ---- batched 2-D transpose -------------------------------------------------------------
struct TrJobs {
    const float* src[16];
    float* dst[16];
    int rows[16];
    int cols[16];
};
__global__ __launch_bounds__(256) void transpose_kernel(TrJobs j) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z;
    const int R = j.rows[z], Cc = j.cols[z];
    const float* __restrict__ s = j.src[z];
    float* d = j.dst[z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (c0 >= Cc || r0 >= R) return;  // whole block out of range for this (smaller) job
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + i][tx] = s[(long)r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (r < R && c < Cc) d[(long)c * R + r] = tile[tx][ty + i];
    }
}

// ---- k-chunk-major re-layout of LSTM weights for the step kernels (skinny.h: skinny_load_tiled) ---------
// mode 0 (forward):  W [4H, K] -> dst[(k/16)][perm(r)][k%16], perm(gate*H + unit) = (unit>>2)*16 + gate*4 + (unit&3)
//                    (the 16 gate columns of a 4-unit forward block become 16 consecutive rows)
// mode 1 (backward): W [4H, H] -> dst[(j/16)][unit][j%16]   (W^T, contraction index j = gate row, tiled)
// modes 2 / 3: the same two layouts in bf16, 32-wide chunks (throughput mode); modes 4 / 5: as three bf16 planes (three-piece form)
struct TileJobs {
    const float* src[16];
    float* dst[16];
    int cols[16];  // K (mode 0) or H (mode 1)
    int mode[16];
};
__global__ __launch_bounds__(256) void tile_weights_kernel(TileJobs j, int H) {
    // Both layouts move 16-byte pieces: mode 0 keeps 4 consecutive k together (a float4 of the source row is a float4
    // of the destination), mode 1 is a [16 rows x 64 k] transpose through LDS (reads coalesced along k, writes 16
    // consecutive rows of one k = 64 B, as float4 per thread).  K % 64 == 0 (H is a multiple of 64).
    __shared__ float tile[16][65];
    const int z = blockIdx.y;
    const int K = j.cols[z];
    const float* __restrict__ src = j.src[z];
    float* dst = j.dst[z];
    const int tid = threadIdx.x;
    if (j.mode[z] == 0) {
        const long n4 = (long)4 * H * K / 4;
        const int k4n = K >> 2;
        for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
            const int r = (int)(i / k4n), k = (int)(i - (long)r * k4n) * 4;
            const int gate = r / H, unit = r - gate * H;
            const int rp = (unit >> 2) * 16 + gate * 4 + (unit & 3);
            *reinterpret_cast<float4*>(dst + ((long)(k >> 4) * 4 * H + rp) * 16 + (k & 15)) =
                *reinterpret_cast<const float4*>(src + (long)r * K + k);
        }
        return;
    }
    if (j.mode[z] == 4 || j.mode[z] == 5) {
        // three-piece form (ARCVAE_LSTM_SPLIT3): the layouts of modes 2 / 3 as THREE bf16 planes (hi, mid, lo: plane stride
        // 4H * K elements) whose sum is the f32 weight to 2^-24 -- the operands of lstm.hip's tile_contract_s
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        auto bf = [](float x) { return __builtin_convertvector(f2v{x, 0.f}, b2v)[0]; };
        auto pk2 = [](__bf16 a, __bf16 b) { return __builtin_bit_cast(unsigned, b2v{a, b}); };
        __bf16* d16 = reinterpret_cast<__bf16*>(dst);
        const long n8 = (long)4 * H * K / 8, plane = (long)4 * H * K;
        const int k8n = K >> 3;
        for (long i = (long)blockIdx.x * 256 + tid; i < n8; i += (long)gridDim.x * 256) {
            float v[8];
            long off;
            if (j.mode[z] == 4) {      // forward layout: 8 consecutive k of a permuted row
                const int r = (int)(i / k8n), k = (int)(i - (long)r * k8n) * 8;
                const int gate = r / H, unit = r - gate * H;
                const int rp = (unit >> 2) * 16 + gate * 4 + (unit & 3);
                const float4 a = *reinterpret_cast<const float4*>(src + (long)r * K + k);
                const float4 b = *reinterpret_cast<const float4*>(src + (long)r * K + k + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                off = ((long)(k >> 5) * 4 * H + rp) * 32 + (k & 31);
            } else {                   // BPTT layout (transposed): 8 consecutive gate rows of one k
                const int g0 = (int)(i / K) * 8, k = (int)(i % K);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[(long)(g0 + e) * K + k];
                off = ((long)(g0 >> 5) * K + k) * 32 + (g0 & 31);
            }
            __bf16 p[3][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[0][e] = bf(v[e]);
                const float r1 = v[e] - (float)p[0][e];
                p[1][e] = bf(r1);
                p[2][e] = bf(r1 - (float)p[1][e]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<u4v*>(d16 + q * plane + off) =
                    u4v{pk2(p[q][0], p[q][1]), pk2(p[q][2], p[q][3]), pk2(p[q][4], p[q][5]), pk2(p[q][6], p[q][7])};
        }
        return;
    }
    if (j.mode[z] == 2 || j.mode[z] == 3) {
        // throughput mode: the same two layouts in bf16 with 32-wide chunks.  mode 2 (forward): 8 consecutive k of a
        // permuted row = two float4 reads, one 16-byte write.  mode 3 (BPTT, transposed): 8 consecutive gate rows g of one
        // k = 8 reads (each coalesced over the lanes' consecutive k), one 16-byte write.
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        auto pk = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, b2v)); };
        __bf16* d16 = reinterpret_cast<__bf16*>(dst);
        const long n8 = (long)4 * H * K / 8;
        if (j.mode[z] == 2) {
            const int k8n = K >> 3;
            for (long i = (long)blockIdx.x * 256 + tid; i < n8; i += (long)gridDim.x * 256) {
                const int r = (int)(i / k8n), k = (int)(i - (long)r * k8n) * 8;
                const int gate = r / H, unit = r - gate * H;
                const int rp = (unit >> 2) * 16 + gate * 4 + (unit & 3);
                const float4 a = *reinterpret_cast<const float4*>(src + (long)r * K + k);
                const float4 b = *reinterpret_cast<const float4*>(src + (long)r * K + k + 4);
                *reinterpret_cast<u4v*>(d16 + ((long)(k >> 5) * 4 * H + rp) * 32 + (k & 31)) =
                    u4v{pk(a.x, a.y), pk(a.z, a.w), pk(b.x, b.y), pk(b.z, b.w)};
            }
        } else {
            for (long i = (long)blockIdx.x * 256 + tid; i < n8; i += (long)gridDim.x * 256) {
                const int g0 = (int)(i / K) * 8, k = (int)(i % K);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[(long)(g0 + e) * K + k];
                *reinterpret_cast<u4v*>(d16 + ((long)(g0 >> 5) * K + k) * 32 + (g0 & 31)) =
                    u4v{pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7])};
            }
        }
        return;
    }
    const int ktiles = K >> 6, ntiles = (4 * H / 16) * ktiles;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int R = t / ktiles, k0 = (t - R * ktiles) * 64;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {                 // 1024 elements: row = (tid >> 6) + 4e, col = tid & 63
            const int rr = (tid >> 6) + 4 * e, kk = tid & 63;
            tile[rr][kk] = src[(long)(R * 16 + rr) * K + k0 + kk];
        }
        __syncthreads();
        const int kk = tid >> 2, r4 = (tid & 3) * 4;
        float4 v = make_float4(tile[r4][kk], tile[r4 + 1][kk], tile[r4 + 2][kk], tile[r4 + 3][kk]);
        *reinterpret_cast<float4*>(dst + ((long)R * K + k0 + kk) * 16 + r4) = v;
    }
}

// ---- segment sum: out[seg[r], :] += X[r, :] --------------------------------------------
// Block = 64 columns x 4 row lanes over a 256-row chunk; per-block [nseg][64] table in LDS
// (ds_add_f32), flushed with one global f32 atomic per touched entry.
__global__ __launch_bounds__(256) void segsum_kernel(const float* __restrict__ X, const int32_t* __restrict__ seg,
                                                     int rows, int nseg, int cols, float* out) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int i = threadIdx.x; i < nseg * 64; i += 256) tab[i] = 0.f;
    __syncthreads();
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int rbeg = blockIdx.y * 256, rend = min(rows, rbeg + 256);
    if (c < cols) {
        for (int r = rbeg + rl; r < rend; r += 4) {
            int sgm = seg[r];
            sgm = min(max(sgm, 0), nseg - 1);
            atomicAdd(&tab[sgm * 64 + cl], X[(long)r * cols + c]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseg * 64; i += 256) {
        const float v = tab[i];
        const int cc = blockIdx.x * 64 + (i & 63);
        if (v != 0.f && cc < cols) atomicAdd(out + (long)(i >> 6) * cols + cc, v);
    }
}

// ---- token-table gradient -> embedding / layer-0 input-weight / bias gradients, ONE launch ------------------
// The layer-0 input projection of an LSTM is a [V,4H] table (emb . Wx0^T + b0), so its gradient arrives as
// dT [V,4H] and is folded back with three tiny products (42 MFLOP at the default shape):
//   dEmb [V,E]   += dT . Wx0[:, :E]          (K = 4H)
//   dWx0 [4H,E]  += dT^T . Emb               (K = V)       (row stride ldw = E or E + C)
//   db0  [4H]    += colsum(dT)
// As three split-K tile-GEMM / colsum launches they cost 13 + 17 + 7 us of pure launch and pipeline latency --
// in the encoder that is the exposed tail of the training step.  Here: one launch of plain f32 FMAs with every
// operand staged through LDS first (no global load inside an FMA loop).  Block roles by blockIdx.x:
//   [0, nA)        dEmb : 4 table rows x 128 embedding columns x one 64-wide slice of K = 4H; f32 atomics over slices
//   [nA, nA+nB)    dWx0 : 16 gate rows x 128 embedding columns, K = V
//   [nA+nB, ...)   db0  : 256 gate columns each
constexpr int TF_ROWS = 4;
constexpr int TF_KS = 64;
__global__ __launch_bounds__(256) void table_finalize_kernel(const float* __restrict__ dT, const float* __restrict__ Wx0,
                                                             int ldw, const float* __restrict__ emb, float* dEmb,
                                                             float* dWx0, float* db0, int V, int E, int G, int nA,
                                                             int nB, int etiles) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int tid = threadIdx.x, e_l = tid & 127, half = tid >> 7;
    int bid = blockIdx.x;
    if (bid < nA) {                                   // ---- dEmb
        const int kslices = G / TF_KS;
        const int ks = bid % kslices, et = (bid / kslices) % etiles, v0 = (bid / (kslices * etiles)) * TF_ROWS;
        const int k0 = ks * TF_KS, e0 = et * 128, e = e0 + e_l;
        float* wt = sh;                               // Wx0 tile [TF_KS][128]
        float* tt = sh + TF_KS * 128;                 // dT tile  [TF_ROWS][TF_KS]
        float* red = tt + TF_ROWS * TF_KS;            // [TF_ROWS][128] partials of half 1
        for (int i = tid; i < TF_KS * 128; i += 256) {
            const int kk = i >> 7, ee = i & 127;
            wt[i] = (e0 + ee < E) ? Wx0[(long)(k0 + kk) * ldw + e0 + ee] : 0.f;
        }
        {
            const int r = tid / TF_KS, kk = tid % TF_KS;  // 256 = TF_ROWS * TF_KS
            tt[tid] = (v0 + r < V) ? dT[(long)(v0 + r) * G + k0 + kk] : 0.f;
        }
        __syncthreads();
        float acc[TF_ROWS];
#pragma unroll
        for (int i = 0; i < TF_ROWS; ++i) acc[i] = 0.f;
        const int kb = half * (TF_KS / 2);
#pragma unroll 8
        for (int kk = kb; kk < kb + TF_KS / 2; ++kk) {
            const float w = wt[kk * 128 + e_l];
#pragma unroll
            for (int i = 0; i < TF_ROWS; ++i) acc[i] = fmaf(tt[i * TF_KS + kk], w, acc[i]);
        }
        if (half == 1)
#pragma unroll
            for (int i = 0; i < TF_ROWS; ++i) red[i * 128 + e_l] = acc[i];
        __syncthreads();
        if (half == 0 && e < E)
#pragma unroll
            for (int i = 0; i < TF_ROWS; ++i)
                if (v0 + i < V) atomicAdd(dEmb + (long)(v0 + i) * E + e, acc[i] + red[i * 128 + e_l]);
        return;
    }
    bid -= nA;
    if (bid < nB) {                                   // ---- dWx0: 16 gate rows (4 per quarter of the block) x 64 e, K = V
        // (64-wide e tiles: with 128 the emb tile alone was 64 KB of LDS, which kept this launch -- three per step, one of
        // them in the step's exposed tail -- from sharing a CU with the GEMM blocks around it: 33-44 us per launch in the
        // step against ~10 alone)
        const int etiles64 = (E + 63) >> 6;
        const int et = bid % etiles64, g0 = (bid / etiles64) * 16;
        const int e64 = tid & 63, q = tid >> 6;
        const int e0 = et * 64, e = e0 + e64;
        float* tt = sh;                               // dT^T tile [128][16]   (K = V in chunks of 128 tokens)
        float* em = sh + 128 * 16;                    // emb tile  [128][64]
        float acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = 0.f;
        for (int vb = 0; vb < V; vb += 128) {
            const int vn = min(128, V - vb);
            if (vb) __syncthreads();
            for (int i = tid; i < vn * 16; i += 256) tt[i] = dT[(long)(vb + (i >> 4)) * G + g0 + (i & 15)];
            for (int i = tid; i < vn * 64; i += 256) {
                const int v = i >> 6, ee = i & 63;
                em[i] = (e0 + ee < E) ? emb[(long)(vb + v) * E + e0 + ee] : 0.f;
            }
            __syncthreads();
            if (e < E) {
#pragma unroll 4
                for (int v = 0; v < vn; ++v) {
                    const float x = em[v * 64 + e64];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(tt[v * 16 + q * 4 + i], x, acc[i]);
                }
            }
        }
        if (e < E) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(dWx0 + (long)(g0 + q * 4 + i) * ldw + e, acc[i]);   // atomic: two
            // chunks' tables may be folded at the same time on different streams (engine: side and main)
        }
        return;
    }
    bid -= nB;
    const int g = bid * 256 + tid;                    // ---- db0
    if (g < G) {
        float a = 0.f;
        for (int v = 0; v < V; ++v) a += dT[(long)v * G + g];
        atomicAdd(db0 + g, a);
    }
}

__global__ void transpose_tokens_kernel(const int32_t* __restrict__ src, int32_t* dst, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * T) {
        const int t = i / B, b = i % B;
        dst[i] = src[b * T + t];
    }
}

// ---- Adam without bias correction (MLX optim.Adam default) ------------------
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p = p - lr*m/(sqrt(v)+eps)
// One launch over the flat [encoder | decoder] parameter buffer: the reference's two optimizers
// share lr/betas/eps and Adam state is elementwise, so one flat pass is the same update.
// HBM-bound: 4 streams read + 3 written = 28 B/param.
// guard_a / guard_b (optional): device error words (an expired gate, a persistent sweep that gave up).  When either
// is non-zero the gradients of this step are not trustworthy: nothing is updated, the parameters stay what they were.
// fin (optional, arcvae_adam_update_finalize): the single-process step's loss finalize -- CE row sums -> stats[2Z+3], recon and
// total scalars, the guard's NaN poison (latent.hip: recon_finalize_kernel) -- done by block 0 of THIS launch: one kernel less in
// the exposed tail of the step (round 4).
struct AdamFinalize {
    const float* rowloss; float* stats; float* scalars; int B, Z, T;
};
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n4, long n,
                                                   float lr, float b1, float b2, float omb1, float omb2, float eps,
                                                   const unsigned* guard_a, const unsigned* guard_b, AdamFinalize fin) {
#pragma clang fp contract(off)
    const bool tripped = (guard_a && __hip_atomic_load(guard_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ||
                         (guard_b && __hip_atomic_load(guard_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
    if (fin.rowloss && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = threadIdx.x; i < fin.B; i += 256) s += fin.rowloss[i];
        __shared__ float red[4];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float* sc = fin.scalars;
            const float ce = (red[0] + red[1]) + (red[2] + red[3]);
            fin.stats[2 * fin.Z + 3] = ce;
            const float recon = ce / (fin.stats[2 * fin.Z + 2] * (float)fin.T);
            sc[1] = recon;
            sc[0] = recon + sc[3] + sc[4] + sc[6] + sc[8];
            sc[15] = 0.0f;
            if (tripped) {
                for (int i = 0; i < 9; ++i) sc[i] = __builtin_nanf("");
                sc[15] = 1.0f;
            }
        }
    }
    if (tripped) return;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(c)                                   \
        mm.c = b1 * mm.c + omb1 * gg.c;            \
        vv.c = b2 * vv.c + omb2 * (gg.c * gg.c);   \
        pp.c = pp.c - (lr * mm.c) / (sqrtf(vv.c) + eps);
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // scalar tail
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (lr * mi) / (sqrtf(vi) + eps);
    }
}

__global__ void zero2d_kernel(float* x, int rows, int cols, int ld) {
    const long n = (long)rows * cols;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[(i / cols) * ld + (i % cols)] = 0.f;
}

// ---- device-side gates: cross-stream ordering WITHOUT a blocked hardware queue --------------------------------
// A stream that waits on another stream's event holds a barrier packet at the head of its hardware queue, and on
// MI355X (ROCm 7.2) every such blocked queue adds ~1 us to EACH dependent dispatch of the chain that is running
// (tools/step_trace.py: 4.72 -> 5.68 us per LSTM step launch).  A gate is instead a one-wave kernel that polls a
// device word: the waiting queue is then "running", not "blocked".
//   signal word  : a monotonically increasing u32 owned by the signalling stream (gate_set adds to it)
//   ticket       : target = (*steps) * stride + offset, where `steps` is the WAITING stream's own count of steps it
//                  has finished (advanced by its last gate of a step).  The kernel arguments are therefore the same
//                  in every step and can live in a captured hipGraph while the host runs many steps ahead.
// The spin is bounded (`max_polls`, ~1 us each): on expiry the gate raises err[0] and returns -- it can never hang
// the device.  Two streams that share one hardware queue would deadlock a gate until that expiry, so the engine
// probes each (waiter, signaller) pair with a short-fused gate before it relies on them (engine.py).
__global__ __launch_bounds__(64) void gate_wait_kernel(const unsigned* flag, unsigned* steps, unsigned stride,
                                                       unsigned offset, int advance, unsigned max_polls,
                                                       unsigned* err) {
    if (threadIdx.x != 0) return;
    const unsigned n = steps ? __hip_atomic_load(steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const unsigned target = n * stride + offset;
    bool ok = false;
    for (unsigned i = 0; i < max_polls; ++i) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) { ok = true; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    if (!ok && err) atomicAdd(err, 1u);
    if (advance && steps) __hip_atomic_store(steps, n + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__global__ __launch_bounds__(64) void gate_set_kernel(unsigned* flag, unsigned value, int add) {
    if (threadIdx.x != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (add) __hip_atomic_fetch_add(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void scale_kernel(float* x, long n, float s) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= s;
}

}  // namespace

extern "C" int arcvae_colsum_accum(const float* X, int rows, int cols, int ld, float* out, float scale,
                                   hipStream_t stream) {
    if (!X || !out || rows <= 0 || cols <= 0 || ld < cols) return ARCVAE_ERR_ARG;
    int nchunks = min(ceil_div(rows, 64), 128);
    const int rpb = ceil_div(rows, nchunks);
    nchunks = ceil_div(rows, rpb);
    dim3 grid(ceil_div(cols, 64), nchunks);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), arcvae_side_lds_pad(1024), stream, X, rows, cols, ld, out, scale, rpb);
    return arcvae_launch_status();
}

extern "C" int arcvae_transpose_batched(const float* const* src, float* const* dst, const int* rows,
                                        const int* cols, int n, hipStream_t stream) {
    if (n <= 0 || n > 16) return ARCVAE_ERR_ARG;
    TrJobs j;
    int maxr = 0, maxc = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0) return ARCVAE_ERR_ARG;
        j.src[i] = src[i]; j.dst[i] = dst[i]; j.rows[i] = rows[i]; j.cols[i] = cols[i];
        maxr = max(maxr, rows[i]); maxc = max(maxc, cols[i]);
    }
    for (int i = n; i < 16; ++i) { j.src[i] = nullptr; j.dst[i] = nullptr; j.rows[i] = 0; j.cols[i] = 0; }
    dim3 grid(ceil_div(maxc, 32), ceil_div(maxr, 32), n);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, j);
    return arcvae_launch_status();
}

extern "C" int arcvae_tile_weights(const float* const* src, float* const* dst, const int* cols, const int* mode,
                                   int n, int H, hipStream_t stream) {
    if (n <= 0 || n > 16 || H <= 0) return ARCVAE_ERR_ARG;
    TileJobs j;
    for (int i = 0; i < 16; ++i) {
        const int k = i < n ? i : 0;
        if (!src[k] || !dst[k] || cols[k] <= 0 || (cols[k] % 64) != 0) return ARCVAE_ERR_ARG;
        j.src[i] = src[k]; j.dst[i] = dst[k]; j.cols[i] = cols[k]; j.mode[i] = mode[k];
    }
    hipLaunchKernelGGL(tile_weights_kernel, dim3(256, n), dim3(256), 0, stream, j, H);
    return arcvae_launch_status();
}

extern "C" int arcvae_segsum_rows_accum(const float* X, const int32_t* seg, int rows, int nseg, int cols,
                                        float* out, hipStream_t stream) {
    if (!X || !seg || !out || rows <= 0 || nseg <= 0 || cols <= 0) return ARCVAE_ERR_ARG;
    const size_t lds = (size_t)nseg * 64 * sizeof(float);
    if (lds > 64 * 1024) return ARCVAE_ERR_ARG;  // nseg <= 256 (vocabulary-sized segment counts)
    dim3 grid(ceil_div(cols, 64), ceil_div(rows, 256));
    hipLaunchKernelGGL(segsum_kernel, grid, dim3(256), lds + arcvae_side_lds_pad((unsigned)lds), stream, X, seg, rows, nseg, cols, out);
    return arcvae_launch_status();
}

// dEmb [V,E] += dT . Wx0[:, :E];  dWx0[:, :E] += dT^T . emb  (row stride ldw >= E);  db0 [4H] += colsum(dT).
// dT [V,4H] (4H % 16 == 0).  One launch (see table_finalize_kernel).
extern "C" int arcvae_table_finalize(const float* dT, const float* Wx0, int ldw, const float* emb, float* dEmb,
                                     float* dWx0, float* db0, int V, int E, int G, hipStream_t stream) {
    if (!dT || !Wx0 || !emb || !dEmb || !dWx0 || !db0) return ARCVAE_ERR_ARG;
    if (V <= 0 || E <= 0 || G <= 0 || (G % 16) != 0 || ldw < E) return ARCVAE_ERR_ARG;
    if ((G % TF_KS) != 0) return ARCVAE_ERR_ARG;
    const int etiles = ceil_div(E, 128);
    const int nA = ceil_div(V, TF_ROWS) * etiles * (G / TF_KS), nB = (G / 16) * ceil_div(E, 64), nC = ceil_div(G, 256);
    const size_t lds = sizeof(float) * (size_t)max(TF_KS * 128 + TF_ROWS * TF_KS + TF_ROWS * 128, 128 * 16 + 128 * 64);
    hipLaunchKernelGGL(table_finalize_kernel, dim3(nA + nB + nC), dim3(256), lds, stream, dT, Wx0, ldw, emb, dEmb, dWx0,
                       db0, V, E, G, nA, nB, etiles);
    return arcvae_launch_status();
}

// ---- the byte-moving launches in front of the persistent forward sweep as ONE (round 2: token transpose, gradient
// memset and the sweep's re-arm were three launch-bound kernels of ~3-5 us each between the optimizer step and the
// first tick; the table0 product stays a skinny MFMA GEMM -- a per-thread dot-product form of it, tried here first,
// took ~70 us on uncoalesced weight rows) -----------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void enc_prologue_kernel(const int32_t* __restrict__ x_bt, int32_t* x_tb, int B, int T,
                                                           float4* zero, long nz4, float4* zero2, long nz4b, unsigned* sync, int nsync, int keep,
                                                           const float* __restrict__ cond, const float* __restrict__ Wc,
                                                           const float* __restrict__ bc, float* comb, float* stats,
                                                           int nstats, int H, int C, float* onehot, int V, int Vp) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x, gs = (long)gridDim.x * 256;
    if (onehot) {   // one-hot token rows [T*B, Vp] (row r = t*B + b) for the token-table gradient GEMMs of the BPTT chunks
        const long n = (long)B * T * Vp;
        for (long i = gt; i < n; i += gs) {
            const int r = (int)(i / Vp), v = (int)(i - (long)r * Vp);
            const int t = r / B, bb = r - t * B;
            const int tk = min(max(x_bt[(long)bb * T + t], 0), V - 1);
            onehot[i] = (v == tk) ? 1.0f : 0.0f;
        }
    }
    if (comb) {   // comb[b, H + u] = bc[u] + sum_c cond[b,c] * Wc[u,c]   (models/encoder.py:109-112); stats = 0
        for (long i = gt; i < (long)B * H; i += gs) {
            const int b = (int)(i / H), u = (int)(i - (long)b * H);
            float a = 0.f;
            for (int c = 0; c < C; ++c) a += cond[b * C + c] * Wc[u * C + c];
            comb[(long)b * 2 * H + H + u] = a + bc[u];
        }
        for (long i = gt; i < nstats; i += gs) stats[i] = 0.f;
    }
    for (long i = gt; i < (long)B * T; i += gs) {
        const int t = (int)(i / B), bb = (int)(i % B);
        x_tb[i] = x_bt[(long)bb * T + t];
    }
    for (long i = gt; i < nsync; i += gs)
        if (i != keep) sync[i] = 0u;          // (the sweeps' sticky error word stays)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = gt; i < nz4; i += gs) zero[i] = z;
    for (long i = gt; i < nz4b; i += gs) zero2[i] = z;
}
}  // namespace

// x_bt [B,T] -> x_tb [T,B];  zero_f32 (optional, 16-byte aligned, n_zero % 4 == 0) = 0;  sync_ws[0 .. n_sync) = 0
// (optional).  Replaces arcvae_transpose_tokens + the gradient memset + the sweep's re-arm in front of
// arcvae_enc_lstm_forward_persistent(flags & 1).
// cond .. comb (optional, all or none): also the condition half of the heads' input, comb[:, H:2H] = condition_fc(cond)
// (models/encoder.py:109-112), and stats[0 .. n_stats) = 0 -- with the sweep storing h_{T-1} into comb[:, :H] the heads
// need no build_comb launch (arcvae_enc_heads_forward(comb_ready = 1)).  onehot_ws (optional, [T*B, roundup(V,4)]): the
// one-hot token rows of arcvae_enc_lstm_wgrad (then called with parts bit 5).
extern "C" int arcvae_enc_prologue(const int32_t* x_bt, int32_t* x_tb, float* zero_f32, long n_zero, float* zero2_f32,
                                   long n_zero2, unsigned* sync_ws, int n_sync, int sync_keep, const float* cond, const float* Wc, const float* bc, float* comb,
                                   float* stats, int n_stats, float* onehot_ws, int V, int B, int T, int H, int C,
                                   hipStream_t stream) {
    if (!x_bt || !x_tb || B <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    if (onehot_ws && V <= 0) return ARCVAE_ERR_ARG;
    if (comb && (!cond || !Wc || !bc || !stats || n_stats <= 0 || H <= 0 || C < 0)) return ARCVAE_ERR_ARG;
    if (zero_f32 && ((n_zero % 4) != 0 || (reinterpret_cast<uintptr_t>(zero_f32) & 15) != 0)) return ARCVAE_ERR_ARG;
    if (zero2_f32 && ((n_zero2 % 4) != 0 || (reinterpret_cast<uintptr_t>(zero2_f32) & 15) != 0)) return ARCVAE_ERR_ARG;
    if (n_sync < 0 || (n_sync > 0 && !sync_ws)) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(enc_prologue_kernel, dim3(256), dim3(256), 0, stream, x_bt, x_tb, B, T,
                       reinterpret_cast<float4*>(zero_f32), zero_f32 ? n_zero / 4 : 0L, reinterpret_cast<float4*>(zero2_f32),
                       zero2_f32 ? n_zero2 / 4 : 0L, sync_ws, n_sync, sync_keep, cond, Wc, bc, comb,
                       stats, n_stats, H, C, onehot_ws, V, (V + 3) & ~3);
    return arcvae_launch_status();
}

extern "C" int arcvae_transpose_tokens(const int32_t* src, int32_t* dst, int B, int T, hipStream_t stream) {
    if (!src || !dst || B <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(transpose_tokens_kernel, dim3(ceil_div(B * T, 256)), dim3(256), 0, stream, src, dst, B, T);
    return arcvae_launch_status();
}

// Reference: trainer.py:320,324 (optimizer.update) with MLX optim.Adam defaults (Q7).
// guard_a / guard_b: optional device words; the update is skipped when either is non-zero (see adam_kernel).
extern "C" int arcvae_adam_update(float* params, const float* grads, float* m, float* v, long n,
                                  double lr, double beta1, double beta2, double eps, const unsigned* guard_a,
                                  const unsigned* guard_b, hipStream_t stream) {
    if (!params || !grads || !m || !v || n <= 0) return ARCVAE_ERR_ARG;
    const uintptr_t al = reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
                         reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v);
    const long n4 = (al & 15) ? 0 : n / 4;
    // (1 - beta) is formed in double and rounded once to f32, as a Python float times an f32 array is.
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const long work = n4 > 0 ? n4 : n;
    const int blocks = (int)min((long)2048, (work + 255) / 256);
    AdamFinalize fin;
    fin.rowloss = nullptr; fin.stats = nullptr; fin.scalars = nullptr; fin.B = fin.Z = fin.T = 0;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, m, v, n4, n, (float)lr,
                       (float)beta1, (float)beta2, omb1, omb2, (float)eps, guard_a, guard_b, fin);
    return arcvae_launch_status();
}

// arcvae_recon_finalize + arcvae_adam_update in ONE launch (the tail of the single-process step: trainer.py:320-366 reads the
// loss after the update was applied): block 0 first sums the CE rows into stats[2Z+3] and writes the recon / total scalars (NaN +
// scalars[15] = 1 when a guard word is set), every block then applies the un-bias-corrected Adam update (skipped on a guard).
extern "C" int arcvae_adam_update_finalize(float* params, const float* grads, float* m, float* v, long n, double lr,
                                           double beta1, double beta2, double eps, const unsigned* guard_a,
                                           const unsigned* guard_b, const float* rowloss, int B, float* stats, float* scalars,
                                           int Z, int T, hipStream_t stream) {
    if (!params || !grads || !m || !v || n <= 0 || !rowloss || !stats || !scalars || B <= 0 || Z <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    const uintptr_t al = reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
                         reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v);
    const long n4 = (al & 15) ? 0 : n / 4;
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const long work = n4 > 0 ? n4 : n;
    const int blocks = (int)min((long)2048, (work + 255) / 256);
    AdamFinalize fin;
    fin.rowloss = rowloss; fin.stats = stats; fin.scalars = scalars; fin.B = B; fin.Z = Z; fin.T = T;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, m, v, n4, n, (float)lr,
                       (float)beta1, (float)beta2, omb1, omb2, (float)eps, guard_a, guard_b, fin);
    return arcvae_launch_status();
}

// x[r*ld + c] = 0 for r < rows, c < cols.  A kernel rather than hipMemsetAsync: memset NODES of a captured
// single-stream segment replayed garbage on ROCm 7.2 (tools/debug_replay.py), kernels replay faithfully.
extern "C" int arcvae_zero(float* x, int rows, int cols, int ld, hipStream_t stream) {
    if (!x || rows <= 0 || cols <= 0 || ld < cols) return ARCVAE_ERR_ARG;
    const long n = (long)rows * cols;
    const int blocks = (int)min((long)1024, (n + 255) / 256);
    hipLaunchKernelGGL(zero2d_kernel, dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld);
    return arcvae_launch_status();
}

// Device-side wait on `stream` until (int)(*flag - ((*steps) * stride + offset)) >= 0 (steps may be null: target =
// offset); advance != 0: ++*steps afterwards.  max_polls bounds the spin (~1 us per poll); on expiry *err += 1.
extern "C" int arcvae_gate_wait(const unsigned* flag, unsigned* steps, unsigned stride, unsigned offset, int advance,
                                unsigned max_polls, unsigned* err, hipStream_t stream) {
    if (!flag || max_polls == 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, stream, flag, steps, stride, offset, advance,
                       max_polls, err);
    return arcvae_launch_status();
}
// *flag = value (add == 0) or *flag += value, after everything earlier on `stream` (agent-scope release).
extern "C" int arcvae_gate_set(unsigned* flag, unsigned value, int add, hipStream_t stream) {
    if (!flag) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(gate_set_kernel, dim3(1), dim3(64), 0, stream, flag, value, add);
    return arcvae_launch_status();
}

// Diagnostic / test entry point (tests/test_occupancy_gpu.py): a stand-in for a communication kernel -- `blocks` workgroups
// of `threads` threads that hold `lds_bytes` of LDS each and do nothing but wait `spin_us` microseconds (wall clock), on
// `stream`.  Used to show that the persistent sweeps (which need a resident block on every CU) survive a neighbour that
// occupies CU resources: RCCL's channel kernels on another stream behave like this while they wait for a peer.
namespace {
__global__ void occupy_kernel(unsigned long long ticks, int lds_floats) {
    extern __shared__ float occ_lds[];
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) occ_lds[i] = 0.f;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace
extern "C" int arcvae_debug_occupy(int blocks, int threads, int lds_bytes, int spin_us, hipStream_t stream) {
    if (blocks <= 0 || blocks > 4096 || threads <= 0 || threads > 1024 || (threads % 64) != 0 || lds_bytes < 0 ||
        lds_bytes > 150 * 1024 || spin_us < 0 || spin_us > 100000)
        return ARCVAE_ERR_ARG;
    (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, stream,
                       (unsigned long long)spin_us * 100ull, lds_bytes / 4);
    return arcvae_launch_status();
}

extern "C" int arcvae_scale_inplace(float* x, long n, float s, hipStream_t stream) {
    if (!x || n <= 0) return ARCVAE_ERR_ARG;
    const int blocks = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, stream, x, n, s);
    return arcvae_launch_status();
}

extern "C" int arcvae_abi_version(int* arch_gfx950) {
    if (arch_gfx950) {
        int dev = 0;
        hipDeviceProp_t prop;
        *arch_gfx950 = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            const char* a = prop.gcnArchName;
            *arch_gfx950 = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
        }
    }
    return 1000;
}

// ---- several small device-to-device copies in ONE launch -------------------------------------------------------------
// A step's inputs (tokens, conditions, eps, teacher-forcing coins) arrive as four buffers of 0.1-32 KB; as four copy
// launches they are ~4 us each of launch-bound time in front of every step (profiles/r02_bench_kernel_stats.csv:
// __amd_rocclr_copyBuffer, 5.8 calls per step).  dst_i[0..nbytes_i) = src_i[0..nbytes_i), i < n <= 8.
namespace {
struct CopyJobs {
    const unsigned char* src[8];
    unsigned char* dst[8];
    long nbytes[8];
    int n;
};
__global__ __launch_bounds__(256) void copy_buffers_kernel(CopyJobs j) {
    const int i = blockIdx.y;
    if (i >= j.n) return;
    const unsigned char* s = j.src[i];
    unsigned char* d = j.dst[i];
    const long nb = j.nbytes[i];
    const long gt = (long)blockIdx.x * 256 + threadIdx.x, gs = (long)gridDim.x * 256;
    long done = 0;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const long n16 = nb >> 4;
        for (long k = gt; k < n16; k += gs) reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(s)[k];
        done = n16 << 4;
    }
    for (long k = done + gt; k < nb; k += gs) d[k] = s[k];
}
}  // namespace

extern "C" int arcvae_copy_buffers(const void* const* src, void* const* dst, const long* nbytes, int n, hipStream_t stream) {
    if (!src || !dst || !nbytes || n <= 0 || n > 8) return ARCVAE_ERR_ARG;
    CopyJobs j;
    j.n = n;
    long mx = 0;
    for (int i = 0; i < 8; ++i) {
        const int k = i < n ? i : 0;
        if (!src[k] || !dst[k] || nbytes[k] < 0) return ARCVAE_ERR_ARG;
        j.src[i] = static_cast<const unsigned char*>(src[k]); j.dst[i] = static_cast<unsigned char*>(dst[k]);
        j.nbytes[i] = nbytes[k];
        if (i < n && nbytes[k] > mx) mx = nbytes[k];
    }
    const int bx = (int)std::min<long>(64, std::max<long>(1, (mx / 16 + 255) / 256));
    hipLaunchKernelGGL(copy_buffers_kernel, dim3(bx, n), dim3(256), 0, stream, j);
    return arcvae_launch_status();
}
