// Encoder stacked-LSTM sweep (reference models/encoder.py:98-101 over MLX nn.LSTM, semantics M1):
// forward and BPTT as a WAVEFRONT of small step launches.
//
// Why launches and not one persistent kernel: each step ends in an all-to-all seam (every
// gate column needs the whole h_{t-1} row block).  On MI355X an in-launch all-gather of that
// size costs about what a kernel boundary costs (MI355X_MICROARCH.md price list: allgather
// 2.4-3.0 us vs boundary 1.5-1.9 us), so the design cuts at the seam and instead shortens the
// dependent chain: launch s computes layer l at time t = s - l for every layer at once
// (T + L - 1 dependent launches instead of T * L), and layers >= 1 fuse their input projection
// h^{l-1}_t . Wx_l^T into the step, so no [B*T,4H] pre-activation tensor is ever materialised
// for them.  Layer 0's input projection is a [V,4H] table lookup (emb . Wx0^T + bias computed
// once per step for the V distinct tokens).
//
// Tiles: forward -- 16 batch rows x 4 hidden units (16 gate columns i,f,g,o interleaved so the
// cell update for a unit is local to the block), K split over 4 waves; backward -- 16 rows x 16
// hidden units, K = 4H (per source) split over 16 waves.  Both on v_mfma_f32_16x16x4_f32 with
// every operand load issued before the first MFMA (skinny.h): a step kernel has nothing to hide
// memory latency behind, so the L2/Infinity-Cache round trip must be paid once, not per K-chunk.
//
// Layout: time-major activations [L][T][B][*] so that each step's operands are contiguous slabs
// and "h shifted by one step" (needed by dWh) is a pointer offset.
#include "ops.h"
#include "skinny.h"
#include "xcd.h"

namespace {

// XCD-aware block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one,
// MI355X_MICROARCH.md "Workgroup dispatch"), and each XCD has its own L2.  A step block writes 16-byte (forward) or
// 64-byte (BPTT) pieces of the activation rows, so with the identity mapping the 8 (2) blocks that together fill one
// 128-B line sit on 8 (2) DIFFERENT XCDs: every L2 then writes back partial lines at the kernel boundary, which is
// on the dependent chain.  With the mapping below neighbouring tiles share an XCD and its L2 merges them into whole
// lines.  gx = gridDim.x must be a multiple of 8 (else identity).
__device__ __forceinline__ int xcd_group_block(int x, int gx, int on) {
    return (on && (gx & 7) == 0) ? (x & 7) * (gx >> 3) + (x >> 3) : x;
}

struct FwdJob {
    const float* xin;    // tiled [H/16][B][16]  h^{l-1}_t, or null for layer 0
    const float* Wx;     // tiled+permuted [H/16][4H][16]  (arcvae_tile_weights mode 0)
    const float* hprev;  // tiled [H/16][B][16]  h^l_{t-1}, or null at t == 0 (MLX: hidden=None skips the term)
    const float* Wh;     // tiled+permuted [H/16][4H][16]
    const float* pre;    // layer 0: table0 [V,4H] (bias folded in); else bias [4H]
    const int32_t* tok;  // layer 0: tokens of this step [B]; else null
    const float* cprev;  // [B,H] or null at t == 0 (MLX: cell=None -> c = i*g)
    float* h;            // [B,H]
    float* ht;           // tiled copy [H/16][B][16] (operand of the next launches)
    float* c;            // [B,H]
    float* gates;        // [B,4H] post-activation i,f,g,o (saved for BPTT)
    void* oct;           // throughput mode (optional): bf16 octet-major copy [B/8][H][8] of h for the weight-gradient kernel
};
struct FwdArgs {
    FwdJob job[ARCVAE_MAX_LAYERS];
    int B, H, V;
    int prio, remap;
    int dbg;                    // timing experiments only (ARCVAE_TILE_DEBUG, bf16 tile kernels): 1 no contraction, 2 no epilogue
    int lite;                   // three-piece tile kernel, forward-only callers: bit 0 = the gates and c are not stored, bit 1 = nor the planes
    unsigned long long* trace;  // diagnostic: {start, end} of block (0,0,0) in 100 MHz ticks, or null
};

// CH = H / 64: each of the 4 waves owns H/4 = 16*CH floats of K per source.
template <int CH>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(FwdArgs a) {
    __shared__ float red[4 * 256];
    __shared__ float act[256];
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    const FwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H;
    const int bx = xcd_group_block(blockIdx.x, gridDim.x, a.remap & 1);
    const int r0 = blockIdx.y * 16, u0 = bx * 4;
    const int arow = min(r0 + (lane & 15), B - 1);
    const int jc = lane & 15;
    const int wrow = bx * 16 + jc;  // permuted weight row of tile column jc (gate = jc>>2, unit = u0+(jc&3))
    const bool s1 = j.xin != nullptr, s2 = j.hprev != nullptr;
    // Epilogue operands are requested FIRST (token -> table row is a dependent pair of loads, c_{t-1}
    // another cold line): their round trip then overlaps the operand loads instead of following the MFMAs.
    const int erow = tid >> 4, ecol = tid & 15;
    const int eb = min(r0 + erow, B - 1);
    const int gcol = (ecol >> 2) * H + u0 + (ecol & 3);
    const float* pre = j.pre;
    if (j.tok) {
        int tk = j.tok[eb];
        tk = min(max(tk, 0), a.V - 1);
        pre += (long)tk * 4 * H;
    }
    const float pre_v = pre[gcol];
    const int crow = tid >> 2, cu = tid & 3;  // cell-update role of threads 0..63
    const long hb = (long)min(r0 + crow, B - 1) * H + u0 + cu;
    float cprev_v = 0.f;
    if (tid < 64 && j.cprev) cprev_v = j.cprev[hb];
    SkinnyFrag<CH> f1, f2;
    if (s1) skinny_load_tiled<CH>(f1, j.xin, arow, B, j.Wx, wrow, 4 * H, wave, lane);
    if (s2) skinny_load_tiled<CH>(f2, j.hprev, arow, B, j.Wh, wrow, 4 * H, wave, lane);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (s1) skinny_mfma<CH>(f1, acc0, acc1);
    if (s2) skinny_mfma<CH>(f2, acc0, acc1);
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    __syncthreads();
    {   // one gate activation per thread: (row, col) -> gate = col>>2, unit = u0 + (col&3)
        const float v = skinny_reduced_n<4>(red, erow, ecol) + pre_v;
        const float av = ((ecol >> 2) == 2) ? tanhf(v) : sigmoidf_acc(v);
        act[tid] = av;
        if (r0 + erow < B) j.gates[(long)(r0 + erow) * 4 * H + gcol] = av;
    }
    __syncthreads();
    if (tid < 64 && r0 + crow < B) {
        const float i = act[crow * 16 + cu], f = act[crow * 16 + 4 + cu], g = act[crow * 16 + 8 + cu],
                    o = act[crow * 16 + 12 + cu];
        const float c = j.cprev ? f * cprev_v + i * g : i * g;
        const float hv = o * tanhf(c);
        const int unit = u0 + cu;
        j.h[hb] = hv;
        j.ht[((long)(unit >> 4) * B + (r0 + crow)) * 16 + (unit & 15)] = hv;
        j.c[hb] = c;
    }
    if (tr) a.trace[1] = wall_clock64();
}

// One BPTT launch carries up to 2L-1 single-source jobs (16 rows x 16 hidden units x K = 4H each):
//   kind 0 "cell":  dh = src . WT (+ ext) -> gate gradients dG^l_t      src = dG^l_{t+1}, WT = Wh_l^T,
//                                                                        ext = dX_l[t] (or d/d(hT) for the top layer)
//   kind 1 "xproj": dX_l[t] = src . WT                                   src = dG^{l+1}_t, WT = Wx_{l+1}^T
// Splitting the lower layers' two contractions into two jobs one launch apart (skew 2 per layer instead of 1)
// makes every block move the same 128 KB and spreads a launch over more CUs: the step is bound by how fast
// a CU can pull its operands (profiles/r01: 10.3 us per launch with the fused two-source job), not by MFMA.
struct BwdJob {
    const float* src;     // tiled [4H/16][B][16]  or null (cell at t == T-1: no recurrent term)
    const float* WT;      // tiled [4H/16][H][16]  (arcvae_tile_weights mode 1: W^T)
    const float* ext;     // [B,ext_ld] added to dh (cell) or null
    const float* gates;   // [B,4H] i,f,g,o at (l,t)            (cell)
    const float* c;       // [B,H] c_t                          (cell)
    const float* cprev;   // [B,H] c_{t-1} or null (t == 0)     (cell)
    const float* dcin;    // [B,H] dc_{t+1} * f_{t+1} or null   (cell)
    float* dcout;         // [B,H] dc_t * f_t                   (cell)
    float* out;           // cell: dG [B,4H];  xproj: dX [B,H]
    float* outt;          // cell: tiled copy of dG [4H/16][B][16] (operand of the next launches)
    void* oct;            // throughput mode (optional, cell): bf16 octet-major copy [B/8][4H][8] of dG (weight-gradient kernel)
    int ext_ld;
    int kind;
};
#define ARCVAE_MAX_BWD_JOBS 16
struct BwdArgs {
    BwdJob job[ARCVAE_MAX_BWD_JOBS];
    int B, H;
    int prio, remap;
    int dbg;                    // timing experiments only (ARCVAE_TILE_DEBUG, bf16 tile kernels): 1 no contraction, 2 no epilogue
    unsigned long long* trace;
    unsigned* signal;  // or null: += 1 (agent scope) when this launch STARTS, i.e. when everything before it on the
                       // stream has completed -- the "sweep chunk done" signal of engine.Gates without a launch of its own
};

// 16 waves: wave w owns 4H/16 = 16*CH floats of the contraction index.
template <int CH>
__global__ __launch_bounds__(1024) void lstm_bwd_step_kernel(BwdArgs a) {
    __shared__ float red[16 * 256];
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const BwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r0 = blockIdx.y * 16, u0 = xcd_group_block(blockIdx.x, gridDim.x, a.remap & 1) * 16;
    const int arow = min(r0 + (lane & 15), B - 1);
    const int wrow = u0 + (lane & 15);
    const bool cell = j.kind == 0;
    // epilogue operands first: their cold round trip overlaps the operand loads (see the forward kernel)
    const int erow = tid >> 4, ecol = tid & 15;
    const int eb = min(r0 + erow, B - 1);
    const int unit = u0 + ecol;
    const long hb = (long)eb * H + unit;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c_v = 0.f, cprev_v = 0.f, dcin_v = 0.f, ext_v = 0.f;
    if (tid < 256 && cell) {
        const float* gp = j.gates + (long)eb * G + unit;
        gi = gp[0]; gf = gp[H]; gg = gp[2 * H]; go = gp[3 * H];
        c_v = j.c[hb];
        if (j.cprev) cprev_v = j.cprev[hb];
        if (j.dcin) dcin_v = j.dcin[hb];
        if (j.ext) ext_v = j.ext[(long)eb * j.ext_ld + unit];
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (j.src) {
        SkinnyFrag<CH> f;
        skinny_load_tiled<CH>(f, j.src, arow, B, j.WT, wrow, H, wave, lane);
        skinny_mfma<CH>(f, acc0, acc1);
    }
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    __syncthreads();
    if (tid < 256 && r0 + erow < B) {
        const float dh = skinny_reduced_n<16>(red, erow, ecol) + ext_v;
        if (!cell) {
            j.out[hb] = dh;
        } else {
            const float i = gi, f = gf, g = gg, o = go;
            const float tc = tanhf(c_v);
            const float d_o = dh * tc * o * (1.f - o);
            const float dc = dh * o * (1.f - tc * tc) + dcin_v;
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * cprev_v * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.out + (long)eb * G + unit;
            dp[0] = d_i;
            dp[H] = d_f;
            dp[2 * H] = d_g;
            dp[3 * H] = d_o;
            // tiled copy: contraction index jj = gate*H + unit -> [(jj/16)][b][jj%16]; H % 16 == 0 so jj%16 = unit%16
            float* tp = j.outt + ((long)(unit >> 4) * B + eb) * 16 + (unit & 15);
            const long gs = (long)(H >> 4) * B * 16;  // one gate block of H contraction indices
            tp[0] = d_i;
            tp[gs] = d_f;
            tp[2 * gs] = d_g;
            tp[3 * gs] = d_o;
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// One-hot rows of the layer-0 tokens: oh[r, v] = (tok[r] == v), r = t*B + b, row stride Vp (V rounded up to 4).
// The token segment-sum dTable0[v] = sum_{r: tok=v} dG_0[r] is then OneHot^T . dG_0 on the matrix cores (split-K
// GEMM, K = rows of the chunk) instead of an LDS-atomic scatter with a global-atomic flush per block: 26 -> ~10 us
// per BPTT chunk and, above all, 38 -> ~10 us in the exposed tail of the step.
__global__ __launch_bounds__(256) void onehot_kernel(const int32_t* __restrict__ tok, int rows, int V, int Vp,
                                                     float* __restrict__ oh) {
    const long n = (long)rows * Vp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / Vp), v = (int)(i - (long)r * Vp);
        const int tk = min(max(tok[r], 0), V - 1);
        oh[i] = (v == tk) ? 1.0f : 0.0f;
    }
}


// ---- large-batch step kernels ("tiled") --------------------------------------------------------------------------
// The 16x16-tile kernels above are built for latency (B = 64: one dependent launch is ~5 us and mostly round trips).
// Their arithmetic intensity is 4 FLOP per byte pulled from L2, so from a few hundred rows per GPU on they are bound
// by L2 -> CU delivery (BASELINE.json configs[2], H512 L4 bs 512: 200 / 156 us per forward / BPTT launch = 24 / 31 %
// of the f32 MFMA peak).  The tiled kernels give every wave a (16*MT rows) x 32 columns register tile over the WHOLE
// contraction: per 16-wide k-chunk a lane loads MT + 2 float4 fragments straight from the k-chunk-major operand
// copies (one wave-instruction = 1 KB contiguous) and issues 8*MT MFMAs (v_mfma_f32_16x16x4_f32), i.e. 10.7 (MT=4),
// 8 (MT=2), 5.3 (MT=1) FLOP per byte, with the next chunk's loads in flight behind the current chunk's MFMAs.  No LDS
// in the main loop; the 4 waves of a block sit side by side in the column direction and share the A rows through L1.
template <int MT, int NT>
struct TileFrag {
    float4 a[MT];
    float4 w[NT];
};

template <int MT, int NT>
__device__ __forceinline__ void tile_load(TileFrag<MT, NT>& f, const float* __restrict__ At, const int* arow, int RA,
                                          const float* __restrict__ Wt, const int* wrow, int RW, int kc, int q4) {
#pragma unroll
    for (int m = 0; m < MT; ++m) f.a[m] = *reinterpret_cast<const float4*>(At + ((long)kc * RA + arow[m]) * 16 + q4);
#pragma unroll
    for (int n = 0; n < NT; ++n) f.w[n] = *reinterpret_cast<const float4*>(Wt + ((long)kc * RW + wrow[n]) * 16 + q4);
}

template <int MT, int NT>
__device__ __forceinline__ void tile_mfma(const TileFrag<MT, NT>& f, f32x4 (&acc)[MT][NT]) {
    // component-major: consecutive MFMAs go to DIFFERENT accumulators, so none waits for its predecessor's result
#define TILE_MFMA_C(c)                                                                                     \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n)          \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[m].c, f.w[n].c, acc[m][n], 0, 0, 0);
    TILE_MFMA_C(x)
    TILE_MFMA_C(y)
    TILE_MFMA_C(z)
    TILE_MFMA_C(w)
#undef TILE_MFMA_C
}

// acc += A[rows, K] . W[cols, K]^T over nch 16-wide chunks (nch % 4 == 0: H is a multiple of 64), operands
// k-chunk-major.  NS-stage register ring: the loads of chunk kc+NS-1 are issued before the MFMAs of chunk kc, because
// one chunk's MFMAs (4*MT*NT x 32 cycles) are shorter than an L2 round trip -- with a single chunk in flight the BPTT
// tile ran 2.8x off its MFMA time (133 us per launch at configs[2] against 48).  NS = 4 (2 for the 4x4 tile: its
// chunk is 2048 MFMA cycles long and its fragments are 32 registers per stage).
template <int MT, int NT, int NS>
__device__ __forceinline__ void tile_contract(f32x4 (&acc)[MT][NT], const float* __restrict__ At, const int* arow,
                                              int RA, const float* __restrict__ Wt, const int* wrow, int RW,
                                              int nch, int q4) {
    TileFrag<MT, NT> f[NS];
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) tile_load<MT, NT>(f[s], At, arow, RA, Wt, wrow, RW, s, q4);
    for (int kc0 = 0; kc0 < nch; kc0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kn = kc0 + s + NS - 1;
            if (kn < nch) tile_load<MT, NT>(f[(s + NS - 1) % NS], At, arow, RA, Wt, wrow, RW, kn, q4);
            tile_mfma<MT, NT>(f[s], acc);
        }
    }
}

// ---- throughput mode (bf16 operands, f32 accumulate): the same tiles on v_mfma_f32_16x16x32_bf16 ---------------------
// The k-chunk-major operand copies hold bf16 in 32-wide chunks ([kc][row][32]: a lane's 16-byte load is its 8 k of one
// MFMA, one wave-instruction = 1 KB contiguous as before); everything a cell keeps (gates, c, h, the accumulators) stays
// f32.  One chunk's MFMAs are 16 x shorter than in f32 (MT*NT x 16 cycles per K = 32), so the ring is NSB stages deep.
typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_l __attribute__((ext_vector_type(4)));
typedef float f32x2_l __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_l __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __bf16 to_bf16(float x) {     // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_convertvector(f32x2_l{x, 0.f}, bf16x2_l)[0];
}
template <int MT, int NT>
struct TileFragB {
    u32x4_l a[MT];
    u32x4_l w[NT];
};
template <int MT, int NT>
__device__ __forceinline__ void tile_load_b(TileFragB<MT, NT>& f, const __bf16* __restrict__ At, const int* arow, int RA,
                                            const __bf16* __restrict__ Wt, const int* wrow, int RW, int kc, int q8) {
#pragma unroll
    for (int m = 0; m < MT; ++m) f.a[m] = *reinterpret_cast<const u32x4_l*>(At + ((long)kc * RA + arow[m]) * 32 + q8);
#pragma unroll
    for (int n = 0; n < NT; ++n) f.w[n] = *reinterpret_cast<const u32x4_l*>(Wt + ((long)kc * RW + wrow[n]) * 32 + q8);
}
template <int MT, int NT, int NS>
__device__ __forceinline__ void tile_contract_b(f32x4 (&acc)[MT][NT], const __bf16* __restrict__ At, const int* arow,
                                                int RA, const __bf16* __restrict__ Wt, const int* wrow, int RW,
                                                int nch, int q8) {
    TileFragB<MT, NT> f[NS];
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nch) tile_load_b<MT, NT>(f[s], At, arow, RA, Wt, wrow, RW, s, q8);
    for (int kc0 = 0; kc0 < nch; kc0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kn = kc0 + s + NS - 1;
            if (kn < nch) tile_load_b<MT, NT>(f[(s + NS - 1) % NS], At, arow, RA, Wt, wrow, RW, kn, q8);
            if (kc0 + s < nch) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_l, f[s].a[m]),
                                                                            __builtin_bit_cast(bf16x8_l, f[s].w[n]),
                                                                            acc[m][n], 0, 0, 0);
            }
        }
    }
}

// ---- three-piece form (ARCVAE_LSTM_SPLIT3: a PARITY path): every f32 operand value as hi + mid + lo bf16 (8 + 8 + 8 bits),
// the six products of weight >= 2^-16 on v_mfma_f32_16x16x32_bf16, f32 accumulate -- fp32-class accuracy (the split GEMMs of
// gemm.hip have carried it since round 2) at 6/16 of the exact-f32 matrix time.  The operand copies are THREE bf16 planes of
// the throughput mode's 32-wide k-chunk-major layout ([plane][kc][row][32]; a plane = rows * K elements), written by the step
// epilogues (activations, gate gradients) and by arcvae_tile_weights modes 4 / 5 (weights).  6 bytes per value instead of 4:
// a 64 x 64 wave tile moves 16 B per matrix cycle and wave -- the CU's 64 B/clk from L2 with its four waves -- so the
// contraction sits where L2 delivery and the matrix pipe meet.
__device__ __forceinline__ void split3_bf16(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = to_bf16(x);
    const float r1 = x - (float)hi;
    mid = to_bf16(r1);
    lo = to_bf16(r1 - (float)mid);
}
template <int MT, int NT>
struct TileFragS {
    u32x4_l a[3][MT];
    u32x4_l w[3][NT];
};
#ifndef ARCVAE_S3_SADDR
#define ARCVAE_S3_SADDR 1
#endif
// Which ring the three-piece contractions run on: 0 the generic NS-stage ring, 1 the branch-free two-stage ring, 2 that ring with
// its stages pinned (tile_contract_s2<.., PIN>: counted vmcnt(24) waits; round 4: 70.0 -> 68.8 us per BPTT tile launch at configs[2],
// 2048 rows 16.17 -> 15.5 ms -- small, because the contraction is bound by what an XCD's L2 delivers to a CU, ~70 GB/s, not by latency)
#ifndef ARCVAE_S3_RING2_FWD
#define ARCVAE_S3_RING2_FWD 2
#endif
#ifndef ARCVAE_S3_RING2_BWD
#define ARCVAE_S3_RING2_BWD 0
#endif
#ifndef ARCVAE_S3_RING2_KS
#define ARCVAE_S3_RING2_KS 2
#endif
#ifndef ARCVAE_DBG_SKIP_A
#define ARCVAE_DBG_SKIP_A 0      // timing experiment (variant builds only): waves 1-3 of a forward tile block do not load the A operand
#endif
template <int MT, int NT, bool SKIPA = false>
__device__ __forceinline__ void tile_load_s(TileFragS<MT, NT>& f, const __bf16* __restrict__ At, long planeA, const int* arow,
                                            int RA, const __bf16* __restrict__ Wt, long planeW, const int* wrow, int RW, int kc,
                                            int q8) {
    // address = wave-uniform base (plane, chunk: scalar registers) + the lane's 32-bit offset (row, k octet): the loads take the
    // scalar-base form and the ring carries MT + NT offsets instead of 3 (MT + NT) 64-bit pointers
#if !ARCVAE_S3_SADDR
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            f.a[p][m] = *reinterpret_cast<const u32x4_l*>(At + p * planeA + ((long)kc * RA + arow[m]) * 32 + q8);
#pragma unroll
        for (int n = 0; n < NT; ++n)
            f.w[p][n] = *reinterpret_cast<const u32x4_l*>(Wt + p * planeW + ((long)kc * RW + wrow[n]) * 32 + q8);
    }
    return;
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const __bf16* ab = At + p * planeA + (long)kc * RA * 32;
        const __bf16* wb = Wt + p * planeW + (long)kc * RW * 32;
        if (!SKIPA || threadIdx.x < 64) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                f.a[p][m] = *reinterpret_cast<const u32x4_l*>(ab + (unsigned)(arow[m] * 32 + q8));
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) f.a[p][m] = u32x4_l{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
            f.w[p][n] = *reinterpret_cast<const u32x4_l*>(wb + (unsigned)(wrow[n] * 32 + q8));
    }
}
template <int MT, int NT, int NS, bool PIN = false>   // PIN: loads and products of a step stay in program order (register footprint)
__device__ __forceinline__ void tile_contract_s(f32x4 (&acc)[MT][NT], const __bf16* __restrict__ At, long planeA,
                                                const int* arow, int RA, const __bf16* __restrict__ Wt, long planeW,
                                                const int* wrow, int RW, int nch, int q8) {
    TileFragS<MT, NT> f[NS];
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nch) tile_load_s<MT, NT>(f[s], At, planeA, arow, RA, Wt, planeW, wrow, RW, s, q8);
    for (int kc0 = 0; kc0 < nch; kc0 += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kn = kc0 + s + NS - 1;
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            if (kn < nch) tile_load_s<MT, NT>(f[(s + NS - 1) % NS], At, planeA, arow, RA, Wt, planeW, wrow, RW, kn, q8);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            if (kc0 + s < nch) {
                // products in ascending weight: hi.lo, lo.hi, mid.mid, hi.mid, mid.hi, hi.hi (pieces 0 = hi, 1 = mid, 2 = lo)
#define TILE_S3(PA, PW)                                                                                              \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n)        \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_l, f[s].a[PA][m]),  \
                                                                        __builtin_bit_cast(bf16x8_l, f[s].w[PW][n]),  \
                                                                        acc[m][n], 0, 0, 0);
                TILE_S3(0, 2) TILE_S3(2, 0) TILE_S3(1, 1) TILE_S3(0, 1) TILE_S3(1, 0) TILE_S3(0, 0)
#undef TILE_S3
            }
        }
    }
}

// The same contraction as a branch-free two-stage ring (nch even, >= 2: H % 64 == 0 makes every caller's count even).  The
// generic ring's `if (kn < nch)` inside the unrolled body costs the register allocator its grip: it renames the stages across
// the branches and shuttles the accumulators between AGPRs and VGPRs every iteration (456 registers for the 4 x 4 tile).
template <int MT, int NT>
__device__ __forceinline__ void tile_mfma_s(const TileFragS<MT, NT>& f, f32x4 (&acc)[MT][NT]) {
#define TILE_S3(PA, PW)                                                                                              \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n)                    \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_l, f.a[PA][m]),                \
                                                            __builtin_bit_cast(bf16x8_l, f.w[PW][n]), acc[m][n], 0, 0, 0);
    TILE_S3(0, 2) TILE_S3(2, 0) TILE_S3(1, 1) TILE_S3(0, 1) TILE_S3(1, 0) TILE_S3(0, 0)
#undef TILE_S3
}
// PIN: a stage's loads and a stage's products stay whole and in program order (scheduling barriers between them) -- the loads of
// chunk k + 1 are then ALL issued in front of the products of chunk k, whose wait is a counted vmcnt(24).  Left to itself the
// compiler interleaves every load just in front of its use (fewer registers, no prefetch distance); the generic ring's
// `if (kn < nch)` makes its wait-count bookkeeping assume the branch not taken, i.e. drain the loads it has just issued.
template <int MT, int NT, bool PIN = false, bool SKIPA = false>
__device__ __forceinline__ void tile_contract_s2(f32x4 (&acc)[MT][NT], const __bf16* __restrict__ At, long planeA,
                                                 const int* arow, int RA, const __bf16* __restrict__ Wt, long planeW,
                                                 const int* wrow, int RW, int nch, int q8) {
    TileFragS<MT, NT> f0, f1;
#define S2_PIN() do { if constexpr (PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
    tile_load_s<MT, NT, SKIPA>(f0, At, planeA, arow, RA, Wt, planeW, wrow, RW, 0, q8);
    for (int kc = 0; kc < nch - 2; kc += 2) {
        S2_PIN();
        tile_load_s<MT, NT, SKIPA>(f1, At, planeA, arow, RA, Wt, planeW, wrow, RW, kc + 1, q8);
        S2_PIN();
        tile_mfma_s<MT, NT>(f0, acc);
        S2_PIN();
        tile_load_s<MT, NT, SKIPA>(f0, At, planeA, arow, RA, Wt, planeW, wrow, RW, kc + 2, q8);
        S2_PIN();
        tile_mfma_s<MT, NT>(f1, acc);
    }
    S2_PIN();
    tile_load_s<MT, NT, SKIPA>(f1, At, planeA, arow, RA, Wt, planeW, wrow, RW, nch - 1, q8);
    S2_PIN();
    tile_mfma_s<MT, NT>(f0, acc);
    S2_PIN();
    tile_mfma_s<MT, NT>(f1, acc);
#undef S2_PIN
}

// Blocks are dealt to the XCDs round-robin by linear id, so the plain (x, y, job) order hands every XCD a slice of EVERY job:
// 7 jobs' weight slices and 28 different row blocks go through each 4 MB L2 per launch (33 MB at configs[2], 264 MB over the
// fabric).  mode bit 1 (ARCVAE_TILE_XCD): XCD k works on a CONTIGUOUS range of the (job, row block, column block) order --
// one job's weights and rows, 12-20 MB per XCD, every weight slice shared by the XCD's row blocks under one L2.
__device__ __forceinline__ void tile_xcd_block(int mode, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (!(mode & 2)) return;
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = id & 7, slot = id >> 3, q = n >> 3, r = n & 7;
    const unsigned nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bx = (int)(nid % gx); by = (int)((nid / gx) % gy); bz = (int)(nid / (gx * gy));
}

// Forward: wave tile = 16*MT rows x 16*NT gate columns (4*NT hidden units); block = 4 waves side by side =
// 16*MT rows x 64*NT columns.  grid (H / (16*NT), ceil(B / (16*MT)), jobs).  The pre-activations go through a
// per-wave LDS tile so that one lane gets the four gates of one (row, unit); then the same fused cell update as
// lstm_fwd_step_kernel.  (MT, NT) = (4, 4): 16 FLOP per byte, one block per CU at configs[2].
#ifndef ARCVAE_S3_LBF
#define ARCVAE_S3_LBF 1
#endif
template <int MT, int NT, int P = 0>   // P: 0 exact-f32 MFMA, 1 throughput mode (bf16 operands), 2 three-piece bf16 (parity path)
__global__ __launch_bounds__(256, (P == 2 ? ARCVAE_S3_LBF : 1)) void lstm_fwd_tile_kernel(FwdArgs a) {
    constexpr bool BF = P == 1, S3 = P == 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDT = P >= 1 ? 16 * NT + 4 : 16 * NT + 1;     // LDS tile row stride (bf16 forms: 16-byte aligned rows for the quad epilogue)
    constexpr int UW = 4 * NT;           // hidden units per wave
    constexpr int NS = (MT * NT >= 16) ? 2 : 4;  // 4x4: a 4-stage ring (364 registers) measured no faster than 2
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    int bx, by, bz;
    tile_xcd_block(a.remap, bx, by, bz);
    const FwdJob& j = a.job[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int row0 = by * 16 * MT;
    const int wcol0 = (bx * 4 + wave) * 16 * NT;  // permuted weight rows: 16 consecutive = 4 units x (i,f,g,o)
    const int ubase = (bx * 4 + wave) * UW;       // first hidden unit of this wave
    int arow[MT], wrow[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) arow[m] = min(row0 + 16 * m + r, B - 1);
#pragma unroll
    for (int n = 0; n < NT; ++n) wrow[n] = wcol0 + 16 * n + r;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (S3) {   // three bf16 planes per operand, six products per 32-wide chunk
#ifndef ARCVAE_S3_NS_FWD
#define ARCVAE_S3_NS_FWD 2
#endif
        constexpr int NSS = (MT * NT >= 16) ? ARCVAE_S3_NS_FWD : ARCVAE_S3_NS_FWD + 1;
        const int q8 = (lane >> 4) * 8;
        const long pA = (long)B * H, pW = (long)G * H;
        if constexpr (NSS == 2 && ARCVAE_S3_RING2_FWD) {
            if (j.xin) tile_contract_s2<MT, NT, ARCVAE_S3_RING2_FWD == 2, (ARCVAE_DBG_SKIP_A != 0)>(acc, reinterpret_cast<const __bf16*>(j.xin), pA, arow, B,
                                                reinterpret_cast<const __bf16*>(j.Wx), pW, wrow, G, H >> 5, q8);
            if (j.hprev) tile_contract_s2<MT, NT, ARCVAE_S3_RING2_FWD == 2, (ARCVAE_DBG_SKIP_A != 0)>(acc, reinterpret_cast<const __bf16*>(j.hprev), pA, arow, B,
                                                  reinterpret_cast<const __bf16*>(j.Wh), pW, wrow, G, H >> 5, q8);
        } else {
        if (j.xin) tile_contract_s<MT, NT, NSS>(acc, reinterpret_cast<const __bf16*>(j.xin), pA, arow, B,
                                                reinterpret_cast<const __bf16*>(j.Wx), pW, wrow, G, H >> 5, q8);
        if (j.hprev) tile_contract_s<MT, NT, NSS>(acc, reinterpret_cast<const __bf16*>(j.hprev), pA, arow, B,
                                                  reinterpret_cast<const __bf16*>(j.Wh), pW, wrow, G, H >> 5, q8);
        }
    } else if constexpr (BF) {   // throughput mode: bf16 operand copies, 32-wide chunks (H % 64 == 0)
        constexpr int NSB = (MT * NT >= 16) ? 4 : 6;
        const int q8 = (lane >> 4) * 8;
        if (j.xin && !(a.dbg & 1)) tile_contract_b<MT, NT, NSB>(acc, reinterpret_cast<const __bf16*>(j.xin), arow, B,
                                                reinterpret_cast<const __bf16*>(j.Wx), wrow, G, H >> 5, q8);
        if (j.hprev && !(a.dbg & 1)) tile_contract_b<MT, NT, NSB>(acc, reinterpret_cast<const __bf16*>(j.hprev), arow, B,
                                                  reinterpret_cast<const __bf16*>(j.Wh), wrow, G, H >> 5, q8);
    } else {
    const int nch = H >> 4;
    if (j.xin) tile_contract<MT, NT, NS>(acc, j.xin, arow, B, j.Wx, wrow, G, nch, q4);
    if (j.hprev) tile_contract<MT, NT, NS>(acc, j.hprev, arow, B, j.Wh, wrow, G, nch, q4);
    }
    // accumulators -> per-wave LDS tile [16*MT][LDT]: D[row = 4*(lane>>4) + reg][col = lane & 15]
    float* t = lds + wave * (16 * MT * LDT);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) t[(16 * m + (lane >> 4) * 4 + reg) * LDT + 16 * n + r] = acc[m][n][reg];
    __syncthreads();
    // (row, unit) pairs of the wave: 16*MT rows x UW units = MT*NT per lane, handled in batches of <= 8.  Within a
    // batch all loads of all pairs are issued before the first store: the output pointers may alias the inputs as far
    // as the compiler knows, so a load placed after a store would wait for it, and the token -> table-row -> cell
    // chain would be paid once per pair in sequence.
    if constexpr (BF) { if (a.dbg & 2) return; }
    if constexpr (S3 || BF) {
        // Quad epilogue (round 3; both bf16 forms): a lane owns FOUR ADJACENT units of a row -- their 16 pre-activations are 64 contiguous bytes of
        // the LDS tile (the permuted gate columns: 4 units x (i, f, g, o)), and every global access is 16 bytes per lane (8 for a
        // bf16 plane): 14 memory instructions per four (row, unit) pairs instead of per pair.  16 MT NT quads per wave.
        constexpr int NQ = 16 * MT * NT, QPL = (NQ + 63) / 64;
        typedef __bf16 bf16x4_l __attribute__((ext_vector_type(4)));
        f32x4 pv[QPL][4], cpv[QPL];
        int qrow[QPL], qunit[QPL];
#pragma unroll
        for (int it = 0; it < QPL; ++it) {
            const int idx = min(it * 64 + lane, NQ - 1);
            qrow[it] = row0 + idx / NT;
            qunit[it] = ubase + 4 * (idx % NT);
            const int rc = min(qrow[it], B - 1);
            long o = 0;
            if (j.tok) {
                int tk = j.tok[rc];
                tk = min(max(tk, 0), a.V - 1);
                o = (long)tk * G;
            }
            const float* pre = j.pre + o + qunit[it];
#pragma unroll
            for (int g = 0; g < 4; ++g) pv[it][g] = *reinterpret_cast<const f32x4*>(pre + g * H);
            cpv[it] = j.cprev ? *reinterpret_cast<const f32x4*>(j.cprev + (long)rc * H + qunit[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < QPL; ++it) {
            const int idx = it * 64 + lane;
            if (idx >= NQ) continue;
            const int row = qrow[it], unit = qunit[it];
            if (row >= B) continue;
            const float* tp = t + (idx / NT) * LDT + 16 * (idx % NT);
            const f32x4 vi = *reinterpret_cast<const f32x4*>(tp), vf = *reinterpret_cast<const f32x4*>(tp + 4),
                        vg = *reinterpret_cast<const f32x4*>(tp + 8), vo = *reinterpret_cast<const f32x4*>(tp + 12);
            f32x4 gi, gf, gg, go, cc, hv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gi[k] = sigmoidf_acc(vi[k] + pv[it][0][k]);
                gf[k] = sigmoidf_acc(vf[k] + pv[it][1][k]);
                gg[k] = tanhf(vg[k] + pv[it][2][k]);
                go[k] = sigmoidf_acc(vo[k] + pv[it][3][k]);
                cc[k] = j.cprev ? gf[k] * cpv[it][k] + gi[k] * gg[k] : gi[k] * gg[k];
                hv[k] = go[k] * tanhf(cc[k]);
            }
            const long hb = (long)row * H + unit;
            *reinterpret_cast<f32x4*>(j.h + hb) = hv;
            if (!(a.lite & 1)) {
                float* gp = j.gates + (long)row * G + unit;
                *reinterpret_cast<f32x4*>(gp) = gi;
                *reinterpret_cast<f32x4*>(gp + H) = gf;
                *reinterpret_cast<f32x4*>(gp + 2 * H) = gg;
                *reinterpret_cast<f32x4*>(gp + 3 * H) = go;
                *reinterpret_cast<f32x4*>(j.c + hb) = cc;
            }
            if (a.lite & 2) continue;
            __bf16* hp = reinterpret_cast<__bf16*>(j.ht) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
            if constexpr (BF) {   // throughput mode: one plane + the octet-major copy of the weight-gradient kernel
                const bf16x4_l b4 = bf16x4_l{to_bf16(hv[0]), to_bf16(hv[1]), to_bf16(hv[2]), to_bf16(hv[3])};
                *reinterpret_cast<bf16x4_l*>(hp) = b4;
                if (j.oct) {
                    __bf16* op = reinterpret_cast<__bf16*>(j.oct) + ((long)(row >> 3) * H + unit) * 8 + (row & 7);
#pragma unroll
                    for (int k = 0; k < 4; ++k) op[k * 8] = b4[k];
                }
                continue;
            }
            __bf16 pc[3][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split3_bf16(hv[k], pc[0][k], pc[1][k], pc[2][k]);
            const long pl = (long)B * H;
#pragma unroll
            for (int pz = 0; pz < 3; ++pz)
                *reinterpret_cast<bf16x4_l*>(hp + pz * pl) = bf16x4_l{pc[pz][0], pc[pz][1], pc[pz][2], pc[pz][3]};
        }
        if (tr) a.trace[1] = wall_clock64();
        return;
    }
    constexpr int NPAIR = MT * NT;
    constexpr int NP = NPAIR < 8 ? NPAIR : 8;
#pragma unroll
    for (int b0 = 0; b0 < NPAIR; b0 += NP) {
        int prow[NP], punit[NP];
        long poff[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = (b0 + it) * 64 + lane;
            prow[it] = row0 + idx / UW;
            punit[it] = ubase + idx % UW;
            const int rc = min(prow[it], B - 1);
            long o = 0;
            if (j.tok) {
                int tk = j.tok[rc];
                tk = min(max(tk, 0), a.V - 1);
                o = (long)tk * G;
            }
            poff[it] = o;
        }
        float pv[NP][4], cpv[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const float* pre = j.pre + poff[it] + punit[it];
            pv[it][0] = pre[0]; pv[it][1] = pre[H]; pv[it][2] = pre[2 * H]; pv[it][3] = pre[3 * H];
            cpv[it] = j.cprev ? j.cprev[(long)min(prow[it], B - 1) * H + punit[it]] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int idx = (b0 + it) * 64 + lane;
            const int rl = idx / UW, ul = idx % UW;
            const int row = prow[it], unit = punit[it];
            if (row >= B) continue;
            const float* tp = t + rl * LDT + 16 * (ul >> 2) + (ul & 3);
            const float gi = sigmoidf_acc(tp[0] + pv[it][0]);
            const float gf = sigmoidf_acc(tp[4] + pv[it][1]);
            const float gg = tanhf(tp[8] + pv[it][2]);
            const float go = sigmoidf_acc(tp[12] + pv[it][3]);
            const long hb = (long)row * H + unit;
            const float c = j.cprev ? gf * cpv[it] + gi * gg : gi * gg;
            const float hv = go * tanhf(c);
            float* gp = j.gates + (long)row * G + unit;
            gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
            j.h[hb] = hv;
            if constexpr (S3) {
                __bf16 p0, p1, p2;
                split3_bf16(hv, p0, p1, p2);
                __bf16* tp = reinterpret_cast<__bf16*>(j.ht) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
                const long pl = (long)B * H;
                tp[0] = p0; tp[pl] = p1; tp[2 * pl] = p2;
            } else if constexpr (BF) {
                const __bf16 hb16 = to_bf16(hv);
                reinterpret_cast<__bf16*>(j.ht)[((long)(unit >> 5) * B + row) * 32 + (unit & 31)] = hb16;
                if (j.oct) reinterpret_cast<__bf16*>(j.oct)[((long)(row >> 3) * H + unit) * 8 + (row & 7)] = hb16;
            } else {
                j.ht[((long)(unit >> 4) * B + row) * 16 + (unit & 15)] = hv;
            }
            j.c[hb] = c;
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// BPTT: block = 16*MT rows x 128 hidden units (wave w: units [32w, 32w+32)), K = 4H.  grid (ceil(H/128),
// ceil(B / (16*MT)), jobs).  An accumulator element IS one (row, unit): the cell epilogue runs on the registers.
#ifndef ARCVAE_S3_LB
#define ARCVAE_S3_LB 1
#endif
template <int MT, int P = 0>   // P as in lstm_fwd_tile_kernel
__global__ __launch_bounds__(256, (P == 2 ? ARCVAE_S3_LB : 1)) void lstm_bwd_tile_kernel(BwdArgs a) {
    constexpr bool BF = P == 1, S3 = P == 2;
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bx, by, bz;
    tile_xcd_block(a.remap, bx, by, bz);
    const BwdJob& j = a.job[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int row0 = by * 16 * MT;
    const int u0 = bx * 128 + wave * 32;
    if (u0 >= H) return;                              // H not a multiple of 128: this wave has no units
    int arow[MT], wrow[2];
#pragma unroll
    for (int m = 0; m < MT; ++m) arow[m] = min(row0 + 16 * m + r, B - 1);
    wrow[0] = min(u0 + r, H - 1);
    wrow[1] = min(u0 + 16 + r, H - 1);
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (S3) {
#ifndef ARCVAE_S3_NS_BWD
#define ARCVAE_S3_NS_BWD 2   /* two stages: a third (368 registers) was no faster alone and 9 % slower in the step -- the GEMMs beside the sweep lose their room */
#endif
        if constexpr (ARCVAE_S3_NS_BWD == 2 && ARCVAE_S3_RING2_BWD) {
            if (j.src && !(a.dbg & 1)) tile_contract_s2<MT, 2>(acc, reinterpret_cast<const __bf16*>(j.src), (long)B * G, arow, B,
                                               reinterpret_cast<const __bf16*>(j.WT), (long)H * G, wrow, H, G >> 5, (lane >> 4) * 8);
        } else {
        if (j.src && !(a.dbg & 1)) tile_contract_s<MT, 2, ARCVAE_S3_NS_BWD>(acc, reinterpret_cast<const __bf16*>(j.src), (long)B * G, arow, B,
                                             reinterpret_cast<const __bf16*>(j.WT), (long)H * G, wrow, H, G >> 5, (lane >> 4) * 8);
        }
    } else if constexpr (BF) {
        if (j.src && !(a.dbg & 1)) tile_contract_b<MT, 2, 6>(acc, reinterpret_cast<const __bf16*>(j.src), arow, B,
                                             reinterpret_cast<const __bf16*>(j.WT), wrow, H, G >> 5, (lane >> 4) * 8);
    } else {
        if (j.src) tile_contract<MT, 2, 4>(acc, j.src, arow, B, j.WT, wrow, H, G >> 4, q4);
    }
    const bool cell = j.kind == 0;
    if constexpr (BF || S3) { if (a.dbg & 2) { if (acc[0][0][0] == 12345.f) j.out[0] = 0.f; return; } }   // (timing experiments)
    // Epilogue per 16-row group m: the 8 (row, unit) elements of a lane are loaded together (clamped indices, no
    // branch between the loads) and only then computed and stored -- see the forward tile kernel.
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float gi[8], gf[8], gg[8], go[8], cv[8], cpv[8], dci[8], exv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = e >> 2, reg = e & 3;
            const int unit = min(u0 + 16 * n + r, H - 1);
            const int row = min(row0 + 16 * m + (lane >> 4) * 4 + reg, B - 1);
            const long hb = (long)row * H + unit;
            exv[e] = j.ext ? j.ext[(long)row * j.ext_ld + unit] : 0.f;
            if (cell) {
                const float* gp = j.gates + (long)row * G + unit;
                gi[e] = gp[0]; gf[e] = gp[H]; gg[e] = gp[2 * H]; go[e] = gp[3 * H];
                cv[e] = j.c[hb];
                cpv[e] = j.cprev ? j.cprev[hb] : 0.f;
                dci[e] = j.dcin ? j.dcin[hb] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = e >> 2, reg = e & 3;
            const int unit = u0 + 16 * n + r;
            const int row = row0 + 16 * m + (lane >> 4) * 4 + reg;
            if (unit >= H || row >= B) continue;
            const long hb = (long)row * H + unit;
            const float dh = acc[m][n][reg] + exv[e];
            if (!cell) {
                j.out[hb] = dh;
                continue;
            }
            const float i = gi[e], f = gf[e], g = gg[e], o = go[e];
            const float tc = tanhf(cv[e]);
            const float d_o = dh * tc * o * (1.f - o);
            const float dc = dh * o * (1.f - tc * tc) + dci[e];
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * cpv[e] * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.out + (long)row * G + unit;
            dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
            if constexpr (S3) {
                __bf16* tp = reinterpret_cast<__bf16*>(j.outt) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
                const long gs = (long)(H >> 5) * B * 32, pl = (long)B * G;
                const float dv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __bf16 p0, p1, p2;
                    split3_bf16(dv[q], p0, p1, p2);
                    tp[q * gs] = p0; tp[q * gs + pl] = p1; tp[q * gs + 2 * pl] = p2;
                }
            } else if constexpr (BF) {
                __bf16* tp = reinterpret_cast<__bf16*>(j.outt) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
                const long gs = (long)(H >> 5) * B * 32;
                const __bf16 b_i = to_bf16(d_i), b_f = to_bf16(d_f), b_g = to_bf16(d_g), b_o = to_bf16(d_o);
                tp[0] = b_i; tp[gs] = b_f; tp[2 * gs] = b_g; tp[3 * gs] = b_o;
                if (j.oct) {
                    __bf16* op = reinterpret_cast<__bf16*>(j.oct) + ((long)(row >> 3) * G + unit) * 8 + (row & 7);
                    op[0] = b_i; op[(long)H * 8] = b_f; op[(long)2 * H * 8] = b_g; op[(long)3 * H * 8] = b_o;
                }
            } else {
            float* tp = j.outt + ((long)(unit >> 4) * B + row) * 16 + (unit & 15);
            const long gs = (long)(H >> 4) * B * 16;
            tp[0] = d_i; tp[gs] = d_f; tp[2 * gs] = d_g; tp[3 * gs] = d_o;
            }
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// BPTT, throughput mode, K split over the waves.  With bf16 operands the 64 x 32 wave tile above moves 48 bytes per MFMA
// cycle and runs one block per CU on a 224-block grid: 49 us per launch at BASELINE.json configs[2], ~6 % of the bf16 MFMA
// peak, bound by the latency of its own 384 KB of operand loads per wave.  Here a block owns 64 rows x 64 units and its
// four waves each contract a QUARTER of K = 4H into a 64 x 64 register tile (16 accumulators, 32 bytes per MFMA cycle,
// a third of the bytes per wave, twice the blocks); the partial tiles are exchanged through LDS (every wave ends up with
// the 16-row group it runs the cell epilogue on).  grid (H/64, ceil(B/64), jobs).
__global__ __launch_bounds__(256) void lstm_bwd_tile_ks_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ksred[];     // [4 src waves][3 foreign groups][16 x 4 regs][64 lanes]
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bx, by, bz;
    tile_xcd_block(a.remap, bx, by, bz);
    const BwdJob& j = a.job[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r = lane & 15;
    const int row0 = by * 64, u0 = bx * 64;
    int arow[4], wrow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        arow[m] = min(row0 + 16 * m + r, B - 1);
        wrow[m] = min(u0 + 16 * m + r, H - 1);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j.src && !(a.dbg & 1)) {
        const int nq = G >> 7;                            // 32-wide chunks per wave (G / 32 / 4; H % 64 == 0)
        const __bf16* At = reinterpret_cast<const __bf16*>(j.src) + (long)wave * nq * B * 32;
        const __bf16* Wt = reinterpret_cast<const __bf16*>(j.WT) + (long)wave * nq * H * 32;
        tile_contract_b<4, 4, 4>(acc, At, arow, B, Wt, wrow, H, nq, (lane >> 4) * 8);
        // exchange: group m of wave w goes to wave m (slot w' = w - (w > m) of its 3 foreign sources)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m == wave) continue;
            float* dst = ksred + ((m * 3 + (wave - (wave > m ? 1 : 0))) * 16) * 64 + lane;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dst[(n * 4 + reg) * 64] = acc[m][n][reg];
        }
        __syncthreads();
    }
    f32x4 mine[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        mine[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m == wave) mine[n] = acc[m][n];           // wave-uniform select: my own 16-row group
    }
    if (j.src && !(a.dbg & 1)) {
#pragma unroll
        for (int sidx = 0; sidx < 3; ++sidx) {
            const float* srcp = ksred + ((wave * 3 + sidx) * 16) * 64 + lane;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) mine[n][reg] += srcp[(n * 4 + reg) * 64];
        }
    }
    const bool cell = j.kind == 0;
    if (a.dbg & 2) { if (mine[0][0] == 12345.f) j.out[0] = 0.f; return; }
    // cell epilogue on rows row0 + 16 wave + 4 (lane >> 4) + reg, units u0 + 16 n + r: two batches of 8 elements (loads of a
    // batch all issued before its first store, as in lstm_bwd_tile_kernel).  Requesting all 16 elements' operands BEFORE the
    // contraction (405 registers) measured the same 54 us per launch: the phase is bound by its stores, not its loads.
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        float gi[8], gf[8], gg[8], go[8], cv[8], cpv[8], dci[8], exv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 2 * hf + (e >> 2), reg = e & 3;
            const int unit = min(u0 + 16 * n + r, H - 1);
            const int row = min(row0 + 16 * wave + (lane >> 4) * 4 + reg, B - 1);
            const long hb = (long)row * H + unit;
            exv[e] = j.ext ? j.ext[(long)row * j.ext_ld + unit] : 0.f;
            if (cell) {
                const float* gp = j.gates + (long)row * G + unit;
                gi[e] = gp[0]; gf[e] = gp[H]; gg[e] = gp[2 * H]; go[e] = gp[3 * H];
                cv[e] = j.c[hb];
                cpv[e] = j.cprev ? j.cprev[hb] : 0.f;
                dci[e] = j.dcin ? j.dcin[hb] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 2 * hf + (e >> 2), reg = e & 3;
            const int unit = u0 + 16 * n + r;
            const int row = row0 + 16 * wave + (lane >> 4) * 4 + reg;
            if (unit >= H || row >= B) continue;
            const long hb = (long)row * H + unit;
            const float dh = mine[n][reg] + exv[e];
            if (!cell) {
                j.out[hb] = dh;
                continue;
            }
            const float i = gi[e], f = gf[e], g = gg[e], o = go[e];
            const float tc = tanhf(cv[e]);
            const float d_o = dh * tc * o * (1.f - o);
            const float dc = dh * o * (1.f - tc * tc) + dci[e];
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * cpv[e] * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.out + (long)row * G + unit;
            dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
            const __bf16 b_i = to_bf16(d_i), b_f = to_bf16(d_f), b_g = to_bf16(d_g), b_o = to_bf16(d_o);
            __bf16* tp = reinterpret_cast<__bf16*>(j.outt) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
            const long gs = (long)(H >> 5) * B * 32;
            if (!(a.dbg & 4)) { tp[0] = b_i; tp[gs] = b_f; tp[2 * gs] = b_g; tp[3 * gs] = b_o; }
            if (j.oct && !(a.dbg & 8)) {
                __bf16* op = reinterpret_cast<__bf16*>(j.oct) + ((long)(row >> 3) * G + unit) * 8 + (row & 7);
                op[0] = b_i; op[(long)H * 8] = b_f; op[(long)2 * H * 8] = b_g; op[(long)3 * H * 8] = b_o;
            }
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// BPTT, three-piece form, K split over the waves (round 3).  The 64 x 32 wave tile of lstm_bwd_tile_kernel<4, 2> asks its CU
// for 96 B per matrix cycle (18 KB of operand planes per 768 cycles of MFMAs and wave) against the 64 B/clk the vector
// memory path delivers, and its epilogue is bound by the NUMBER of memory instructions (25 per element: 2-byte plane stores,
// 64-byte segments).  Here, as in lstm_bwd_tile_ks_kernel: a block owns 64 rows x 64 units, every wave contracts a quarter
// of K = 4H into a 64 x 64 register tile (24 KB per 1536 matrix cycles: 64 B/clk per CU); ALL four partial tiles go through
// LDS and come back in the epilogue's layout -- a lane owns FOUR ADJACENT units of a row, so every load and store of the
// cell epilogue is 16 bytes per lane (8 for the bf16 planes) in whole 128-byte lines: 25 memory instructions per FOUR
// elements.  grid (H/64, ceil(B/64), jobs), 64 KB of LDS.
#ifndef ARCVAE_KS3_PIN
#define ARCVAE_KS3_PIN 1
#endif
template <int P = 2>   // P = 2: three bf16 pieces (parity path); P = 1: throughput mode (one bf16 plane + the octet-major copy)
__global__ __launch_bounds__(256) void lstm_bwd_tile_ks3_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ksred[];     // [4 dst groups][4 src waves][4 n][4 regs][64 lanes]
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bx, by, bz;
    tile_xcd_block(a.remap, bx, by, bz);
    const BwdJob& j = a.job[bz];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int row0 = by * 64, u0 = bx * 64;
    f32x4 mine[4];                                        // [reg: row 16 wave + 4 (lane >> 4) + reg][k: unit u0 + 4 (lane & 15) + k]
#pragma unroll
    for (int g = 0; g < 4; ++g) mine[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j.src && !(a.dbg & 1)) {                          // block-uniform (dbg: timing experiments only)
        const int r = lane & 15;
        int arow[4], wrow[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            arow[m] = min(row0 + 16 * m + r, B - 1);
            wrow[m] = u0 + 16 * m + r;                    // (H % 64 == 0: always a unit)
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nq = G >> 7;                            // 32-wide chunks per wave (G / 32 / 4)
        const __bf16* At = reinterpret_cast<const __bf16*>(j.src) + (long)wave * nq * B * 32;
        const __bf16* Wt = reinterpret_cast<const __bf16*>(j.WT) + (long)wave * nq * H * 32;
        if constexpr (P == 1) {
            tile_contract_b<4, 4, 4>(acc, At, arow, B, Wt, wrow, H, nq, (lane >> 4) * 8);
        } else {
#if ARCVAE_S3_RING2_KS == 2
        tile_contract_s2<4, 4, true>(acc, At, (long)B * G, arow, B, Wt, (long)H * G, wrow, H, nq, (lane >> 4) * 8);
#elif ARCVAE_S3_RING2_KS
        tile_contract_s2<4, 4>(acc, At, (long)B * G, arow, B, Wt, (long)H * G, wrow, H, nq, (lane >> 4) * 8);
#else
        tile_contract_s<4, 4, 2>(acc, At, (long)B * G, arow, B, Wt, (long)H * G, wrow, H, nq, (lane >> 4) * 8);
#endif
        }
        // every partial tile through LDS: group m of wave w -> slot [m][w]; element (n, reg) of lane = row 16 m + 4 (lane >> 4) + reg,
        // unit u0 + 16 n + (lane & 15)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float* dst = ksred + ((m * 4 + wave) * 16) * 64 + lane;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dst[(n * 4 + reg) * 64] = acc[m][n][reg];
        }
        __syncthreads();
        // ... and back, transposed: my four adjacent units 4 (lane & 15) + k sit in tile n = (lane & 15) >> 2 at lanes
        // (lane >> 4) * 16 + 4 (lane & 3) + k of the source wave: one 16-byte LDS read per (source, reg)
        const int nL = (lane & 15) >> 2, l0 = (lane >> 4) * 16 + 4 * (lane & 3);
#pragma unroll
        for (int sw = 0; sw < 4; ++sw)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ksred + ((wave * 4 + sw) * 16 + nL * 4 + reg) * 64 + l0);
                mine[reg] += v;
            }
    }
    const bool cell = j.kind == 0;
    if (a.dbg & 2) { if (mine[0][0] == 12345.f) j.out[0] = 0.f; return; }
    const int unit = u0 + 4 * (lane & 15);
    // two rows per batch: all operand loads of a batch are issued before its first store
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 gi[2], gf[2], gg[2], go[2], cv[2], cpv[2], dci[2], exv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = min(row0 + 16 * wave + (lane >> 4) * 4 + 2 * hf + e, B - 1);
            const long hb = (long)row * H + unit;
            exv[e] = j.ext ? *reinterpret_cast<const f32x4*>(j.ext + (long)row * j.ext_ld + unit) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (cell) {
                const float* gp = j.gates + (long)row * G + unit;
                gi[e] = *reinterpret_cast<const f32x4*>(gp);
                gf[e] = *reinterpret_cast<const f32x4*>(gp + H);
                gg[e] = *reinterpret_cast<const f32x4*>(gp + 2 * H);
                go[e] = *reinterpret_cast<const f32x4*>(gp + 3 * H);
                cv[e] = *reinterpret_cast<const f32x4*>(j.c + hb);
                cpv[e] = j.cprev ? *reinterpret_cast<const f32x4*>(j.cprev + hb) : f32x4{0.f, 0.f, 0.f, 0.f};
                dci[e] = j.dcin ? *reinterpret_cast<const f32x4*>(j.dcin + hb) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = row0 + 16 * wave + (lane >> 4) * 4 + 2 * hf + e;
            if (row >= B) continue;
            const long hb = (long)row * H + unit;
            const f32x4 dh = mine[2 * hf + e] + exv[e];
            if (!cell) {
                *reinterpret_cast<f32x4*>(j.out + hb) = dh;
                continue;
            }
            f32x4 d_i, d_f, d_g, d_o, dcf;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float i = gi[e][k], f = gf[e][k], g = gg[e][k], o = go[e][k];
                const float tc = tanhf(cv[e][k]);
                d_o[k] = dh[k] * tc * o * (1.f - o);
                const float dc = dh[k] * o * (1.f - tc * tc) + dci[e][k];
                d_i[k] = dc * g * i * (1.f - i);
                d_f[k] = j.cprev ? dc * cpv[e][k] * f * (1.f - f) : 0.f;
                d_g[k] = dc * i * (1.f - g * g);
                dcf[k] = dc * f;
            }
            *reinterpret_cast<f32x4*>(j.dcout + hb) = dcf;
            float* dp = j.out + (long)row * G + unit;
            *reinterpret_cast<f32x4*>(dp) = d_i;
            *reinterpret_cast<f32x4*>(dp + H) = d_f;
            *reinterpret_cast<f32x4*>(dp + 2 * H) = d_g;
            *reinterpret_cast<f32x4*>(dp + 3 * H) = d_o;
            // the three bf16 planes of the next launches' operand copy: 4 units = 8 bytes per (gate, plane)
            __bf16* tp = reinterpret_cast<__bf16*>(j.outt) + ((long)(unit >> 5) * B + row) * 32 + (unit & 31);
            const long gs = (long)(H >> 5) * B * 32, pl = (long)B * G;
            const f32x4 dv[4] = {d_i, d_f, d_g, d_o};
            typedef __bf16 bf16x4_l __attribute__((ext_vector_type(4)));
            if constexpr (P == 1) {   // throughput mode: one plane (8 bytes per gate) + the octet-major copy of the weight-gradient kernel
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x4_l b4 = bf16x4_l{to_bf16(dv[q][0]), to_bf16(dv[q][1]), to_bf16(dv[q][2]), to_bf16(dv[q][3])};
                    *reinterpret_cast<bf16x4_l*>(tp + q * gs) = b4;
                    if (j.oct) {
                        __bf16* op = reinterpret_cast<__bf16*>(j.oct) + ((long)(row >> 3) * G + q * H + unit) * 8 + (row & 7);
#pragma unroll
                        for (int k = 0; k < 4; ++k) op[k * 8] = b4[k];
                    }
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __bf16 pc[3][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) split3_bf16(dv[q][k], pc[0][k], pc[1][k], pc[2][k]);
#pragma unroll
                for (int pz = 0; pz < 3; ++pz)
                    *reinterpret_cast<bf16x4_l*>(tp + q * gs + pz * pl) = bf16x4_l{pc[pz][0], pc[pz][1], pc[pz][2], pc[pz][3]};
            }
            }
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// Largest register tile whose grid still fills the chip (>= minb blocks: 200; 140 in the three-piece form, see
// tile_min_blocks), or 0 = keep the latency-oriented 16x16 kernels.  ARCVAE_STEP_TILE: -1 auto (default), 0 never, 1 / 2 / 4
// force that MT.
inline int choose_tile_mt(int B, int col_blocks, int jobs, int minb = 200) {
    const int force = arcvae_env_int("ARCVAE_STEP_TILE", -1);  // read per sweep call (tests toggle it)
    if (force == 0 || force == 22) return 0;
    if (force == 1 || force == 2 || force == 4) return force;
    if (force == 44) return 4;
    const int mts[2] = {4, 2};  // MT = 1 never pays: at that size the all-loads-first 16x16 kernels are ahead
    for (int i = 0; i < 2; ++i)
        if (ceil_div(B, 16 * mts[i]) * col_blocks * jobs >= minb) return mts[i];
    return 0;
}

template <int MT, int NT, int P = 0>
void launch_fwd_tile(const FwdArgs& a, int B, int H, int nj, hipStream_t s) {
    dim3 grid(H / (16 * NT), ceil_div(B, 16 * MT), nj);
    hipLaunchKernelGGL((lstm_fwd_tile_kernel<MT, NT, P>), grid, dim3(256), 4 * 16 * MT * (16 * NT + (P >= 1 ? 4 : 1)) * sizeof(float), s, a);
}
template <int MT, int P = 0>
void launch_bwd_tile(const BwdArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((lstm_bwd_tile_kernel<MT, P>), grid, dim3(256), 0, s, a);
}


// ---- mid-batch BPTT step kernel ("2x2") ---------------------------------------------------------------------------
// Between the latency regime (B = 64) and the register-tiled one (grids of >= 200 blocks) -- per-GPU batches of
// 256..512 rows, e.g. the 256 rows per GPU of BASELINE.json configs[3] -- the 16x16 kernel is bound by L2 -> CU
// delivery (each block pulls 64 KB of A and 64 KB of W for a 16x16 tile: 98 MB per launch at B = 256) while the
// tiled kernel's grid is too small.  This kernel keeps the latency kernel's shape (K = 4H split over 16 waves, every
// load issued before the first MFMA, epilogue operands requested first) but gives each wave 2 x 2 fragments: a block
// covers 32 rows x 32 hidden units, i.e. half the bytes per output, and its 1024 threads each own exactly one
// (row, unit) of the epilogue.  Measured (MI355X, H256 L2): the launch itself barely moves (10.8 -> 10.7 us at bs 256,
// 17.9 -> 17.0 at bs 512: its 192..384 blocks are MFMA-serialised, 16 waves on 4 SIMDs), the step does (4.02 -> 3.87 ms
// and 7.35 -> 6.70 ms) because the sweep takes half the L2 bandwidth away from the GEMMs beside it; at bs 128 it loses
// (6.9 -> 9.5 us), hence the threshold.  A 2-D XCD partition of the blocks (A/p + W/q fabric bytes per XCD instead of
// all of A) was tried on top and changed nothing: the launch is not fabric-bound.
template <int CH>
__global__ __launch_bounds__(1024) void lstm_bwd_step2_kernel(BwdArgs a) {
    __shared__ float red[16 * 1024];  // [wave][32 rows][32 units]
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const BwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r0 = blockIdx.y * 32, u0 = blockIdx.x * 32;
    const bool cell = j.kind == 0;
    // epilogue role: thread -> (row = tid >> 5, unit = tid & 31); operands requested first
    const int erow = tid >> 5, ecol = tid & 31;
    const int eb = min(r0 + erow, B - 1);
    const int unit = min(u0 + ecol, H - 1);
    const long hb = (long)eb * H + unit;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c_v = 0.f, cprev_v = 0.f, dcin_v = 0.f, ext_v = 0.f;
    if (j.ext) ext_v = j.ext[(long)eb * j.ext_ld + unit];
    if (cell) {
        const float* gp = j.gates + (long)eb * G + unit;
        gi = gp[0]; gf = gp[H]; gg = gp[2 * H]; go = gp[3 * H];
        c_v = j.c[hb];
        if (j.cprev) cprev_v = j.cprev[hb];
        if (j.dcin) dcin_v = j.dcin[hb];
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j.src) {
        const int r = lane & 15, q4 = (lane >> 4) * 4;
        const int arow0 = min(r0 + r, B - 1), arow1 = min(r0 + 16 + r, B - 1);
        const int wrow0 = min(u0 + r, H - 1), wrow1 = min(u0 + 16 + r, H - 1);
        float4 fa[2][CH], fw[2][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {   // wave w owns chunks [w*CH, (w+1)*CH) of the 4H/16 chunks
            const long kc = wave * CH + c;
            fa[0][c] = *reinterpret_cast<const float4*>(j.src + (kc * B + arow0) * 16 + q4);
            fa[1][c] = *reinterpret_cast<const float4*>(j.src + (kc * B + arow1) * 16 + q4);
            fw[0][c] = *reinterpret_cast<const float4*>(j.WT + (kc * H + wrow0) * 16 + q4);
            fw[1][c] = *reinterpret_cast<const float4*>(j.WT + (kc * H + wrow1) * 16 + q4);
        }
#define STEP2_MFMA(comp)                                                                                      \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)               \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][c].comp, fw[n][c].comp, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            STEP2_MFMA(x) STEP2_MFMA(y) STEP2_MFMA(z) STEP2_MFMA(w)
        }
#undef STEP2_MFMA
    }
    {   // partial tiles -> red[wave][row][unit]: D[row = 4*(lane>>4) + reg][col = lane & 15] per 16x16 fragment
        float* p = red + wave * 1024;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    p[(16 * m + (lane >> 4) * 4 + reg) * 32 + 16 * n + (lane & 15)] = acc[m][n][reg];
    }
    __syncthreads();
    if (r0 + erow < B && u0 + ecol < H) {
        float dh = ext_v;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += red[w * 1024 + erow * 32 + ecol];
        if (!cell) {
            j.out[hb] = dh;
        } else {
            const float i = gi, f = gf, g = gg, o = go;
            const float tc = tanhf(c_v);
            const float d_o = dh * tc * o * (1.f - o);
            const float dc = dh * o * (1.f - tc * tc) + dcin_v;
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * cprev_v * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.out + (long)eb * G + unit;
            dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
            float* tp = j.outt + ((long)(unit >> 4) * B + eb) * 16 + (unit & 15);
            const long gs = (long)(H >> 4) * B * 16;
            tp[0] = d_i; tp[gs] = d_f; tp[2 * gs] = d_g; tp[3 * gs] = d_o;
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}

// Mid-batch forward step kernel ("2x2"): see lstm_bwd_step2_kernel.  Block = 32 rows x 32 gate columns (8 hidden
// units, permuted weight rows: 16 consecutive = 4 units x i,f,g,o), K split over the 4 waves, 2 x 2 fragments per wave
// and source, every load before the first MFMA; the 256 threads each own one (row, unit) of the cell update.
template <int CH>
__global__ __launch_bounds__(256) void lstm_fwd_step2_kernel(FwdArgs a) {
    __shared__ float red[4 * 1024];  // [wave][32 rows][32 cols]
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    const FwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int bx = xcd_group_block(blockIdx.x, gridDim.x, a.remap & 1);  // 4 consecutive blocks fill a 128-B line of h / c
    const int r0 = blockIdx.y * 32, u0 = bx * 8;
    // epilogue role first: thread -> (row = tid >> 3, unit = u0 + (tid & 7)); token -> table row is a dependent pair
    const int erow = tid >> 3, ul = tid & 7;
    const int eb = min(r0 + erow, B - 1);
    const int unit = u0 + ul;
    const float* pre = j.pre;
    if (j.tok) {
        int tk = j.tok[eb];
        tk = min(max(tk, 0), a.V - 1);
        pre += (long)tk * G;
    }
    const float p0 = pre[unit], p1 = pre[H + unit], p2 = pre[2 * H + unit], p3 = pre[3 * H + unit];
    const long hb = (long)eb * H + unit;
    const float cprev_v = j.cprev ? j.cprev[hb] : 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int arow0 = min(r0 + r, B - 1), arow1 = min(r0 + 16 + r, B - 1);
    const int wrow0 = bx * 32 + r, wrow1 = bx * 32 + 16 + r;
    const bool s1 = j.xin != nullptr, s2 = j.hprev != nullptr;
    float4 fa1[2][CH], fw1[2][CH], fa2[2][CH], fw2[2][CH];
    if (s1) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {   // wave w owns chunks [w*CH, (w+1)*CH) of the H/16 chunks of a source
            const long kc = wave * CH + c;
            fa1[0][c] = *reinterpret_cast<const float4*>(j.xin + (kc * B + arow0) * 16 + q4);
            fa1[1][c] = *reinterpret_cast<const float4*>(j.xin + (kc * B + arow1) * 16 + q4);
            fw1[0][c] = *reinterpret_cast<const float4*>(j.Wx + (kc * G + wrow0) * 16 + q4);
            fw1[1][c] = *reinterpret_cast<const float4*>(j.Wx + (kc * G + wrow1) * 16 + q4);
        }
    }
    if (s2) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const long kc = wave * CH + c;
            fa2[0][c] = *reinterpret_cast<const float4*>(j.hprev + (kc * B + arow0) * 16 + q4);
            fa2[1][c] = *reinterpret_cast<const float4*>(j.hprev + (kc * B + arow1) * 16 + q4);
            fw2[0][c] = *reinterpret_cast<const float4*>(j.Wh + (kc * G + wrow0) * 16 + q4);
            fw2[1][c] = *reinterpret_cast<const float4*>(j.Wh + (kc * G + wrow1) * 16 + q4);
        }
    }
#define STEP2F_MFMA(FA, FW, comp)                                                                             \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)               \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[m][c].comp, FW[n][c].comp, acc[m][n], 0, 0, 0);
    if (s1) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { STEP2F_MFMA(fa1, fw1, x) STEP2F_MFMA(fa1, fw1, y) STEP2F_MFMA(fa1, fw1, z) STEP2F_MFMA(fa1, fw1, w) }
    }
    if (s2) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { STEP2F_MFMA(fa2, fw2, x) STEP2F_MFMA(fa2, fw2, y) STEP2F_MFMA(fa2, fw2, z) STEP2F_MFMA(fa2, fw2, w) }
    }
#undef STEP2F_MFMA
    {
        float* p = red + wave * 1024;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    p[(16 * m + (lane >> 4) * 4 + reg) * 32 + 16 * n + (lane & 15)] = acc[m][n][reg];
    }
    __syncthreads();
    if (r0 + erow < B) {
        // tile column of (unit ul, gate g): 16 * (ul >> 2) + 4 * g + (ul & 3)
        const int cb = erow * 32 + 16 * (ul >> 2) + (ul & 3);
        float v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            v[g] = (red[cb + 4 * g] + red[1024 + cb + 4 * g]) + (red[2048 + cb + 4 * g] + red[3072 + cb + 4 * g]);
        const float gi = sigmoidf_acc(v[0] + p0), gf = sigmoidf_acc(v[1] + p1), gg = tanhf(v[2] + p2),
                    go = sigmoidf_acc(v[3] + p3);
        const float c = j.cprev ? gf * cprev_v + gi * gg : gi * gg;
        const float hv = go * tanhf(c);
        float* gp = j.gates + (long)eb * G + unit;
        gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
        j.h[hb] = hv;
        j.ht[((long)(unit >> 4) * B + eb) * 16 + (unit & 15)] = hv;
        j.c[hb] = c;
    }
    if (tr) a.trace[1] = wall_clock64();
}
template <int CH>
void launch_fwd2(const FwdArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(lstm_fwd_step2_kernel<CH>, grid, dim3(256), 0, s, a);
}
template <int CH>
void launch_bwd2(const BwdArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(lstm_bwd_step2_kernel<CH>, grid, dim3(1024), 0, s, a);
}
// The same kernel in the three-piece form (round 3): a third of a mid-batch BPTT launch is MATRIX time -- 64 exact-f32
// instructions of 32 cycles per wave, four waves per SIMD: 3.5 us of a 10.7 us launch at 256 rows -- and it sits behind the
// operand loads, on the chain.  With the operand copies as hi / mid / lo bf16 planes (the layout of the tiled sweeps:
// [plane][k >> 5][row][32]) a wave's 64 k are two 32-wide chunks, six products each on v_mfma_f32_16x16x32_bf16: 48
// instructions of 16 cycles, 1.3 us per launch; 1.5 x the operand bytes.  CH even (H % 128 == 0).  The epilogue writes the next
// launches' planes.
template <int CH>
__global__ __launch_bounds__(1024) void lstm_bwd_step2s_kernel(BwdArgs a) {
    static_assert(CH % 2 == 0, "two 16-wide chunks make one 32-wide plane chunk");
    __shared__ float red[16 * 1024];  // [wave][32 rows][32 units]
    arcvae_set_prio(a.prio);
    const bool tr = a.trace && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (tr) a.trace[0] = wall_clock64();
    if (a.signal && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const BwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r0 = blockIdx.y * 32, u0 = blockIdx.x * 32;
    const bool cell = j.kind == 0;
    const int erow = tid >> 5, ecol = tid & 31;
    const int eb = min(r0 + erow, B - 1);
    const int unit = min(u0 + ecol, H - 1);
    const long hb = (long)eb * H + unit;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c_v = 0.f, cprev_v = 0.f, dcin_v = 0.f, ext_v = 0.f;
    if (j.ext) ext_v = j.ext[(long)eb * j.ext_ld + unit];
    if (cell) {
        const float* gp = j.gates + (long)eb * G + unit;
        gi = gp[0]; gf = gp[H]; gg = gp[2 * H]; go = gp[3 * H];
        c_v = j.c[hb];
        if (j.cprev) cprev_v = j.cprev[hb];
        if (j.dcin) dcin_v = j.dcin[hb];
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j.src) {
        constexpr int NC = CH / 2;                        // 32-wide chunks per wave
        const int r = lane & 15, q8 = (lane >> 4) * 8;
        const int arow[2] = {min(r0 + r, B - 1), min(r0 + 16 + r, B - 1)};
        const int wrow[2] = {min(u0 + r, H - 1), min(u0 + 16 + r, H - 1)};
        const __bf16* At = reinterpret_cast<const __bf16*>(j.src);
        const __bf16* Wt = reinterpret_cast<const __bf16*>(j.WT);
        const long pA = (long)B * G, pW = (long)H * G;
        u32x4_l fa[3][2][NC], fw[3][2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const long kc = wave * NC + c;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    fa[p][m][c] = *reinterpret_cast<const u32x4_l*>(At + p * pA + (kc * B + arow[m]) * 32 + q8);
                    fw[p][m][c] = *reinterpret_cast<const u32x4_l*>(Wt + p * pW + (kc * H + wrow[m]) * 32 + q8);
                }
        }
#define STEP2_S3(PA, PW)                                                                                               \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)                    \
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_l, fa[PA][m][c]),            \
                                                                __builtin_bit_cast(bf16x8_l, fw[PW][n][c]), acc[m][n], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            STEP2_S3(0, 2) STEP2_S3(2, 0) STEP2_S3(1, 1) STEP2_S3(0, 1) STEP2_S3(1, 0) STEP2_S3(0, 0)
        }
#undef STEP2_S3
    }
    {
        float* p = red + wave * 1024;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    p[(16 * m + (lane >> 4) * 4 + reg) * 32 + 16 * n + (lane & 15)] = acc[m][n][reg];
    }
    __syncthreads();
    if (r0 + erow < B && u0 + ecol < H) {
        float dh = ext_v;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += red[w * 1024 + erow * 32 + ecol];
        if (!cell) {
            j.out[hb] = dh;
        } else {
            const float i = gi, f = gf, g = gg, o = go;
            const float tc = tanhf(c_v);
            const float d_o = dh * tc * o * (1.f - o);
            const float dc = dh * o * (1.f - tc * tc) + dcin_v;
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * cprev_v * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.out + (long)eb * G + unit;
            dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
            __bf16* tp = reinterpret_cast<__bf16*>(j.outt) + ((long)(unit >> 5) * B + eb) * 32 + (unit & 31);
            const long gs = (long)(H >> 5) * B * 32, pl = (long)B * G;
            const float dv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __bf16 p0, p1, p2;
                split3_bf16(dv[q], p0, p1, p2);
                tp[q * gs] = p0; tp[q * gs + pl] = p1; tp[q * gs + 2 * pl] = p2;
            }
        }
    }
    if (tr) a.trace[1] = wall_clock64();
}
template <int CH>
void launch_bwd2s(const BwdArgs& a, dim3 grid, hipStream_t s) {
    if constexpr (CH % 2 == 0) hipLaunchKernelGGL(lstm_bwd_step2s_kernel<CH>, grid, dim3(1024), 0, s, a);
}

// Mid-batch kernels when the batch is >= 256 rows (and the tiled form was not chosen).  ARCVAE_STEP_TILE=22 forces
// them, 0 disables them together with the tiled kernels.
inline bool choose_step2(int B) {
    const int force = arcvae_env_int("ARCVAE_STEP_TILE", -1);
    if (force == 22) return true;
    if (force >= 0) return false;
    return B >= 256;
}


// ---- persistent forward sweep (latency regime) ---------------------------------------------------------------------
// One launch for the whole forward wavefront instead of T + L - 1 dependent launches.  What makes it cheaper than the
// launches (which pay a ~1.6 us seam plus a cold L2 -- the L2s are invalidated at every kernel boundary -- per step)
// is the PARTITION: the batch rows are split over the 8 XCDs, so a row's recurrence lives in ONE XCD from the first
// step to the last.  h_{t-1} is then produced and consumed under one L2: stores are write-through to that L2, agent-scope
// (sc1) loads read it there (ps_load_sc1*: the protocol's assumptions are listed at that helper), no fence, no
// cross-XCD traffic.  The 32 CUs of an XCD split the 4H gate columns, keep their
// slice of every weight matrix stationary in LDS (default shape: 3 x 32 KB), and meet once per tick at a barrier that
// is one 128-B flag line in their own L2 (0.44 us per tick measured; a chip-wide barrier costs 4-7).
// tools/probe_persist.hip: 2.67 us per tick for the default shape against 4.85 us per launch.
//
// Block roles are taken at run time: a block reads its XCC id and draws a slot from that XCD's counter (the dispatcher
// deals blocks round-robin over the XCDs; with one block per CU -- the LDS footprint forces that -- every XCD receives
// exactly 32).  Every spin is bounded; a block that gives up raises sync[PS_ERR] and all blocks drain.
// Shapes: H = 128 * NT (NT = 1..3), L <= 4, ceil(B / 8) <= 16 rows per XCD, weight slices within LDS.
constexpr int PS_FLAGS = 0, PS_CNT = 256, PS_WORDS = 272;   // re-armed (zeroed) before every sweep: flags, role counters
// (PS_ERR = 500, the sticky error word outside the re-armed ranges: xcd.h)
constexpr int PS_BWD = 512;                                  // the BPTT sweeps' flags / role counters live at sync_ws + PS_BWD (their own
                                                             // words: both sweeps of a step can be re-armed by ONE zero fill ahead of the
                                                             // forward, arcvae_enc_prologue); the error word is shared (buffer: 1024 words)
// Two-group form (NG = 2: up to 32 rows per XCD as two independent 16-row recurrences, two blocks per CU): its own words
// behind the one-group regions -- forward flags [g][xcc][32] and role counters [g][xcc]; the same for the BPTT sweep with a
// fresh set of role counters per chunk; and one table of per-CU arrival counters (8 XCDs x 256 CU keys) shared by every
// launch of a step (a block's group = parity of its arrival on its CU; each launch adds two arrivals per CU).
constexpr int PS2_FWD_FLAGS = 1024, PS2_FWD_CNT = 1536, PS2_BWD_FLAGS = 1552, PS2_BWD_CNT = 2064, PS2_CU = 2304;
constexpr int PS2_WORDS = PS2_CU + 8 * 256;                  // 4352: what a two-group step re-arms (arcvae_enc_prologue)
// "gathered" words of the reduce-scatter BPTT sweep's single-buffered exchange: [NG][8 XCDs][32], the tick whose partial sums a
// CU has finished reading (lstm_bwd_persist_rs_kernel, PersistRsArgs::parmask = 0)
constexpr int PS3_DONE = PS2_WORDS, PS3_WORDS = PS3_DONE + 512;   // 4864
struct PersistArgs {
    const int32_t* x_tb;
    const float* table0;
    const float* bias[ARCVAE_MAX_LAYERS];   // l >= 1
    const float* W[7];                       // the row-major weights [4H,H] themselves: Wh_l at l, Wx_l (l >= 1) at L+l-1 (L <= 4)
    float* hseq;
    float* cseq;
    float* gseq;
    float* comb;                             // or null: [B,2H], its first H columns receive h_{T-1} of the top layer (heads' input)
    unsigned* sync;                          // PS_WORDS words, zeroed before the launch
    unsigned* flags;                         // [NG][8 XCDs][32] published-tick words (one 128-B line per XCD and group)
    unsigned* cnt;                           // [NG][8] role counters
    unsigned* cucnt;                         // NG = 2: [8][256] arrivals per (XCD, CU key), or null (groups by arrival order)
    unsigned* start_signal;                  // or null: += 1 once, when the sweep starts
    unsigned long long* trace;               // or null: {start, end} per tick of block (xcc 0, role 0)
    int B, T, H, V, RX, prio;
    int stagger;                             // two-group form: group 1 starts this many 10 ns units late (see ps_stagger)
};

// Two blocks of a CU that run the same tick loop fall into lockstep: both reach their matrix work together (and halve each
// other's rate) and both wait for their flag lines together, so nothing hides behind anything.  Starting the second group
// about half a tick late puts one group's exchange under the other's matrix work; the offset then holds by itself (a group
// whose partner is in its wait phase finds the matrix pipe free and keeps its phase).
__device__ __forceinline__ void ps_stagger(unsigned grp, int units) {
    if (grp != 0 && units > 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)units) __builtin_amdgcn_s_sleep(8);
    }
}

typedef short s16x4_l __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_l __attribute__((ext_vector_type(2)));
// four f32 -> four packed bf16 (round to nearest even, two v_cvt_pk_bf16_f32): the operand of v_mfma_f32_4x4x4_16b_bf16
__device__ __forceinline__ s16x4_l pk4_bf16(f32x4 v) {
    const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_l{v.x, v.y}, bf16x2_l));
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_l{v.z, v.w}, bf16x2_l));
    return __builtin_bit_cast(s16x4_l, u32x2_l{lo, hi});
}

constexpr bool persist_wreg(int NT, int LL) { return (2 * LL - 1) * (2 * NT) * NT <= 32; }

// Diagnostic build only (-DARCVAE_PS_STAMPS, tools/tick_stamps.py; never the shipped library): eight stamps per tick of one
// block (100 MHz wall clock) at trace[8 s + k] instead of the {start, end} pair at trace[2 s ..]: where a tick's time goes.
#ifdef ARCVAE_PS_STAMPS
#define PS_STAMP(k) do { if (tr) a.trace[8 * (long)s + (k)] = wall_clock64(); } while (0)
#define PS_STAMP_START() PS_STAMP(1)
#define PS_STAMP_END() PS_STAMP(7)
#else
#define PS_STAMP(k) do { } while (0)
#define PS_STAMP_START() do { if (tr) a.trace[2 * s] = wall_clock64(); } while (0)
#define PS_STAMP_END() do { if (tr) a.trace[2 * s + 1] = wall_clock64(); } while (0)
#endif

// Role of a block in a persistent sweep: (group, slot) on its XCD.  NG = 1: slot = arrival order on the XCD.  NG = 2 (two
// blocks per CU, the XCD's rows as two independent 16-row recurrences): the two blocks of a CU should serve DIFFERENT groups
// -- then one group's exchange latency (store acknowledgement, flag line, gather) sits under the other group's matrix work
// on every CU, scheduled by the hardware between the SIMD's two waves.  Placement is speed only: any split of the XCD's 64
// blocks into 32 + 32 is correct.  cucnt != null: arrivals per CU (key = the se / sh / cu fields of HW_ID), parity picks the
// group and a full group sends the block to the other one; cucnt == null: the first 32 arrivals of the XCD take group 0.
template <int NG>
__device__ __forceinline__ void ps_take_role(unsigned* cnt, unsigned* cucnt, unsigned xcc, unsigned& grp, unsigned& role) {
    if constexpr (NG == 1) {
        grp = 0;
        role = atomicAdd(cnt + (xcc & 7), 1u);
    } else {
        unsigned g, r;
        if (cucnt) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g = atomicAdd(cucnt + (xcc & 7) * 256 + ((hwid >> 8) & 0xff), 1u) & 1u;
            r = atomicAdd(cnt + g * 8 + (xcc & 7), 1u);
            if (r >= 32) { g ^= 1u; r = atomicAdd(cnt + g * 8 + (xcc & 7), 1u); }
        } else {
            g = 0;
            r = atomicAdd(cnt + (xcc & 7), 1u);
            if (r >= 32) { g = 1; r = atomicAdd(cnt + 8 + (xcc & 7), 1u); }
        }
        grp = g; role = r;
    }
}

// MF = 1 (H = 256, at most 8 rows per XCD, L <= 2): the contraction on v_mfma_f32_4x4x1 blocks instead of 16x16x4 tiles.
// With 8 rows a 16-row tile is half empty: 96 instructions of 32 cycles per wave and tick, 1.28 us of the tick, for
// half that much useful work.  Blocks = 2 row groups x 8 column groups: ONE 4x4x1 instruction (8 cycles) is the rank-1
// update of the CU's whole 8 x 32 output for one k -- 192 instructions per wave and tick, 0.64 us.  h is the A operand:
// a lane loads 4 consecutive k of its row (block cg holds k = 4cg .. 4cg+3), and the instruction's A-broadcast
// (cbsz = 3, abid = source block) hands one block's values to the 8 blocks of its row group, so the operand is loaded
// once, without padded rows (2 16-byte loads per source instead of 4).  Weights are the B operand, 192 VGPRs.
// NG = 2: the two-group form -- 512 blocks, two per CU (the register budget of launch bounds (256, 2) and an LDS floor that
// admits exactly two), every XCD's rows as two independent recurrences of up to 16 rows with their own flag lines
// (ps_take_role); a block is the RT = 1 kernel on its group's rows.
template <int NT, int LL, int RT, int MF = 0, int NG = 1>   // RT = 16-row MFMA tiles per XCD (rows per XCD RX <= 16 * RT)
__global__ __launch_bounds__(256, NG) void lstm_fwd_persist_kernel(PersistArgs a) {
    static_assert(MF == 0 || (NT == 2 && RT == 1 && LL <= 2), "4x4x1 form: H = 256, one row tile, L <= 2");
    static_assert(NG == 1 || (RT == 1 && MF == 0), "two groups: one 16-row tile per group");
    // MF = 2 (throughput mode, ARCVAE_PERSIST_BF16): the same blocks on v_mfma_f32_4x4x4_16b_bf16 -- a lane's four consecutive
    // k (one 16-byte load of h, one float4 of its weight column) are ONE instruction instead of four: 48 instead of 192
    // matrix instructions per wave and tick, weights in 96 instead of 192 VGPRs (packed bf16), h and the weights rounded
    // to bf16 (round to nearest even) on their way into the instruction; accumulators, gates, c and the stored h stay f32.
    constexpr int CW = 16 * NT;          // gate columns per CU
    constexpr int UW = 4 * NT;           // hidden units per CU
    constexpr int NCH = 8 * NT;          // 16-wide k-chunks of a source (H / 16)
    constexpr int CHW = NCH / 4;         // chunks per wave (K split over the 4 waves)
    constexpr int S = 2 * LL - 1;        // weight matrices
    constexpr int TPL = 256 / LL;        // epilogue threads per layer
    constexpr int MAXP = (16 * RT * UW + TPL - 1) / TPL;
    // Where the stationary weights live: in REGISTERS when a wave's share (its K quarter of every source:
    // S * CHW * NT float4 per lane) is at most 32 float4 = 128 VGPRs -- the default shape needs 24 -- else in LDS.
    // Registers take the 96 KB of LDS reads per tick off the LDS pipe (which the GEMM blocks sharing the CU also use):
    // 1.400 -> 1.382 ms per step at the default shape (A/B on one box).
    constexpr bool WREG = (persist_wreg(NT, LL) && RT == 1) || MF;   // two row tiles: 288 VGPRs and 0.8 % slower than LDS
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                 // [S][NCH][CW][16]   (LDS variant only)
    float* red = WREG ? lds : lds + S * NCH * CW * 16;   // [4 waves][LL][16*RT][CW]
    constexpr int RW = LL * 16 * RT * CW;            // floats per wave in red
    __shared__ unsigned s_role, s_xcc, s_ok, s_grp;
    arcvae_set_prio(a.prio);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, T = a.T, H = a.H, G = 4 * a.H;
    if (tid == 0) {
        s_xcc = ps_xcc_id();
        unsigned g_, r_;
        ps_take_role<NG>(a.cnt, a.cucnt, s_xcc, g_, r_);
        s_grp = g_; s_role = r_;
        s_ok = 1;
        if (blockIdx.x == 0 && a.start_signal)
            __hip_atomic_fetch_add(a.start_signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned xcc = __builtin_amdgcn_readfirstlane(s_xcc), role = __builtin_amdgcn_readfirstlane(s_role);   // wave-uniform (SGPRs)
    const unsigned grp = NG == 1 ? 0u : __builtin_amdgcn_readfirstlane(s_grp);
    if (xcc >= 8 || role >= 32) {                    // no slot on this XCD: the others will time out and drain
        if (tid == 0) atomicAdd(a.sync + PS_ERR, 1u);
        return;
    }
    // rows of this block: the XCD's a.RX rows, or (two groups) the group's up to 16 of them (possibly none: the block then
    // only keeps the protocol going -- its loads repeat row 0 of the XCD, its epilogue owns no pair)
    const int RX = NG == 1 ? a.RX : max(0, min(16, a.RX - 16 * (int)grp));
    const bool tr = a.trace && xcc == 0 && role == 0 && grp == 0 && tid == 0;
    // stationary weights: permuted rows [role*CW, +CW) of every source, read ONCE per launch straight from the row-major
    // parameters (no k-chunk-major copy, i.e. no re-layout launch in front of the sweep: round 2).  Permuted row rp of a
    // source = unit 4(rp >> 4) + (rp & 3), gate (rp >> 2) & 3: the 16 gate columns of 4 units are consecutive.
    auto wsrc = [&](int si, int kc, int rp, int ko) -> const float* {
        return a.W[si] + (long)(((rp >> 2) & 3) * H + (rp >> 4) * 4 + (rp & 3)) * H + 16 * kc + ko;
    };
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int rg = lane >> 5, cg = (lane >> 2) & 7, ij = lane & 3;   // 4x4x1 blocks: (row group, column group), index in block
    f32x4 wr[(WREG && !MF) ? S : 1][(WREG && !MF) ? CHW : 1][(WREG && !MF) ? NT : 1];
    f32x4 wq[MF == 1 ? S : 1][MF == 1 ? CHW : 1][MF == 1 ? 4 : 1];   // [source][16-wide k chunk of my quarter][4-group]: W[k..k+3][4cg + ij]
    s16x4_l wqb[MF == 2 ? S : 1][MF == 2 ? CHW : 1][MF == 2 ? 4 : 1];   // the same, packed bf16
    {
        if constexpr (MF == 2) {
#pragma unroll
            for (int si = 0; si < S; ++si)
#pragma unroll
                for (int c = 0; c < CHW; ++c)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        wqb[si][c][g4] = pk4_bf16(*reinterpret_cast<const f32x4*>(wsrc(si, wave * CHW + c, role * CW + 4 * cg + ij, 4 * g4)));
        } else if constexpr (MF == 1) {
#pragma unroll
            for (int si = 0; si < S; ++si)
#pragma unroll
                for (int c = 0; c < CHW; ++c)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        wq[si][c][g4] = *reinterpret_cast<const f32x4*>(wsrc(si, wave * CHW + c, role * CW + 4 * cg + ij, 4 * g4));
        } else if constexpr (WREG) {
#pragma unroll
            for (int si = 0; si < S; ++si)
#pragma unroll
                for (int c = 0; c < CHW; ++c)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        wr[si][c][n] = *reinterpret_cast<const f32x4*>(wsrc(si, wave * CHW + c, role * CW + 16 * n + r, q4));
        } else {
            for (int i = tid; i < S * NCH * CW * 4; i += 256) {
                const int si = i / (NCH * CW * 4), rem = i - si * (NCH * CW * 4);
                const int kc = rem / (CW * 4), c4 = rem - kc * (CW * 4);
                reinterpret_cast<float4*>(wl)[i] = *reinterpret_cast<const float4*>(wsrc(si, kc, role * CW + (c4 >> 2), (c4 & 3) * 4));
            }
        }
    }
    __syncthreads();
    const int row0 = xcc * a.RX + 16 * (int)grp;
    int arow[RT];                                               // tile rows beyond this XCD's rows repeat the last one
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = min(row0 + max(min(16 * rt + r, RX - 1), 0), B - 1);
    const int mrow = min(row0 + max(min(4 * rg + ij, RX - 1), 0), B - 1);   // 4x4x1 form: the row of this lane's A values
    // epilogue ownership: layer el, pairs p = tl + i*TPL of the RX x UW (row, unit) pairs of this block
    const int el = min(tid / TPL, LL - 1), tl = tid - el * TPL;
    const bool eactive = tid < LL * TPL;
    float cst[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) cst[i] = 0.f;
    const long sH = (long)B * H, sG = (long)B * G, lH = (long)T * sH, lG = (long)T * sG;
    unsigned* my_flag = a.flags + (grp * 8 + xcc) * 32 + role;
    const unsigned* xflags = a.flags + (grp * 8 + xcc) * 32;
    if constexpr (NG == 2) ps_stagger(grp, a.stagger);

    for (int s = 0; s < T + LL - 1; ++s) {
        PS_STAMP(0);
        // The additive terms of this tick's gates (layer 0: token -> row of table0, a dependent pair of loads; else the
        // bias) do not depend on the previous tick: request them BEFORE waiting at the barrier.
        float pv[MAXP][4];
        {
            const int t = s - el;
            if (eactive && t >= 0 && t < T) {
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int p = tl + i * TPL;
                    if (p < RX * UW) {
                        const int erow = p / UW, ul = p - erow * UW;
                        const int b = min(row0 + erow, B - 1);
                        const int unit = role * UW + ul;
                        const float* pre;
                        if (el == 0) {
                            int tk = a.x_tb[(long)t * B + b];
                            tk = min(max(tk, 0), a.V - 1);
                            pre = a.table0 + (long)tk * G;
                        } else {
                            pre = a.bias[el];
                        }
                        pv[i][0] = pre[unit]; pv[i][1] = pre[H + unit]; pv[i][2] = pre[2 * H + unit]; pv[i][3] = pre[3 * H + unit];
                    }
                }
            }
        }
        if (s > 0) {   // every CU of my XCD has published tick s-1
            // (polled by wave 1: waves 0 and 2 run the cell epilogues and have just requested next tick's gate terms -- a
            // dependent token -> table-row pair of loads; the flag loads of a wave queue behind ITS older loads.  Forward
            // sweep alone 419 -> 398 us at the default shape)
#ifndef ARCVAE_FWD_POLL_WAVE
#define ARCVAE_FWD_POLL_WAVE 1
#endif
            if (wave == ARCVAE_FWD_POLL_WAVE) {
                unsigned spins = 0;
                while (true) {
                    const unsigned v = (lane < 32) ? __hip_atomic_load(xflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : (unsigned)s;
                    if (__all((int)(v - (unsigned)s) >= 0)) break;
                    if (++spins > 4000000u || __hip_atomic_load(a.sync + PS_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        if (lane == 0) { atomicAdd(a.sync + PS_ERR, 1u); s_ok = 0; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (!s_ok) return;
        }
        PS_STAMP_START();
        if constexpr (MF != 0) {
            // ---- A operands: 4 consecutive k of my row per load; block cg of my row group holds k = 64w + 32m + 4cg + e
            f32x4 qx[LL][2], qh[LL][2];
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                const int t = s - l;
                const bool act = t >= 0 && t < T;
                const unsigned off = (unsigned)(((long)mrow * H + 64 * wave + 4 * cg) * 4);
                // (All 32 CUs of the XCD read the same 24 KB per tick and 32 readers per line queue at that line's L2 channel:
                // 0.80 us from issue to landed.  Measured and dropped: the exchanged h ALSO stored in 2 / 4 / 8 replicas, a CU
                // reading replica role % R -- the loads land in 0.56-0.60 us, the extra stores give it back: 0.982 / 0.992 /
                // 1.014 ms per step against 0.983.)
                if (act && l > 0) {
                    const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.hseq + (l - 1) * lH + (long)t * sH, sH * 4);
                    qx[l][0] = ps_load_sc1_x4(rs, off);
                    qx[l][1] = ps_load_sc1_x4(rs, off + 128);
                }
                if (act && t > 0) {
                    const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.hseq + l * lH + (long)(t - 1) * sH, sH * 4);
                    qh[l][0] = ps_load_sc1_x4(rs, off);
                    qh[l][1] = ps_load_sc1_x4(rs, off + 128);
                }
            }
            PS_STAMP(2);
#ifdef ARCVAE_PS_STAMPS   // diagnostic build: when have the operand loads LANDED (forces the wait; slot 4 = the barrier otherwise)
            if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            PS_STAMP(4);
#endif
            // ---- rank-1 updates: for every k of my quarter, the A values of block (k >> 2) & 7 broadcast to its row group
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                const int t = s - l;
                if (t < 0 || t >= T) continue;            // block-uniform
                f32x4 acc[4];                              // four independent chains (one per k mod 4; bf16 form: two, per source block parity)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (MF == 2) {
#define PS_B1(QB, SIDX, M_, AB_) acc[(AB_) & 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(QB, wqb[MF == 2 ? (SIDX) : 0][MF == 2 ? 2 * M_ + (AB_ >> 2) : 0][MF == 2 ? (AB_ & 3) : 0], acc[(AB_) & 1], 3, AB_, 0);
#define PS_B8(Q, SIDX, M_) { const s16x4_l qb_ = pk4_bf16(Q[l][M_]);                                                  \
                             PS_B1(qb_, SIDX, M_, 0) PS_B1(qb_, SIDX, M_, 1) PS_B1(qb_, SIDX, M_, 2) PS_B1(qb_, SIDX, M_, 3) \
                             PS_B1(qb_, SIDX, M_, 4) PS_B1(qb_, SIDX, M_, 5) PS_B1(qb_, SIDX, M_, 6) PS_B1(qb_, SIDX, M_, 7) }
                    if (l > 0) { PS_B8(qx, LL + l - 1, 0) PS_B8(qx, LL + l - 1, 1) }
                    if (t > 0) { PS_B8(qh, l, 0) PS_B8(qh, l, 1) }
#undef PS_B8
#undef PS_B1
                } else {
#define PS_Q1(Q, SIDX, M_, AB_)                                                                                          \
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(Q[l][M_].x, wq[MF == 1 ? (SIDX) : 0][MF == 1 ? 2 * M_ + (AB_ >> 2) : 0][MF == 1 ? (AB_ & 3) : 0].x, acc[0], 3, AB_, 0); \
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(Q[l][M_].y, wq[MF == 1 ? (SIDX) : 0][MF == 1 ? 2 * M_ + (AB_ >> 2) : 0][MF == 1 ? (AB_ & 3) : 0].y, acc[1], 3, AB_, 0); \
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(Q[l][M_].z, wq[MF == 1 ? (SIDX) : 0][MF == 1 ? 2 * M_ + (AB_ >> 2) : 0][MF == 1 ? (AB_ & 3) : 0].z, acc[2], 3, AB_, 0); \
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(Q[l][M_].w, wq[MF == 1 ? (SIDX) : 0][MF == 1 ? 2 * M_ + (AB_ >> 2) : 0][MF == 1 ? (AB_ & 3) : 0].w, acc[3], 3, AB_, 0);
#define PS_Q8(Q, SIDX, M_) PS_Q1(Q, SIDX, M_, 0) PS_Q1(Q, SIDX, M_, 1) PS_Q1(Q, SIDX, M_, 2) PS_Q1(Q, SIDX, M_, 3) \
                           PS_Q1(Q, SIDX, M_, 4) PS_Q1(Q, SIDX, M_, 5) PS_Q1(Q, SIDX, M_, 6) PS_Q1(Q, SIDX, M_, 7)
                if (l > 0) { PS_Q8(qx, LL + l - 1, 0) PS_Q8(qx, LL + l - 1, 1) }
                if (t > 0) { PS_Q8(qh, l, 0) PS_Q8(qh, l, 1) }
#undef PS_Q8
#undef PS_Q1
                }
                // lane (rg, cg, ij) holds rows 4rg + i (register i) of column 4cg + ij
                float* rp = red + wave * RW + l * 16 * RT * CW + (4 * rg) * CW + 4 * cg + ij;
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[i * CW] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
            }
        } else {
        // ---- A operands: rows of my XCD from the slabs the previous tick wrote (L1-bypassing loads)
        f32x4 fx[LL][RT][CHW], fh[LL][RT][CHW];
#pragma unroll
        for (int l = 0; l < LL; ++l) {
            const int t = s - l;
            const bool act = t >= 0 && t < T;
            // (tile rows beyond this XCD's RX rows repeat its last row; masking those loads off was measured: slower)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned off = (unsigned)(((long)arow[rt] * H + wave * CHW * 16 + q4) * 4);
                if (act && l > 0) {
                    const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.hseq + (l - 1) * lH + (long)t * sH, sH * 4);
#pragma unroll
                    for (int c = 0; c < CHW; ++c) fx[l][rt][c] = ps_load_sc1_x4(rs, off + c * 64);
                }
                if (act && t > 0) {
                    const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.hseq + l * lH + (long)(t - 1) * sH, sH * 4);
#pragma unroll
                    for (int c = 0; c < CHW; ++c) fh[l][rt][c] = ps_load_sc1_x4(rs, off + c * 64);
                }
            }
        }
        PS_STAMP(2);
        // ---- MFMA: pre-activations of my CW columns, K quarter of this wave, weights from LDS
#pragma unroll
        for (int l = 0; l < LL; ++l) {
            const int t = s - l;
            if (t < 0 || t >= T) continue;            // block-uniform
            f32x4 acc[RT][NT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#define PS_SRC(FR, SIDX)                                                                                         \
            _Pragma("unroll") for (int c = 0; c < CHW; ++c) {                                                    \
                const int kc = wave * CHW + c;                                                                   \
                _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                 \
                    f32x4 w;                                                                                      \
                    if constexpr (WREG) w = wr[WREG ? (SIDX) : 0][WREG ? c : 0][WREG ? n : 0];                     \
                    else w = *reinterpret_cast<const f32x4*>(wl + (((SIDX) * NCH + kc) * CW + 16 * n + r) * 16 + q4); \
                    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                          \
                        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(FR[l][rt][c].x, w.x, acc[rt][n], 0, 0, 0); \
                        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(FR[l][rt][c].y, w.y, acc[rt][n], 0, 0, 0); \
                        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(FR[l][rt][c].z, w.z, acc[rt][n], 0, 0, 0); \
                        acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(FR[l][rt][c].w, w.w, acc[rt][n], 0, 0, 0); \
                    }                                                                                            \
                }                                                                                                \
            }
            if (l > 0) { PS_SRC(fx, LL + l - 1) }
            if (t > 0) { PS_SRC(fh, l) }
#undef PS_SRC
            float* rp = red + wave * RW + l * 16 * RT * CW;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        rp[(16 * rt + (lane >> 4) * 4 + reg) * CW + 16 * n + r] = acc[rt][n][reg];
        }
        }
        PS_STAMP(3);
        __syncthreads();
#ifndef ARCVAE_PS_STAMPS
        PS_STAMP(4);
#endif
        // ---- cell update of my (row, unit) pairs; c stays in a register from tick to tick.  (Round 3, measured and dropped: a
        // pair on TWO adjacent lanes -- i, f and c on one, g, o and h on the other, the activation chain 2 transcendentals +
        // tanh(c) deep instead of 5, all four waves busy: the forward sweep ALONE ran 398.7 -> 390.8 us, the step stayed at
        // 0.970-0.972 ms: in the step the sweep's pace is set by the decoder's GEMMs beside it.)  (Storing the saved gates
        // and c after the flag, under the next tick's barrier wait, was measured twice: no gain without the store wait
        // below, 0.5 % slower with it -- 1.083-1.085 vs 1.077-1.078 ms per step.)
        const int te = s - el;
        if (eactive && te >= 0 && te < T) {
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int p = tl + i * TPL;
                if (p >= RX * UW) break;
                const int erow = p / UW, ul = p - erow * UW;
                const int b = row0 + erow;
                if (b >= B) continue;
                const int unit = role * UW + ul;
                const int cb = erow * CW + 16 * (ul >> 2) + (ul & 3);
                float v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = el * 16 * RT * CW + cb + 4 * g;
                    v[g] = (red[o] + red[RW + o]) + (red[2 * RW + o] + red[3 * RW + o]);
                }
                const float gi = chain_sigmoid(v[0] + pv[i][0]);
                const float gf = chain_sigmoid(v[1] + pv[i][1]);
                const float gg = chain_tanh(v[2] + pv[i][2]);
                const float go = chain_sigmoid(v[3] + pv[i][3]);
                const float c = te > 0 ? gf * cst[i] + gi * gg : gi * gg;   // MLX: cell=None at t == 0 -> c = i*g
                cst[i] = c;
                const long hb = (long)b * H + unit;
                const float hval = go * chain_tanh(c);
                a.hseq[el * lH + (long)te * sH + hb] = hval;
                if (a.comb && el == LL - 1 && te == T - 1) a.comb[(long)b * 2 * H + unit] = hval;
                float* gp = a.gseq + el * lG + (long)te * sG + (long)b * G + unit;
                gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
                a.cseq[el * lH + (long)te * sH + hb] = c;
            }
        }
        PS_STAMP(5);
        ps_stores_in_l2();                                       // my h stores have reached the XCD's L2
        PS_STAMP(6);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PS_STAMP_END();
    }
}

template <int NT, int LL, int RT, int MF = 0, int NG = 1>
void launch_persist(const PersistArgs& a, size_t lds, hipStream_t s) {
    // > 64 KB of dynamic LDS has to be allowed per kernel; set on every call (idempotent, no host state kept)
    (void)hipFuncSetAttribute((const void*)lstm_fwd_persist_kernel<NT, LL, RT, MF, NG>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL((lstm_fwd_persist_kernel<NT, LL, RT, MF, NG>), dim3(256 * NG), dim3(256), lds, s, a);
}
// LDS allocation of a two-group block (ARCVAE_PERSIST2_LDS_KB, default 56): more than a third of a CU's 160 KB and at most
// half of it, so that a CU admits exactly two of them
inline size_t persist2_lds_bytes() {
    int kb = arcvae_env_int("ARCVAE_PERSIST2_LDS_KB", 56);
    kb = kb < 54 ? 54 : (kb > 80 ? 80 : kb);
    return (size_t)kb * 1024;
}
// Two-group form of the persistent sweeps: H = 256, L <= 2, 17..32 rows per XCD (129..256 rows per GPU: the 256-row shard of
// BASELINE.json configs[3]).  ARCVAE_PERSIST2 (bit 0 forward, bit 1 BPTT; 0 keeps the one-group kernels).
inline bool persist_two_groups(int B, int H, int L, int which /* 1 forward sweep, 2 BPTT sweep, 3 either */) {
    return (arcvae_env_int("ARCVAE_PERSIST2", 1) & which) != 0 && H == 256 && L >= 1 && L <= 2 && ceil_div(B, 8) > 16 && ceil_div(B, 8) <= 32;
}
inline int persist_row_tiles(int B) { return ceil_div(B, 8) > 16 ? 2 : 1; }
// LDS allocation floor of the register-stationary persistent kernels (ARCVAE_PERSIST_LDS_KB, default 81 = more than half
// of a CU's 160 KB): two of their blocks can then never share a CU, whatever else is resident.
inline size_t persist_lds_floor() { return (size_t)arcvae_env_int("ARCVAE_PERSIST_LDS_KB", 81) * 1024; }
inline size_t persist_lds_bytes(int B, int H, int L) {
    const int NT = H / 128, CW = 16 * NT;
    const size_t red = sizeof(float) * (size_t)4 * L * 16 * persist_row_tiles(B) * CW;
    if (persist_wreg(NT, L) && persist_row_tiles(B) == 1)   // weights in registers: only the reduction buffer is used, but the allocation stays above
        return red > persist_lds_floor() ? red : persist_lds_floor();   // half the CU's LDS so that the blocks land one per CU
    return sizeof(float) * (size_t)(2 * L - 1) * (H / 16) * CW * 16 + red;
}
// up to 32 rows per XCD (two MFMA row tiles): B <= 256, i.e. also the 256-row shard of BASELINE.json configs[3]
inline bool persist_shape_ok(int B, int T, int H, int L) {
    if (arcvae_env_int("ARCVAE_PERSIST", 1) == 0) return false;
    if (H % 128 != 0 || H / 128 > 3 || L < 1 || L > 4 || B < 1 || T < 1) return false;
    if (ceil_div(B, 8) > 32) return false;
    return persist_lds_bytes(B, H, L) <= 150 * 1024;
}


// ---- persistent BPTT sweep (latency regime) ------------------------------------------------------------------------
// The BPTT wavefront with the same row partition as lstm_fwd_persist_kernel: XCD g owns rows [g*RX, (g+1)*RX), its 32
// CUs split the H hidden units, the W^T slices are stationary in LDS, the per-tick barrier is the XCD's flag line.
// One launch covers a CHUNK of the sweep ([s_begin, s_end) of the T + 2(L-1) ticks of arcvae_enc_lstm_backward's
// schedule): the kernel boundary between chunks is what writes the gate gradients back from the XCD's L2 for the
// weight-gradient GEMMs that overlap the sweep on other XCDs, and what carries the chunk signal.  State that crosses a
// tick stays on chip (dc*f of a (layer, row, unit) in a register of the thread that owns it, dX in LDS); it is also
// written to the dcs / dxs rings every tick so that the next chunk can pick it up.
// Slots (fixed thread groups of 64): cell(l) -> LL-1-l, xproj(l) -> LL + l.  L <= 2, RX * UW <= 64 pairs per slot.
struct PersistBwdArgs {
    const float* wT;          // k-chunk-major transposed weights [(2L-1)][4H/16][H][16]: WhT_l at l, WxT_{l+1} at L + l
    const float* cseq;
    const float* gseq;        // saved gates (dG may alias it)
    const float* dh_top;
    float* dG;
    float* dcs;               // [L][RS][B][H] ring
    float* dxs;               // [L][RS][B][H] ring
    unsigned* sync;           // PS_WORDS words; flags hold the global tick index, role counters per chunk launch
    unsigned* flags;          // [NG][8 XCDs][32] published-tick words of this sweep
    unsigned* cnt;            // [NG][8] role counters of THIS chunk launch
    unsigned* cucnt;          // two-group form: [8][256] arrivals per (XCD, CU key), or null (ps_take_role)
    unsigned* err;            // the sticky error word (sync_ws + PS_ERR, shared with the forward sweep)
    unsigned* start_signal;
    unsigned long long* trace;
    int B, T, H, RX, RS, ld_dh_top, s_begin, s_end, cnt_off, prio;
    int stagger;              // two-group form: group 1 starts this many 10 ns units late (ps_stagger)
    int row_base, row_xs;     // reduce-scatter sweep: XCD x works on rows row_base + x * row_xs + [0, RX) (default 0, RX; the
                              // half-batch form of round 4: row_xs = 32, row_base = 16 * half, RX = 16)
};

template <int NT, int LL, int RT>   // RT = 16-row MFMA tiles per XCD (rows per XCD RX <= 16 * RT)
__global__ __launch_bounds__(256) void lstm_bwd_persist_kernel(PersistBwdArgs a) {
    constexpr int UW = 4 * NT;           // hidden units per CU
    constexpr int NCHB = 32 * NT;        // 16-wide k-chunks of a source (4H / 16)
    constexpr int CHB = NCHB / 4;        // chunks per wave
    constexpr int S = 2 * LL - 1;        // sources = slots
    constexpr int NH = RT;               // a source's chunks are loaded in NH pieces (register budget: 2 x RT x CHP float4)
    constexpr int CHP = CHB / NH;        // chunks per piece
    constexpr int NP = 16 * RT * UW;     // (row, unit) pairs of a slot
    constexpr int MAXP = (NP + 63) / 64; // pairs per thread of a slot
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                  // [S][NCHB][UW][16]
    float* red = wl + S * NCHB * UW * 16;             // [4 waves][S][16*RT][16]
    float* dxl = red + 4 * S * 256 * RT;              // [2][LL][NP]  dX of the previous / this tick
    __shared__ unsigned s_role, s_xcc, s_ok;
    arcvae_set_prio(a.prio);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, T = a.T, H = a.H, G = 4 * a.H, RX = a.RX, RS = a.RS;
    if (tid == 0) {
        s_xcc = ps_xcc_id();
        s_role = atomicAdd(a.cnt + (s_xcc & 7), 1u);
        s_ok = 1;
        if (blockIdx.x == 0 && a.start_signal)
            __hip_atomic_fetch_add(a.start_signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned xcc = __builtin_amdgcn_readfirstlane(s_xcc), role = __builtin_amdgcn_readfirstlane(s_role);   // wave-uniform (SGPRs)
    if (xcc >= 8 || role >= 32) {
        if (tid == 0) atomicAdd(a.err, 1u);
        return;
    }
    const bool tr = a.trace && xcc == 0 && role == 0 && tid == 0;
    {
        const long wsz = (long)H * G;
        for (int i = tid; i < S * NCHB * UW * 4; i += 256) {
            const int sidx = i / (NCHB * UW * 4), rem = i - sidx * (NCHB * UW * 4);
            const int kc = rem / (UW * 4), c4 = rem - kc * (UW * 4);
            // slot -> weight matrix: cell(l) (slot LL-1-l) uses WhT_l, xproj(l) (slot LL+l) uses WxT_{l+1} at LL + l
            const int widx = sidx < LL ? (LL - 1 - sidx) : sidx;
            reinterpret_cast<float4*>(wl)[i] =
                *reinterpret_cast<const float4*>(a.wT + widx * wsz + ((long)kc * H + role * UW) * 16 + c4 * 4);
        }
    }
    __syncthreads();
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int row0 = xcc * RX;
    int arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = min(row0 + min(16 * rt + r, RX - 1), B - 1);
    const int ub = min(r, UW - 1);
    const long sH = (long)B * H, sG = (long)B * G, lH = (long)T * sH, lG = (long)T * sG;
    // epilogue ownership: slot = tid >> 6 (threads of slots >= S idle), pairs p = (tid & 63) + 64 i
    const int slot = tid >> 6, p0 = tid & 63;
    const bool is_cell = slot < LL;
    const int el = is_cell ? (LL - 1 - slot) : (slot - LL);   // layer of the slot
    float dcst[MAXP];                                  // dc_{t+1} * f_{t+1} of my (layer, row, unit) pairs
#pragma unroll
    for (int i = 0; i < MAXP; ++i) dcst[i] = 0.f;
    unsigned* my_flag = a.flags + xcc * 32 + role;
    const unsigned* xflags = a.flags + xcc * 32;

    for (int s = a.s_begin; s < a.s_end; ++s) {
        // this tick's job of my slot
        const int skew = 2 * (LL - 1 - el);
        const int t = is_cell ? T - 1 - (s - skew) : T - 1 - (s + 1 - skew);
        const bool jact = slot < S && t >= 0 && t < T;
        // forward values of the cell epilogue (static): requested before the barrier wait
        float gi[MAXP], gf[MAXP], gg[MAXP], go[MAXP], c_v[MAXP], cprev_v[MAXP], ext_v[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            gi[i] = gf[i] = gg[i] = go[i] = c_v[i] = cprev_v[i] = ext_v[i] = 0.f;
            const int p = p0 + 64 * i;
            const int erow = p / UW, ul = p - erow * UW;
            if (is_cell && jact && p < RX * UW && row0 + erow < B) {
                const int eb = row0 + erow, unit = role * UW + ul;
                const long hb = (long)eb * H + unit;
                const float* gp = a.gseq + el * lG + (long)t * sG + (long)eb * G + unit;
                gi[i] = gp[0]; gf[i] = gp[H]; gg[i] = gp[2 * H]; go[i] = gp[3 * H];
                c_v[i] = a.cseq[el * lH + (long)t * sH + hb];
                if (t > 0) cprev_v[i] = a.cseq[el * lH + (long)(t - 1) * sH + hb];
                if (el == LL - 1 && t == T - 1) ext_v[i] = a.dh_top[(long)eb * a.ld_dh_top + unit];
                if (s == a.s_begin) {   // state of the previous chunk (or none at the very first tick of a layer)
                    dcst[i] = (t < T - 1) ? a.dcs[((long)el * RS + ((t + 1) % RS)) * sH + hb] : 0.f;
                    if (el < LL - 1) ext_v[i] = a.dxs[((long)el * RS + (t % RS)) * sH + hb];
                }
            }
        }
        if (s > 0) {   // every CU of my XCD has published tick s-1 (flags carry the global tick index)
            if (wave == 0) {
                unsigned spins = 0;
                while (true) {
                    const unsigned v = (lane < 32) ? __hip_atomic_load(xflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : (unsigned)s;
                    if (__all((int)(v - (unsigned)s) >= 0)) break;
                    if (++spins > 4000000u || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        if (lane == 0) { atomicAdd(a.err, 1u); s_ok = 0; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (!s_ok) return;
        }
        if (tr) a.trace[2 * s] = wall_clock64();
        // ---- contractions of the S slots, piece by piece, the next piece's operand loads in flight behind the MFMAs
        const float* srcp[S];
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const bool cellj = j < LL;
            const int lj = cellj ? (LL - 1 - j) : (j - LL);
            const int skj = 2 * (LL - 1 - lj);
            const int tj = cellj ? T - 1 - (s - skj) : T - 1 - (s + 1 - skj);
            const float* pj = nullptr;
            if (tj >= 0 && tj < T) {
                if (cellj) { if (tj < T - 1) pj = a.dG + lj * lG + (long)(tj + 1) * sG; }   // dG^l_{t+1}
                else pj = a.dG + (lj + 1) * lG + (long)tj * sG;                                // dG^{l+1}_{tx}
            }
            srcp[j] = pj;
        }
        f32x4 fa[2][RT][CHP];
#define PB_LOAD(BUF, J, HALF)                                                                                  \
        if (srcp[J]) {                                                                                         \
            const __amdgpu_buffer_rsrc_t rs = ps_rsrc(srcp[J], sG * 4);                                        \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                \
                const unsigned off = (unsigned)(((long)arow[rt] * G + (wave * CHB + (HALF) * CHP) * 16 + q4) * 4); \
                _Pragma("unroll") for (int c = 0; c < CHP; ++c)                                                \
                    fa[BUF][rt][c] = ps_load_sc1_x4(rs, off + c * 64);                                         \
            }                                                                                                  \
        }
        PB_LOAD(0, 0, 0)
        f32x4 acc[RT];
#pragma unroll
        for (int k = 0; k < S * NH; ++k) {            // piece k = (source j = k / NH, part h = k % NH)
            const int j = k / NH, h = k % NH;
            if (k + 1 < S * NH) { PB_LOAD((k + 1) & 1, (k + 1 < S * NH ? (k + 1) / NH : j), (k + 1) % NH) }
            if (h == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (srcp[j]) {
#pragma unroll
                for (int c = 0; c < CHP; ++c) {
                    const int kc = wave * CHB + h * CHP + c;
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + ((j * NCHB + kc) * UW + ub) * 16 + q4);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k & 1][rt][c].x, w.x, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k & 1][rt][c].y, w.y, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k & 1][rt][c].z, w.z, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[k & 1][rt][c].w, w.w, acc[rt], 0, 0, 0);
                    }
                }
            }
            if (h == NH - 1) {
                float* rp = red + (wave * S + j) * 256 * RT;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) rp[(16 * rt + (lane >> 4) * 4 + reg) * 16 + r] = acc[rt][reg];
            }
        }
#undef PB_LOAD
        __syncthreads();
        // ---- epilogue of my (slot, row, unit) pairs
        if (jact) {
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int p = p0 + 64 * i;
                const int erow = p / UW, ul = p - erow * UW;
                if (p >= RX * UW || row0 + erow >= B) continue;
                const int eb = row0 + erow, unit = role * UW + ul;
                const long hb = (long)eb * H + unit;
                const int o = slot * 256 * RT + erow * 16 + ul;
                float dh = (red[o] + red[S * 256 * RT + o]) + (red[2 * S * 256 * RT + o] + red[3 * S * 256 * RT + o]);
                if (!is_cell) {                                 // dX_l[tx]: consumed by cell(l, tx) at the next tick
                    dxl[(((s + 1) & 1) * LL + el) * NP + p] = dh;
                    a.dxs[((long)el * RS + (t % RS)) * sH + hb] = dh;
                } else {
                    float ext = ext_v[i];
                    if (el < LL - 1 && s != a.s_begin) ext = dxl[((s & 1) * LL + el) * NP + p];
                    dh += ext;
                    const float tc = tanhf(c_v[i]);
                    const float d_o = dh * tc * go[i] * (1.f - go[i]);
                    const float dc = dh * go[i] * (1.f - tc * tc) + dcst[i];
                    const float d_i = dc * gg[i] * gi[i] * (1.f - gi[i]);
                    const float d_f = t > 0 ? dc * cprev_v[i] * gf[i] * (1.f - gf[i]) : 0.f;
                    const float d_g = dc * gi[i] * (1.f - gg[i] * gg[i]);
                    dcst[i] = dc * gf[i];
                    a.dcs[((long)el * RS + (t % RS)) * sH + hb] = dcst[i];
                    float* dp = a.dG + el * lG + (long)t * sG + (long)eb * G + unit;
                    dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
                }
            }
        }
        ps_stores_in_l2();                                       // my dG stores have reached the XCD's L2
        __syncthreads();
        if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tr) a.trace[2 * s + 1] = wall_clock64();
    }
}


// ---- persistent BPTT sweep, reduce-scatter form (default shape) -------------------------------------------------------
// The forward sweep is cheap per tick because a CU needs 24 KB of operand: it owns 32 GATE columns and the contraction
// runs over the hidden units.  In the BPTT step dh = dG . Wh the contraction runs over the gate columns and the outputs
// are hidden units, so "each CU owns some outputs" (lstm_bwd_persist_kernel) makes every CU pull all of dG: 96 KB per
// tick.  Turn it round: a CU keeps the gate gradients of ITS 32 gate columns (the ones its own epilogue produced one
// tick earlier: they never leave the CU) and multiplies them with its 32 ROWS of Wh -- a partial dh over ALL hidden
// units, K = 32.  The partials are reduce-scattered through the XCD's L2: every CU writes 8 rows x 8 units for each of
// the 32 CUs (laid out so that a consumer's 32 pieces are one contiguous 8 KB), one barrier, every CU sums the 32
// pieces of its own units.  Per tick and source a CU writes 8 KB and reads 8 KB; the MFMA work is 32 instructions
// per wave and source.  Weights: the CU's 32 rows of the ORIGINAL row-major Wh / Wx (no transposed copy), in LDS.
// Shape: H = 256 (32 CUs x 8 units), L <= 2, at most 8 rows per XCD.  Chunks, signals, rings: as lstm_bwd_persist_kernel.
struct PersistRsArgs {
    PersistBwdArgs b;
    const float* W[3];        // row-major [4H,H]: cell(l=1) Wh_1, cell(l=0) Wh_0, xproj(0) Wx_1   (slot order; L = 1: W[0] = Wh_0)
    float* part;              // [2 parity][S][8 XCDs][32 consumers][32 producers][64]  partial sums in flight
    // fused weight gradients (FW variant): accumulated on chip over the launch's ticks, added ("+=") at its end
    const float* hseq;        // [L,T,B,H] hidden states of the forward sweep
    const int32_t* x_tb;      // [T,B] tokens (layer-0 token-table gradient)
    float* dW[3];             // gradient buffers [4H,H] of: Wh_top; then (L = 2) Wx_1, Wh_0
    float* dbias1;            // [4H] gradient of bias_1 (L = 2), else null
    float* dtable;            // [V,4H] token-table gradient workspace
    int V;
    int fw_dbg;               // timing experiments only (ARCVAE_FW_DEBUG): 1 no h loads, 2 no products, 4 no token table
    int parmask;              // 1: the exchange buffers alternate with the tick's parity; 0: ONE buffer, guarded by `done`
    unsigned* done;           // parmask = 0: [NG][8 XCDs][32] "tick whose partial sums this CU has finished gathering" words
};

// MF = 1 (registers only): the contraction on v_mfma_f32_4x4x1 (16 blocks of 4x4, K = 1) instead of 16x16x4.  An XCD owns
// at most 8 batch rows, so a 16-row tile is half empty and the 96 16x16x4 instructions per wave and tick (32 cycles
// each: 1.28 us of a 3.4 us tick) do twice the useful work.  With blocks = 2 row groups x 8 unit groups one 4x4x1
// instruction (8 cycles) covers 8 rows x 32 units for one k without a padded row: 192 instructions, 0.64 us.  Weights
// are the A operand (lane (rg, ug, i): W[k][32ch + 4ug + i], 192 VGPRs), the local gate gradients the B operand
// (lane (rg, ug, j): dG[4rg + j][k], broadcast reads from LDS); a lane ends up with 4 consecutive units of one row.
//
// FW = true: the WEIGHT GRADIENTS ride in the same kernel.  The separate weight-gradient GEMMs were what the step lost most
// on: run beside the sweep they doubled its tick (2.47 -> 4.3-5.2 us over the second chunk: streaming loads in the memory
// queue of every CU that is an endpoint of the exchange), run after it they are the step's tail -- 120 us of a 1.06 ms
// step either way (profiles/r02_wgrad_experiments.txt).  But a CU already HOLDS one operand: the gate gradients of its 32
// gate columns and its XCD's 8 rows, one tick old, in LDS.  dWh_l[my 32 rows, :] += dG_l[t]^T . h_l[t-1] needs only those 8
// rows of h (8 KB per source and tick, prefetched a tick ahead, plain cached loads: nobody waits for them) -- an outer
// product with K = 8 per tick on v_mfma_f32_32x32x2: 24 instructions per wave and tick, issued where the chain leaves the
// matrix pipe idle (behind the flag store while the flags travel, and under the gather's loads), accumulated in 96
// registers per lane over the whole sweep and added to the gradient buffers once, at the end (8 XCDs x 3 sources:
// float atomics, full 128-B rows).  The token-table gradient (layer 0's input side) is an LDS scatter-add by token, the
// bias gradient a per-thread running sum.  No aux-stream GEMMs, no chunks, nothing beside the sweep.
// RG = groups of 8 rows per XCD (rows per XCD RX <= 8 RG: B <= 64 RG; round 2: RG = 2, 4 for 65..256 rows per GPU, e.g.
// the 256-row shard of BASELINE.json configs[3]): a tick walks the groups -- products and partial stores of every
// group, ONE store wait / flag / barrier, then gather and epilogue of every group -- so the per-tick exchange cost is
// paid once for up to 32 rows.
// MF = 0 with RG = 2 ("R16"): up to 16 rows as ONE 16x16x4 row tile -- the products of both 8-row halves in one pass (96
// instructions per wave and tick, no padded rows), each half's partial sums into that half's exchange buffer, gather and
// epilogue per half as in the RG = 2 walk.  NG = 2: the two-group form (see lstm_fwd_persist_kernel): 512 blocks, two per
// CU, the XCD's up to 32 rows as two independent R16 recurrences with their own flag lines and exchange buffers.
// SB = true: the single-buffered exchange (see `single` below; compiled in only where it is used -- its poll and check cost the
// default shape's tick 0.35 us when they were a run-time branch of every variant).
template <int LL, bool WR, int MF, bool FW = false, int RG = 1, int NG = 1, bool SB = false>   // WR: the weight slices live in registers, else in LDS
__global__ __launch_bounds__(256, NG) void lstm_bwd_persist_rs_kernel(PersistRsArgs ar) {
    static_assert(MF == 0 || WR, "the 4x4x1 form keeps its weights in registers");
    static_assert(!FW || MF == 1, "fused weight gradients: 4x4x1 form only");
    static_assert(RG == 1 || (MF != 0 && !FW) || (MF == 0 && (RG == 2 || RG == 4) && WR && !FW), "row groups: 4x4 forms or 16-row tiles; weight gradients by GEMM");
    static_assert(NG == 1 || (MF == 0 && RG == 2), "two groups: the 16-row tile form");
    constexpr bool R16 = MF == 0 && RG >= 2;          // the tile form: RG / 2 tiles of 16 rows (RG = 4: the XCD's 32 rows in one block)
    constexpr int NRT = R16 ? RG / 2 : 1;
    // MF = 2 (throughput mode): the 4x4 blocks on v_mfma_f32_4x4x4_16b_bf16 -- four consecutive gate columns k per
    // instruction (48 instead of 192 per wave and tick), the weight rows packed to bf16 in 96 VGPRs, the local gate
    // gradients rounded to bf16 as they leave LDS; partial sums, exchange and cell epilogue stay f32.
    constexpr int DGL = LL * 16 * 32;                 // floats of one group's gate-gradient image
    const PersistBwdArgs& a = ar.b;
    constexpr int UW = 8, S = 2 * LL - 1, WS = 36;   // WS: padded row stride of the weight image (bank-conflict-free b128 reads)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wloc = lds;                                // [S][256 units][WS]: W[my gate col k][unit], k-contiguous (LDS variant)
    float* dgl = wloc + (WR ? 0 : S * 256 * WS);      // [LL][16 rows][32 gate cols of mine]  (rows >= 8 stay zero)
    float* dxl = dgl + RG * DGL;                      // [RG][2][64]
    float* dtl = dxl + RG * 128;                           // FW: [V][33] token-table gradient of my 32 gate columns
    __shared__ unsigned s_role, s_xcc, s_ok, s_grp;
    arcvae_set_prio(a.prio);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, T = a.T, H = a.H, G = 4 * a.H, RS = a.RS;
    if (tid == 0) {
        s_xcc = ps_xcc_id();
        unsigned g_, r_;
        ps_take_role<NG>(a.cnt, a.cucnt, s_xcc, g_, r_);
        s_grp = g_; s_role = r_;
        s_ok = 1;
        if (blockIdx.x == 0 && a.start_signal)
            __hip_atomic_fetch_add(a.start_signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned xcc = __builtin_amdgcn_readfirstlane(s_xcc), role = __builtin_amdgcn_readfirstlane(s_role);   // wave-uniform (SGPRs)
    const unsigned grp = NG == 1 ? 0u : __builtin_amdgcn_readfirstlane(s_grp);
    if (xcc >= 8 || role >= 32) {
        if (tid == 0) atomicAdd(a.err, 1u);
        return;
    }
    const int RX = NG == 1 ? a.RX : max(0, min(16, a.RX - 16 * (int)grp));   // rows of this block (two groups: the group's)
    const bool tr = a.trace && xcc == 0 && role == 0 && grp == 0 && tid == 0;
    // my 32 rows of every weight matrix as the MFMA's B operand: element (k, unit) = W[gate*H + 8*role + ul][unit] with
    // k = gate * 8 + ul.  Register variant: lane (r, q4) of wave w keeps k = 16c + q4 + e, unit = 64w + 16n + r.
    // LDS variant: transposed image [unit][k]; thread tid takes unit column tid of every row, the 32 loads of a source
    // all in flight before the first LDS store (one load -> store round trip per element made a chunk launch cost 55 us).
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    f32x4 wr[(WR && !MF) ? S : 1][(WR && !MF) ? 4 : 1][(WR && !MF) ? 2 : 1];
    float wq[MF == 1 ? S : 1][MF == 1 ? 2 : 1][MF == 1 ? 32 : 1];
    s16x4_l wqb[MF == 2 ? S : 1][MF == 2 ? 2 : 1][MF == 2 ? 8 : 1];     // bf16 form: W[4 k4 .. 4 k4 + 3][my unit], packed
    const int rg = lane >> 5, ug = (lane >> 2) & 7, ij = lane & 3;     // 4x4x1 blocks: (row group, unit group), index in block
    if constexpr (MF == 2) {
#pragma unroll
        for (int si = 0; si < S; ++si)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * k4 + e;
                        w[e] = ar.W[si][(long)((k >> 3) * H + role * UW + (k & 7)) * H + 64 * wave + 32 * ch + 4 * ug + ij];
                    }
                    wqb[si][ch][k4] = pk4_bf16(w);
                }
    } else if constexpr (MF == 1) {
#pragma unroll
        for (int si = 0; si < S; ++si)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int k = 0; k < 32; ++k)
                    wq[si][ch][k] = ar.W[si][(long)((k >> 3) * H + role * UW + (k & 7)) * H + 64 * wave + 32 * ch + 4 * ug + ij];
    } else if constexpr (WR) {
#pragma unroll
        for (int si = 0; si < S; ++si)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 16 * c + q4 + e;
                        wr[si][n][c][e] = ar.W[si][(long)((k >> 3) * H + role * UW + (k & 7)) * H + 64 * wave + 16 * n + r];
                    }
    } else {
#pragma unroll
        for (int si = 0; si < S; ++si) {
            float wv[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) wv[k] = ar.W[si][(long)((k >> 3) * H + role * UW + (k & 7)) * H + tid];
#pragma unroll
            for (int k = 0; k < 32; k += 4)
                *reinterpret_cast<f32x4*>(wloc + (si * 256 + tid) * WS + k) = f32x4{wv[k], wv[k + 1], wv[k + 2], wv[k + 3]};
        }
    }
    for (int i = tid; i < RG * DGL; i += 256) dgl[i] = 0.f;
    if constexpr (FW)
        for (int i = tid; i < ar.V * 33; i += 256) dtl[i] = 0.f;   // row stride 33: rows of different tokens on different banks
    __syncthreads();
    const int row0 = a.row_base + (int)xcc * a.row_xs + 16 * (int)grp;
    const long sH = (long)B * H, sG = (long)B * G, lH = (long)T * sH, lG = (long)T * sG;
    // the gate-gradient image of row `rw` of 8-row half g, layer l: one 16-row image (R16) or an image per half
    auto dgl_row = [&](int g, int l, int rw) -> float* {
        return R16 ? dgl + (l * 16 * NRT + 8 * g + rw) * 32 : dgl + g * DGL + (l * 16 + rw) * 32;
    };
    // ---- fused weight gradients: sources q = 0: dWh_top (A = dG_top[t], B = h_top[t-1]); L = 2: q = 1: dWx_1 (dG_1[t],
    // h_0[t]), q = 2: dWh_0 (dG_0[t], h_0[t-1]).  32x32x2: A lane (c = lane & 31, k = lane >> 5) = dG[row 2kk + k][c],
    // B lane = h[row 2kk + k][64 wave + 32 tile + (lane & 31)]; D register g, lane: gate column (g&3) + 8(g>>2) + 4(lane>>5).
    constexpr int NQ = FW ? S : 1;
    f32x16 wacc[NQ][2];
    float wa[NQ][4], wb[NQ][4][2];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    bool wq_on[NQ];
    if constexpr (FW) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            wq_on[q] = false;
#pragma unroll
            for (int tl_ = 0; tl_ < 2; ++tl_)
#pragma unroll
                for (int g = 0; g < 16; ++g) wacc[q][tl_][g] = 0.f;
        }
    }
    // operands of the gate gradients that tick sp left in dgl: A from LDS now (the next epilogue overwrites it), B requested
    auto wg_prepare = [&](int sp, int q_lo, int q_hi) {
        if constexpr (FW) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q < q_lo || q >= q_hi) continue;
                const int lq = (q == 2) ? 0 : LL - 1;                          // layer of the gate gradients
                const int tq = T - 1 - (sp - 2 * (LL - 1 - lq));               // their time step
                const int lh = (q == 1) ? 0 : lq;                              // layer / time of the hidden states
                const int th = (q == 1) ? tq : tq - 1;
                wq_on[q] = tq >= 0 && tq < T && th >= 0;                       // block-uniform
                if (!wq_on[q]) continue;
                const float* hp = ar.hseq + lh * lH + (long)th * sH + 64 * wave + (lane & 31);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int row = 2 * kk + (lane >> 5);
                    wa[q][kk] = dgl[(lq * 16 + row) * 32 + (lane & 31)];
                    const bool rok = row < RX && row0 + row < B && !(ar.fw_dbg & 1);
                    const float* hr = hp + (long)min(row0 + row, B - 1) * H;
                    wb[q][kk][0] = rok ? hr[0] : 0.f;
                    wb[q][kk][1] = rok ? hr[32] : 0.f;
                }
            }
        }
    };
    auto wg_mfma = [&](int q) {
        if constexpr (FW) {
            if (q < NQ && wq_on[q < NQ ? q : 0] && !(ar.fw_dbg & 2)) {
                const int qq = q < NQ ? q : 0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    wacc[qq][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[qq][kk], wb[qq][kk][0], wacc[qq][0], 0, 0, 0);
                    wacc[qq][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[qq][kk], wb[qq][kk][1], wacc[qq][1], 0, 0, 0);
                }
            }
        }
    };
    const int slot = tid >> 6, p = tid & 63;          // slots: cell(l) -> LL-1-l, xproj(0) -> LL (as lstm_bwd_persist_kernel)
    const bool is_cell = slot < LL;
    const int el = is_cell ? (LL - 1 - slot) : (slot - LL);
    const int erow = p >> 3, ul = p & 7;
    const int unit = role * UW + ul;
    bool eact[RG];
    int eb[RG];
    long hb[RG];
    float dcst[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        eact[g] = slot < S && 8 * g + erow < RX && row0 + 8 * g + erow < B;
        eb[g] = min(row0 + 8 * g + erow, B - 1);
        hb[g] = (long)eb[g] * H + unit;
        dcst[g] = 0.f;
    }
    unsigned* my_flag = a.flags + (grp * 8 + xcc) * 32 + role;
    const unsigned* xflags = a.flags + (grp * 8 + xcc) * 32;
    const long part_src = (long)8 * 32 * 32 * 64;     // floats per (parity, source)
    float* const part_grp = ar.part + (long)grp * RG * 2 * S * part_src;   // two-group form: each group its own exchange buffers
    if constexpr (NG == 2) ps_stagger(grp, a.stagger);
    // Single-buffered exchange (ar.parmask == 0): half the footprint in the XCD's L2 -- at 32 rows per XCD the two parities
    // of the exchange are 6 MB against 4 MB of L2 and every gather missed (tick 10.0 -> 7.6 us isolated).  What the parity
    // bought is restored by a second flag line: a CU raises done[role] = s + 1 behind the tick's closing barrier (all its
    // gathers of tick s have returned); a producer looks at the 32 done words at the TOP of tick s + 1 and checks them just
    // before its first partial store -- they were raised a whole product phase earlier, so the check costs no time.
    constexpr bool single = SB;
    unsigned* my_done = single ? ar.done + (grp * 8 + xcc) * 32 + role : nullptr;
    const unsigned* xdone = single ? ar.done + (grp * 8 + xcc) * 32 : nullptr;
    auto wait_gathered = [&](unsigned dv, int s) {    // every consumer of my XCD has finished reading tick s - 1
        unsigned spins = 0;
        while (!__all((int)(dv - (unsigned)s) >= 0)) {
            if (++spins > 4000000u || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (lane == 0) atomicAdd(a.err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            dv = (lane < 32) ? __hip_atomic_load(xdone + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)s;
        }
    };

    for (int s = a.s_begin; s < a.s_end; ++s) {
        PS_STAMP(0);
        unsigned dv = (unsigned)s;                                // single buffer: the consumers' "gathered" words, requested now
        if constexpr (SB) { if (lane < 32) dv = __hip_atomic_load(xdone + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        bool may_store = !SB;
        const int skew = 2 * (LL - 1 - el);
        const int t = is_cell ? T - 1 - (s - skew) : T - 1 - (s + 1 - skew);
        const bool jact = slot < S && t >= 0 && t < T;
        // forward values of my epilogue (static).  ARCVAE_RS_LATE_PREFETCH = 0: requested first, at the top of the tick (they
        // have landed by the gather); 1: requested behind the flag store, under the flag poll -- the products start at once
        float gi[RG], gf[RG], gg[RG], go[RG], c_v[RG], cprev_v[RG], ext_v[RG];
        auto prefetch = [&]() {
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                gi[g] = gf[g] = gg[g] = go[g] = c_v[g] = cprev_v[g] = ext_v[g] = 0.f;
                if (eact[g] && is_cell && jact) {
                    const float* gp = a.gseq + el * lG + (long)t * sG + (long)eb[g] * G + unit;
                    gi[g] = gp[0]; gf[g] = gp[H]; gg[g] = gp[2 * H]; go[g] = gp[3 * H];
                    c_v[g] = a.cseq[el * lH + (long)t * sH + hb[g]];
                    if (t > 0) cprev_v[g] = a.cseq[el * lH + (long)(t - 1) * sH + hb[g]];
                    if (el == LL - 1 && t == T - 1) ext_v[g] = a.dh_top[(long)eb[g] * a.ld_dh_top + unit];
                    if (s == a.s_begin) {
                        dcst[g] = (t < T - 1) ? a.dcs[((long)el * RS + ((t + 1) % RS)) * sH + hb[g]] : 0.f;
                        if (el < LL - 1) ext_v[g] = a.dxs[((long)el * RS + (t % RS)) * sH + hb[g]];
                    }
                }
            }
        };
#ifndef ARCVAE_RS_LATE_PREFETCH
#define ARCVAE_RS_LATE_PREFETCH 1
#endif
        if constexpr (ARCVAE_RS_LATE_PREFETCH == 0 || FW) prefetch();
        int tok = 0;
        if constexpr (FW) {
            if (eact[0] && is_cell && jact && el == 0) tok = min(max(ar.x_tb[(long)t * B + eb[0]], 0), ar.V - 1);
        }
        // first tick of a chunk: my gate columns of the gradients the previous chunk left in memory
        if (s == a.s_begin) {
            for (int i = tid; i < RG * LL * 8 * 32; i += 256) {
                const int g = i / (LL * 256), ig = i - g * (LL * 256);
                const int l = ig / 256, rem = ig - l * 256, rw = rem >> 5, k = rem & 31;
                const int tl_ = T - 1 - (s - 2 * (LL - 1 - l));          // cell(l, tl_) of this tick reads dG^l_{tl_ + 1}
                float v = 0.f;
                if (tl_ + 1 >= 0 && tl_ + 1 < T && 8 * g + rw < RX && row0 + 8 * g + rw < B)
                    v = a.dG[l * lG + (long)(tl_ + 1) * sG + (long)(row0 + 8 * g + rw) * G + (k >> 3) * H + role * UW + (k & 7)];
                dgl_row(g, l, rw)[k] = v;
            }
            __syncthreads();
        }
        if (s > a.s_begin) wg_prepare(s - 1, 0, 2);   // FW (sources 0, 1; source 2 behind the flag store): operands of the weight-gradient products of the previous tick's gate gradients
        PS_STAMP_START();
        // ---- partial products of the S slots from my local gate gradients; wave w covers units [64w, 64w + 64)
        float* pbase = part_grp + (long)(SB ? 0 : (s & 1)) * S * part_src + (long)xcc * 32 * 32 * 64;   // 8-row half g: + g * 2 S part_src
#pragma unroll
        for (int g = 0; g < (R16 ? 1 : RG); ++g)   // (R16: both halves in one pass of 16-row tiles)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const bool cellj = j < LL;
            const int lj = cellj ? (LL - 1 - j) : (j - LL);
            const int skj = 2 * (LL - 1 - lj);
            const int tj = cellj ? T - 1 - (s - skj) : T - 1 - (s + 1 - skj);
            const bool actj = tj >= 0 && tj < T && (!cellj || tj < T - 1);
            const int ls = cellj ? lj : lj + 1;                         // layer whose local gradients feed this slot
            if constexpr (MF != 0) {
                // lane (rg, ug, ij): row 4rg + ij, units ju0 .. ju0+3 -> consumer ju0>>3, piece [row][ju0&7 ..]
                float* pdst = pbase + (long)g * 2 * S * part_src + j * part_src + ((long)((64 * wave + 4 * ug) >> 3) * 32 + role) * 64 + (4 * rg + ij) * 8 + ((4 * ug) & 7);
                if (actj) {
                    f32x4 bq[8];
#pragma unroll
                    for (int k4 = 0; k4 < 8; ++k4)
                        bq[k4] = *reinterpret_cast<const f32x4*>(dgl + g * DGL + (ls * 16 + 4 * rg + ij) * 32 + 4 * k4);
                    // (round 3, measured and dropped: all eight LDS reads of a slot pinned in front of its first MFMA --
                    // isolated tick 2.85 -> 2.95 us; the four accumulators in rotation, distance 4 instead of 2 -- 2.84: the
                    // compiler's one-read-ahead schedule of this phase is not what holds it at 1.2 us)
                    f32x4 acc[2][2];      // two independent chains per unit chunk (first link: C = 0)
                    if constexpr (MF == 2) {
#pragma unroll
                        for (int k4 = 0; k4 < 8; ++k4) {
                            const s16x4_l bb = pk4_bf16(bq[k4]);
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch) {
                                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                                acc[ch][k4 & 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(
                                    wqb[MF == 2 ? j : 0][MF == 2 ? ch : 0][MF == 2 ? k4 : 0], bb, k4 > 1 ? acc[ch][k4 & 1] : z, 0, 0, 0);
                            }
                        }
                    } else
#pragma unroll
                    for (int k4 = 0; k4 < 8; ++k4)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) {
                            const int sj = MF == 1 ? j : 0, sc = MF == 1 ? ch : 0, sk = MF == 1 ? 4 * k4 : 0;
                            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                            acc[ch][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[sj][sc][sk + 0], bq[k4].x, k4 ? acc[ch][0] : z, 0, 0, 0);
                            acc[ch][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[sj][sc][sk + (MF == 1 ? 1 : 0)], bq[k4].y, k4 ? acc[ch][1] : z, 0, 0, 0);
                            acc[ch][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[sj][sc][sk + (MF == 1 ? 2 : 0)], bq[k4].z, acc[ch][0], 0, 0, 0);
                            acc[ch][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[sj][sc][sk + (MF == 1 ? 3 : 0)], bq[k4].w, acc[ch][1], 0, 0, 0);
                        }
                    if constexpr (SB) { if (!may_store) { wait_gathered(dv, s); may_store = true; } }
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)   // unit chunk ch: units + 32 -> consumer + 4
                        *reinterpret_cast<f32x4*>(pdst + (long)ch * 4 * 32 * 64) =
                            f32x4{acc[ch][0].x + acc[ch][1].x, acc[ch][0].y + acc[ch][1].y,
                                  acc[ch][0].z + acc[ch][1].z, acc[ch][0].w + acc[ch][1].w};
                } else {
                    if constexpr (SB) { if (!may_store) { wait_gathered(dv, s); may_store = true; } }
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        *reinterpret_cast<f32x4*>(pdst + (long)ch * 4 * 32 * 64) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            } else {
            f32x4 acc[NRT][4];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (actj) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 av[NRT];
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
                        av[rt] = *reinterpret_cast<const f32x4*>(dgl + (ls * 16 * NRT + 16 * rt + r) * 32 + 16 * c + q4);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        f32x4 w;
                        if constexpr (WR) w = wr[WR ? j : 0][WR ? n : 0][WR ? c : 0];
                        else w = *reinterpret_cast<const f32x4*>(wloc + (j * 256 + 64 * wave + 16 * n + r) * WS + 16 * c + q4);
                        // weights as the A operand: the product comes out transposed, lane (row r, units 4*(lane>>4)+reg)
#pragma unroll
                        for (int rt = 0; rt < NRT; ++rt) {
                            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, av[rt].x, acc[rt][n], 0, 0, 0);
                            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, av[rt].y, acc[rt][n], 0, 0, 0);
                            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, av[rt].z, acc[rt][n], 0, 0, 0);
                            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, av[rt].w, acc[rt][n], 0, 0, 0);
                        }
                    }
                }
            }
            // scatter: rows 0..7 (lanes with r < 8; tile form: all 16 rows of every tile, each 8-row half into its own
            // buffers); a lane holds 4 consecutive units ju0.. of its row -> consumer ju0>>3, piece [row][ju0&7 ..]: one
            // 16-byte store per tile (4 per slot instead of 16 dword stores)
            if constexpr (SB) { if (!may_store) { wait_gathered(dv, s); may_store = true; } }
            if (R16 || r < 8) {
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    float* ph = pbase + (R16 ? (long)(2 * rt + (r >> 3)) * 2 * S * part_src : 0L);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int ju0 = 64 * wave + 16 * n + q4;
                        *reinterpret_cast<f32x4*>(ph + j * part_src + ((long)(ju0 >> 3) * 32 + role) * 64 + (r & 7) * 8 + (ju0 & 7)) = acc[rt][n];
                    }
                }
            }
                    }
        }
        PS_STAMP(2);
        ps_stores_in_l2();                                       // my partials have reached the XCD's L2
        PS_STAMP(3);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (ARCVAE_RS_LATE_PREFETCH != 0 && !FW) prefetch();
        // FW: the flags are travelling and the matrix pipe is idle: first part of the weight-gradient products
        if constexpr (LL > 1) { if (s > a.s_begin) wg_prepare(s - 1, 2, 3); }   // its loads fly while the flags are polled
        wg_mfma(0);
        if constexpr (LL > 1) wg_mfma(1);
        // ---- every CU of my XCD has published its partials of tick s (polled by a wave without an epilogue job where there
        // is one -- its flag loads then do not queue behind the epilogue's operand loads: ARCVAE_RS_POLL_WAVE)
#ifndef ARCVAE_RS_POLL_WAVE
#define ARCVAE_RS_POLL_WAVE 3
#endif
        if (wave == (S < 4 ? ARCVAE_RS_POLL_WAVE : 0)) {
            unsigned spins = 0;
            while (true) {
                const unsigned v = (lane < 32) ? __hip_atomic_load(xflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : (unsigned)(s + 1);
                if (__all((int)(v - (unsigned)(s + 1)) >= 0)) break;
                if (++spins > 4000000u || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) { atomicAdd(a.err, 1u); s_ok = 0; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!s_ok) return;
        PS_STAMP(4);
        // ---- gather: the 32 pieces of my (slot, row, unit), group by group, then that group's epilogue
#pragma unroll
        for (int g = 0; g < RG; ++g) {
        float dh = 0.f;
        {
            // my role's 32 pieces of this slot: 8 KB contiguous, [producer][64 (row, unit) values]
            // (slot = tid >> 6 is wave-uniform: say so, or the compiler wraps every load in a waterfall loop over the
            // descriptor's lanes with a vmcnt(0) inside -- 32 serialised round trips)
            const int slot_u = __builtin_amdgcn_readfirstlane(slot);
            const __amdgpu_buffer_rsrc_t rs = ps_rsrc(pbase + (long)g * 2 * S * part_src + (slot_u < S ? slot_u : 0) * part_src + ((long)role * 32) * 64, 32 * 64 * 4);
            float v[32];
            const bool gact = eact[g] && jact;
            if (gact) {
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = ps_load_sc1(rs, (unsigned)((i * 64 + p) * 4));
            }
            if constexpr (LL > 1) { if (g == 0) wg_mfma(2); }   // FW: the rest of the weight-gradient products, under the gather's loads
            if (gact) {
#pragma unroll
                for (int i = 0; i < 32; i += 4) dh += (v[i] + v[i + 1]) + (v[i + 2] + v[i + 3]);
            }
#ifdef ARCVAE_PS_STAMPS
            if (g == 0 && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PS_STAMP(5); }
#endif
        }
        float* dxg = dxl + g * 128;
        if (eact[g] && jact) {
            if (!is_cell) {
                dxg[((s + 1) & 1) * 64 + p] = dh;
                a.dxs[((long)el * RS + (t % RS)) * sH + hb[g]] = dh;
            } else {
                if (el < LL - 1 && s != a.s_begin) ext_v[g] = dxg[(s & 1) * 64 + p];
                dh += ext_v[g];
                const float tc = chain_tanh(c_v[g]);
                const float d_o = dh * tc * go[g] * (1.f - go[g]);
                const float dc = dh * go[g] * (1.f - tc * tc) + dcst[g];
                const float d_i = dc * gg[g] * gi[g] * (1.f - gi[g]);
                const float d_f = t > 0 ? dc * cprev_v[g] * gf[g] * (1.f - gf[g]) : 0.f;
                const float d_g = dc * gi[g] * (1.f - gg[g] * gg[g]);
                dcst[g] = dc * gf[g];
                a.dcs[((long)el * RS + (t % RS)) * sH + hb[g]] = dcst[g];
                float* dp = a.dG + el * lG + (long)t * sG + (long)eb[g] * G + unit;
                dp[0] = d_i; dp[H] = d_f; dp[2 * H] = d_g; dp[3 * H] = d_o;
                float* dl = dgl_row(g, el, erow) + ul;   // my gate columns stay on the CU for the next tick
                dl[0] = d_i; dl[8] = d_f; dl[16] = d_g; dl[24] = d_o;
                if constexpr (FW) {
                    if (el > 0) {                                    // bias gradient of the layers above 0: running sum
                        bsum[0] += d_i; bsum[1] += d_f; bsum[2] += d_g; bsum[3] += d_o;
                    } else if (!(ar.fw_dbg & 4)) {                   // layer 0: token-table gradient (LDS scatter-add)
                        float* tp = dtl + tok * 33 + ul;
                        atomicAdd(tp, d_i); atomicAdd(tp + 8, d_f); atomicAdd(tp + 16, d_g); atomicAdd(tp + 24, d_o);
                    }
                }
            }
        }
        }
        PS_STAMP(6);
        __syncthreads();
        if constexpr (SB) { if (tid == 0) __hip_atomic_store(my_done, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        PS_STAMP_END();
    }
    if constexpr (FW) {
        // ---- the last tick's gate gradients, then everything this launch accumulated goes to the gradient buffers
        wg_prepare(a.s_end - 1, 0, NQ);
#pragma unroll
        for (int q = 0; q < NQ; ++q) wg_mfma(q);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int tl_ = 0; tl_ < 2; ++tl_)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int c = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);              // my gate column (MFMA row)
                    atomicAdd(ar.dW[q] + (long)((c >> 3) * H + role * UW + (c & 7)) * H + 64 * wave + 32 * tl_ + (lane & 31),
                              wacc[q][tl_][g]);
                }
        if (LL > 1 && ar.dbias1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = bsum[g];
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);   // over my XCD's rows
                if (is_cell && el > 0 && erow == 0) atomicAdd(ar.dbias1 + g * H + unit, v);
            }
        }
        for (int i = tid; i < ar.V * 32; i += 256) {
            const float v = dtl[(i >> 5) * 33 + (i & 31)];
            if (v != 0.f) atomicAdd(ar.dtable + (long)(i >> 5) * G + ((i & 31) >> 3) * H + role * UW + (i & 7), v);
        }
    }
}

template <int NT, int LL, int RT>
void launch_persist_bwd(const PersistBwdArgs& a, size_t lds, hipStream_t s) {
    (void)hipFuncSetAttribute((const void*)lstm_bwd_persist_kernel<NT, LL, RT>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL((lstm_bwd_persist_kernel<NT, LL, RT>), dim3(256), dim3(256), lds, s, a);
}
inline size_t persist_bwd_lds_bytes(int B, int H, int L) {
    const int NT = H / 128, UW = 4 * NT, S = 2 * L - 1, RT = persist_row_tiles(B);
    return sizeof(float) * ((size_t)S * (4 * H / 16) * UW * 16 + (size_t)4 * S * 256 * RT + (size_t)2 * L * 16 * RT * UW);
}
// Opt-in (ARCVAE_PERSIST_BWD=1): measured SLOWER than the launches at every batch tried -- bs 64: tick 5.4 us alone and
// 7-8 us beside the weight-gradient GEMMs against 5.0 / 5.4-6.3 us per launch (step 1.64 vs 1.40 ms); bs 128: 2.38 vs
// 2.03 ms; bs 256 (two row tiles): 4.28 vs 3.30 ms.  A tick's operand is 4x the forward's (K = 4H: 96-192 KB into every
// CU in dependent load -> MFMA rounds), its outputs are 8 units wide (half of a 16 x 16 tile at best), and blocks that
// sit on every CU for the whole sweep feel the GEMMs' LDS / MFMA traffic in every tick.  Kept: parity-green, and the
// base for a version with an LDS-staged operand and two sources packed into one tile.
inline bool persist_bwd_shape_ok(int B, int T, int H, int L) {
    if (arcvae_env_int("ARCVAE_PERSIST", 1) == 0 || arcvae_env_int("ARCVAE_PERSIST_BWD", 0) == 0) return false;
    if (H % 128 != 0 || H / 128 > 3 || L < 1 || L > 2 || B < 1 || T < 1) return false;
    const int RX = ceil_div(B, 8);
    if (RX > 32) return false;
    return persist_bwd_lds_bytes(B, H, L) <= 150 * 1024;
}

template <int CH>
void launch_fwd(const FwdArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(lstm_fwd_step_kernel<CH>, grid, dim3(256), 0, s, a);
}
template <int CH>
void launch_bwd(const BwdArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL(lstm_bwd_step_kernel<CH>, grid, dim3(1024), 0, s, a);
}

#define DISPATCH_CH(H, FN, ...)                         \
    switch ((H) / 64) {                                 \
        case 1: FN<1>(__VA_ARGS__); break;              \
        case 2: FN<2>(__VA_ARGS__); break;              \
        case 3: FN<3>(__VA_ARGS__); break;              \
        case 4: FN<4>(__VA_ARGS__); break;              \
        case 5: FN<5>(__VA_ARGS__); break;              \
        case 6: FN<6>(__VA_ARGS__); break;              \
        case 7: FN<7>(__VA_ARGS__); break;              \
        case 8: FN<8>(__VA_ARGS__); break;              \
        default: return ARCVAE_ERR_ARG;                 \
    }

inline bool hidden_ok(int H) { return H > 0 && (H % 64) == 0 && H <= 512; }

}  // namespace

// Throughput mode (ARCVAE_LSTM_BF16): the sweeps' operand copies and products in bf16 -- where the shape runs on the
// register-tiled step kernels (the MFMA-bound regime, BASELINE.json configs[2]); the latency-regime kernels stay f32 (a
// tick there is round trips, not matrix time: DESIGN.md section 10).  Forward and BPTT decide separately (their grids differ).
// Where the tile regime starts.  With the three-piece form and what round 3 built on it (K-split BPTT tile, weight gradients
// from the operand planes, the dense decoder on the same kernels) the whole step is ahead of the mid-batch kernels from grids
// of ~140 blocks on -- measured at H 256 / L 2: 640 rows 7.96 vs 7.69 ms (120 blocks: not yet), 768 rows 8.40 vs 8.82, 896 rows
// 9.03 vs 10.09, 1024 rows 9.43 vs 10.94 (192 blocks).  The exact-f32 and throughput-mode forms keep the 200 they were tuned
// with.  ARCVAE_S3_MIN_BLOCKS overrides.
static inline bool s3_flags(int flags) { return (flags & ARCVAE_LSTM_SPLIT3) && !(flags & ARCVAE_LSTM_BF16); }
static inline int tile_min_blocks(int flags) { return s3_flags(flags) ? arcvae_env_int("ARCVAE_S3_MIN_BLOCKS", 140) : 200; }
static inline int fwd_tile_choice(int B, int H, int L, int flags) {
    // (the earlier start only together with the BPTT's: a 64-row forward tile on a 160-block grid beside the mid-batch BPTT kernels
    // loses to the 32-row tile -- 640 rows: 8.04 vs 7.69 ms)
    const bool early = s3_flags(flags) && arcvae_env_int("ARCVAE_STEP_TILE", -1) < 0 &&
                       (H / 64) * ceil_div(B, 64) * (2 * L - 1) >= tile_min_blocks(flags);
    int tile_mt = choose_tile_mt(B, H / 32, L, early ? tile_min_blocks(flags) : 200);
    const int tile_env = arcvae_env_int("ARCVAE_STEP_TILE", -1);
    if (tile_env == 44 || (tile_mt == 4 && tile_env != 4 && ceil_div(B, 64) * (H / 64) * L >= 200))
        tile_mt = 44;  // the 64 x 64 wave tile (16 FLOP per byte) when even its grid fills the chip; 44 forces it
    return tile_mt;
}
// BPTT: in the three-piece form the K-split 64 x 64 tile decides (its grid: H/64 x ceil(B/64) x jobs); else the 64 x 32 wave tile's
static inline int bwd_tile_choice(int B, int H, int L, int flags) {
    if (s3_flags(flags)) {
        const int force = arcvae_env_int("ARCVAE_STEP_TILE", -1);
        if (force < 0 && (H / 64) * ceil_div(B, 64) * (2 * L - 1) >= tile_min_blocks(flags)) return 4;
    }
    return choose_tile_mt(B, ceil_div(H, 128), 2 * L - 1);
}
static inline bool fwd_bf16(int B, int H, int L, int flags) { return (flags & ARCVAE_LSTM_BF16) && fwd_tile_choice(B, H, L, flags) != 0; }
static inline bool bwd_bf16(int B, int H, int L, int flags) {
    return (flags & ARCVAE_LSTM_BF16) && bwd_tile_choice(B, H, L, flags) != 0;
}
// Three-piece form (ARCVAE_LSTM_SPLIT3, a parity path): where the sweep runs on the register-tiled kernels and the
// throughput mode is not asked for.  The operand copies then hold three bf16 planes: 1.5 x the f32 copies' size.
static inline bool fwd_split3(int B, int H, int L, int flags) {
    return s3_flags(flags) && fwd_tile_choice(B, H, L, flags) != 0;
}
// ... and, OPT-IN (ARCVAE_STEP2_SPLIT3=1), the mid-batch 2x2 BPTT kernel where H is a multiple of 128.  Parity-green, measured
// SLOWER than its exact-f32 form: 10.87 vs 10.64 us per launch alone at 256 rows (15.6 vs 14.2 in the step, 3.19 vs 3.03 ms),
// 18.8 vs 17.1 at 512 -- the launch is bound by operand delivery and its stores, not by the 3.5 us of matrix time the form
// removes (1.5 x the operand bytes, twelve 2-byte plane stores per element instead of four 4-byte ones).
static inline bool bwd_step2_split3(int B, int H, int L, int flags) {
    // (not where the opt-in output-split persistent BPTT would run instead: it reads the f32 weight layouts)
    return bwd_tile_choice(B, H, L, flags) == 0 && choose_step2(B) && (H % 128) == 0 &&
           !persist_bwd_shape_ok(B, 1, H, L) && arcvae_env_int("ARCVAE_STEP2_SPLIT3", 0) != 0;
}
static inline bool bwd_split3(int B, int H, int L, int flags) {
    return s3_flags(flags) && (bwd_tile_choice(B, H, L, flags) != 0 || bwd_step2_split3(B, H, L, flags));
}
// Operand-plane weight gradients (gemm.hip: wgrad_planes_kernel): both sweeps on the three-piece tile kernels and whole
// 32-row K-steps.  The operand rings (hseq_t, dG_t) then keep ALL T time slots -- the planes a launch writes for the next
// launch are the weight-gradient GEMM's operands as well; the small dc / dX rings stay short.  ARCVAE_WGRAD_PLANES=0: off.
static inline bool wgrad_planes_mode(int B, int H, int L, int flags) {
    return fwd_split3(B, H, L, flags) && bwd_split3(B, H, L, flags) && (B % 32) == 0 && arcvae_env_int("ARCVAE_WGRAD_PLANES", 1) != 0;
}
static inline int operand_ring_slots(int B, int T, int H, int L, int flags) {
    return wgrad_planes_mode(B, H, L, flags) ? T : arcvae_ring_slots(T);
}

// tiled weights: Wh_t[l] at wt + l*wsz, Wx_t[l] (l >= 1) at wt + (L + l - 1)*wsz; with wT_bwd also the BPTT layouts of
// the same weights (keeps that launch off the chain between the sweeps) -- one launch for both when the 2(2L-1) jobs
// fit (L <= 4)
// fwd_p / bwd_p: 0 f32 copies, 1 bf16 (throughput mode), 2 three bf16 planes (slot stride 1.5 x the f32 copy's)
static int tile_all_weights(const float* const* Wx, const float* const* Wh, float* wt, float* wT_bwd, int H, int L,
                            int fwd_p, int bwd_p, hipStream_t stream) {
    const long wsz0 = (long)H * 4 * H;
    const float* src[32];
    float* dst[32];
    int cols[32], mode[32];
    int n = 0;
    for (int pass = 0; pass < (wT_bwd ? 2 : 1); ++pass) {
        float* base = pass == 0 ? wt : wT_bwd;
        if (!base) continue;                       // wt == null: only the BPTT layouts (persistent forward reads row-major)
        const int pp = pass == 0 ? fwd_p : bwd_p;
        const long wsz = pp == 2 ? wsz0 * 3 / 2 : wsz0;
        for (int l = 0; l < L; ++l) {
            const int md = pass == 0 ? (pp == 2 ? 4 : (pp == 1 ? 2 : 0)) : (pp == 2 ? 5 : (pp == 1 ? 3 : 1));
            src[n] = Wh[l]; dst[n] = base + l * wsz; cols[n] = H; mode[n] = md; ++n;
            if (l > 0) { src[n] = Wx[l]; dst[n] = base + (L + l - 1) * wsz; cols[n] = H; mode[n] = md; ++n; }
        }
    }
    for (int i = 0; i < n; i += 16) {
        const int rc = arcvae_tile_weights(src + i, dst + i, cols + i, mode + i, n - i < 16 ? n - i : 16, H, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    return ARCVAE_OK;
}

// bit 0: arcvae_enc_lstm_forward runs this shape on the register-tiled step kernels, bit 1: arcvae_enc_lstm_backward does
// -- i.e. where ARCVAE_LSTM_BF16 takes effect (and the octet-major copies get written: both bits needed for the
// weight-gradient kernel that reads them).
// Time slots of the operand rings hseq_t / dG_t the sweeps will use for this shape and these flags (T when the weight
// gradients read the operand planes, else the short ring of common.h): what the caller sizes those workspaces with.
extern "C" int arcvae_enc_lstm_operand_slots(int B, int T, int H, int L, int flags) {
    if (B <= 0 || T <= 0 || L <= 0 || !hidden_ok(H)) return 0;
    return operand_ring_slots(B, T, H, L, flags);
}

// Floats of {hseq_t, dG_t, wt, wT} the launch-based sweeps of this shape use under the CURRENT kernel-family knobs: ring slots x
// slab size, slabs and weight copies at 3/2 of their f32 size where that sweep runs in the three-piece form.  What
// arcvae_enc_lstm_ws_floats reports and what every sweep call checks its caller's capacities against (ws_floats): the family is
// re-decided at every call (the tests toggle the knobs), so an undersized buffer is refused here instead of being written past.
static inline void lstm_ws_need(int B, int T, int H, int L, int flags, long need[4]) {
    const long sH = (long)B * H, sG = (long)B * 4 * H, w = (long)H * 4 * H;
    const bool f3 = fwd_split3(B, H, L, flags), b3 = bwd_split3(B, H, L, flags);
    const long slots = operand_ring_slots(B, T, H, L, flags);
    need[0] = (long)L * slots * (f3 ? sH * 3 / 2 : sH);
    need[1] = (long)L * slots * (b3 ? sG * 3 / 2 : sG);
    need[2] = (long)(2 * L - 1) * (f3 ? w * 3 / 2 : w);
    need[3] = (long)(2 * L - 1) * (b3 ? w * 3 / 2 : w);
}
extern "C" int arcvae_enc_lstm_ws_floats(int B, int T, int H, int L, int flags, long* floats) {
    if (B <= 0 || T <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H) || !floats) return ARCVAE_ERR_ARG;
    lstm_ws_need(B, T, H, L, flags, floats);
    return ARCVAE_OK;
}

extern "C" int arcvae_enc_lstm_tiled(int B, int H, int L) {
    if (B <= 0 || L <= 0 || !hidden_ok(H)) return 0;
    return (fwd_tile_choice(B, H, L, 0) != 0 ? 1 : 0) | (bwd_tile_choice(B, H, L, 0) != 0 ? 2 : 0);
}
// ... for the sweeps called with these flags (the three-piece form enters the tile regime earlier: tile_min_blocks)
extern "C" int arcvae_enc_lstm_tiled_for(int B, int H, int L, int flags) {
    if (B <= 0 || L <= 0 || !hidden_ok(H)) return 0;
    return (fwd_tile_choice(B, H, L, flags) != 0 ? 1 : 0) | (bwd_tile_choice(B, H, L, flags) != 0 ? 2 : 0);
}

// Reference: models/encoder.py:98-101 (L stacked nn.LSTM over the full padded sequence, Q3).
//   x_tb   [T,B] tokens (time-major)          table0 [V,4H] = emb . Wx_0^T + bias_0
//   Wx[l]  [4H,H] (l >= 1), Wh[l] [4H,H], bias[l] [4H] (l >= 1): HOST arrays of device pointers
//   hseq/cseq [L,T,B,H], gseq [L,T,B,4H] outputs (gseq = post-activation i,f,g,o)
//   hseq_t [L,RS,B*H] workspace: k-chunk-major copy of the h slab of time t in slot t % RS, RS = min(T, 16)
//          (common.h: arcvae_ring_slots).  A slab is only read by the NEXT launch (same layer at t+1, layer above at
//          t), so a short ring suffices and stays cache-resident, unlike a [L,T,..] copy that streams 17 MB through
//          the Infinity Cache
//   wt     [(2L-1),4H*H] workspace: k-chunk-major weight copies, refreshed here (weights change every step)
//   wT_bwd optional [(2L-1),H*4H]: also write the BPTT layouts (then call the backward with retile = 0)
//   hidden_dim: multiple of 64, <= 512.
extern "C" int arcvae_enc_lstm_forward(const int32_t* x_tb, const float* table0, const float* const* Wx,
                                       const float* const* Wh, const float* const* bias, float* hseq,
                                       float* hseq_t, float* cseq, float* gseq, float* wt, float* wT_bwd, int B,
                                       int T, int V, int H, int L, int flags, const long* ws_floats, void* h_oct,
                                       unsigned long long* trace, hipStream_t stream) {
    if (!x_tb || !table0 || !Wx || !Wh || !bias || !hseq || !hseq_t || !cseq || !gseq || !wt || !ws_floats) return ARCVAE_ERR_ARG;
    if (B <= 0 || T <= 0 || V <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H)) return ARCVAE_ERR_ARG;
    {   // the caller's buffers against what the kernel family chosen at THIS call writes (rings: slots x slab, planes at 3/2)
        long need[4];
        lstm_ws_need(B, T, H, L, flags, need);
        if (ws_floats[0] < need[0] || ws_floats[2] < need[2] || (wT_bwd && ws_floats[3] < need[3])) return ARCVAE_ERR_ARG;
    }
    for (int l = 0; l < L; ++l)
        if (!Wh[l] || (l > 0 && (!Wx[l] || !bias[l]))) return ARCVAE_ERR_ARG;
    const long sH = (long)B * H, sG = (long)B * 4 * H;
    const long lH = (long)T * sH, lG = (long)T * sG;
    const bool b16 = fwd_bf16(B, H, L, flags), s3 = fwd_split3(B, H, L, flags);
    const long wsz = s3 ? (long)H * 4 * H * 3 / 2 : (long)H * 4 * H;     // (three-piece form: three bf16 planes per copy)
    const long sHt = s3 ? sH * 3 / 2 : sH;
    {
        const int rc = tile_all_weights(Wx, Wh, wt, wT_bwd, H, L, s3 ? 2 : (b16 ? 1 : 0),
                                        bwd_split3(B, H, L, flags) ? 2 : (bwd_bf16(B, H, L, flags) ? 1 : 0), stream);
        if (rc != ARCVAE_OK) return rc;
    }
    const int tile_mt = fwd_tile_choice(B, H, L, flags);
    const bool oct = b16 && h_oct && (B % 16) == 0;      // the octet-major copy groups 8 batch rows of one time step
    const bool step2 = !tile_mt && choose_step2(B);
    const int RS = operand_ring_slots(B, T, H, L, flags);
    for (int s = 0; s < T + L - 1; ++s) {
        FwdArgs a;
        a.B = B; a.H = H; a.V = V; a.prio = arcvae_step_prio(); a.remap = arcvae_xcd_remap() | (arcvae_env_int("ARCVAE_TILE_XCD", 0) != 0 ? 2 : 0); a.trace = trace ? trace + 2 * (long)s : nullptr;
        a.dbg = arcvae_env_int("ARCVAE_TILE_DEBUG", 0); a.lite = 0;
        int nj = 0;
        for (int l = 0; l < L; ++l) {
            const int t = s - l;
            if (t < 0 || t >= T) continue;
            FwdJob& j = a.job[nj++];
            j.xin = l > 0 ? hseq_t + ((long)(l - 1) * RS + (t % RS)) * sHt : nullptr;
            j.Wx = l > 0 ? wt + (L + l - 1) * wsz : nullptr;
            j.hprev = t > 0 ? hseq_t + ((long)l * RS + ((t - 1) % RS)) * sHt : nullptr;
            j.Wh = wt + l * wsz;
            j.pre = l > 0 ? bias[l] : table0;
            j.tok = l > 0 ? nullptr : x_tb + (long)t * B;
            j.cprev = t > 0 ? cseq + l * lH + (t - 1) * sH : nullptr;
            j.h = hseq + l * lH + t * sH;
            j.ht = hseq_t + ((long)l * RS + (t % RS)) * sHt;
            j.c = cseq + l * lH + t * sH;
            j.gates = gseq + l * lG + t * sG;
            j.oct = oct ? static_cast<char*>(h_oct) + 2 * (l * lH + t * sH) : nullptr;
        }
        for (int k = nj; k < ARCVAE_MAX_LAYERS; ++k) a.job[k] = a.job[0];
        if (tile_mt && s3) {
            if (tile_mt == 44) launch_fwd_tile<4, 4, 2>(a, B, H, nj, stream);
            else if (tile_mt == 4) launch_fwd_tile<4, 2, 2>(a, B, H, nj, stream);
            else if (tile_mt == 2) launch_fwd_tile<2, 2, 2>(a, B, H, nj, stream);
            else launch_fwd_tile<1, 2, 2>(a, B, H, nj, stream);
            continue;
        }
        if (tile_mt && b16) {
            if (tile_mt == 44) launch_fwd_tile<4, 4, 1>(a, B, H, nj, stream);
            else if (tile_mt == 4) launch_fwd_tile<4, 2, 1>(a, B, H, nj, stream);
            else if (tile_mt == 2) launch_fwd_tile<2, 2, 1>(a, B, H, nj, stream);
            else launch_fwd_tile<1, 2, 1>(a, B, H, nj, stream);
            continue;
        }
        if (tile_mt) {
            if (tile_mt == 44) launch_fwd_tile<4, 4>(a, B, H, nj, stream);
            else if (tile_mt == 4) launch_fwd_tile<4, 2>(a, B, H, nj, stream);
            else if (tile_mt == 2) launch_fwd_tile<2, 2>(a, B, H, nj, stream);
            else launch_fwd_tile<1, 2>(a, B, H, nj, stream);
            continue;
        }
        if (step2) {
            dim3 grid2(H / 8, ceil_div(B, 32), nj);
            DISPATCH_CH(H, launch_fwd2, a, grid2, stream)
            continue;
        }
        dim3 grid(H / 4, ceil_div(B, 16), nj);
        DISPATCH_CH(H, launch_fwd, a, grid, stream)
    }
    return arcvae_launch_status();
}


// 1 if arcvae_enc_lstm_forward_persistent supports the shape (and ARCVAE_PERSIST != 0), else 0.
extern "C" int arcvae_enc_lstm_persistent_ok(int B, int T, int H, int L) { return persist_shape_ok(B, T, H, L) ? 1 : 0; }

// The forward sweep of arcvae_enc_lstm_forward as ONE persistent launch (lstm_fwd_persist_kernel) for the latency
// regime.  Same outputs hseq / cseq / gseq (no k-chunk-major h copy is needed); sync_ws: 512 u32 of scratch (the first
// 272 are re-armed here); start_signal (optional): += 1 when the sweep starts.  After the stream has drained,
// sync_ws[500] != 0 means a block gave up waiting (results invalid: fall back to arcvae_enc_lstm_forward).
extern "C" int arcvae_enc_lstm_forward_persistent(const int32_t* x_tb, const float* table0, const float* const* Wx,
                                                  const float* const* Wh, const float* const* bias, float* hseq,
                                                  float* cseq, float* gseq, float* wT_bwd, long wT_bwd_floats, float* comb,
                                                  unsigned* sync_ws, unsigned* start_signal, int B, int T, int V, int H,
                                                  int L, int flags, unsigned long long* trace, hipStream_t stream) {
    if (!x_tb || !table0 || !Wx || !Wh || !bias || !hseq || !cseq || !gseq || !sync_ws) return ARCVAE_ERR_ARG;
    if (V <= 0 || !persist_shape_ok(B, T, H, L)) return ARCVAE_ERR_ARG;
    if (wT_bwd) {
        long need[4];
        lstm_ws_need(B, T, H, L, flags, need);
        if (wT_bwd_floats < need[3]) return ARCVAE_ERR_ARG;
    }
    for (int l = 0; l < L; ++l)
        if (!Wh[l] || (l > 0 && (!Wx[l] || !bias[l]))) return ARCVAE_ERR_ARG;
    int rc;
    if (wT_bwd) {      // BPTT layouts for a launch-based backward of the same step (the sweep itself reads row-major weights)
        // (in the precision the launch-based backward of this step will read them in: ADVICE r2)
        rc = tile_all_weights(Wx, Wh, nullptr, wT_bwd, H, L, 0,
                              bwd_split3(B, H, L, flags) ? 2 : (bwd_bf16(B, H, L, flags) ? 1 : 0), stream);
        if (rc != ARCVAE_OK) return rc;
    }
    if (!(flags & 1)) {   // bit 0: sync_ws was re-armed by arcvae_enc_prologue
        rc = arcvae_zero(reinterpret_cast<float*>(sync_ws), 1, PS_WORDS, PS_WORDS, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    PersistArgs a;
    a.x_tb = x_tb; a.table0 = table0; a.hseq = hseq; a.cseq = cseq; a.gseq = gseq; a.comb = comb;
    for (int i = 0; i < 7; ++i) a.W[i] = Wh[0];
    for (int l = 0; l < L; ++l) { a.W[l] = Wh[l]; if (l > 0) a.W[L + l - 1] = Wx[l]; }
    for (int l = 0; l < ARCVAE_MAX_LAYERS; ++l) a.bias[l] = (l > 0 && l < L) ? bias[l] : nullptr;
    a.sync = sync_ws; a.start_signal = start_signal; a.trace = trace;
    a.flags = sync_ws + PS_FLAGS; a.cnt = sync_ws + PS_CNT; a.cucnt = nullptr; a.stagger = 0;
    a.B = B; a.T = T; a.H = H; a.V = V; a.RX = ceil_div(B, 8); a.prio = arcvae_step_prio();
    if (persist_two_groups(B, H, L, 1)) {   // two blocks per CU, two independent 16-row recurrences per XCD (the caller's
        if (!(flags & 1)) {              // re-arm covers PS2_WORDS: arcvae_enc_prologue, or here)
            rc = arcvae_zero(reinterpret_cast<float*>(sync_ws + PS2_FWD_FLAGS), 1, PS2_WORDS - PS2_FWD_FLAGS, PS2_WORDS - PS2_FWD_FLAGS, stream);
            if (rc != ARCVAE_OK) return rc;
        }
        a.flags = sync_ws + PS2_FWD_FLAGS; a.cnt = sync_ws + PS2_FWD_CNT;
        a.stagger = arcvae_env_int("ARCVAE_PERSIST2_STAGGER_FWD", 200);
        a.cucnt = arcvae_env_int("ARCVAE_PERSIST2_ASSIGN", 1) != 0 ? sync_ws + PS2_CU : nullptr;
        if (L == 1) launch_persist<2, 1, 1, 0, 2>(a, persist2_lds_bytes(), stream);
        else launch_persist<2, 2, 1, 0, 2>(a, persist2_lds_bytes(), stream);
        return arcvae_launch_status();
    }
    const size_t lds = persist_lds_bytes(B, H, L);
    const int NT = H / 128;
    const int RTn = persist_row_tiles(B);
#define PS_LAUNCH(N_, L_) do { if (RTn == 2) launch_persist<N_, L_, 2>(a, lds, stream); else launch_persist<N_, L_, 1>(a, lds, stream); } while (0)
#define PS_BY_L(N_) switch (L) { case 1: PS_LAUNCH(N_, 1); break; case 2: PS_LAUNCH(N_, 2); break; case 3: PS_LAUNCH(N_, 3); break; default: PS_LAUNCH(N_, 4); break; }
    // ARCVAE_FWD_MFMA: 1 (default) = 4x4x1 blocks where the shape allows (H 256, <= 8 rows per XCD, L <= 2), 0 = 16x16x4
    if (NT == 2 && L <= 2 && a.RX <= 8 && arcvae_env_int("ARCVAE_FWD_MFMA", 1) != 0) {
        if (flags & ARCVAE_PERSIST_BF16) {   // throughput mode: the 4x4x4 bf16 form of the same blocks
            if (L == 1) launch_persist<2, 1, 1, 2>(a, lds, stream); else launch_persist<2, 2, 1, 2>(a, lds, stream);
        } else {
            if (L == 1) launch_persist<2, 1, 1, 1>(a, lds, stream); else launch_persist<2, 2, 1, 1>(a, lds, stream);
        }
    } else if (NT == 1) { PS_BY_L(1) } else if (NT == 2) { PS_BY_L(2) } else { PS_BY_L(3) }
#undef PS_BY_L
#undef PS_LAUNCH
    return arcvae_launch_status();
}


// 1 if arcvae_enc_lstm_backward_persistent supports the shape (and ARCVAE_PERSIST / ARCVAE_PERSIST_BWD != 0).
extern "C" int arcvae_enc_lstm_bwd_persistent_ok(int B, int T, int H, int L) { return persist_bwd_shape_ok(B, T, H, L) ? 1 : 0; }

// Ticks [s_begin, s_end) of the BPTT wavefront of arcvae_enc_lstm_backward as ONE persistent launch
// (lstm_bwd_persist_kernel).  Same schedule, same outputs dG / dcs / dxs (no k-chunk-major dG copy is written, so a
// sweep must use this entry point for all its chunks or for none); the BPTT weight layouts wT must be current (written
// by the forward); sync_ws as for the forward (its flags are re-armed when s_begin == 0).
extern "C" int arcvae_enc_lstm_backward_persistent(const float* cseq, const float* gseq, const float* dh_top,
                                                   int ld_dh_top, float* dG, float* dcs, float* dxs, const float* wT,
                                                   unsigned* sync_ws, unsigned* start_signal, int B, int T, int H,
                                                   int L, int s_begin, int s_end, int chunk_index,
                                                   unsigned long long* trace, hipStream_t stream) {
    if (!cseq || !gseq || !dh_top || !dG || !dcs || !dxs || !wT || !sync_ws) return ARCVAE_ERR_ARG;
    if (!persist_bwd_shape_ok(B, T, H, L) || ld_dh_top < H) return ARCVAE_ERR_ARG;
    const int S = T + 2 * (L - 1);
    if (s_begin < 0 || s_end > S || s_begin >= s_end) return ARCVAE_ERR_ARG;
    // role counters: a fresh set of 8 words per chunk launch of a sweep (all zeroed by chunk 0)
    if (chunk_index < 0 || chunk_index >= 8 || (chunk_index == 0) != (s_begin == 0)) return ARCVAE_ERR_ARG;
    if (chunk_index == 0) {
        const int rc = arcvae_zero(reinterpret_cast<float*>(sync_ws + PS_BWD), 1, PS_WORDS + 64, PS_WORDS + 64, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    PersistBwdArgs a;
    a.wT = wT; a.cseq = cseq; a.gseq = gseq; a.dh_top = dh_top; a.dG = dG; a.dcs = dcs; a.dxs = dxs;
    a.sync = sync_ws + PS_BWD; a.err = sync_ws + PS_ERR; a.start_signal = start_signal; a.trace = trace;
    a.B = B; a.T = T; a.H = H; a.RX = ceil_div(B, 8); a.RS = arcvae_ring_slots(T); a.ld_dh_top = ld_dh_top;
    a.row_base = 0; a.row_xs = a.RX;
    a.s_begin = s_begin; a.s_end = s_end; a.prio = arcvae_step_prio();
    a.cnt_off = chunk_index == 0 ? 0 : (PS_WORDS - PS_CNT) + 8 * (chunk_index - 1);   // chunk 0: words 256..263, then 272..
    a.flags = a.sync + PS_FLAGS; a.cnt = a.sync + PS_CNT + a.cnt_off; a.cucnt = nullptr; a.stagger = 0;
    const size_t lds = persist_bwd_lds_bytes(B, H, L);
    const int NT = H / 128;
    const int RTn = persist_row_tiles(B);
#define PB_BY_L(N_)                                                                                          \
    if (RTn == 2) { if (L == 1) launch_persist_bwd<N_, 1, 2>(a, lds, stream); else launch_persist_bwd<N_, 2, 2>(a, lds, stream); } \
    else { if (L == 1) launch_persist_bwd<N_, 1, 1>(a, lds, stream); else launch_persist_bwd<N_, 2, 1>(a, lds, stream); }
    if (NT == 1) { PB_BY_L(1) } else if (NT == 2) { PB_BY_L(2) } else { PB_BY_L(3) }
#undef PB_BY_L
    return arcvae_launch_status();
}


// 1 if arcvae_enc_lstm_backward_persistent_rs supports the shape (and ARCVAE_PERSIST != 0).
// 1 if the reduce-scatter BPTT of this shape runs as two half-batch sweeps (ARCVAE_RS_HALVES, 129..256 rows; see launch_bwd_rs):
// the caller then makes two calls per chunk, flags ARCVAE_RS_HALF and ARCVAE_RS_HALF | ARCVAE_RS_HALF1.
extern "C" int arcvae_enc_lstm_bwd_rs_halves(int B, int T, int H, int L) {
    return (arcvae_env_int("ARCVAE_PERSIST", 1) != 0 && arcvae_env_int("ARCVAE_RS_HALVES", 0) != 0 && H == 256 && L >= 1 && L <= 2 &&
            B > 128 && B <= 256 && T >= 1 && !persist_two_groups(B, H, L, 2)) ? 1 : 0;
}
extern "C" int arcvae_enc_lstm_bwd_rs_ok(int B, int T, int H, int L) {
    // ARCVAE_RS_MAX_B
    return (arcvae_env_int("ARCVAE_PERSIST", 1) != 0 && H == 256 && L >= 1 && L <= 2 && B >= 1 &&
            (B <= arcvae_env_int("ARCVAE_RS_MAX_B", 128) || persist_two_groups(B, H, L, 2) ||
             arcvae_enc_lstm_bwd_rs_halves(B, T, H, L) == 1) && B <= 256 && T >= 1) ? 1 : 0;
}

// 2 if the persistent sweeps of this shape run in their two-group form (two blocks per CU, the XCD's 17..32 rows as two
// independent 16-row recurrences: H = 256, L <= 2, 129 <= B <= 256), else 1.  A two-group step re-arms 4352 words of
// sync_ws (arcvae_enc_prologue n_sync), a one-group step 848.
// floats of part_ws the reduce-scatter sweep of this shape uses: groups of 8 rows per XCD (1 / 2 / 4; the two-group form: 2 x 2)
// x two tick parities x (2L-1) slots x [8 XCDs][32 consumers][32 producers][64]
static inline long bwd_rs_part_floats(int B, int H, int L) {
    const int rows_x = ceil_div(B, 8);
    const int rgn = persist_two_groups(B, H, L, 2) ? 4 : (rows_x <= 8 ? 1 : (rows_x <= 16 ? 2 : 4));
    return (long)rgn * 2 * (2 * L - 1) * 8 * 32 * 32 * 64;
}
// Floats of part_ws (the partial sums in flight) arcvae_enc_lstm_backward_persistent_rs / _fused use for this shape under the
// current knobs, or 0 where that sweep does not run.
extern "C" long arcvae_enc_lstm_bwd_rs_part_floats(int B, int T, int H, int L) {
    return arcvae_enc_lstm_bwd_rs_ok(B, T, H, L) == 1 ? bwd_rs_part_floats(B, H, L) : 0;
}

extern "C" int arcvae_enc_lstm_persist_groups(int B, int H, int L) {
    return (arcvae_env_int("ARCVAE_PERSIST", 1) != 0 && B > 0 && persist_two_groups(B, H, L, 3)) ? 2 : 1;
}

// Reduce-scatter form of the persistent BPTT sweep (lstm_bwd_persist_rs_kernel): H = 256, L <= 2, B <= 256.
//   Wx / Wh: HOST arrays of the row-major weights (as arcvae_enc_lstm_backward);  part_ws: RG * 2 * (2L-1) * 8 * 32 * 32 * 64
//   floats of scratch for the partial sums in flight, RG = 1 / 2 / 4 for B <= 64 / 128 / 256.  Everything else as arcvae_enc_lstm_backward_persistent.
// fused != null: the FW variant (weight gradients accumulated in the kernel, see lstm_bwd_persist_rs_kernel).
namespace {
struct FusedWgrad {
    const float* hseq; const int32_t* x_tb; float* const* dWx; float* const* dWh; float* const* dbias; float* dtable; int V;
};
int launch_bwd_rs(const float* const* Wx, const float* const* Wh, const float* cseq, const float* gseq,
                  const float* dh_top, int ld_dh_top, float* dG, float* dcs, float* dxs, float* part_ws, long part_ws_floats,
                  unsigned* sync_ws, unsigned* start_signal, int B, int T, int H, int L, int s_begin, int s_end,
                  int chunk_index, unsigned long long* trace, const FusedWgrad* fused, int flags, hipStream_t stream) {
    if (!Wx || !Wh || !cseq || !gseq || !dh_top || !dG || !dcs || !dxs || !part_ws || !sync_ws) return ARCVAE_ERR_ARG;
    if (H != 256 || L < 1 || L > 2 || B < 1 || B > 256 || T < 1 || ld_dh_top < H) return ARCVAE_ERR_ARG;
    if (arcvae_env_int("ARCVAE_PERSIST", 1) == 0) return ARCVAE_ERR_ARG;
    if (part_ws_floats < bwd_rs_part_floats(B, H, L)) return ARCVAE_ERR_ARG;   // (the form is re-decided at every call: refuse, never overrun)
    // Half-batch form (round 4; flags bit 4, bit 5 = which half): 129..256 rows as TWO sweeps of 16 rows per XCD -- this call walks
    // rows 32 x + 16 half + [0, 16) of XCD x, the other half is a second call behind it on the same stream (its sync words: the
    // two-group BPTT's, unused in this form).  A 16-row tick is 4.0 us against 9.7 for 32 rows in one block.
    const bool halves = (flags & ARCVAE_RS_HALF) != 0;
    const int half = (flags & ARCVAE_RS_HALF1) ? 1 : 0;
    if (halves && (fused || B <= 128 || (flags & ARCVAE_PERSIST_BF16))) return ARCVAE_ERR_ARG;
    const int rows_x = halves ? 16 : ceil_div(B, 8);                    // rows per XCD (of this call)
    // two-group form (17..32 rows per XCD): two blocks per CU, each group an independent 16-row recurrence (R16 tile form)
    const bool two = !halves && !fused && !(flags & ARCVAE_PERSIST_BF16) && persist_two_groups(B, H, L, 2);
    const int rgn = two ? 2 : (rows_x <= 8 ? 1 : (rows_x <= 16 ? 2 : 4));   // groups of 8 rows a tick walks (per block)
    if (fused && rgn != 1) return ARCVAE_ERR_ARG;
    const int S = T + 2 * (L - 1);
    if (s_begin < 0 || s_end > S || s_begin >= s_end) return ARCVAE_ERR_ARG;
    if (chunk_index < 0 || chunk_index >= 8 || (chunk_index == 0) != (s_begin == 0)) return ARCVAE_ERR_ARG;
    const int bwd_base = (halves && half) ? PS2_BWD_FLAGS : PS_BWD;       // where this sweep's flags / role counters live
    if (chunk_index == 0 && !(flags & 1)) {   // flags bit 0: the words were zeroed ahead of the step (arcvae_enc_prologue)
        int rc = two ? arcvae_zero(reinterpret_cast<float*>(sync_ws + PS2_BWD_FLAGS), 1, PS2_CU - PS2_BWD_FLAGS, PS2_CU - PS2_BWD_FLAGS, stream)
                     : arcvae_zero(reinterpret_cast<float*>(sync_ws + bwd_base), 1, PS_WORDS + 64, PS_WORDS + 64, stream);
        if (rc == ARCVAE_OK) rc = arcvae_zero(reinterpret_cast<float*>(sync_ws + PS3_DONE), 1, 512, 512, stream);   // the "gathered" words
        if (rc != ARCVAE_OK) return rc;
    }
    PersistRsArgs ar;
    PersistBwdArgs& a = ar.b;
    a.wT = nullptr; a.cseq = cseq; a.gseq = gseq; a.dh_top = dh_top; a.dG = dG; a.dcs = dcs; a.dxs = dxs;
    a.sync = sync_ws + bwd_base; a.err = sync_ws + PS_ERR; a.start_signal = start_signal; a.trace = trace;
    a.B = B; a.T = T; a.H = H; a.RX = rows_x; a.RS = arcvae_ring_slots(T); a.ld_dh_top = ld_dh_top;
    a.row_base = halves ? 16 * half : 0; a.row_xs = halves ? 32 : rows_x;
    a.s_begin = s_begin; a.s_end = s_end; a.prio = arcvae_step_prio();
    a.cnt_off = chunk_index == 0 ? 0 : (PS_WORDS - PS_CNT) + 8 * (chunk_index - 1);
    a.flags = a.sync + PS_FLAGS; a.cnt = a.sync + PS_CNT + a.cnt_off; a.cucnt = nullptr; a.stagger = 0;
    if (two) {
        a.stagger = arcvae_env_int("ARCVAE_PERSIST2_STAGGER_BWD", 200);
        a.flags = sync_ws + PS2_BWD_FLAGS; a.cnt = sync_ws + PS2_BWD_CNT + 16 * chunk_index;
        a.cucnt = arcvae_env_int("ARCVAE_PERSIST2_ASSIGN", 1) != 0 ? sync_ws + PS2_CU : nullptr;
    }
    if (L == 2) { ar.W[0] = Wh[1]; ar.W[1] = Wh[0]; ar.W[2] = Wx[1]; }
    else { ar.W[0] = Wh[0]; ar.W[1] = Wh[0]; ar.W[2] = Wh[0]; }
    for (int i = 0; i < 2 * L - 1; ++i) if (!ar.W[i]) return ARCVAE_ERR_ARG;
    ar.part = part_ws;
    ar.fw_dbg = arcvae_env_int("ARCVAE_FW_DEBUG", 0);
    // ARCVAE_RS_SINGLE=0: two parities also from 17 rows per XCD on (there they are 6 MB against the XCD's 4 MB of L2; the
    // single-buffered form, guarded by the "gathered" words, is the default: isolated tick 10.0 -> 7.9 us)
    const int single_env = arcvae_env_int("ARCVAE_RS_SINGLE", -1);
    const bool single = !fused && single_env != 0 && rows_x > 16;     // (compiled for the 32-row tile form only)
    ar.parmask = single ? 0 : 1;
    ar.done = sync_ws + PS3_DONE;
    ar.hseq = nullptr; ar.x_tb = nullptr; ar.dW[0] = ar.dW[1] = ar.dW[2] = nullptr; ar.dbias1 = nullptr; ar.dtable = nullptr; ar.V = 0;
    if (fused) {
        if (!fused->hseq || !fused->x_tb || !fused->dWx || !fused->dWh || !fused->dbias || !fused->dtable || fused->V < 1 ||
            fused->V > 128)
            return ARCVAE_ERR_ARG;
        ar.hseq = fused->hseq; ar.x_tb = fused->x_tb; ar.dtable = fused->dtable; ar.V = fused->V;
        ar.dW[0] = fused->dWh[L - 1];                                   // q = 0: Wh of the top layer
        if (L == 2) { ar.dW[1] = fused->dWx[1]; ar.dW[2] = fused->dWh[0]; ar.dbias1 = fused->dbias[1]; }
        else { ar.dW[1] = ar.dW[0]; ar.dW[2] = ar.dW[0]; }
        for (int i = 0; i < 2 * L - 1; ++i) if (!ar.dW[i]) return ARCVAE_ERR_ARG;
        if (L == 2 && !ar.dbias1) return ARCVAE_ERR_ARG;
        if (chunk_index == 0) {                                          // the token table accumulates over the sweep's chunks
            const int rc = arcvae_zero(fused->dtable, fused->V, 4 * H, 4 * H, stream);
            if (rc != ARCVAE_OK) return rc;
        }
    }
    // ARCVAE_RS_WREG=0: weight slices in LDS instead of registers.  Either way at least 81 KB of LDS: one block per CU.
    const bool wreg = arcvae_env_int("ARCVAE_RS_WREG", 1) != 0 || fused || rgn > 1;
    size_t lds = sizeof(float) * ((wreg ? 0 : (size_t)(2 * L - 1) * 256 * 36) + (size_t)rgn * (L * 16 * 32 + 128) + (fused ? 128 * 33 : 0));
    if (lds < persist_lds_floor()) lds = persist_lds_floor();
    if (two) lds = persist2_lds_bytes();      // exactly two blocks per CU
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL(kern, dim3(two ? 512 : 256), dim3(256), lds, stream, ar);
    };
    if (two) {
        if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 0, false, 2, 2>); else launch(lstm_bwd_persist_rs_kernel<2, true, 0, false, 2, 2>);
        return arcvae_launch_status();
    }
    // The tile form for 9..32 rows per XCD (ARCVAE_RS_R16=0: the 4x4x1 blocks walking groups of 8 rows): all rows of the XCD
    // in ONE pass of 16x16x4 tiles with full rows -- isolated tick 4.0 vs 5.4 us at 16 rows per XCD (bs 128)
    if (rgn >= 2 && !(flags & ARCVAE_PERSIST_BF16) && arcvae_env_int("ARCVAE_RS_R16", 1) != 0) {
        // (the 32-row form holds more than 256 registers per lane: two of its blocks can never share a SIMD, so it needs no
        // LDS floor to land one per CU -- and what it does not allocate stays with the GEMM blocks beside it;
        // ARCVAE_RS_R32_LDS_KB > 0 restores a floor)
        if (rgn == 4) {
            const size_t need = sizeof(float) * (size_t)rgn * (L * 16 * 32 + 128), kb = (size_t)arcvae_env_int("ARCVAE_RS_R32_LDS_KB", 0) * 1024;
            lds = need > kb ? need : kb;
        }
        if (rgn == 2) { if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 0, false, 2>); else launch(lstm_bwd_persist_rs_kernel<2, true, 0, false, 2>); }
        else if (single) { if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 0, false, 4, 1, true>); else launch(lstm_bwd_persist_rs_kernel<2, true, 0, false, 4, 1, true>); }
        else { if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 0, false, 4>); else launch(lstm_bwd_persist_rs_kernel<2, true, 0, false, 4>); }
        return arcvae_launch_status();
    }
    // ARCVAE_RS_MFMA: 1 (default) = 4x4x1 blocks, 0 = 16x16x4 tiles (registers only; the LDS variant always uses 16x16x4)
    const bool mf = (wreg && arcvae_env_int("ARCVAE_RS_MFMA", 1) != 0) || fused || rgn > 1;
    if ((flags & ARCVAE_PERSIST_BF16) && mf && !fused && rgn <= 2) {   // throughput mode: the 4x4x4 bf16 form of the same blocks
        if (rgn == 2) {
            if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 2, false, 2>); else launch(lstm_bwd_persist_rs_kernel<2, true, 2, false, 2>);
        } else {
            if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 2>); else launch(lstm_bwd_persist_rs_kernel<2, true, 2>);
        }
    } else
    if (rgn == 2) {
        if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 1, false, 2>); else launch(lstm_bwd_persist_rs_kernel<2, true, 1, false, 2>);
    } else if (rgn == 4) {
        if (L == 1) launch(lstm_bwd_persist_rs_kernel<1, true, 1, false, 4>); else launch(lstm_bwd_persist_rs_kernel<2, true, 1, false, 4>);
    } else if (L == 1) {
        if (fused) launch(lstm_bwd_persist_rs_kernel<1, true, 1, true>);
        else if (mf) launch(lstm_bwd_persist_rs_kernel<1, true, 1>);
        else if (wreg) launch(lstm_bwd_persist_rs_kernel<1, true, 0>);
        else launch(lstm_bwd_persist_rs_kernel<1, false, 0>);
    } else {
        if (fused) launch(lstm_bwd_persist_rs_kernel<2, true, 1, true>);
        else if (mf) launch(lstm_bwd_persist_rs_kernel<2, true, 1>);
        else if (wreg) launch(lstm_bwd_persist_rs_kernel<2, true, 0>);
        else launch(lstm_bwd_persist_rs_kernel<2, false, 0>);
    }
    return arcvae_launch_status();
}
}  // namespace

extern "C" int arcvae_enc_lstm_backward_persistent_rs(const float* const* Wx, const float* const* Wh, const float* cseq,
                                                      const float* gseq, const float* dh_top, int ld_dh_top, float* dG,
                                                      float* dcs, float* dxs, float* part_ws, long part_ws_floats,
                                                      unsigned* sync_ws, unsigned* start_signal, int B, int T, int H, int L,
                                                      int s_begin, int s_end, int chunk_index, int flags,
                                                      unsigned long long* trace, hipStream_t stream) {
    return launch_bwd_rs(Wx, Wh, cseq, gseq, dh_top, ld_dh_top, dG, dcs, dxs, part_ws, part_ws_floats, sync_ws, start_signal, B, T, H, L,
                         s_begin, s_end, chunk_index, trace, nullptr, flags, stream);
}

// The same sweep with the weight gradients of the stack formed INSIDE it (FW variant of lstm_bwd_persist_rs_kernel):
// replaces arcvae_enc_lstm_backward_persistent_rs + the per-layer GEMMs, bias sums and token segment-sum of
// arcvae_enc_lstm_wgrad for the same tick range.  "+=" into dWh[l] (all l), dWx[l] and dbias[l] (l >= 1); the layer-0 input
// side arrives as dtable_ws [V,4H] (zeroed by the chunk_index == 0 call), to be folded by arcvae_table_finalize.
//   hseq [L,T,B,H], x_tb [T,B]: the forward sweep's outputs / inputs;  dWx, dWh, dbias: HOST arrays of device pointers.
extern "C" int arcvae_enc_lstm_backward_fused(const float* const* Wx, const float* const* Wh, const float* cseq,
                                              const float* gseq, const float* hseq, const int32_t* x_tb,
                                              const float* dh_top, int ld_dh_top, float* dG, float* dcs, float* dxs,
                                              float* part_ws, long part_ws_floats, unsigned* sync_ws, unsigned* start_signal,
                                              float* const* dWx, float* const* dWh, float* const* dbias,
                                              float* dtable_ws, int B, int T, int V, int H, int L, int s_begin, int s_end,
                                              int chunk_index, unsigned long long* trace, hipStream_t stream) {
    FusedWgrad f;
    f.hseq = hseq; f.x_tb = x_tb; f.dWx = dWx; f.dWh = dWh; f.dbias = dbias; f.dtable = dtable_ws; f.V = V;
    return launch_bwd_rs(Wx, Wh, cseq, gseq, dh_top, ld_dh_top, dG, dcs, dxs, part_ws, part_ws_floats, sync_ws, start_signal, B, T, H, L,
                         s_begin, s_end, chunk_index, trace, &f, 0, stream);
}

// BPTT for the stack.  Only h_{T-1} of the top layer receives an external gradient
// (models/encoder.py:106).  Produces dG [L,T,B,4H] (pre-activation gate gradients); weight
// gradients are formed from dG by arcvae_enc_lstm_wgrad.
//   dcs, dxs  workspaces [L,RS,B,H];  dG_t workspace [L,RS,B*4H] (k-chunk-major copy of dG): rings indexed by t % RS,
//   RS = min(T, 16) -- each slab is produced by one launch and consumed by the next only.  dG may alias gseq (the gate
//   gradients of (l,t) then overwrite the saved gates of (l,t), which the same thread has just read);  wT workspace
//   [(2L-1),H*4H] (k-chunk-major Wh_l^T, Wx_l^T copies, refreshed when s_begin == 0 and retile != 0).
//   start_signal (optional): device word that the FIRST launch of this call bumps by 1 when it starts, i.e. once all
//   earlier work of the stream (the previous sub-range) is complete: engine.Gates' chunk signal without its own launch.  The sweep is T + 2(L-1) dependent launches; [s_begin, s_end) selects a sub-range.
//   Schedule: cell(l,t) at launch (T-1-t) + 2(L-1-l); xproj_l(t) (dX_l[t] = dG^{l+1}_t . Wx_{l+1}) one launch earlier.
//   After launches [0, s_end) every layer has finished all t >= T - s_end + 2(L-1).
extern "C" int arcvae_enc_lstm_backward(const float* const* Wx, const float* const* Wh, const float* cseq,
                                        const float* gseq, const float* dh_top, int ld_dh_top, float* dG,
                                        float* dG_t, float* dcs, float* dxs, float* wT, int B, int T, int H,
                                        int L, int s_begin, int s_end, int flags, const long* ws_floats, void* dG_oct,
                                        unsigned* start_signal, unsigned long long* trace, hipStream_t stream) {
    const int retile = flags & ARCVAE_LSTM_RETILE;
    const bool b16 = bwd_bf16(B, H, L, flags), s3 = bwd_split3(B, H, L, flags);
    const bool oct = b16 && dG_oct && (B % 16) == 0;
    if (!Wx || !Wh || !cseq || !gseq || !dh_top || !dG || !dG_t || !dcs || !dxs || !wT || !ws_floats) return ARCVAE_ERR_ARG;
    if (B <= 0 || T <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H) || ld_dh_top < H)
        return ARCVAE_ERR_ARG;
    {   // (as in arcvae_enc_lstm_forward: an undersized ring or weight copy is an argument error, never an overrun)
        long need[4];
        lstm_ws_need(B, T, H, L, flags, need);
        if (ws_floats[1] < need[1] || ws_floats[3] < need[3]) return ARCVAE_ERR_ARG;
    }
    const int S = T + 2 * (L - 1);
    if (s_begin < 0 || s_end > S || s_begin >= s_end) return ARCVAE_ERR_ARG;
    const long sH = (long)B * H, sG = (long)B * 4 * H;
    const long lH = (long)T * sH, lG = (long)T * sG;
    const long wsz = s3 ? (long)H * 4 * H * 3 / 2 : (long)H * 4 * H;     // (three-piece form: three bf16 planes per copy)
    const long sGt = s3 ? sG * 3 / 2 : sG;
    // tiled transposed weight copies: WhT[l] at wT + l*wsz, WxT[l] (l>=1) at wT + (L + l - 1)*wsz
    if (s_begin == 0 && retile) {
        const float* src[16];
        float* dst[16];
        int cols[16], mode[16];
        int n = 0;
        for (int l = 0; l < L; ++l) {
            if (!Wh[l] || (l > 0 && !Wx[l])) return ARCVAE_ERR_ARG;
            src[n] = Wh[l]; dst[n] = wT + l * wsz; cols[n] = H; mode[n] = s3 ? 5 : (b16 ? 3 : 1); ++n;
            if (l > 0) { src[n] = Wx[l]; dst[n] = wT + (L + l - 1) * wsz; cols[n] = H; mode[n] = s3 ? 5 : (b16 ? 3 : 1); ++n; }
        }
        const int rc = arcvae_tile_weights(src, dst, cols, mode, n, H, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    const int tile_mt = bwd_tile_choice(B, H, L, flags);
    const bool step2 = !tile_mt && choose_step2(B);
    const int RS = arcvae_ring_slots(T), RSo = operand_ring_slots(B, T, H, L, flags);   // dc / dX rings, operand-plane ring
    for (int s = s_begin; s < s_end; ++s) {
        BwdArgs a;
        a.B = B; a.H = H; a.prio = arcvae_step_prio(); a.remap = arcvae_xcd_remap() | (arcvae_env_int("ARCVAE_TILE_XCD", 0) != 0 ? 2 : 0); a.trace = trace ? trace + 2 * (long)s : nullptr;
        a.dbg = arcvae_env_int("ARCVAE_TILE_DEBUG", 0);
        a.signal = (s == s_begin) ? start_signal : nullptr;
        int nj = 0;
        for (int l = L - 1; l >= 0; --l) {
            const int skew = 2 * (L - 1 - l);
            const int t = T - 1 - (s - skew);          // cell(l, t)
            if (t >= 0 && t < T) {
                BwdJob& j = a.job[nj++];
                const bool top = (l == L - 1), last = (t == T - 1);
                j.kind = 0;
                j.src = last ? nullptr : dG_t + ((long)l * RSo + ((t + 1) % RSo)) * sGt;
                j.WT = wT + l * wsz;
                if (top) { j.ext = last ? dh_top : nullptr; j.ext_ld = ld_dh_top; }
                else { j.ext = dxs + ((long)l * RS + (t % RS)) * sH; j.ext_ld = H; }
                j.gates = gseq + l * lG + t * sG;
                j.c = cseq + l * lH + t * sH;
                j.cprev = t > 0 ? cseq + l * lH + (t - 1) * sH : nullptr;
                j.dcin = last ? nullptr : dcs + ((long)l * RS + ((t + 1) % RS)) * sH;
                j.dcout = dcs + ((long)l * RS + (t % RS)) * sH;
                j.out = dG + l * lG + t * sG;
                j.outt = dG_t + ((long)l * RSo + (t % RSo)) * sGt;
                j.oct = oct ? static_cast<char*>(dG_oct) + 2 * (l * lG + t * sG) : nullptr;
            }
            const int tx = T - 1 - (s + 1 - skew);     // xproj_l(tx): feeds cell(l, tx) at the next launch
            if (l < L - 1 && tx >= 0 && tx < T) {
                BwdJob& j = a.job[nj++];
                j.kind = 1;
                j.src = dG_t + ((long)(l + 1) * RSo + (tx % RSo)) * sGt;
                j.WT = wT + (L + l) * wsz;             // WxT[l+1]
                j.ext = nullptr; j.ext_ld = H;
                j.gates = nullptr; j.c = nullptr; j.cprev = nullptr; j.dcin = nullptr; j.dcout = nullptr;
                j.out = dxs + ((long)l * RS + (tx % RS)) * sH;
                j.outt = nullptr; j.oct = nullptr;
            }
        }
        if (nj == 0) continue;
        for (int k = nj; k < ARCVAE_MAX_BWD_JOBS; ++k) a.job[k] = a.job[0];
        if (tile_mt) {
            dim3 tgrid(ceil_div(H, 128), ceil_div(B, 16 * tile_mt), nj);
            // ARCVAE_BWD_KSPLIT (default 1): the K-split 64 x 64 form where its grid fills the chip
            if (b16 && tile_mt == 4 && arcvae_env_int("ARCVAE_BWD_KSPLIT", 1) != 0 && (H / 64) * ceil_div(B, 64) * nj >= 200) {
                // (ARCVAE_BF16_KS_QUAD, default 1: the K-split tile with the four-unit epilogue of the three-piece form; 0: the
                // round-2 kernel with one (row, unit) per lane and element)
                if (arcvae_env_int("ARCVAE_BF16_KS_QUAD", 1) != 0 && (ld_dh_top % 4) == 0 && (reinterpret_cast<uintptr_t>(dh_top) % 16) == 0)
                    hipLaunchKernelGGL(lstm_bwd_tile_ks3_kernel<1>, dim3(H / 64, ceil_div(B, 64), nj), dim3(256), 64 * 1024, stream, a);
                else
                hipLaunchKernelGGL(lstm_bwd_tile_ks_kernel, dim3(H / 64, ceil_div(B, 64), nj), dim3(256), 48 * 1024, stream, a);
                continue;
            }
            if (s3) {
                // ARCVAE_BWD_KSPLIT3 (default 1; 2 forces it: tests): the K-split 64 x 64 three-piece form where its grid fills the chip (its
                // 16-byte epilogue accesses need dh_top on a 16-byte grid)
                const int ks3 = arcvae_env_int("ARCVAE_BWD_KSPLIT3", 1);   // (read per call: tests toggle it)
                if (ks3 != 0 && (ks3 == 2 || (H / 64) * ceil_div(B, 64) * nj >= tile_min_blocks(flags)) && (ld_dh_top % 4) == 0 &&
                    (reinterpret_cast<uintptr_t>(dh_top) % 16) == 0) {
                    hipLaunchKernelGGL(lstm_bwd_tile_ks3_kernel<2>, dim3(H / 64, ceil_div(B, 64), nj), dim3(256), 64 * 1024, stream, a);
                    continue;
                }
                if (tile_mt == 4) launch_bwd_tile<4, 2>(a, tgrid, stream);
                else if (tile_mt == 2) launch_bwd_tile<2, 2>(a, tgrid, stream);
                else launch_bwd_tile<1, 2>(a, tgrid, stream);
                continue;
            }
            if (b16) {
                if (tile_mt == 4) launch_bwd_tile<4, 1>(a, tgrid, stream);
                else if (tile_mt == 2) launch_bwd_tile<2, 1>(a, tgrid, stream);
                else launch_bwd_tile<1, 1>(a, tgrid, stream);
                continue;
            }
            if (tile_mt == 4) launch_bwd_tile<4>(a, tgrid, stream);
            else if (tile_mt == 2) launch_bwd_tile<2>(a, tgrid, stream);
            else launch_bwd_tile<1>(a, tgrid, stream);
            continue;
        }
        if (step2) {
            dim3 grid2(ceil_div(H, 32), ceil_div(B, 32), nj);
            if (s3) { DISPATCH_CH(H, launch_bwd2s, a, grid2, stream) }
            else { DISPATCH_CH(H, launch_bwd2, a, grid2, stream) }
            continue;
        }
        dim3 grid(H / 16, ceil_div(B, 16), nj);
        DISPATCH_CH(H, launch_bwd, a, grid, stream)
    }
    return arcvae_launch_status();
}

// ---- the dense decoder's layers l >= 1 on the three-piece tile kernels (round 3) ---------------------------------------------
// models/decoder.py:134-188 with the reference's quirk (Q1/Q2): the decoder's recurrence is dead, every layer l >= 1 is a
// ZERO-STATE cell over R = B*V rows -- h_l = o . tanh(i . g) of W_l h_{l-1} + b_l -- i.e. an L-1 layer stack at T = 1, which is
// exactly what lstm_fwd_tile_kernel (cprev = null), lstm_bwd_tile_ks3_kernel (cprev = dcin = null) and wgrad_planes_kernel
// compute.  In the MFMA-bound regime (configs[2]: R = 40960 rows) the decoder's exact-f32 GEMMs were 11 ms of kernel time per
// step beside the forward sweep; here they run as three bf16 pieces per operand with fp32-class accuracy, every activation
// split once by the epilogue that produces it.  Workspace (floats), in this order:
//   hact_t [L][R*H*3/2] operand planes of every layer's h;  cseq [L-1][R*H];  dG_t [2][R*4H*3/2];  dc [R*H];
//   wt [L-1][4H*H*3/2] forward weight planes;  wT [L-1][H*4H*3/2] transposed weight planes
namespace {
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ X, __bf16* __restrict__ P, long R, int H) {
    // X [slots][R, H] f32 -> [slots][plane][H >> 5][R][32] bf16 (hi, mid, lo); a thread: four adjacent units of a row; blockIdx.y = slot
    X += (long)blockIdx.y * R * H;
    P += (long)blockIdx.y * 3 * R * H;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int hq = H >> 2;
    if (i >= R * hq) return;
    const long row = i / hq;
    const int unit = 4 * (int)(i - row * hq);
    const f32x4 v = *reinterpret_cast<const f32x4*>(X + row * H + unit);
    typedef __bf16 bf16x4_l __attribute__((ext_vector_type(4)));
    __bf16 pc[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) split3_bf16(v[k], pc[0][k], pc[1][k], pc[2][k]);
    __bf16* tp = P + ((long)(unit >> 5) * R + row) * 32 + (unit & 31);
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) *reinterpret_cast<bf16x4_l*>(tp + pz * R * H) = bf16x4_l{pc[pz][0], pc[pz][1], pc[pz][2], pc[pz][3]};
}
__global__ __launch_bounds__(256) void bf16_copies_from_f32_kernel(const float* __restrict__ X, __bf16* __restrict__ P,
                                                                   __bf16* __restrict__ O8, long R, int H) {
    // throughput mode: X [R, H] f32 -> the tile kernels' operand copy [H >> 5][R][32] and the octet-major copy [R >> 3][H][8]
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H) return;
    const long row = i / H;
    const int unit = (int)(i - row * H);
    const __bf16 v = to_bf16(X[i]);
    P[((long)(unit >> 5) * R + row) * 32 + (unit & 31)] = v;
    O8[((row >> 3) * H + unit) * 8 + (row & 7)] = v;
}
struct DenseWs {
    float *hact_t, *cseq, *dG_t, *dc, *wt, *wT, *oct_h, *oct_g;   // (oct_*: throughput mode's octet-major bf16 copies)
    long sHt, sGt, wsz, total;
};
inline DenseWs dense_ws(float* base, long R, int H, int L) {
    DenseWs w;
    const long G = 4L * H;
    w.sHt = R * H * 3 / 2; w.sGt = R * G * 3 / 2; w.wsz = G * H * 3 / 2;
    long o = 0;
    w.hact_t = base + o; o += (long)L * w.sHt;
    w.cseq = base + o; o += (long)(L - 1) * R * H;
    w.dG_t = base + o; o += 2 * w.sGt;
    w.dc = base + o; o += R * H;
    w.wt = base + o; o += (long)(L - 1) * w.wsz;
    w.wT = base + o; o += (long)(L - 1) * w.wsz;
    w.oct_h = base + o; o += (long)L * R * H / 2;
    w.oct_g = base + o; o += R * G / 2;
    w.total = o;
    return w;
}
}  // namespace

// 1: layers 1 .. L-1 of a dense stack over R rows run on the tile kernels (grids that fill the chip, whole 32-row K-steps);
// ARCVAE_DENSE_TILED: 1 auto (default), 0 never, 2 whenever the shape allows (tests).
extern "C" int arcvae_dense_stack_ok(long R, int H, int L) {
    const int mode = arcvae_env_int("ARCVAE_DENSE_TILED", 1);
    if (mode == 0 || L < 2 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H) || R <= 0 || R > 0x3fffffff || (R % 32) != 0) return 0;
    if (mode == 2) return 1;
    return (R / 64) * (H / 64) >= 512 ? 1 : 0;
}
extern "C" int arcvae_dense_stack_ws_floats(long R, int H, int L, long* floats) {
    if (R <= 0 || H <= 0 || L < 2 || !floats) return ARCVAE_ERR_ARG;
    *floats = dense_ws(nullptr, R, H, L).total;
    return ARCVAE_OK;
}

// hact [L, R, H]: layer 0 given (dec_l0_fwd_kernel), layers 1 .. L-1 written;  gates [L-1, R, 4H] POST-activation i, f, g, o of
// layers 1 .. (what the backward below reads -- not the pre-activations arcvae_dec_forward_dense keeps);  Wx / bias: HOST arrays
// [L] of device pointers (entry 0 unused).
// flags bit 0: forward only (sampler) -- gates / cell states / the backward's weight planes / the top layer's planes are not written.
extern "C" int arcvae_dense_stack_forward(const float* const* Wx, const float* const* bias, float* hact, float* gates, float* ws,
                                          long ws_floats, long R, int H, int L, int flags, hipStream_t stream) {
    if (!Wx || !bias || !hact || !ws || !arcvae_dense_stack_ok(R, H, L)) return ARCVAE_ERR_ARG;
    if (ws_floats < dense_ws(nullptr, R, H, L).total) return ARCVAE_ERR_ARG;
    const bool fwd_only = (flags & 1) != 0, b16 = (flags & ARCVAE_LSTM_BF16) != 0;   // bit 1: throughput mode (bf16 operands)
    if ((!fwd_only && !gates) || (fwd_only && b16)) return ARCVAE_ERR_ARG;
    const int G = 4 * H, Ri = (int)R;
    const DenseWs w = dense_ws(ws, R, H, L);
    {   // weight planes, both layouts (arcvae_tile_weights modes 4 / 5)
        const float* src[2 * ARCVAE_MAX_LAYERS];
        float* dst[2 * ARCVAE_MAX_LAYERS];
        int cols[2 * ARCVAE_MAX_LAYERS], mode[2 * ARCVAE_MAX_LAYERS];
        int n = 0;
        for (int l = 1; l < L; ++l) {
            if (!Wx[l] || !bias[l]) return ARCVAE_ERR_ARG;
            src[n] = Wx[l]; dst[n] = w.wt + (l - 1) * w.wsz; cols[n] = H; mode[n] = b16 ? 2 : 4; ++n;
            if (!fwd_only) { src[n] = Wx[l]; dst[n] = w.wT + (l - 1) * w.wsz; cols[n] = H; mode[n] = b16 ? 3 : 5; ++n; }
        }
        const int rc = arcvae_tile_weights(src, dst, cols, mode, n, H, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    if (b16)
        hipLaunchKernelGGL(bf16_copies_from_f32_kernel, dim3((unsigned)((R * H + 255) / 256)), dim3(256), 0, stream, hact,
                           reinterpret_cast<__bf16*>(w.hact_t), reinterpret_cast<__bf16*>(w.oct_h), R, H);
    else
    hipLaunchKernelGGL(planes_from_f32_kernel, dim3((unsigned)((R * (H >> 2) + 255) / 256)), dim3(256), 0, stream, hact,
                       reinterpret_cast<__bf16*>(w.hact_t), R, H);
    for (int l = 1; l < L; ++l) {
        FwdArgs a;
        a.B = Ri; a.H = H; a.V = 1; a.prio = 0; a.remap = 0; a.trace = nullptr; a.dbg = 0;
        a.lite = fwd_only ? (l == L - 1 ? 3 : 1) : 0;
        FwdJob& j = a.job[0];
        j.xin = w.hact_t + (long)(l - 1) * w.sHt; j.Wx = w.wt + (l - 1) * w.wsz;
        j.hprev = nullptr; j.Wh = nullptr;
        j.pre = bias[l]; j.tok = nullptr; j.cprev = nullptr;
        j.h = hact + (long)l * R * H;
        j.ht = w.hact_t + (long)l * w.sHt;
        j.c = w.cseq + (long)(l - 1) * R * H;
        j.gates = gates ? gates + (long)(l - 1) * R * G : nullptr;
        j.oct = b16 ? reinterpret_cast<__bf16*>(w.oct_h) + (long)l * R * H : nullptr;
        for (int k = 1; k < ARCVAE_MAX_LAYERS; ++k) a.job[k] = a.job[0];
        if (b16) launch_fwd_tile<4, 4, 1>(a, Ri, H, 1, stream);
        else launch_fwd_tile<4, 4, 2>(a, Ri, H, 1, stream);
    }
    return arcvae_launch_status();
}

// Backward of the above from dh_top [R, H] (gradient of the top layer's h): dWx_l += dG_l^T h_{l-1}, dbias_l += colsum(dG_l)
// for l = L-1 .. 1 and dh0 [R, H] = dG_1 . Wx_1 (the input of dec_l0_bwd_kernel).  dG [R, 4H]: scratch (every layer's gate
// gradients pass through it).  `ws` as the forward left it.
extern "C" int arcvae_dense_stack_backward(const float* gates, const float* dh_top, float* dG, float* dh0, float* const* dWx,
                                           float* const* dbias, float* ws, long ws_floats, long R, int H, int L, int flags,
                                           hipStream_t stream) {
    if (!gates || !dh_top || !dG || !dh0 || !dWx || !dbias || !ws || !arcvae_dense_stack_ok(R, H, L)) return ARCVAE_ERR_ARG;
    if (ws_floats < dense_ws(nullptr, R, H, L).total) return ARCVAE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(dh_top) % 16) != 0) return ARCVAE_ERR_ARG;
    const int G = 4 * H, Ri = (int)R;
    const DenseWs w = dense_ws(ws, R, H, L);
    const bool b16 = (flags & ARCVAE_LSTM_BF16) != 0;       // throughput mode: the bf16 K-split tile + the octet weight-gradient kernel
    auto launch = [&](BwdArgs& a) {
        for (int k = 1; k < ARCVAE_MAX_BWD_JOBS; ++k) a.job[k] = a.job[0];
        if (b16) hipLaunchKernelGGL(lstm_bwd_tile_ks3_kernel<1>, dim3(H / 64, ceil_div(Ri, 64), 1), dim3(256), 64 * 1024, stream, a);
        else hipLaunchKernelGGL(lstm_bwd_tile_ks3_kernel<2>, dim3(H / 64, ceil_div(Ri, 64), 1), dim3(256), 64 * 1024, stream, a);
    };
    for (int l = L - 1; l >= 1; --l) {
        if (!dWx[l] || !dbias[l]) return ARCVAE_ERR_ARG;
        BwdArgs a;
        a.B = Ri; a.H = H; a.prio = 0; a.remap = 0; a.trace = nullptr; a.dbg = 0; a.signal = nullptr;
        BwdJob& j = a.job[0];
        j.kind = 0;
        const bool top = l == L - 1;
        j.src = top ? nullptr : w.dG_t + ((l + 1) & 1) * w.sGt;          // dh_l = dG_{l+1} . Wx_{l+1}
        j.WT = top ? nullptr : w.wT + (long)l * w.wsz;                    // (slot l = layer l + 1)
        j.ext = top ? dh_top : nullptr; j.ext_ld = H;
        j.gates = gates + (long)(l - 1) * R * G;
        j.c = w.cseq + (long)(l - 1) * R * H;
        j.cprev = nullptr; j.dcin = nullptr; j.dcout = w.dc;
        j.out = dG; j.outt = w.dG_t + (l & 1) * w.sGt; j.oct = b16 ? w.oct_g : nullptr;
        launch(a);
        if (b16) {   // dWx_l += dG_l^T . h_{l-1} from the octet-major copies, dbias_l += colsum(dG_l)
            const void* Ao = w.oct_g;
            const void* Bo = reinterpret_cast<const __bf16*>(w.oct_h) + (long)(l - 1) * R * H;
            float* Co = dWx[l];
            int rc = arcvae_wgrad_octet_group(1, G, H, &Ri, &Ao, &Bo, &Co, H, stream);
            if (rc) return rc;
            rc = arcvae_colsum_accum(dG, Ri, G, G, dbias[l], 1.0f, stream);
            if (rc) return rc;
            continue;
        }
        // dWx_l += dG_l^T . h_{l-1} from the planes (one time slot of R rows), dbias_l += colsum(dG_l)
        const void* Ap = w.dG_t + (l & 1) * w.sGt;
        const void* Bp = w.hact_t + (long)(l - 1) * w.sHt;
        const int zero = 0, one = 1;
        float* Cp = dWx[l];
        float* Sp = dbias[l];
        const int rc = arcvae_wgrad_planes_group(1, G, H, Ri, &Ap, &zero, &Bp, &zero, &one, &Cp, H, &Sp, stream);
        if (rc) return rc;
    }
    {   // dh0 = dG_1 . Wx_1
        BwdArgs a;
        a.B = Ri; a.H = H; a.prio = 0; a.remap = 0; a.trace = nullptr; a.dbg = 0; a.signal = nullptr;
        BwdJob& j = a.job[0];
        j.kind = 1;
        j.src = w.dG_t + (1 & 1) * w.sGt; j.WT = w.wT;
        j.ext = nullptr; j.ext_ld = H;
        j.gates = nullptr; j.c = nullptr; j.cprev = nullptr; j.dcin = nullptr; j.dcout = nullptr;
        j.out = dh0; j.outt = nullptr; j.oct = nullptr;
        launch(a);
    }
    return arcvae_launch_status();
}

// Weight gradients of the stack from dG over the time range [t_lo, t_hi) (all "+=" into the caller's
// gradient buffers), so that chunks can run on another stream while the BPTT sweep is still going:
//   l >= 1: dWx_l += dG_l[t]^T . hseq_{l-1}[t];  all l: dWh_l += dG_l[t]^T . hseq_l[t-1] (t >= 1);  dbias_l += colsum
//   l == 0: dTable0[v] += sum_{(t,b): x=v} dG_0[t,b]   (dtable_ws [V,4H], zeroed when `first` != 0)
//   when `last` != 0 (all ranges done): dEmb += dTable0 . Wx_0;  dWx_0 += dTable0^T . Emb;  dbias_0 += colsum(dTable0)
//   parts: bit 0 = the per-layer GEMMs (= bits 2 | 3), bit 1 = the layer-0 token-table path, bit 2 = only the dWx_l
//   GEMMs (l >= 1) and the bias column sums, bit 3 = only the dWh_l GEMMs (disjoint outputs: up to three streams);
//   bit 5 = onehot_ws already holds the one-hot rows (arcvae_enc_prologue);  bit 6 = 128-row split-bf16 tile (this range
//   runs behind the sweep: no sweep block is resident, the 322-register tile fits);
//   bit 4 = exact-f32 MFMA tile GEMMs instead of the split-bf16 kernel (45 instead of 208 registers per lane: what
//   fits on a SIMD beside a persistent sweep wave of more than 296 registers, i.e. the 2 / 4 row-group sweeps)
//   bit 11 = h_oct / dG_oct are the three-plane operand rings hseq_t / dG_t with all T slots: GEMMs from the planes
//   bit 12 (with 11) = those rings are scratch: this call first splits dG / h of its time range into them (mid-size batches)
//   onehot_ws [T*B, roundup(V,4)] workspace: one-hot token rows, written when `first` != 0 (token-table part)
extern "C" int arcvae_enc_lstm_wgrad(const int32_t* x_tb, const float* emb, const float* Wx0,
                                     const float* hseq, const float* dG, float* dtable_ws, float* onehot_ws,
                                     float* dEmb, float* const* dWx, float* const* dWh, float* const* dbias,
                                     int B, int T, int V, int E, int H, int L, int t_lo, int t_hi, int first,
                                     int last, int parts, const void* h_oct, const void* dG_oct, const long* ws_floats,
                                     hipStream_t stream) {
    if (!x_tb || !emb || !Wx0 || !hseq || !dG || !dtable_ws || !onehot_ws || !dEmb || !dWx || !dWh || !dbias)
        return ARCVAE_ERR_ARG;
    if (t_lo < 0 || t_hi > T || t_lo > t_hi) return ARCVAE_ERR_ARG;
    if ((parts & 2048) && h_oct && dG_oct) {   // plane rings with ALL T slots: [L,T,B*H*3/2] and [L,T,B*4H*3/2] floats
        if (!ws_floats || ws_floats[0] < (long)L * T * B * H * 3 / 2 || ws_floats[1] < (long)L * T * B * 4 * H * 3 / 2) return ARCVAE_ERR_ARG;
    }
    const int G = 4 * H, TB = T * B;
    const long lH = (long)TB * H, lG = (long)TB * G;
    int rc;
    const bool do_wx = (parts & (1 | 4)) != 0, do_wh = (parts & (1 | 8)) != 0, do_table = (parts & 2) != 0;
    const bool b16 = (parts & 128) != 0;                       // throughput mode: one bf16 product per GEMM step
    const bool exact_f32 = (parts & 16) != 0 && !b16, wide = (parts & 64) != 0;
    const bool do_layers = do_wx || do_wh;
    const int Vp = (V + 3) & ~3;
    if (do_table && first) {
        if (!(parts & 256))   // bit 8: the table workspace was zeroed ahead of the call (arcvae_enc_prologue)
        if (arcvae_zero(dtable_ws, V, G, G, stream) != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
        const long n = (long)TB * Vp;
        if (!(parts & 32))   // bit 5: the one-hot rows were written by arcvae_enc_prologue
        hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)min((long)1024, (n + 255) / 256)), dim3(256), 0, stream, x_tb,
                           TB, V, Vp, onehot_ws);
    }
    if (t_hi > t_lo) {
        const int nt = t_hi - t_lo;
        if (do_layers) {  // all per-layer weight-gradient GEMMs of this time range in ONE grouped launch
            const float* Ag[2 * ARCVAE_MAX_LAYERS];
            const float* Bg[2 * ARCVAE_MAX_LAYERS];
            float* Cg[2 * ARCVAE_MAX_LAYERS];
            float* Sg[2 * ARCVAE_MAX_LAYERS];    // bias gradient that rides with the problem (dWx_l: dbias_l), or null
            int Kg[2 * ARCVAE_MAX_LAYERS];
            int n = 0;
            const int t1 = t_lo > 1 ? t_lo : 1;  // dWh pairs dG[t] with h[t-1]
            for (int l = 0; l < L; ++l) {
                const float* dGl = dG + l * lG;
                if (do_wh && t_hi > t1) {
                    Ag[n] = dGl + (long)t1 * B * G; Bg[n] = hseq + l * lH + (long)(t1 - 1) * B * H;
                    Cg[n] = dWh[l]; Sg[n] = nullptr; Kg[n] = (t_hi - t1) * B; ++n;
                }
                if (do_wx && l > 0) {
                    Ag[n] = dGl + (long)t_lo * B * G; Bg[n] = hseq + (l - 1) * lH + (long)t_lo * B * H;
                    Cg[n] = dWx[l]; Sg[n] = dbias[l]; Kg[n] = nt * B; ++n;   // same rows as the bias sum: colsum(dG_l[t_lo..t_hi))
                }
            }
            if ((parts & 4096) && (parts & 2048) && h_oct && dG_oct) {
                // bit 12 (with bit 11): the plane rings are the CALLER's scratch -- the sweeps of this shape do not write planes (the
                // persistent / per-step-launch kernels of 256..512 rows): split dG[t_lo, t_hi) and the h slots they pair with
                // here, one elementwise pass (10 bytes per value), then the GEMMs from the planes as below
                if ((B % 32) != 0 || (H % 64) != 0) return ARCVAE_ERR_ARG;
                __bf16* hp = static_cast<__bf16*>(const_cast<void*>(h_oct));
                __bf16* gp = static_cast<__bf16*>(const_cast<void*>(dG_oct));
                const int h_lo = t_lo > 0 ? t_lo - 1 : 0;                         // dWh pairs dG[t] with h[t-1]
                // (a call that forms only the dWx_l or only the dWh_l splits what IT reads; where two calls on two streams split the
                // same slab they write the same bytes)
                for (int l = 0; l < L; ++l) {
                    if (do_wh || l > 0)
                        hipLaunchKernelGGL(planes_from_f32_kernel, dim3((unsigned)(((long)B * (G >> 2) + 255) / 256), nt), dim3(256), 0, stream,
                                           dG + l * lG + (long)t_lo * B * G, gp + 3 * (l * lG + (long)t_lo * B * G), (long)B, G);
                    const int a0 = do_wh ? h_lo : t_lo, a1 = (do_wx && l < L - 1) ? t_hi : t_hi - 1;   // slots of h_l this call reads
                    if (a1 > a0 && (do_wh || l < L - 1))
                        hipLaunchKernelGGL(planes_from_f32_kernel, dim3((unsigned)(((long)B * (H >> 2) + 255) / 256), a1 - a0), dim3(256), 0, stream,
                                           hseq + l * lH + (long)a0 * B * H, hp + 3 * (l * lH + (long)a0 * B * H), (long)B, H);
                }
            }
            if ((parts & 2048) && h_oct && dG_oct) {
                // bit 11: h_oct / dG_oct are the sweeps' three-plane operand rings with all T slots (arcvae_enc_lstm_operand_slots
                // == T): the weight gradients straight from the planes (gemm.hip: wgrad_planes_kernel)
                if ((B % 32) != 0 || (H % 64) != 0) return ARCVAE_ERR_ARG;
                const void* Ap[2 * ARCVAE_MAX_LAYERS];
                const void* Bp[2 * ARCVAE_MAX_LAYERS];
                int ta[2 * ARCVAE_MAX_LAYERS], tb[2 * ARCVAE_MAX_LAYERS], nts[2 * ARCVAE_MAX_LAYERS];
                for (int i = 0; i < n; ++i) {
                    const long ao = Ag[i] - dG, bo = Bg[i] - hseq;                  // element offsets: (layer, t) of each operand
                    const int la = (int)(ao / lG), lb = (int)(bo / lH);
                    ta[i] = (int)((ao - la * lG) / ((long)B * G)); tb[i] = (int)((bo - lb * lH) / ((long)B * H));
                    nts[i] = Kg[i] / B;
                    Ap[i] = static_cast<const char*>(dG_oct) + 2 * ((long)la * T * 3 * B * G);
                    Bp[i] = static_cast<const char*>(h_oct) + 2 * ((long)lb * T * 3 * B * H);
                }
                for (int i = 0; i < n; i += 8) {
                    // (the bias gradients ride in the GEMM: dbias_l += sum over rows of dG_l, a product with a column of ones)
                    rc = arcvae_wgrad_planes_group(n - i < 8 ? n - i : 8, G, H, B, Ap + i, ta + i, Bp + i, tb + i, nts + i, Cg + i, H,
                                                   Sg + i, stream);
                    if (rc) return rc;
                }
            } else if (b16 && h_oct && dG_oct && (B % 16) == 0) {
                // throughput mode with the octet-major bf16 copies the tiled sweeps left (same ranges, same targets)
                const void* Ao[2 * ARCVAE_MAX_LAYERS];
                const void* Bo[2 * ARCVAE_MAX_LAYERS];
                for (int i = 0; i < n; ++i) {
                    Ao[i] = static_cast<const char*>(dG_oct) + 2 * (Ag[i] - dG);
                    Bo[i] = static_cast<const char*>(h_oct) + 2 * (Bg[i] - hseq);
                }
                for (int i = 0; i < n; i += 8) {
                    rc = arcvae_wgrad_octet_group(n - i < 8 ? n - i : 8, G, H, Kg + i, Ao + i, Bo + i, Cg + i, H, stream);
                    if (rc) return rc;
                }
                for (int i = 0; i < n; ++i)      // (the octet kernel has no bias-sum rider)
                    if (Sg[i]) { rc = arcvae_colsum_accum(Ag[i], Kg[i], G, G, Sg[i], 1.0f, stream); if (rc) return rc; }
            } else
            for (int i = 0; i < n; i += 8) {
                rc = arcvae_gemm_tn_group_accum(n - i < 8 ? n - i : 8, G, H, Kg + i, Ag + i, G, Bg + i, H, Cg + i, H,
                                                (parts & 1024) ? 8 : (exact_f32 ? 0 : (1 | (wide ? 2 : 0) | (b16 ? 4 : 0))), Sg + i, stream);
                if (rc) return rc;       // (bias gradients dbias_l += colsum(dG_l): inside the split kernel, else by launch)
            }
        }
        if (do_table) {  // dTable0 += OneHot[rows]^T . dG_0[rows]   (see onehot_kernel)
            rc = arcvae_gemm_f32(1, 0, V, G, nt * B, onehot_ws + (long)t_lo * B * Vp, Vp, dG + (long)t_lo * B * G, G,
                                 dtable_ws, G, nullptr,
                                 ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK | (exact_f32 ? ARCVAE_GEMM_TILE64 : 0) |
                                     (wide ? ARCVAE_GEMM_TILE_WIDE : 0) | (b16 ? ARCVAE_GEMM_BF16 : 0), stream);
            if (rc) return rc;
        }
    }
    if (last && do_table)
        return arcvae_table_finalize(dtable_ws, Wx0, E, emb, dEmb, dWx[0], dbias[0], V, E, G, stream);
    return arcvae_launch_status();
}
