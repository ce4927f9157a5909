// Decoder path (reference models/decoder.py:113-190, losses/recon.py:29-64,
// models/decoder_sampling.py:48-128), evaluated VOCABULARY-DENSE.
//
// In the reference every decoder step calls each nn.LSTM with hidden=None/cell=None on a
// length-1 sequence (decoder.py:165-168), so no state is carried and z never reaches the LSTM
// (SURVEY Q1/Q2): logits_t is a pure function F(tok_t, cond_b).  There are only V distinct
// inputs per batch row, so instead of T dependent steps this file evaluates F for all B*V
// (row, token) pairs in one batched pass (V <= T at the default shape: fewer rows than B*T),
// and the whole autoregressive loop -- teacher-forcing coins, argmax feedback, EOS tracking --
// collapses into an integer table walk per row (dec_chain / dec_sample_chain).  The backward
// is dense over the same B*V rows: positions that fed the same token share one gradient row.
//
//   layer 0:  pre = TableD[v] + cond_b . Wc^T + bias_0   (TableD = emb . Wx_0[:, :E]^T, [V,4H])
//             c = i*g, h = o*tanh(c)        (zero-state cell: forget gate and Wh are dead)
//   layer l:  G_l = h_{l-1} . Wx_l^T + bias_l  (tile GEMM, M = B*V) -> zero-state cell
//   logits = h_{L-1} . Wout^T + bout ;  lse = logsumexp ; nxt = first argmax
#include "ops.h"

namespace {

#define MAXC 8

// ---- layer 0 forward: block = 64 units x 4 row lanes of one token v; a thread keeps its unit's three table entries and
// condition-weight columns in registers and walks up to 64 batch rows.  (The first form, a thread per (row, unit), read
// Wx0[j, E + c] -- stride E + C floats between neighbouring lanes -- 3 C times per OUTPUT: 248 us per launch at the sampler's
// bs 1024, 10x its bytes; profiles/r02_sampler_kernel_stats.csv.)  Same operation order as dec_l0_bwd_kernel's recomputation.
__global__ __launch_bounds__(256) void dec_l0_fwd_kernel(const float* __restrict__ tableD,
                                                         const float* __restrict__ cond,
                                                         const float* __restrict__ Wx0,
                                                         const float* __restrict__ bias0, float* h0, int B,
                                                         int V, int E, int C, int H) {
    const int ul = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int v = blockIdx.y, unit = blockIdx.x * 64 + ul;
    if (unit >= H) return;
    const int ld = E + C;
    const int gsel[3] = {0, 2, 3};  // i, g, o (forget gate is unused by the zero-state cell)
    float base[3], bs[3], wc[3][MAXC];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = gsel[k] * H + unit;
        base[k] = tableD[(long)v * 4 * H + j];
        bs[k] = bias0[j];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) wc[k][c] = c < C ? Wx0[(long)j * ld + E + c] : 0.f;
    }
    const int bend = min(B, (int)(blockIdx.z + 1) * 64);
    for (int b = blockIdx.z * 64 + bl; b < bend; b += 4) {
        float pre[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t = base[k];
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) t += cond[b * C + c] * wc[k][c];
            pre[k] = t + bs[k];
        }
        const float i = sigmoidf_acc(pre[0]), g = tanhf(pre[1]), o = sigmoidf_acc(pre[2]);
        h0[((long)b * V + v) * H + unit] = o * tanhf(i * g);
    }
}

// ---- zero-state cell on pre-activations G [R,4H] ------------------------------------------
__global__ __launch_bounds__(256) void cell_zero_fwd_kernel(const float* __restrict__ G, float* h, long R, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * H) return;
    const int unit = (int)(idx % H);
    const long r = idx / H;
    const float* g4 = G + r * 4 * H + unit;
    const float i = sigmoidf_acc(g4[0]), g = tanhf(g4[2 * H]), o = sigmoidf_acc(g4[3 * H]);
    h[idx] = o * tanhf(i * g);
}

// dG (pre-activation gradients; forget-gate columns are exactly 0) from dh and the saved pre-activations.
__global__ __launch_bounds__(256) void cell_zero_bwd_kernel(const float* __restrict__ G,
                                                            const float* __restrict__ dh, float* dG, long R,
                                                            int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * H) return;
    const int unit = (int)(idx % H);
    const long r = idx / H;
    const float* g4 = G + r * 4 * H + unit;
    const float i = sigmoidf_acc(g4[0]), g = tanhf(g4[2 * H]), o = sigmoidf_acc(g4[3 * H]);
    const float tc = tanhf(i * g);
    const float d = dh[idx];
    const float dc = d * o * (1.f - tc * tc);
    float* o4 = dG + r * 4 * H + unit;
    o4[0] = dc * g * i * (1.f - i);
    o4[H] = 0.f;
    o4[2 * H] = dc * i * (1.f - g * g);
    o4[3 * H] = d * tc * o * (1.f - o);
}

// ---- layer 0 backward: block per (v, 64-unit slab); 4 batch lanes reduce over b ----------------
//   dTableD[v, j]      = sum_b dG0[b,v,j]
//   wcpart[v, j, c]    = sum_b dG0[b,v,j] * cond[b,c]
__global__ __launch_bounds__(256) void dec_l0_bwd_kernel(const float* __restrict__ tableD,
                                                         const float* __restrict__ cond,
                                                         const float* __restrict__ Wx0,
                                                         const float* __restrict__ bias0,
                                                         const float* __restrict__ dh0, float* dTableD,
                                                         float* wcpart, int B, int V, int E, int C, int H) {
    __shared__ float red[4][64][3 * (1 + MAXC)];
    const int ul = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int v = blockIdx.y, unit = blockIdx.x * 64 + ul;
    const int ld = E + C;
    const int gsel[3] = {0, 2, 3};
    float base[3], wc[3][MAXC];
    float s[3] = {0.f, 0.f, 0.f}, sc[3][MAXC];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { sc[k][c] = 0.f; wc[k][c] = 0.f; }
    if (unit < H) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = gsel[k] * H + unit;
            base[k] = tableD[(long)v * 4 * H + j];
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) wc[k][c] = Wx0[(long)j * ld + E + c];
        }
        for (int b = bl; b < B; b += 4) {
            float cb[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) cb[c] = c < C ? cond[b * C + c] : 0.f;
            float pre[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float t = base[k];
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) t += cb[c] * wc[k][c];
                pre[k] = t + bias0[gsel[k] * H + unit];
            }
            const float i = sigmoidf_acc(pre[0]), g = tanhf(pre[1]), o = sigmoidf_acc(pre[2]);
            const float tc = tanhf(i * g);
            const float d = dh0[((long)b * V + v) * H + unit];
            const float dc = d * o * (1.f - tc * tc);
            float dg3[3];
            dg3[0] = dc * g * i * (1.f - i);
            dg3[1] = dc * i * (1.f - g * g);
            dg3[2] = d * tc * o * (1.f - o);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s[k] += dg3[k];
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) sc[k][c] += dg3[k] * cb[c];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        red[bl][ul][k * (1 + MAXC)] = s[k];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) red[bl][ul][k * (1 + MAXC) + 1 + c] = sc[k][c];
    }
    __syncthreads();
    if (bl == 0 && unit < H) {
        float* dt = dTableD + (long)v * 4 * H;
        dt[H + unit] = 0.f;  // forget gate
        for (int c = 0; c < C; ++c) wcpart[((long)v * 4 * H + H + unit) * C + c] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int o = k * (1 + MAXC);
            const int j = gsel[k] * H + unit;
            dt[j] = (red[0][ul][o] + red[1][ul][o]) + (red[2][ul][o] + red[3][ul][o]);
            for (int c = 0; c < C; ++c)
                wcpart[((long)v * 4 * H + j) * C + c] = (red[0][ul][o + 1 + c] + red[1][ul][o + 1 + c]) +
                                                        (red[2][ul][o + 1 + c] + red[3][ul][o + 1 + c]);
        }
    }
}

// dWx0[j, E + c] += sum_v wcpart[v, j, c]   (one wave per (j,c): lanes stride over v)
__global__ __launch_bounds__(256) void dec_wc_reduce_kernel(const float* __restrict__ wcpart, float* dWx0, int V,
                                                            int G, int E, int C) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= G * C) return;
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += wcpart[(long)v * G * C + idx];
    s = wave_sum(s);
    if (lane == 0) {
        const int j = idx / C, c = idx % C;
        dWx0[(long)j * (E + C) + E + c] += s;
    }
}

// ---- per-row statistics of the dense logits: lse and the fed-back token ---------------------
// mode 0 (training, decoder.py:185): nxt = first argmax(logits).
// mode 1 (sampling, decoder_sampling.py:110-117): nxt = first argmax(softmax(logits / temperature)).
__global__ __launch_bounds__(256) void dec_rowstats_kernel(const float* __restrict__ logits, float* lse,
                                                           int32_t* nxt, long R, int V, int mode, float temp) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* x = logits + r * V;
    float m = -INFINITY;
    for (int w = lane; w < V; w += 64) m = fmaxf(m, x[w]);
    m = wave_max(m);
    float s = 0.f;
    for (int w = lane; w < V; w += 64) s += expf(x[w] - m);
    s = wave_sum(s);
    if (lane == 0) lse[r] = m + logf(s);
    // argmax (first maximal index) of the score the reference feeds back
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (mode == 0) {
        for (int w = lane; w < V; w += 64) {
            const float sc = x[w];
            if (sc > bv) { bv = sc; bi = w; }
        }
    } else {
        float ms = -INFINITY;
        for (int w = lane; w < V; w += 64) ms = fmaxf(ms, x[w] / temp);
        ms = wave_max(ms);
        float ss = 0.f;
        for (int w = lane; w < V; w += 64) ss += expf(x[w] / temp - ms);
        ss = wave_sum(ss);
        for (int w = lane; w < V; w += 64) {
            const float sc = expf(x[w] / temp - ms) / ss;
            if (sc > bv) { bv = sc; bi = w; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) nxt[r] = bi == 0x7fffffff ? 0 : bi;
}

// ---- training-time autoregressive walk (decoder.py:146-185) + CE row sums (recon.py) ---------
// Block per batch row.  fed[b,t] = token fed at step t: fed[b,0] = 0 (start = pad, Q4),
// fed[b,t+1] = coins[t] ? x[b,t] : nxt[b, fed[b,t]].   rowloss[b] = sum_t (lse - logit[target]).
__global__ __launch_bounds__(64) void dec_chain_kernel(const int32_t* __restrict__ x,
                                                       const uint8_t* __restrict__ coins,
                                                       const int32_t* __restrict__ nxt,
                                                       const float* __restrict__ lse,
                                                       const float* __restrict__ logits, int32_t* fed,
                                                       float* rowloss, int T, int V) {
    extern __shared__ int32_t sm[];
    int32_t* s_nxt = sm;          // [V]
    int32_t* s_x = sm + V;        // [T]
    int32_t* s_fed = sm + V + T;  // [T]
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < V; i += 64) s_nxt[i] = min(max(nxt[(long)b * V + i], 0), V - 1);
    for (int t = lane; t < T; t += 64) {
        const int tk = min(max(x[(long)b * T + t], 0), V - 1);
        // teacher-forced steps carry the target itself, free-running steps carry -1
        s_x[t] = tk | (coins[t] ? 0 : (int32_t)0x40000000);
    }
    __syncthreads();
    if (lane == 0) {
        int cur = 0;
        for (int t = 0; t < T; ++t) {
            s_fed[t] = cur;
            const int e = s_x[t];
            cur = (e & 0x40000000) ? s_nxt[cur] : e;
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int t = lane; t < T; t += 64) {
        const int f = s_fed[t];
        const int tgt = s_x[t] & 0x3fffffff;
        fed[(long)b * T + t] = f;
        const long r = (long)b * V + f;
        acc += lse[r] - logits[r * V + tgt];
    }
    acc = wave_sum(acc);
    if (lane == 0) rowloss[b] = acc;
}

// ---- d(recon)/d(dense logits) ---------------------------------------------------------------
// recon = inv_count * sum_{b,t} CE(logits[b,fed[b,t]], x[b,t])  (inv_count = 1/(B_global*T), Q3)
// dlogits[b,v,w] = inv_count * (cnt[b,v] * softmax(logits[b,v])[w] - hist[b,v,w])
// PACK (vocabularies of 128..255 tokens): the V x V histogram as 16-bit counts, two per LDS word (a count is at most T <
// 65536, so a packed add never carries into its neighbour): 131 KB at V = 255 instead of 260 KB.
template <bool PACK>
__global__ __launch_bounds__(256) void dec_ce_bwd_kernel(const int32_t* __restrict__ x,
                                                         const int32_t* __restrict__ fed,
                                                         const float* __restrict__ logits,
                                                         const float* __restrict__ lse, float* dlogits, int T,
                                                         int V, float inv_count) {
    extern __shared__ int32_t sm[];
    const int nh = PACK ? (V * V + 1) / 2 : V * V;
    int32_t* hist = sm;          // [V*V] counts (PACK: 16 bits each)
    int32_t* cnt = sm + nh;      // [V]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nh + V; i += 256) sm[i] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const int f = min(max(fed[(long)b * T + t], 0), V - 1);
        const int tg = min(max(x[(long)b * T + t], 0), V - 1);
        const int i = f * V + tg;
        if (PACK) atomicAdd(&hist[i >> 1], 1 << (16 * (i & 1)));
        else atomicAdd(&hist[i], 1);
        atomicAdd(&cnt[f], 1);
    }
    __syncthreads();
    for (int i = tid; i < V * V; i += 256) {
        const int v = i / V;
        const long r = (long)b * V + v;
        const int cv = cnt[v];
        float d = 0.f;
        if (cv) {
            const int h = PACK ? (int)(((unsigned)hist[i >> 1] >> (16 * (i & 1))) & 0xffffu) : hist[i];
            d = ((float)cv * expf(logits[r * V + (i - v * V)] - lse[r]) - (float)h) * inv_count;
        }
        dlogits[r * V + (i - v * V)] = d;
    }
}

// out[b,t,:] = dense[b, fed[b,t], :]   (materialises the reference's [B,T,V] logits for the API)
__global__ __launch_bounds__(256) void dec_gather_logits_kernel(const float* __restrict__ dense,
                                                                const int32_t* __restrict__ fed, float* out,
                                                                long BT, int T, int V) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BT * V) return;
    const long bt = idx / V;
    const int w = (int)(idx - bt * V);
    const long b = bt / T;
    const int f = min(max(fed[bt], 0), V - 1);
    out[idx] = dense[(b * V + f) * V + w];
}

// ---- greedy sampler walk (decoder_sampling.py:85-123, Q9) ---------------------------------------
// tokens[b,t] = nxt[b, cur], cur <- tokens[b,t]; first_end[b] = first t with tokens[b,t] == end_token
// (max_len if none).  The host turns max_b(first_end)+1 into the reference's early-stop length.
__global__ __launch_bounds__(64) void dec_sample_chain_kernel(const int32_t* __restrict__ nxt, int32_t* tokens,
                                                              int32_t* first_end, int B, int V, int max_len,
                                                              int end_token) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int32_t* nr = nxt + (long)b * V;
    int cur = 0, fe = max_len;
    for (int t = 0; t < max_len; ++t) {
        cur = min(max(nr[cur], 0), V - 1);
        tokens[(long)b * max_len + t] = cur;
        if (cur == end_token && fe == max_len) fe = t;
    }
    first_end[b] = fe;
}

// ---- categorical sampler walk (round 4; the sampling the reference marks TODO, models/decoder_sampling.py:115-116: "for now, use
// argmax") -- an EXTENSION, not a parity path: tokens[b,t] ~ Categorical(softmax(logits[b, cur, :] / temperature)).  The decoder is
// stateless (Q1/Q2), so the dense pass has every distribution a row can ever meet; the walk is one wave per batch row: a lane holds
// up to four vocabulary entries, the row's cumulative distribution is a wave scan, the uniform number of step t comes from a
// counter-based generator keyed by (seed, row, t) (splitmix64 finaliser: reproducible for a seed whatever the launch shape), and the
// next token is the first entry whose cumulative mass exceeds u * total.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void dec_sample_categorical_kernel(const float* __restrict__ logits, int32_t* tokens,
                                                                    int32_t* first_end, int B, int V, int max_len,
                                                                    int end_token, float inv_temp, unsigned long long seed) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                   // wave-uniform
    int cur = 0, fe = max_len;
    for (int t = 0; t < max_len; ++t) {
        const float* row = logits + ((long)b * V + cur) * V;
        float x[4], mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * lane + e;
            x[e] = j < V ? row[j] * inv_temp : -3.0e38f;
            mx = fmaxf(mx, x[e]);
        }
        mx = wave_max(mx);
        float c[4], run = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * lane + e;
            run += j < V ? expf(x[e] - mx) : 0.f;
            c[e] = run;                                   // inclusive sums inside my four entries
        }
        float incl = run;                                 // inclusive scan of the lanes' totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const float total = __shfl(incl, 63, 64), before = incl - run;
        const unsigned long long r = mix64(mix64(seed ^ ((unsigned long long)b << 32)) + (unsigned long long)t);
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f) * total;     // 24 uniform bits in [0, 1)
        int pick = V - 1;                                 // (u * total == total cannot happen; the last entry is the guard)
#pragma unroll
        for (int e = 3; e >= 0; --e)
            if (4 * lane + e < V && before + c[e] > u) pick = 4 * lane + e;
        // first entry over the threshold = the smallest qualifying index over the wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pick = min(pick, __shfl_xor(pick, o, 64));
        cur = pick;
        if (lane == 0) tokens[(long)b * max_len + t] = cur;
        if (cur == end_token && fe == max_len) fe = t;
    }
    if (lane == 0) first_end[b] = fe;
}

inline int blocks_for(long n) { return (int)((n + 255) / 256); }

}  // namespace

// Dense decoder forward.  Pointer arrays Wx/bias are HOST arrays [L] of device pointers.
//   ws/out: tableD [V,4H], hact [L,B*V,H], gpre [max(L-1,1),B*V,4H], logits [B*V,V], lse/nxt [B*V]
extern "C" int arcvae_dec_forward_dense(const float* emb, const float* const* Wx, const float* const* bias,
                                        const float* Wout, const float* bout, const float* cond, float* tableD,
                                        float* hact, float* gpre, float* logits, float* lse, int32_t* nxt, int B,
                                        int V, int E, int C, int H, int L, int mode, float temperature,
                                        hipStream_t stream) {
    if (!emb || !Wx || !bias || !Wout || !bout || !cond || !tableD || !hact || !gpre || !logits || !lse || !nxt)
        return ARCVAE_ERR_ARG;
    if (B <= 0 || V <= 0 || E <= 0 || C < 0 || C > MAXC || H <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS)
        return ARCVAE_ERR_ARG;
    // the B*V-row products: throughput mode = bf16 operands; ARCVAE_DEC_SPLIT3 = three bf16 pieces, six products (fp32-class)
    const int BF = (mode & ARCVAE_DEC_BF16) ? ARCVAE_GEMM_BF16 : ((mode & ARCVAE_DEC_SPLIT3) ? ARCVAE_GEMM_SPLIT3 : 0);
    const bool no_gpre = (mode & ARCVAE_DEC_NO_GPRE) != 0 && BF == 0;   // forward only: fuse GEMM + cell (exact-f32 path)
    // ARCVAE_DEC_PART_HEAD: only the token table and layer 0; ARCVAE_DEC_PART_TAIL: only the logits and the row statistics -- the
    // caller runs layers 1 .. L-1 in between (arcvae_dense_stack_forward: the three-piece tile kernels)
    const bool head_only = (mode & ARCVAE_DEC_PART_HEAD) != 0, tail_only = (mode & ARCVAE_DEC_PART_TAIL) != 0;
    mode &= ~(ARCVAE_DEC_BF16 | ARCVAE_DEC_SPLIT3 | ARCVAE_DEC_NO_GPRE | ARCVAE_DEC_PART_HEAD | ARCVAE_DEC_PART_TAIL);
    if (mode == 1 && !(temperature > 0.f)) return ARCVAE_ERR_ARG;
    for (int l = 0; l < L; ++l)
        if (!Wx[l] || !bias[l]) return ARCVAE_ERR_ARG;
    const int G = 4 * H;
    const long R = (long)B * V;
    int rc = ARCVAE_OK;
    if (!tail_only) {
        rc = arcvae_gemm_f32(0, 1, V, G, E, emb, E, Wx[0], E + C, tableD, G, nullptr, 0, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(dec_l0_fwd_kernel, dim3(ceil_div(H, 64), V, ceil_div(B, 64)), dim3(256), 0, stream, tableD, cond, Wx[0],
                           bias[0], hact, B, V, E, C, H);
        if (head_only) return arcvae_launch_status();
    }
    for (int l = 1; l < L && !tail_only; ++l) {
        float* Gl = gpre + (long)(l - 1) * R * G;
        if (no_gpre && R >= 4096 &&
            arcvae_gemm_cell_zero((int)R, H, H, hact + (long)(l - 1) * R * H, H, Wx[l], H, bias[l], hact + (long)l * R * H,
                                  stream) == ARCVAE_OK)
            continue;                                      // (the pre-activations of this layer were never written)
        rc = arcvae_gemm_f32(0, 1, (int)R, G, H, hact + (long)(l - 1) * R * H, H, Wx[l], H, Gl, G, bias[l], BF,
                             stream);
        if (rc) return rc;
        hipLaunchKernelGGL(cell_zero_fwd_kernel, dim3(blocks_for(R * H)), dim3(256), 0, stream, Gl,
                           hact + (long)l * R * H, R, H);
    }
    rc = arcvae_gemm_f32(0, 1, (int)R, V, H, hact + (long)(L - 1) * R * H, H, Wout, H, logits, V, bout, BF, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_rowstats_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, logits, lse, nxt,
                       R, V, mode, temperature);
    return arcvae_launch_status();
}

// Teacher-forcing walk + CE row sums.  x [B,T] tokens, coins [T] bytes (1 = teacher forced, i.e.
// np.random.rand() < ratio at that step), outputs fed [B,T], rowloss [B].
extern "C" int arcvae_dec_chain_ce(const int32_t* x, const uint8_t* coins, const int32_t* nxt, const float* lse,
                                   const float* logits, int32_t* fed, float* rowloss, int B, int T, int V,
                                   hipStream_t stream) {
    if (!x || !coins || !nxt || !lse || !logits || !fed || !rowloss || B <= 0 || T <= 0 || V <= 0)
        return ARCVAE_ERR_ARG;
    const size_t lds = (size_t)(V + 2 * T) * sizeof(int32_t);
    if (lds > 64 * 1024) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(dec_chain_kernel, dim3(B), dim3(64), lds, stream, x, coins, nxt, lse, logits, fed, rowloss,
                       T, V);
    return arcvae_launch_status();
}

extern "C" int arcvae_dec_ce_backward(const int32_t* x, const int32_t* fed, const float* logits, const float* lse,
                                      float* dlogits, int B, int T, int V, float inv_count, hipStream_t stream) {
    if (!x || !fed || !logits || !lse || !dlogits || B <= 0 || T <= 0 || V <= 0) return ARCVAE_ERR_ARG;
    if (V > 255 || T > 65535) return ARCVAE_ERR_ARG;   // vocab_size <= 255 (the V x V LDS histogram; 16-bit counts from 128 on)
    const size_t lds = (size_t)(V * V + V) * sizeof(int32_t);
    if (lds <= 64 * 1024) {                            // V <= 127: 32-bit counts
        hipLaunchKernelGGL(dec_ce_bwd_kernel<false>, dim3(B), dim3(256), lds, stream, x, fed, logits, lse, dlogits, T, V,
                           inv_count);
        return arcvae_launch_status();
    }
    const size_t lds16 = (size_t)((V * V + 1) / 2 + V) * sizeof(int32_t);   // 131 KB at V = 255
    (void)hipFuncSetAttribute((const void*)dec_ce_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(dec_ce_bwd_kernel<true>, dim3(B), dim3(256), lds16, stream, x, fed, logits, lse, dlogits, T, V,
                       inv_count);
    return arcvae_launch_status();
}

extern "C" int arcvae_dec_gather_logits(const float* dense, const int32_t* fed, float* out, int B, int T, int V,
                                        hipStream_t stream) {
    if (!dense || !fed || !out || B <= 0 || T <= 0 || V <= 0) return ARCVAE_ERR_ARG;
    const long BT = (long)B * T;
    hipLaunchKernelGGL(dec_gather_logits_kernel, dim3(blocks_for(BT * V)), dim3(256), 0, stream, dense, fed, out,
                       BT, T, V);
    return arcvae_launch_status();
}

extern "C" int arcvae_dec_sample_chain(const int32_t* nxt, int32_t* tokens, int32_t* first_end, int B, int V,
                                       int max_len, int end_token, hipStream_t stream) {
    if (!nxt || !tokens || !first_end || B <= 0 || V <= 0 || max_len <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(dec_sample_chain_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, stream, nxt, tokens, first_end,
                       B, V, max_len, end_token);
    return arcvae_launch_status();
}

// Categorical sampling walk over the dense logits [B*V, V] of arcvae_dec_forward_dense (see dec_sample_categorical_kernel): an
// extension beyond the reference (its sampler is greedy, Q9); vocab_size <= 256, temperature > 0.
extern "C" int arcvae_dec_sample_chain_categorical(const float* logits, int32_t* tokens, int32_t* first_end, int B, int V,
                                                   int max_len, int end_token, float temperature, unsigned long long seed,
                                                   hipStream_t stream) {
    if (!logits || !tokens || !first_end || B <= 0 || V <= 0 || V > 256 || max_len <= 0 || !(temperature > 0.f)) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(dec_sample_categorical_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, logits, tokens, first_end, B, V,
                       max_len, end_token, 1.0f / temperature, seed);
    return arcvae_launch_status();
}

// Dense decoder backward from dlogits [B*V,V].  All parameter gradients are "+=".
//   ws: dh [2,B*V,H] ping-pong, dG [B*V,4H], dtableD [V,4H], wcpart [V,4H,max(C,1)]
//   dead parameters (Wh, z_to_hidden, condition_to_hidden) are never touched: their grads stay 0.
extern "C" int arcvae_dec_backward_dense(const float* emb, const float* const* Wx, const float* const* bias,
                                         const float* Wout, const float* cond, const float* tableD,
                                         const float* hact, const float* gpre, const float* dlogits, float* dh,
                                         float* dG, float* dtableD, float* wcpart, float* dEmb,
                                         float* const* dWx, float* const* dbias, float* dWout, float* dbout,
                                         int B, int V, int E, int C, int H, int L, int flags, hipStream_t stream) {
    if (!emb || !Wx || !bias || !Wout || !cond || !tableD || !hact || !gpre || !dlogits || !dh || !dG ||
        !dtableD || !wcpart || !dEmb || !dWx || !dbias || !dWout || !dbout)
        return ARCVAE_ERR_ARG;
    if (B <= 0 || V <= 0 || E <= 0 || C < 0 || C > MAXC || H <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS)
        return ARCVAE_ERR_ARG;
    const int G = 4 * H;
    const long R = (long)B * V;
    const int Ri = (int)R;
    const int BF = (flags & ARCVAE_DEC_BF16) ? ARCVAE_GEMM_BF16 : ((flags & ARCVAE_DEC_SPLIT3) ? ARCVAE_GEMM_SPLIT3 : 0);
    const int SK = ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK | (BF & ARCVAE_GEMM_BF16);   // (TN "+=": the split TN kernel either way)
    // ARCVAE_DEC_PART_TAIL: only fc_out's gradients and dh_top = dlogits . Wout (into dh[0 .. R*H)); ARCVAE_DEC_PART_HEAD: only
    // layer 0 and the token table, from dh_0 in dh[0 .. R*H) -- the caller runs layers L-1 .. 1 in between
    // (arcvae_dense_stack_backward)
    const bool head_only = (flags & ARCVAE_DEC_PART_HEAD) != 0, tail_only = (flags & ARCVAE_DEC_PART_TAIL) != 0;
    float* dhA = dh;
    float* dhB = dh + R * H;
    const float* hTop = hact + (long)(L - 1) * R * H;
    int rc = ARCVAE_OK;
    if (!head_only) {
        rc = arcvae_gemm_f32(1, 0, V, H, Ri, dlogits, V, hTop, H, dWout, H, nullptr, SK, stream);  // dWout += dL^T h
        if (rc) return rc;
        rc = arcvae_colsum_accum(dlogits, Ri, V, V, dbout, 1.0f, stream);
        if (rc) return rc;
        rc = arcvae_gemm_f32(0, 0, Ri, H, V, dlogits, V, Wout, H, dhA, H, nullptr, BF, stream);  // dh = dL Wout
        if (rc) return rc;
        if (tail_only) return arcvae_launch_status();
    }
    for (int l = L - 1; l >= 1 && !head_only; --l) {
        const float* Gl = gpre + (long)(l - 1) * R * G;
        hipLaunchKernelGGL(cell_zero_bwd_kernel, dim3(blocks_for(R * H)), dim3(256), 0, stream, Gl, dhA, dG, R, H);
        rc = arcvae_gemm_f32(1, 0, G, H, Ri, dG, G, hact + (long)(l - 1) * R * H, H, dWx[l], H, nullptr, SK, stream);
        if (rc) return rc;
        rc = arcvae_colsum_accum(dG, Ri, G, G, dbias[l], 1.0f, stream);
        if (rc) return rc;
        rc = arcvae_gemm_f32(0, 0, Ri, H, G, dG, G, Wx[l], H, dhB, H, nullptr, BF, stream);  // dh_{l-1} = dG Wx_l
        if (rc) return rc;
        float* t = dhA; dhA = dhB; dhB = t;
    }
    hipLaunchKernelGGL(dec_l0_bwd_kernel, dim3(ceil_div(H, 64), V), dim3(256), 0, stream, tableD, cond, Wx[0],
                       bias[0], dhA, dtableD, wcpart, B, V, E, C, H);
    if (C > 0)
        hipLaunchKernelGGL(dec_wc_reduce_kernel, dim3(ceil_div(G * C, 4)), dim3(256), 0, stream, wcpart, dWx[0],
                           V, G, E, C);
    // dEmb += dTableD . Wx0[:, :E];  dWx0[:, :E] += dTableD^T . Emb (ld E + C);  dbias_0 += colsum(dTableD)
    rc = arcvae_table_finalize(dtableD, Wx[0], E + C, emb, dEmb, dWx[0], dbias[0], V, E, G, stream);
    if (rc) return rc;
    return arcvae_launch_status();
}
