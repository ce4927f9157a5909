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

namespace {

struct FwdJob {
    const float* xin;    // [B,H]   h^{l-1}_t, or null for layer 0
    const float* Wx;     // [4H,H]
    const float* hprev;  // [B,H]   h^l_{t-1}, or null at t == 0 (MLX: hidden=None skips the term)
    const float* Wh;     // [4H,H]
    const float* pre;    // layer 0: table0 [V,4H] (bias folded in); else bias [4H]
    const int32_t* tok;  // layer 0: tokens of this step [B]; else null
    const float* cprev;  // [B,H] or null at t == 0 (MLX: cell=None -> c = i*g)
    float* h;            // [B,H]
    float* c;            // [B,H]
    float* gates;        // [B,4H] post-activation i,f,g,o (saved for BPTT)
};
struct FwdArgs {
    FwdJob job[ARCVAE_MAX_LAYERS];
    int B, H, V;
};

// CH = H / 64: each of the 4 waves owns H/4 = 16*CH floats of K per source.
template <int CH>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(FwdArgs a) {
    __shared__ float red[4 * 256];
    __shared__ float act[256];
    const FwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H;
    const int r0 = blockIdx.y * 16, u0 = blockIdx.x * 4;
    const int arow = min(r0 + (lane & 15), B - 1);
    const int jc = lane & 15;
    const long wrow = (long)(jc >> 2) * H + u0 + (jc & 3);  // gate-major weight row of tile column jc
    const bool s1 = j.xin != nullptr, s2 = j.hprev != nullptr;
    SkinnyFrag<CH> f1, f2;
    if (s1) skinny_load<CH>(f1, j.xin, (long)arow * H, j.Wx, wrow * H, wave, lane);
    if (s2) skinny_load<CH>(f2, j.hprev, (long)arow * H, j.Wh, wrow * H, wave, lane);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (s1) skinny_mfma<CH>(f1, acc0, acc1);
    if (s2) skinny_mfma<CH>(f2, acc0, acc1);
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    __syncthreads();
    {   // one gate activation per thread: (row, col) -> gate = col>>2, unit = u0 + (col&3)
        const int row = tid >> 4, col = tid & 15;
        const int b = r0 + row;
        if (b < B) {
            const int gcol = (col >> 2) * H + u0 + (col & 3);
            const float* pre = j.pre;
            if (j.tok) {
                int tk = j.tok[b];
                tk = min(max(tk, 0), a.V - 1);
                pre += (long)tk * 4 * H;
            }
            const float v = skinny_reduced_n<4>(red, row, col) + pre[gcol];
            const float av = ((col >> 2) == 2) ? tanhf(v) : sigmoidf_acc(v);
            act[tid] = av;
            j.gates[(long)b * 4 * H + gcol] = av;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int row = tid >> 2, u = tid & 3;
        const int b = r0 + row;
        if (b < B) {
            const float i = act[row * 16 + u], f = act[row * 16 + 4 + u], g = act[row * 16 + 8 + u],
                        o = act[row * 16 + 12 + u];
            const long hb = (long)b * H + u0 + u;
            const float c = j.cprev ? f * j.cprev[hb] + i * g : i * g;
            j.h[hb] = o * tanhf(c);
            j.c[hb] = c;
        }
    }
}

struct BwdJob {
    const float* dGup;    // [B,4H] dG^{l+1}_t or null (top layer)
    const float* WxTup;   // [H,4H] = Wx_{l+1}^T
    const float* dGnext;  // [B,4H] dG^l_{t+1} or null (t == T-1)
    const float* WhT;     // [H,4H] = Wh_l^T
    const float* dhext;   // [B,dhext_ld] external gradient (top layer, t == T-1) or null
    const float* gates;   // [B,4H] i,f,g,o at (l,t)
    const float* c;       // [B,H] c_t
    const float* cprev;   // [B,H] c_{t-1} or null (t == 0)
    const float* dcin;    // [B,H] dc_{t+1} * f_{t+1} or null (t == T-1)
    float* dcout;         // [B,H] dc_t * f_t
    float* dG;            // [B,4H] pre-activation gate gradients
    int dhext_ld;
    int pad;
};
struct BwdArgs {
    BwdJob job[ARCVAE_MAX_LAYERS];
    int B, H;
};

// 16 waves: wave w owns 4H/16 = 16*CH floats of the contraction index per source.
template <int CH>
__global__ __launch_bounds__(1024) void lstm_bwd_step_kernel(BwdArgs a) {
    __shared__ float red[16 * 256];
    const BwdJob& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H, G = 4 * a.H;
    const int r0 = blockIdx.y * 16, u0 = blockIdx.x * 16;
    const int arow = min(r0 + (lane & 15), B - 1);
    const long wrow = u0 + (lane & 15);
    const bool s1 = j.dGup != nullptr, s2 = j.dGnext != nullptr;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CH <= 4) {  // both sources' operands fit the 128-VGPR budget of a 1024-thread block
        SkinnyFrag<CH> f1, f2;
        if (s1) skinny_load<CH>(f1, j.dGup, (long)arow * G, j.WxTup, wrow * G, wave, lane);
        if (s2) skinny_load<CH>(f2, j.dGnext, (long)arow * G, j.WhT, wrow * G, wave, lane);
        if (s1) skinny_mfma<CH>(f1, acc0, acc1);
        if (s2) skinny_mfma<CH>(f2, acc0, acc1);
    } else {
        SkinnyFrag<CH> f;
        if (s1) {
            skinny_load<CH>(f, j.dGup, (long)arow * G, j.WxTup, wrow * G, wave, lane);
            skinny_mfma<CH>(f, acc0, acc1);
        }
        if (s2) {
            skinny_load<CH>(f, j.dGnext, (long)arow * G, j.WhT, wrow * G, wave, lane);
            skinny_mfma<CH>(f, acc0, acc1);
        }
    }
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    __syncthreads();
    if (tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        const int b = r0 + row;
        if (b < B) {
            const int unit = u0 + col;
            const long hb = (long)b * H + unit;
            float dh = skinny_reduced_n<16>(red, row, col);
            if (j.dhext) dh += j.dhext[(long)b * j.dhext_ld + unit];
            const float* gp = j.gates + (long)b * G + unit;
            const float i = gp[0], f = gp[H], g = gp[2 * H], o = gp[3 * H];
            const float tc = tanhf(j.c[hb]);
            const float d_o = dh * tc * o * (1.f - o);
            float dc = dh * o * (1.f - tc * tc);
            if (j.dcin) dc += j.dcin[hb];
            const float d_i = dc * g * i * (1.f - i);
            const float d_f = j.cprev ? dc * j.cprev[hb] * f * (1.f - f) : 0.f;
            const float d_g = dc * i * (1.f - g * g);
            j.dcout[hb] = dc * f;
            float* dp = j.dG + (long)b * G + unit;
            dp[0] = d_i;
            dp[H] = d_f;
            dp[2 * H] = d_g;
            dp[3 * H] = d_o;
        }
    }
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

// Reference: models/encoder.py:98-101 (L stacked nn.LSTM over the full padded sequence, Q3).
//   x_tb   [T,B] tokens (time-major)          table0 [V,4H] = emb . Wx_0^T + bias_0
//   Wx[l]  [4H,H] (l >= 1), Wh[l] [4H,H], bias[l] [4H] (l >= 1): HOST arrays of device pointers
//   hseq/cseq [L,T,B,H], gseq [L,T,B,4H] outputs (gseq = post-activation i,f,g,o)
//   hidden_dim: multiple of 64, <= 512.
extern "C" int arcvae_enc_lstm_forward(const int32_t* x_tb, const float* table0, const float* const* Wx,
                                       const float* const* Wh, const float* const* bias, float* hseq,
                                       float* cseq, float* gseq, int B, int T, int V, int H, int L,
                                       hipStream_t stream) {
    if (!x_tb || !table0 || !Wx || !Wh || !bias || !hseq || !cseq || !gseq) return ARCVAE_ERR_ARG;
    if (B <= 0 || T <= 0 || V <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H)) return ARCVAE_ERR_ARG;
    for (int l = 0; l < L; ++l)
        if (!Wh[l] || (l > 0 && (!Wx[l] || !bias[l]))) return ARCVAE_ERR_ARG;
    const long sH = (long)B * H, sG = (long)B * 4 * H;
    const long lH = (long)T * sH, lG = (long)T * sG;
    for (int s = 0; s < T + L - 1; ++s) {
        FwdArgs a;
        a.B = B; a.H = H; a.V = V;
        int nj = 0;
        for (int l = 0; l < L; ++l) {
            const int t = s - l;
            if (t < 0 || t >= T) continue;
            FwdJob& j = a.job[nj++];
            j.xin = l > 0 ? hseq + (l - 1) * lH + t * sH : nullptr;
            j.Wx = l > 0 ? Wx[l] : nullptr;
            j.hprev = t > 0 ? hseq + l * lH + (t - 1) * sH : nullptr;
            j.Wh = Wh[l];
            j.pre = l > 0 ? bias[l] : table0;
            j.tok = l > 0 ? nullptr : x_tb + (long)t * B;
            j.cprev = t > 0 ? cseq + l * lH + (t - 1) * sH : nullptr;
            j.h = hseq + l * lH + t * sH;
            j.c = cseq + l * lH + t * sH;
            j.gates = gseq + l * lG + t * sG;
        }
        for (int k = nj; k < ARCVAE_MAX_LAYERS; ++k) a.job[k] = a.job[0];
        dim3 grid(H / 4, ceil_div(B, 16), nj);
        DISPATCH_CH(H, launch_fwd, a, grid, stream)
    }
    return arcvae_launch_status();
}

// BPTT for the stack.  Only h_{T-1} of the top layer receives an external gradient
// (models/encoder.py:106).  Produces dG [L,T,B,4H] (pre-activation gate gradients); weight
// gradients are formed from dG by arcvae_enc_lstm_wgrad.
//   dcs  workspace [L,T,B,H];  wT workspace [(2L-1),H,4H] (transposed Wh_l, Wx_l copies, refreshed here)
extern "C" int arcvae_enc_lstm_backward(const float* const* Wx, const float* const* Wh, const float* cseq,
                                        const float* gseq, const float* dh_top, int ld_dh_top, float* dG,
                                        float* dcs, float* wT, int B, int T, int H, int L, hipStream_t stream) {
    if (!Wx || !Wh || !cseq || !gseq || !dh_top || !dG || !dcs || !wT) return ARCVAE_ERR_ARG;
    if (B <= 0 || T <= 0 || L <= 0 || L > ARCVAE_MAX_LAYERS || !hidden_ok(H) || ld_dh_top < H)
        return ARCVAE_ERR_ARG;
    const long sH = (long)B * H, sG = (long)B * 4 * H;
    const long lH = (long)T * sH, lG = (long)T * sG;
    const long wsz = (long)H * 4 * H;
    // transposed weight copies: WhT[l] at wT + l*wsz, WxT[l] (l>=1) at wT + (L + l - 1)*wsz
    {
        const float* src[16];
        float* dst[16];
        int rows[16], cols[16];
        int n = 0;
        for (int l = 0; l < L; ++l) {
            if (!Wh[l] || (l > 0 && !Wx[l])) return ARCVAE_ERR_ARG;
            src[n] = Wh[l]; dst[n] = wT + l * wsz; rows[n] = 4 * H; cols[n] = H; ++n;
            if (l > 0) { src[n] = Wx[l]; dst[n] = wT + (L + l - 1) * wsz; rows[n] = 4 * H; cols[n] = H; ++n; }
        }
        const int rc = arcvae_transpose_batched(src, dst, rows, cols, n, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    for (int s = 0; s < T + L - 1; ++s) {
        BwdArgs a;
        a.B = B; a.H = H;
        int nj = 0;
        for (int l = L - 1; l >= 0; --l) {
            const int t = T - 1 - (s - (L - 1 - l));
            if (t < 0 || t >= T) continue;
            BwdJob& j = a.job[nj++];
            const bool top = (l == L - 1), last = (t == T - 1);
            j.dGup = top ? nullptr : dG + (l + 1) * lG + t * sG;
            j.WxTup = top ? nullptr : wT + (L + l) * wsz;  // WxT[l+1]
            j.dGnext = last ? nullptr : dG + l * lG + (t + 1) * sG;
            j.WhT = wT + l * wsz;
            j.dhext = (top && last) ? dh_top : nullptr;
            j.gates = gseq + l * lG + t * sG;
            j.c = cseq + l * lH + t * sH;
            j.cprev = t > 0 ? cseq + l * lH + (t - 1) * sH : nullptr;
            j.dcin = last ? nullptr : dcs + l * lH + (t + 1) * sH;
            j.dcout = dcs + l * lH + t * sH;
            j.dG = dG + l * lG + t * sG;
            j.dhext_ld = ld_dh_top;
            j.pad = 0;
        }
        for (int k = nj; k < ARCVAE_MAX_LAYERS; ++k) a.job[k] = a.job[0];
        dim3 grid(H / 16, ceil_div(B, 16), nj);
        DISPATCH_CH(H, launch_bwd, a, grid, stream)
    }
    return arcvae_launch_status();
}

// Weight gradients of the stack from dG (all "+=" into the caller's gradient buffers):
//   l >= 1: dWx_l += dG_l^T . hseq_{l-1};   all l: dWh_l += dG_l[1:]^T . hseq_l[:-1];  dbias_l += colsum(dG_l)
//   l == 0: dTable0[v] = sum_{(t,b): x=v} dG_0[t,b]  ->  dEmb += dTable0 . Wx_0;  dWx_0 += dTable0^T . Emb;
//           dbias_0 += colsum(dTable0)
//   dtable_ws: workspace [V,4H]
extern "C" int arcvae_enc_lstm_wgrad(const int32_t* x_tb, const float* emb, const float* Wx0,
                                     const float* hseq, const float* dG, float* dtable_ws, float* dEmb,
                                     float* const* dWx, float* const* dWh, float* const* dbias, int B, int T,
                                     int V, int E, int H, int L, hipStream_t stream) {
    if (!x_tb || !emb || !Wx0 || !hseq || !dG || !dtable_ws || !dEmb || !dWx || !dWh || !dbias)
        return ARCVAE_ERR_ARG;
    const int G = 4 * H, TB = T * B;
    const long lH = (long)TB * H, lG = (long)TB * G;
    int rc;
    for (int l = 0; l < L; ++l) {
        const float* dGl = dG + l * lG;
        if (T > 1) {
            rc = arcvae_gemm_f32(1, 0, G, H, (T - 1) * B, dGl + (long)B * G, G, hseq + l * lH, H, dWh[l], H,
                                 nullptr, ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK, stream);
            if (rc) return rc;
        }
        if (l > 0) {
            rc = arcvae_gemm_f32(1, 0, G, H, TB, dGl, G, hseq + (l - 1) * lH, H, dWx[l], H, nullptr,
                                 ARCVAE_GEMM_ACCUMULATE | ARCVAE_GEMM_SPLITK, stream);
            if (rc) return rc;
            rc = arcvae_colsum_accum(dGl, TB, G, G, dbias[l], 1.0f, stream);
            if (rc) return rc;
        }
    }
    if (hipMemsetAsync(dtable_ws, 0, (size_t)V * G * sizeof(float), stream) != hipSuccess) return ARCVAE_ERR_LAUNCH;
    rc = arcvae_segsum_rows_accum(dG, x_tb, TB, V, G, dtable_ws, stream);
    if (rc) return rc;
    rc = arcvae_gemm_f32(0, 0, V, E, G, dtable_ws, G, Wx0, E, dEmb, E, nullptr, ARCVAE_GEMM_ACCUMULATE, stream);
    if (rc) return rc;
    rc = arcvae_gemm_f32(1, 0, G, E, V, dtable_ws, G, emb, E, dWx[0], E, nullptr, ARCVAE_GEMM_ACCUMULATE, stream);
    if (rc) return rc;
    return arcvae_colsum_accum(dtable_ws, V, G, G, dbias[0], 1.0f, stream);
}
