// Encoder heads, reparameterisation and the latent part of the loss
// (reference models/encoder.py:106-153, losses/kl.py:35-66, losses/info.py:23-78,
//  complete_vae_loss.py:45-99), forward and hand-written backward.
//
// The mutual-information term uses BATCH statistics, so the loss is computed in
// two halves with a reduction seam between them: `heads_forward` leaves per-rank partial sums
// in `stats` (2Z+4 floats; under data parallelism these are all-reduced), `latent_loss` turns
// the (global) sums into the loss scalars and d/d(mu_raw), d/d(logvar_raw) for the local rows.
//
// stats layout: [0,Z) sum_b mu   [Z,2Z) sum_b exp(logvar)   [2Z] sum_b KL_b (unclamped, MI form)
//               [2Z+1] sum_b KL_b (clamped >= 0, free bits)   [2Z+2] rows   [2Z+3] sum_b CE row sums
// hyper layout (device, so a captured graph follows the epoch schedules):
//               [0] beta [1] lambda_collapse [2] lambda_mi [3] target_mi [4] free_bits
// scalars out:  [0] total [1] recon [2] kl [3] beta*kl [4] collapse [5] prop(=0) [6] lambda_prop*prop(=0)
//               [7] mutual_info [8] mi_penalty [9] d(total)/d(mi_raw) [10] global rows
#include "ops.h"
#include "skinny.h"
#include "xcd.h"

namespace {

// comb[b, :H] = hT[b, :];  comb[b, H + j] = bc[j] + sum_c cond[b,c] * Wc[j,c]   (encoder.py:106-112)
__global__ __launch_bounds__(256) void build_comb_kernel(const float* __restrict__ hT,
                                                         const float* __restrict__ cond,
                                                         const float* __restrict__ Wc,
                                                         const float* __restrict__ bc, float* comb, float* stats,
                                                         int nstats, int B, int H, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nstats) stats[idx] = 0.f;  // zero the partial-sum seam here: one launch less on the critical chain
    if (idx >= B * 2 * H) return;
    const int b = idx / (2 * H), j = idx % (2 * H);
    float v;
    if (j < H) {
        v = hT[(long)b * H + j];
    } else {
        const int u = j - H;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += cond[b * C + c] * Wc[u * C + c];
        v = s + bc[u];
    }
    comb[idx] = v;
}

__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// mu = 2 tanh(mu_raw/2); logvar = tanh(lv_raw/2) - 1; z = mu + eps*exp(logvar/2); partial stats.
// Block = 4 batch rows (one per wave); lanes stride over Z.  Column sums go to stats with one f32
// atomic per (row, column) -- B adds per address, a few microseconds at B = 64..2048 -- and the
// per-row KL sums are wave-reduced first.
__global__ __launch_bounds__(256) void latent_apply_kernel(const float* __restrict__ mu_raw,
                                                           const float* __restrict__ lv_raw,
                                                           const float* __restrict__ eps, float* mu, float* logvar,
                                                           float* z, float* stats, int B, int Z, float fb_min) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float klmi = 0.f, klfb = 0.f;
    for (int j = lane; j < Z; j += 64) {
        const long i = (long)b * Z + j;
        const float m = tanhf(mu_raw[i] / 2.0f) * 2.0f;
        const float lv = tanhf(lv_raw[i] / 2.0f) * 1.0f - 1.0f;
        mu[i] = m;
        logvar[i] = lv;
        z[i] = m + eps[i] * expf(0.5f * lv);
        const float mc = clipf(m, -3.0f, 3.0f), lc = clipf(lv, -6.0f, 3.0f);
        const float var = expf(lc);
        const float k = -0.5f * (1.0f + lc - mc * mc - var);
        atomicAdd(stats + j, mc);
        atomicAdd(stats + Z + j, var);
        klmi += k;
        float kf = fmaxf(k, 0.0f);
        if (fb_min > 0.0f) kf = fmaxf(kf, fb_min);
        klfb += kf;
    }
    klmi = wave_sum(klmi);
    klfb = wave_sum(klfb);
    if (lane == 0) {
        atomicAdd(stats + 2 * Z, klmi);
        atomicAdd(stats + 2 * Z + 1, klfb);
        atomicAdd(stats + 2 * Z + 2, 1.0f);
    }
}

// lv_raw = lh . Wlv^T + blv (a 16 x 16 tile per block, K over the 4 waves: the skinny contraction) with latent_apply as
// its epilogue: the tile's mu / logvar / z and its share of the batch statistics -- one launch less on the step's
// critical chain (round 2).  The tile reduces its own sums first: 16 column atomics per quantity and two scalar
// atomics per block instead of one per element.
__global__ __launch_bounds__(256) void heads_lv_apply_kernel(const float* __restrict__ lh, const float* __restrict__ Wlv,
                                                             const float* __restrict__ blv, const float* __restrict__ mu_raw,
                                                             const float* __restrict__ eps, float* lv_raw, float* mu,
                                                             float* logvar, float* z, float* stats, int B, int Z, int K,
                                                             float fb_min) {
    __shared__ float red[4 * 256];
    __shared__ float cs[2][16][17];     // per-element terms of the column sums (mu, var)
    __shared__ float rs[2][4];          // per-wave partial sums of the two KL forms
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int arow = min(m0 + (lane & 15), B - 1), bcol = min(n0 + (lane & 15), Z - 1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    skinny_accum_kk(acc0, acc1, lh, (long)arow * K, Wlv, (long)bcol * K, K, wave, lane);
    skinny_store_partial(red, acc0, acc1, wave, lane);
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const int gr = m0 + row, gc = n0 + col;
    float klmi = 0.f, klfb = 0.f, mc_v = 0.f, var_v = 0.f;
    if (gr < B && gc < Z) {
        const long i = (long)gr * Z + gc;
        const float lvr = skinny_reduced(red, row, col) + blv[gc];
        lv_raw[i] = lvr;
        const float m = tanhf(mu_raw[i] / 2.0f) * 2.0f;
        const float lv = tanhf(lvr / 2.0f) * 1.0f - 1.0f;
        mu[i] = m;
        logvar[i] = lv;
        z[i] = m + eps[i] * expf(0.5f * lv);
        const float mc = clipf(m, -3.0f, 3.0f), lc = clipf(lv, -6.0f, 3.0f);
        const float var = expf(lc);
        const float k = -0.5f * (1.0f + lc - mc * mc - var);
        mc_v = mc; var_v = var;
        klmi = k;
        float kf = fmaxf(k, 0.0f);
        if (fb_min > 0.0f) kf = fmaxf(kf, fb_min);
        klfb = kf;
    }
    cs[0][row][col] = mc_v;
    cs[1][row][col] = var_v;
    klmi = wave_sum(klmi);
    klfb = wave_sum(klfb);
    if (lane == 0) { rs[0][wave] = klmi; rs[1][wave] = klfb; }
    __syncthreads();
    if (tid < 32) {                      // column sums of the tile: 16 columns x {mu, var}
        const int q = tid >> 4, c = tid & 15;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += cs[q][r][c];
        if (n0 + c < Z) atomicAdd(stats + q * Z + n0 + c, a);
    } else if (tid == 32) {
        atomicAdd(stats + 2 * Z, (rs[0][0] + rs[0][1]) + (rs[0][2] + rs[0][3]));
        atomicAdd(stats + 2 * Z + 1, (rs[1][0] + rs[1][1]) + (rs[1][2] + rs[1][3]));
        if (blockIdx.x == 0) atomicAdd(stats + 2 * Z + 2, (float)min(16, B - m0));
    }
}

__global__ __launch_bounds__(256) void sum_to_kernel(const float* __restrict__ x, int n, float* out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}

// Loss scalars from the (global) stats -- evaluated by EVERY block that needs them (Z-length reduction, a
// few hundred cycles) so that scalars and latent gradients are one launch; block 0 writes the scalars out.
struct LatentScalars {
    float total, recon, kl, wkl, collapse, mi, mi_pen, cmi, Bg;
};
__device__ __forceinline__ LatentScalars latent_scalars_block(const float* __restrict__ stats,
                                                              const float* __restrict__ hyper, int Z, int T,
                                                              float* red /* [4] shared */) {
    const float Bg = stats[2 * Z + 2];
    float a = 0.f;
    for (int j = threadIdx.x; j < Z; j += blockDim.x) {
        const float mm = stats[j] / Bg, mv = stats[Z + j] / Bg;
        a += 1.0f + logf(mv) - mm * mm - mv;
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    LatentScalars r;
    const float agg = -0.5f * tot;
    const float beta = hyper[0], lc = hyper[1], lmi = hyper[2], target = hyper[3];
    const float mean_kl = stats[2 * Z] / Bg;
    const float mi_raw = mean_kl - agg;
    r.mi = mi_raw > 0.0f ? mi_raw : 0.0f;              // mx.maximum(mi, 0.0): grad iff mi_raw > 0
    const float d = target - r.mi;
    const bool gate = !(0.0f > d);                     // mx.maximum(0.0, d): grad to d iff 0 <= d
    const float dpos = gate ? d : 0.0f;
    r.collapse = lc * dpos;
    r.mi_pen = lmi * dpos;
    r.kl = stats[2 * Z + 1] / Bg;
    r.recon = stats[2 * Z + 3] / (Bg * (float)T);
    r.wkl = beta * r.kl;
    r.total = r.recon + r.wkl + r.collapse + 0.0f + r.mi_pen;
    r.cmi = (gate && mi_raw > 0.0f) ? -(lc + lmi) : 0.0f;
    r.Bg = Bg;
    return r;
}
__device__ __forceinline__ void latent_scalars_write(const LatentScalars& r, float* scalars) {
    scalars[0] = r.total; scalars[1] = r.recon; scalars[2] = r.kl; scalars[3] = r.wkl; scalars[4] = r.collapse;
    scalars[5] = 0.0f; scalars[6] = 0.0f; scalars[7] = r.mi; scalars[8] = r.mi_pen; scalars[9] = r.cmi;
    scalars[10] = r.Bg;
}

__global__ __launch_bounds__(256) void latent_scalars_kernel(const float* __restrict__ stats,
                                                             const float* __restrict__ hyper, float* scalars,
                                                             int Z, int T) {
    __shared__ float red[4];
    const LatentScalars r = latent_scalars_block(stats, hyper, Z, T, red);
    if (threadIdx.x == 0) latent_scalars_write(r, scalars);
}

// d(total)/d(mu_raw), d(total)/d(lv_raw) for the local rows (z carries no gradient, Q2).
__global__ __launch_bounds__(256) void latent_grad_kernel(const float* __restrict__ mu,
                                                          const float* __restrict__ logvar,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ hyper,
                                                          float* scalars, float* dmu_raw,
                                                          float* dlv_raw, int B, int Z, int T, float fb_min) {
    __shared__ float red[4];
    const LatentScalars sc = latent_scalars_block(stats, hyper, Z, T, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) latent_scalars_write(sc, scalars);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Z) return;
    const int j = idx % Z;
    const float Bg = sc.Bg;
    const float beta = hyper[0], cmi = sc.cmi;
    const float m = mu[idx], lv = logvar[idx];
    // clip pass-through masks (mx.clip = minimum(maximum(x, lo), hi); never active inside the tanh bounds)
    const float pm = (m > -3.0f && m < 3.0f) ? 1.0f : 0.0f;
    const float pl = (lv > -6.0f && lv < 3.0f) ? 1.0f : 0.0f;
    const float mc = clipf(m, -3.0f, 3.0f), lc = clipf(lv, -6.0f, 3.0f);
    const float var = expf(lc);
    const float k = -0.5f * (1.0f + lc - mc * mc - var);
    const bool live = (k > 0.0f) && !(fb_min > 0.0f && !(k > fb_min));
    float gm = 0.f, gl = 0.f;
    if (live) {
        gm += beta * mc / Bg;
        gl += beta * 0.5f * (var - 1.0f) / Bg;
    }
    if (cmi != 0.0f) {
        const float mm = stats[j] / Bg, mv = stats[Z + j] / Bg;
        gm += cmi * (mc - mm) / Bg;
        gl += cmi * 0.5f * (var / mv - 1.0f) / Bg;
    }
    gm *= pm;
    gl *= pl;
    const float hm = 0.5f * m;          // tanh(mu_raw/2)
    const float hl = lv + 1.0f;         // tanh(lv_raw/2)
    dmu_raw[idx] = gm * (1.0f - hm * hm);
    dlv_raw[idx] = gl * 0.5f * (1.0f - hl * hl);
}

// y = dy_in * (1 - t^2)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ t, float* d, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float tv = t[i];
        d[i] = d[i] * (1.0f - tv * tv);
    }
}

}  // namespace

// Reference models/encoder.py:106-153 (+ the per-rank halves of kl.py / info.py reductions).
//   hT [B,H] = top-layer h at the LAST padded position (Q3); eps [B,Z] injected N(0,1) noise (Q18)
//   ws/out: comb [B,2H], lh [B,2H], mu_raw/lv_raw [B,Z], mu/logvar/z [B,Z], stats [2Z+4] (zeroed here)
extern "C" int arcvae_enc_heads_forward(const float* hT, const float* cond, const float* Wc, const float* bc,
                                        const float* Wmu, const float* bmu, const float* Wlh, const float* blh,
                                        const float* Wlv, const float* blv, const float* eps, float* comb,
                                        float* lh, float* mu_raw, float* lv_raw, float* mu, float* logvar,
                                        float* z, float* stats, int B, int H, int Z, int C, float free_bits,
                                        int comb_ready, hipStream_t stream) {
    if (!Wmu || !bmu || !Wlh || !blh || !Wlv || !blv || !eps || !comb || !lh || !mu_raw || !lv_raw || !mu || !logvar ||
        !z || !stats)
        return ARCVAE_ERR_ARG;
    if (!comb_ready && (!hT || !cond || !Wc || !bc)) return ARCVAE_ERR_ARG;
    if (B <= 0 || H <= 0 || Z <= 0 || C < 0) return ARCVAE_ERR_ARG;
    const int H2 = 2 * H;
    if (2 * Z + 4 > B * H2) return ARCVAE_ERR_ARG;
    // comb_ready: comb [B,2H] and the zeroed stats were written ahead of this call (arcvae_enc_prologue wrote the
    // condition half and cleared stats, the persistent forward sweep stored h_{T-1} of the top layer into the other half)
    if (!comb_ready)
        hipLaunchKernelGGL(build_comb_kernel, dim3(ceil_div(B * H2, 256)), dim3(256), 0, stream, hT, cond, Wc, bc, comb,
                           stats, 2 * Z + 4, B, H, C);
    const float fb_min = free_bits > 0.0f ? free_bits / (float)Z : 0.0f;
    int rc;
    const bool fast = B <= 256 && (H2 % 64) == 0 && (reinterpret_cast<uintptr_t>(comb) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(Wmu) & 15) == 0 && (reinterpret_cast<uintptr_t>(Wlh) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(Wlv) & 15) == 0 && (reinterpret_cast<uintptr_t>(lh) & 15) == 0;
    if (fast) {
        // [mu_raw | lh = tanh(.)] in one launch, then lv_raw with latent_apply as its epilogue: 2 launches instead of 4
        const int M[2] = {B, B}, N[2] = {Z, H2}, K[2] = {H2, H2}, lda[2] = {H2, H2}, ldb[2] = {H2, H2}, ldc[2] = {Z, H2};
        const float* A[2] = {comb, comb};
        const float* Bm[2] = {Wmu, Wlh};
        float* Cm[2] = {mu_raw, lh};
        const float* bias[2] = {bmu, blh};
        const int fl[2] = {0, ARCVAE_GEMM_TANH};
        rc = arcvae_gemm_skinny_pair(1, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias, fl, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(heads_lv_apply_kernel, dim3(ceil_div(Z, 16), ceil_div(B, 16)), dim3(256), 0, stream, lh, Wlv, blv,
                           mu_raw, eps, lv_raw, mu, logvar, z, stats, B, Z, H2, fb_min);
        return arcvae_launch_status();
    }
    rc = arcvae_gemm_f32(0, 1, B, Z, H2, comb, H2, Wmu, H2, mu_raw, Z, bmu, 0, stream);
    if (rc) return rc;
    rc = arcvae_gemm_f32(0, 1, B, H2, H2, comb, H2, Wlh, H2, lh, H2, blh, ARCVAE_GEMM_TANH, stream);
    if (rc) return rc;
    rc = arcvae_gemm_f32(0, 1, B, Z, H2, lh, H2, Wlv, H2, lv_raw, Z, blv, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(latent_apply_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, mu_raw, lv_raw, eps, mu,
                       logvar, z, stats, B, Z, fb_min);
    return arcvae_launch_status();
}

// stats[2Z+3] = sum_b rowloss[b]  (CE row sums from arcvae_dec_chain_ce)
extern "C" int arcvae_stats_set_recon(const float* rowloss, int B, float* stats, int Z, hipStream_t stream) {
    if (!rowloss || !stats || B <= 0 || Z <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(sum_to_kernel, dim3(1), dim3(256), 0, stream, rowloss, B, stats + 2 * Z + 3);
    return arcvae_launch_status();
}

// Loss scalars (complete_vae_loss.py:45-99) from GLOBAL stats, and the latent gradients of the
// local rows.  hyper/scalars/stats are device arrays (layouts at the top of this file).
extern "C" int arcvae_latent_loss(const float* stats, const float* hyper, const float* mu, const float* logvar,
                                  float* scalars, float* dmu_raw, float* dlv_raw, int B, int Z, int T,
                                  float free_bits, hipStream_t stream) {
    if (!stats || !hyper || !mu || !logvar || !scalars || B <= 0 || Z <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    if (dmu_raw && dlv_raw) {
        const float fb_min = free_bits > 0.0f ? free_bits / (float)Z : 0.0f;
        hipLaunchKernelGGL(latent_grad_kernel, dim3(ceil_div(B * Z, 256)), dim3(256), 0, stream, mu, logvar, stats,
                           hyper, scalars, dmu_raw, dlv_raw, B, Z, T, fb_min);
    } else {
        hipLaunchKernelGGL(latent_scalars_kernel, dim3(1), dim3(256), 0, stream, stats, hyper, scalars, Z, T);
    }
    return arcvae_launch_status();
}

// Recon half of the scalars, once the decoder's CE row sums have been reduced into stats[2Z+3]
// (arcvae_stats_set_recon, + all-reduce under data parallelism): recon = sum / (B_global*T) (mean over ALL
// positions, Q3) and total = recon + beta*kl + collapse + lambda_prop*0 + mi_penalty (complete_vae_loss.py:76-82).
// arcvae_latent_loss may therefore run BEFORE the decoder has finished: nothing on the encoder's backward
// path depends on the reconstruction term (Q2).
namespace {
// guard words (optional): an expired gate / a persistent sweep that gave up.  Non-zero -> this step's numbers are
// not ordered results: the nine loss scalars become NaN and scalars[15] = 1 (the trainer reads both in its per-batch
// D2H copy and raises at that batch; arcvae_adam_update skips the update on the same words).
__device__ __forceinline__ bool step_guard_tripped(const unsigned* ga, const unsigned* gb) {
    return (ga && __hip_atomic_load(ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ||
           (gb && __hip_atomic_load(gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
}
__device__ __forceinline__ void poison_scalars(float* scalars) {
    for (int i = 0; i < 9; ++i) scalars[i] = __builtin_nanf("");
    scalars[15] = 1.0f;
}
__global__ void loss_finalize_kernel(const float* __restrict__ stats, float* scalars, int Z, int T,
                                     const unsigned* ga, const unsigned* gb) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float recon = stats[2 * Z + 3] / (stats[2 * Z + 2] * (float)T);
        scalars[1] = recon;
        scalars[0] = recon + scalars[3] + scalars[4] + scalars[6] + scalars[8];
        scalars[15] = 0.0f;
        if (step_guard_tripped(ga, gb)) poison_scalars(scalars);
    }
}
}  // namespace

namespace {
// CE row sums -> stats[2Z+3], then the recon / total scalars: arcvae_stats_set_recon + arcvae_loss_finalize in one
// launch (both sit in the exposed tail of the single-process step)
__global__ __launch_bounds__(256) void recon_finalize_kernel(const float* __restrict__ rowloss, int n, float* stats,
                                                             float* scalars, int Z, int T, const unsigned* ga,
                                                             const unsigned* gb) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += rowloss[i];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ce = (red[0] + red[1]) + (red[2] + red[3]);
        stats[2 * Z + 3] = ce;
        const float recon = ce / (stats[2 * Z + 2] * (float)T);
        scalars[1] = recon;
        scalars[0] = recon + scalars[3] + scalars[4] + scalars[6] + scalars[8];
        scalars[15] = 0.0f;
        if (step_guard_tripped(ga, gb)) poison_scalars(scalars);
    }
}
}  // namespace

// stats[2Z+3] = sum_b rowloss[b], then as arcvae_loss_finalize: one launch (single-process step; under data
// parallelism the CE sum is all-reduced between the two, so the two separate entry points stay)
extern "C" int arcvae_recon_finalize(const float* rowloss, int B, float* stats, float* scalars, int Z, int T,
                                     const unsigned* guard_a, const unsigned* guard_b, hipStream_t stream) {
    if (!rowloss || !stats || !scalars || B <= 0 || Z <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(recon_finalize_kernel, dim3(1), dim3(256), 0, stream, rowloss, B, stats, scalars, Z, T, guard_a,
                       guard_b);
    return arcvae_launch_status();
}

extern "C" int arcvae_loss_finalize(const float* stats, float* scalars, int Z, int T, const unsigned* guard_a,
                                    const unsigned* guard_b, hipStream_t stream) {
    if (!stats || !scalars || Z <= 0 || T <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, stats, scalars, Z, T, guard_a, guard_b);
    return arcvae_launch_status();
}

// Backward of the heads in two phases so the critical path (dcomb -> BPTT) does not wait for the
// parameter-gradient GEMMs:
//   phase 1: dcomb [B,2H] = dmu_raw . Wmu + (dlv_raw . Wlv * (1 - lh^2)) . Wlh ; leaves dlh = d(pre-tanh) in ws
//            (dcomb's first H columns are d/d(hT), consumed by arcvae_enc_lstm_backward with ld = 2H)
//   phase 2: parameter gradients ("+=") from dmu_raw, dlv_raw, dlh, dcomb  -- any stream, after phase 1

// ---- the heads' parameter gradients in ONE launch -----------------------------------------------------------------
// dWmu += dmu_raw^T . comb, dWlv += dlv_raw^T . lh, dWlh += dlh^T . comb, dWc += dcr^T . cond and the four bias gradients
// (= the column sums of the four left operands) were four GEMM launches + four column-sum launches of K = B <= a few
// hundred: launch-bound work that runs on aux beside the persistent BPTT sweep, where every small dispatch waits for CU
// resources -- the eight of them dribbled over ~180 us, sat in front of the first chunk's weight-gradient GEMMs and made
// aux, not the chain, end the step (profiles/r02_tail_timeline.txt).  Here: 64 x 64 output tiles, K staged 16 rows at a
// time in LDS, 4 x 4 outputs per thread as a k-ordered fmaf chain (the order of the f32 MFMA kernels it replaces); a
// problem's first column of tiles also sums its left operand over k (the bias gradient).  Every output has one owner:
// plain "+=", no atomics.
namespace {
struct HeadsWg {
    const float* A[4];   // [K, M] (lda): dmu_raw, dlv_raw, dlh, dcomb + H
    const float* B[4];   // [K, N] (ldb): comb, lh, comb, cond
    float* C[4];         // [M, N] (ldc), +=
    float* bias[4];      // [M], += column sums of A
    int M[4], N[4], lda[4], ldb[4], ldc[4];
    int tile0[5];        // first tile of problem i in blockIdx.x; tile0[4] = total
    int K;
};
__global__ __launch_bounds__(256) void heads_wgrad_kernel(HeadsWg g) {
    __shared__ __attribute__((aligned(16))) float As[16][64];
    __shared__ __attribute__((aligned(16))) float Bs[16][64];
    int pi = 0;
    while (pi < 3 && (int)blockIdx.x >= g.tile0[pi + 1]) ++pi;
    const int t = blockIdx.x - g.tile0[pi];
    const int M = g.M[pi], N = g.N[pi], ntn = N > 0 ? (N + 63) >> 6 : 1;
    const int m0 = (t / ntn) * 64, n0 = (t % ntn) * 64;
    const float* __restrict__ A = g.A[pi];
    const float* __restrict__ Bm = g.B[pi];
    const int lda = g.lda[pi], ldb = g.ldb[pi];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;          // outputs: rows m0 + 4 ty + i, columns n0 + 4 tx + j
    float acc[4][4], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e, kk = idx >> 6, c = idx & 63;
            const bool kin = k0 + kk < g.K;
            As[kk][c] = (kin && m0 + c < M) ? A[(long)(k0 + kk) * lda + m0 + c] : 0.f;
            Bs[kk][c] = (kin && n0 + c < N) ? Bm[(long)(k0 + kk) * ldb + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&As[kk][4 * ty]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[kk][4 * tx]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bsum[i] += a[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
        float* c = g.C[pi] + (long)m * g.ldc[pi] + n0 + 4 * tx;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + 4 * tx + j < N) c[j] += acc[i][j];
        if (n0 == 0 && tx == 0 && g.bias[pi]) g.bias[pi][m] += bsum[i];
    }
}
}  // namespace

extern "C" int arcvae_enc_heads_backward(const float* cond, const float* Wmu, const float* Wlh, const float* Wlv,
                                         const float* comb, const float* lh, const float* dmu_raw,
                                         const float* dlv_raw, float* dlh, float* dcomb, float* dWc, float* dbc,
                                         float* dWmu, float* dbmu, float* dWlh, float* dblh, float* dWlv,
                                         float* dblv, int B, int H, int Z, int C, int phase, hipStream_t stream) {
    if (!cond || !Wmu || !Wlh || !Wlv || !comb || !lh || !dmu_raw || !dlv_raw || !dlh || !dcomb || !dWc || !dbc ||
        !dWmu || !dbmu || !dWlh || !dblh || !dWlv || !dblv)
        return ARCVAE_ERR_ARG;
    if (phase < 0 || phase > 2) return ARCVAE_ERR_ARG;  // 0 = both
    const int H2 = 2 * H;
    const int ACC = ARCVAE_GEMM_ACCUMULATE;
    int rc;
    if (phase == 0 || phase == 1) {
        const bool pair = B <= 256 && (Z % 64) == 0 && (reinterpret_cast<uintptr_t>(dmu_raw) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(dlv_raw) & 15) == 0;
        if (pair) {   // dcomb = dmu_raw . Wmu and dlh = (dlv_raw . Wlv) * (1 - lh^2) in one launch
            const int M[2] = {B, B}, N[2] = {H2, H2}, K[2] = {Z, Z}, lda[2] = {Z, Z}, ldb[2] = {H2, H2}, ldc[2] = {H2, H2};
            const float* A[2] = {dmu_raw, dlv_raw};
            const float* Bm[2] = {Wmu, Wlv};
            float* Cm[2] = {dcomb, dlh};
            const float* bias[2] = {nullptr, lh};
            const int fl[2] = {0, ARCVAE_GEMM_DTANH};
            if ((rc = arcvae_gemm_skinny_pair(0, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias, fl, stream))) return rc;
        } else {
        if ((rc = arcvae_gemm_f32(0, 0, B, H2, Z, dmu_raw, Z, Wmu, H2, dcomb, H2, nullptr, 0, stream))) return rc;
        // dlh = (dlv_raw . Wlv) * (1 - lh^2): tanh backward fused into the GEMM epilogue
        if ((rc = arcvae_gemm_f32(0, 0, B, H2, Z, dlv_raw, Z, Wlv, H2, dlh, H2, lh, ARCVAE_GEMM_DTANH, stream)))
            return rc;
        }
        if ((rc = arcvae_gemm_f32(0, 0, B, H2, H2, dlh, H2, Wlh, H2, dcomb, H2, nullptr, ACC, stream))) return rc;
    }
    if ((phase == 0 || phase == 2) && arcvae_env_int("ARCVAE_HEADS_WGRAD_FUSED", 1) != 0) {   // one launch (heads_wgrad_kernel)
        HeadsWg g;
        const float* As_[4] = {dmu_raw, dlv_raw, dlh, dcomb + H};
        const float* Bs_[4] = {comb, lh, comb, cond};
        float* Cs_[4] = {dWmu, dWlv, dWlh, dWc};
        float* bs_[4] = {dbmu, dblv, dblh, dbc};
        const int Ms[4] = {Z, Z, H2, H}, Ns[4] = {H2, H2, H2, C}, la[4] = {Z, Z, H2, H2}, lb[4] = {H2, H2, H2, C > 0 ? C : 1},
                  lc[4] = {H2, H2, H2, C > 0 ? C : 1};
        int tiles = 0;
        for (int i = 0; i < 4; ++i) {
            g.A[i] = As_[i]; g.B[i] = Bs_[i]; g.C[i] = Cs_[i]; g.bias[i] = bs_[i];
            g.M[i] = Ms[i]; g.N[i] = Ns[i]; g.lda[i] = la[i]; g.ldb[i] = lb[i]; g.ldc[i] = lc[i];
            g.tile0[i] = tiles;
            // condition_fc with C == 0 has no weight, but its bias gradient (column sums of dcomb[:, H:]) still exists:
            // one column of (empty) tiles
            tiles += ceil_div(Ms[i], 64) * (Ns[i] > 0 ? ceil_div(Ns[i], 64) : 1);
        }
        g.tile0[4] = tiles;
        g.K = B;
        hipLaunchKernelGGL(heads_wgrad_kernel, dim3(tiles), dim3(256), 0, stream, g);
        return arcvae_launch_status();
    }
    if (phase == 0 || phase == 2) {
        if ((rc = arcvae_gemm_f32(1, 0, Z, H2, B, dmu_raw, Z, comb, H2, dWmu, H2, nullptr, ACC, stream))) return rc;
        if ((rc = arcvae_colsum_accum(dmu_raw, B, Z, Z, dbmu, 1.0f, stream))) return rc;
        if ((rc = arcvae_gemm_f32(1, 0, Z, H2, B, dlv_raw, Z, lh, H2, dWlv, H2, nullptr, ACC, stream))) return rc;
        if ((rc = arcvae_colsum_accum(dlv_raw, B, Z, Z, dblv, 1.0f, stream))) return rc;
        if ((rc = arcvae_gemm_f32(1, 0, H2, H2, B, dlh, H2, comb, H2, dWlh, H2, nullptr, ACC, stream))) return rc;
        if ((rc = arcvae_colsum_accum(dlh, B, H2, H2, dblh, 1.0f, stream))) return rc;
        if (C > 0) {  // condition_fc: dcr = dcomb[:, H:]
            if ((rc = arcvae_gemm_f32(1, 0, H, C, B, dcomb + H, H2, cond, C, dWc, C, nullptr, ACC, stream))) return rc;
        }
        if ((rc = arcvae_colsum_accum(dcomb + H, B, H, H2, dbc, 1.0f, stream))) return rc;
    }
    return arcvae_launch_status();
}

// =============================================================================================
// The SEAM between the two sweeps as ONE per-XCD kernel (round 4).  At the default shape the chain between
// the forward sweep's last tick and the BPTT sweep's first was five launches -- [mu_raw | lh], lv_raw + bounds + reparameterise +
// statistics, latent loss + gradients, [dmu_raw . Wmu | dlh], dcomb -- of 5-12 us each for ~0.1 GFLOP: 36 us of launch seams, cold
// operand fetches and pipeline fill on the step's critical chain (profiles/r04_timeline_base.txt).  Here the batch rows are
// partitioned over the 8 XCDs exactly as in the sweeps (XCD x owns rows [x RX, x RX + RX), RX = ceil(B / 8) <= 8), an XCD's 32 CUs
// split the OUTPUT COLUMNS of every product (Z/32 latent columns, 2H/32 hidden columns each), every weight slice a CU will need in
// any phase is requested into LDS when the block starts (96 KB at the default shape: the loads of all five products fly at once),
// activations are exchanged through the XCD's L2 behind the sweeps' flag-line barrier, and the only cross-XCD step -- the batch
// statistics of the MI term (Q11) -- is f32 atomics into `stats` plus one device-wide arrival counter.
//   phases bit 0: forward part (models/encoder.py:106-153: heads, tanh bounds, z, this process's partial `stats`)
//   phases bit 1: backward part from GLOBAL stats (complete_vae_loss.py:45-99 scalars, d/d(mu_raw, lv_raw), dlh, dcomb)
//   3: both in one launch (single process); data-parallel steps launch 1, all-reduce `stats`, launch 2.
// Same arithmetic as the launches it replaces (k-ordered f32 FMA chains; libm tanhf / expf / logf); statistics by atomics as before.
namespace {
struct SeamArgs {
    const float *comb, *Wmu, *bmu, *Wlh, *blh, *Wlv, *blv, *eps, *hyper;
    float *lh, *mu_raw, *lv_raw, *mu, *logvar, *z, *stats, *scalars, *dmu_raw, *dlv_raw, *dlh, *dcomb;
    unsigned* sync;      // sync_ws (xcd.h: PSEAM words, PS_ERR)
    int B, H2, Z, T, RX, phases, launch;
    int dbg;             // timing experiments only (ARCVAE_SEAM_DEBUG = n: every block returns at stop point n; results are WRONG)
    float fb_min;
};
constexpr int SEAM_R = 8;     // rows per XCD at most

// per-XCD barrier (the sweeps' protocol: every storing thread drains its stores to the L2, then ONE plain flag store per CU on
// the XCD's flag line, polled with agent-scope loads by one wave); false: a bounded spin expired (error word raised)
__device__ __forceinline__ bool seam_xcd_barrier(unsigned* my_flag, const unsigned* xflags, unsigned epoch, unsigned* err,
                                                 unsigned* s_ok) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ps_stores_in_l2();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (wave == 1) {
        unsigned spins = 0;
        while (true) {
            const unsigned v = (lane < 32) ? __hip_atomic_load(xflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all((int)(v - epoch) >= 0)) break;
            if (++spins > 4000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (lane == 0) { atomicAdd(err, 1u); *s_ok = 0; }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return *s_ok != 0;
}

__global__ __launch_bounds__(256) void enc_seam_kernel(SeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ unsigned s_role, s_xcc, s_ok;
    __shared__ float s_red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H2 = a.H2, Z = a.Z, Zc = Z >> 5, Hc = H2 >> 5, LDW = H2 + 4, LDZ = Z + 4;
    unsigned* err = a.sync + PS_ERR;
    if (tid == 0) {
        s_xcc = ps_xcc_id();
        s_role = atomicAdd(a.sync + PSEAM_CNT + 8 * a.launch + (s_xcc & 7), 1u);
        s_ok = 1;
    }
    __syncthreads();
    const unsigned xcc = __builtin_amdgcn_readfirstlane(s_xcc), role = __builtin_amdgcn_readfirstlane(s_role);
    if (xcc >= 8 || role >= 32) {                     // no slot on this XCD: the others time out and drain
        if (tid == 0) atomicAdd(err, 1u);
        return;
    }
    const int row0 = xcc * a.RX, RXl = max(0, min(a.RX, B - row0));
    unsigned* my_flag = a.sync + PSEAM + xcc * 32 + role;
    const unsigned* xflags = a.sync + PSEAM + xcc * 32;
    const bool fwd = (a.phases & 1) != 0, bwd = (a.phases & 2) != 0;
    // LDS: weight slices of every phase, the XCD's activation rows, the exchanged gradient rows
    float* wA = sm;                                   // [(Zc + Hc)][LDW]: rows of Wmu (my Zc), then of Wlh (my Hc)
    float* wB = wA + (Zc + Hc) * LDW;                 // [Zc][LDW]: rows of Wlv
    float* wD = wB + Zc * LDW;                        // [Z][Hc]: Wlv[:, my columns]
    float* wE1 = wD + Z * Hc;                         // [Z][Hc]: Wmu[:, my columns]
    float* wE2 = wE1 + Z * Hc;                        // [H2][Hc]: Wlh[:, my columns]
    float* act = wE2 + H2 * Hc;                       // [8][LDW]: comb rows, then lh rows, then dlh rows of my XCD
    float* g1 = act + SEAM_R * LDW;                   // [8][LDZ] dmu_raw rows
    float* g2 = g1 + SEAM_R * LDZ;                    // [8][LDZ] dlv_raw rows
    float* mr = g2 + SEAM_R * LDZ;                    // [8][8] my mu_raw values
    float* lhown = mr + SEAM_R * 8;                   // [8][Hc] my lh values
    // ---- every weight slice, requested now (16-byte loads; the phases find them in LDS)
    const int q4 = H2 >> 2;                           // float4 per weight row
    if (fwd) {
        for (int i = tid; i < (Zc + Hc) * q4; i += 256) {
            const int r = i / q4, k4 = i - r * q4;
            const float* src = r < Zc ? a.Wmu + (long)(role * Zc + r) * H2 : a.Wlh + (long)(role * Hc + r - Zc) * H2;
            *reinterpret_cast<f32x4*>(wA + r * LDW + 4 * k4) = *reinterpret_cast<const f32x4*>(src + 4 * k4);
        }
        for (int i = tid; i < Zc * q4; i += 256) {
            const int r = i / q4, k4 = i - r * q4;
            *reinterpret_cast<f32x4*>(wB + r * LDW + 4 * k4) = *reinterpret_cast<const f32x4*>(a.Wlv + (long)(role * Zc + r) * H2 + 4 * k4);
        }
        for (int i = tid; i < RXl * q4; i += 256) {   // comb rows of my XCD (written by the sweep's last tick + the prologue)
            const int r = i / q4, k4 = i - r * q4;
            *reinterpret_cast<f32x4*>(act + r * LDW + 4 * k4) = *reinterpret_cast<const f32x4*>(a.comb + (long)(row0 + r) * H2 + 4 * k4);
        }
    }
    if (bwd) {
        const int c4 = Hc >> 2;
        for (int i = tid; i < Z * c4; i += 256) {
            const int j = i / c4, c = i - j * c4;
            *reinterpret_cast<f32x4*>(wD + j * Hc + 4 * c) = *reinterpret_cast<const f32x4*>(a.Wlv + (long)j * H2 + role * Hc + 4 * c);
            *reinterpret_cast<f32x4*>(wE1 + j * Hc + 4 * c) = *reinterpret_cast<const f32x4*>(a.Wmu + (long)j * H2 + role * Hc + 4 * c);
        }
        for (int i = tid; i < H2 * c4; i += 256) {
            const int k = i / c4, c = i - k * c4;
            *reinterpret_cast<f32x4*>(wE2 + k * Hc + 4 * c) = *reinterpret_cast<const f32x4*>(a.Wlh + (long)k * H2 + role * Hc + 4 * c);
        }
    }
    __syncthreads();
    if (a.dbg == 1) return;
    // thread roles: phase B / C: (row i, my latent column jq), K split over 8 lanes;  phases D / E: (row i, my hidden column q), K over 2
    const int oB = tid >> 3, sB = tid & 7, iB = oB & 7, jq = oB >> 3;
    const bool ownB = sB == 0 && jq < Zc && iB < RXl;
    const int jcol = role * Zc + min(jq, Zc - 1);
    const int oD = tid >> 1, sD = tid & 1, iD = oD & 7, qD = oD >> 3;
    const bool actD = qD < Hc && iD < RXl;
    float m_v = 0.f, lv_v = 0.f;                      // my (row, latent column): mu, logvar
    if (fwd) {
        // ---- phase A: [mu_raw | lh = tanh(.)] of my XCD's rows, my columns
        if (tid < 8 * (Zc + Hc)) {
            const int i = tid & 7, c = tid >> 3;
            if (i < RXl) {
                const f32x4* w4 = reinterpret_cast<const f32x4*>(wA + c * LDW);
                const f32x4* x4 = reinterpret_cast<const f32x4*>(act + i * LDW);
                // (four independent chains, eight 16-byte LDS reads in flight: one dependent chain over K ran at an LDS round trip
                // per step -- 5.3 us for this phase; ARCVAE_SEAM_DEBUG stop points, tools/r4_seam_alone.py)
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
                for (int k = 0; k < q4; k += 4) {
                    const f32x4 w0 = w4[k], x0 = x4[k], w1 = w4[k + 1], x1 = x4[k + 1], w2 = w4[k + 2], x2 = x4[k + 2], w3 = w4[k + 3], x3 = x4[k + 3];
                    a0 = fmaf(x0[0], w0[0], a0); a0 = fmaf(x0[1], w0[1], a0); a0 = fmaf(x0[2], w0[2], a0); a0 = fmaf(x0[3], w0[3], a0);
                    a1 = fmaf(x1[0], w1[0], a1); a1 = fmaf(x1[1], w1[1], a1); a1 = fmaf(x1[2], w1[2], a1); a1 = fmaf(x1[3], w1[3], a1);
                    a2 = fmaf(x2[0], w2[0], a2); a2 = fmaf(x2[1], w2[1], a2); a2 = fmaf(x2[2], w2[2], a2); a2 = fmaf(x2[3], w2[3], a2);
                    a3 = fmaf(x3[0], w3[0], a3); a3 = fmaf(x3[1], w3[1], a3); a3 = fmaf(x3[2], w3[2], a3); a3 = fmaf(x3[3], w3[3], a3);
                }
                const float acc = (a0 + a1) + (a2 + a3);
                if (c < Zc) {
                    const int col = role * Zc + c;
                    const float v = acc + a.bmu[col];
                    a.mu_raw[(long)(row0 + i) * Z + col] = v;
                    mr[i * 8 + c] = v;
                } else {
                    const int col = role * Hc + c - Zc;
                    const float v = tanhf(acc + a.blh[col]);
                    a.lh[(long)(row0 + i) * H2 + col] = v;
                    lhown[i * Hc + c - Zc] = v;
                }
            }
        }
        if (a.dbg == 2) return;
        if (!seam_xcd_barrier(my_flag, xflags, 1u, err, &s_ok)) return;
        if (a.dbg == 3) return;
        // ---- phase B: lv_raw from the XCD's lh rows, then bounds, z and my share of the batch statistics
        if (RXl > 0) {
            const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.lh + (long)row0 * H2, (long)RXl * H2 * 4);
            for (int i = tid; i < RXl * q4; i += 256) {
                const int r = i / q4, k4 = i - r * q4;
                *reinterpret_cast<f32x4*>(act + r * LDW + 4 * k4) = ps_load_sc1_x4(rs, (unsigned)((r * H2 + 4 * k4) * 4));
            }
        }
        __syncthreads();
        float part = 0.f;
        if (jq < Zc && iB < RXl) {
            const int kn = H2 >> 3, k0 = sB * kn;      // my eighth of K
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wB + jq * LDW + k0);
            const f32x4* x4 = reinterpret_cast<const f32x4*>(act + iB * LDW + k0);
            float p0 = 0.f, p1 = 0.f;
#pragma unroll 4
            for (int k = 0; k < (kn >> 2); k += 2) {
                const f32x4 w0 = w4[k], x0 = x4[k], w1 = w4[k + 1], x1 = x4[k + 1];
                p0 = fmaf(x0[0], w0[0], p0); p0 = fmaf(x0[1], w0[1], p0); p0 = fmaf(x0[2], w0[2], p0); p0 = fmaf(x0[3], w0[3], p0);
                p1 = fmaf(x1[0], w1[0], p1); p1 = fmaf(x1[1], w1[1], p1); p1 = fmaf(x1[2], w1[2], p1); p1 = fmaf(x1[3], w1[3], p1);
            }
            part = p0 + p1;
        }
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
        float mc_v = 0.f, var_v = 0.f, k_v = 0.f, kf_v = 0.f;
        if (ownB) {
            const long idx = (long)(row0 + iB) * Z + jcol;
            const float lvr = part + a.blv[jcol];
            a.lv_raw[idx] = lvr;
            const float m = tanhf(mr[iB * 8 + jq] / 2.0f) * 2.0f;
            const float lv = tanhf(lvr / 2.0f) * 1.0f - 1.0f;
            a.mu[idx] = m; a.logvar[idx] = lv;
            a.z[idx] = m + a.eps[idx] * expf(0.5f * lv);
            m_v = m; lv_v = lv;
            const float mc = clipf(m, -3.0f, 3.0f), lc = clipf(lv, -6.0f, 3.0f);
            var_v = expf(lc); mc_v = mc;
            k_v = -0.5f * (1.0f + lc - mc * mc - var_v);
            kf_v = fmaxf(k_v, 0.0f);
            if (a.fb_min > 0.0f) kf_v = fmaxf(kf_v, a.fb_min);
        }
        // wave jq holds the 8 rows of latent column jq (lanes with sB == 0): column sums and KL sums by wave reduction
        mc_v = wave_sum(mc_v); var_v = wave_sum(var_v); k_v = wave_sum(k_v); kf_v = wave_sum(kf_v);
        // column sums: a column has ONE owner per XCD -> 8 adds per address.  The two KL sums and the row count would take an add from
        // every wave of every CU (1024 per address, serialised at the memory side: this phase measured 28 us): a CU's partials go
        // to its own slot, the XCD's CU 0 adds the 32 slots up behind one more flag barrier and issues ONE add per XCD.
        if (lane == 0 && wave < Zc && RXl > 0) {
            atomicAdd(a.stats + jcol, mc_v);
            atomicAdd(a.stats + Z + jcol, var_v);
        }
        if (lane == 0) { s_red[wave] = wave < Zc ? k_v : 0.f; s_red[4 + wave] = wave < Zc ? kf_v : 0.f; }
        __syncthreads();
        float* xpart = reinterpret_cast<float*>(a.sync + PSEAM_PART) + xcc * 64;
        if (tid == 0) {
            xpart[2 * role] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
            xpart[2 * role + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        }
        if (!seam_xcd_barrier(my_flag, xflags, 2u, err, &s_ok)) return;
        if (role == 0 && wave == 0 && RXl > 0) {
            const __amdgpu_buffer_rsrc_t rp = ps_rsrc(xpart, 64 * 4);
            float v = ps_load_sc1(rp, (unsigned)(lane * 4));          // lane 2r: CU r's k sum, lane 2r + 1: its free-bits sum
            v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 2) atomicAdd(a.stats + 2 * Z + lane, v);
            if (lane == 2) atomicAdd(a.stats + 2 * Z + 2, (float)RXl);
        }
        if (!bwd || a.dbg == 4) return;
        // ---- every block's statistics are in: device-wide arrival counter (the one cross-XCD step of the seam)
        ps_stores_in_l2();
        __syncthreads();
        if (wave == 1) {
            unsigned* gcnt = a.sync + PSEAM_GCNT;
            if (lane == 0) __hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(gcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 256u) {
                if (++spins > 4000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) { atomicAdd(err, 1u); s_ok = 0; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (!s_ok || a.dbg == 5) return;
    } else {
        // backward part on its own (data-parallel step: `stats` was all-reduced in between): my values back from memory
        if (ownB) {
            const long idx = (long)(row0 + iB) * Z + jcol;
            m_v = a.mu[idx]; lv_v = a.logvar[idx];
        }
        if (sD == 0 && actD) lhown[iD * Hc + qD] = a.lh[(long)(row0 + iD) * H2 + role * Hc + qD];
    }
    // ---- phase C: loss scalars from the GLOBAL statistics (agent-scope loads: other XCDs' atomics), d/d(mu_raw, lv_raw) of mine
    float cmi, Bg;
    {
        const float bg = __hip_atomic_load(a.stats + 2 * Z + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float t = 0.f;
        for (int j = tid; j < Z; j += 256) {
            const float mm = __hip_atomic_load(a.stats + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / bg;
            const float mv = __hip_atomic_load(a.stats + Z + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / bg;
            t += 1.0f + logf(mv) - mm * mm - mv;
        }
        t = wave_sum(t);
        if (lane == 0) s_red[wave] = t;
        __syncthreads();
        const float tot = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        const float agg = -0.5f * tot;
        const float beta = a.hyper[0], lc = a.hyper[1], lmi = a.hyper[2], target = a.hyper[3];
        const float mean_kl = __hip_atomic_load(a.stats + 2 * Z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / bg;
        const float mi_raw = mean_kl - agg;
        const float mi = mi_raw > 0.0f ? mi_raw : 0.0f;
        const float d = target - mi;
        const bool gate = !(0.0f > d);
        const float dpos = gate ? d : 0.0f;
        cmi = (gate && mi_raw > 0.0f) ? -(lc + lmi) : 0.0f;
        Bg = bg;
        if (xcc == 0 && role == 0 && tid == 0) {
            LatentScalars r;
            r.mi = mi; r.collapse = lc * dpos; r.mi_pen = lmi * dpos;
            r.kl = __hip_atomic_load(a.stats + 2 * Z + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / bg;
            r.recon = __hip_atomic_load(a.stats + 2 * Z + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (bg * (float)a.T);
            r.wkl = beta * r.kl;
            r.total = r.recon + r.wkl + r.collapse + 0.0f + r.mi_pen;
            r.cmi = cmi; r.Bg = bg;
            latent_scalars_write(r, a.scalars);
        }
    }
    if (ownB) {
        const long idx = (long)(row0 + iB) * Z + jcol;
        const float beta = a.hyper[0];
        const float m = m_v, lv = lv_v;
        const float pm = (m > -3.0f && m < 3.0f) ? 1.0f : 0.0f;
        const float pl = (lv > -6.0f && lv < 3.0f) ? 1.0f : 0.0f;
        const float mc = clipf(m, -3.0f, 3.0f), lc = clipf(lv, -6.0f, 3.0f);
        const float var = expf(lc);
        const float k = -0.5f * (1.0f + lc - mc * mc - var);
        const bool live = (k > 0.0f) && !(a.fb_min > 0.0f && !(k > a.fb_min));
        float gm = 0.f, gl = 0.f;
        if (live) {
            gm += beta * mc / Bg;
            gl += beta * 0.5f * (var - 1.0f) / Bg;
        }
        if (cmi != 0.0f) {
            const float mm = __hip_atomic_load(a.stats + jcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / Bg;
            const float mv = __hip_atomic_load(a.stats + Z + jcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / Bg;
            gm += cmi * (mc - mm) / Bg;
            gl += cmi * 0.5f * (var / mv - 1.0f) / Bg;
        }
        gm *= pm;
        gl *= pl;
        const float hm = 0.5f * m, hl = lv + 1.0f;
        a.dmu_raw[idx] = gm * (1.0f - hm * hm);
        a.dlv_raw[idx] = gl * 0.5f * (1.0f - hl * hl);
    }
    if (a.dbg == 6) return;
    if (!seam_xcd_barrier(my_flag, xflags, 3u, err, &s_ok)) return;
    if (a.dbg == 7) return;
    // ---- phase D: dlh = (dlv_raw . Wlv) (1 - lh^2) for my columns; the dmu_raw . Wmu half of dcomb rides along
    if (RXl > 0) {
        const __amdgpu_buffer_rsrc_t r1 = ps_rsrc(a.dmu_raw + (long)row0 * Z, (long)RXl * Z * 4);
        const __amdgpu_buffer_rsrc_t r2 = ps_rsrc(a.dlv_raw + (long)row0 * Z, (long)RXl * Z * 4);
        const int z4 = Z >> 2;
        for (int i = tid; i < RXl * z4; i += 256) {
            const int r = i / z4, k4 = i - r * z4;
            *reinterpret_cast<f32x4*>(g1 + r * LDZ + 4 * k4) = ps_load_sc1_x4(r1, (unsigned)((r * Z + 4 * k4) * 4));
            *reinterpret_cast<f32x4*>(g2 + r * LDZ + 4 * k4) = ps_load_sc1_x4(r2, (unsigned)((r * Z + 4 * k4) * 4));
        }
    }
    __syncthreads();
    float dl = 0.f, dcm = 0.f;
    if (actD) {
        const int jn = Z >> 1, j0 = sD * jn;
        float d0 = 0.f, d1 = 0.f, c0 = 0.f, c1 = 0.f;
#pragma unroll 2
        for (int j = j0; j < j0 + jn; j += 4) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g2 + iD * LDZ + j), mv = *reinterpret_cast<const f32x4*>(g1 + iD * LDZ + j);
            const float u0 = wD[j * Hc + qD], u1 = wD[(j + 1) * Hc + qD], u2 = wD[(j + 2) * Hc + qD], u3 = wD[(j + 3) * Hc + qD];
            const float t0 = wE1[j * Hc + qD], t1 = wE1[(j + 1) * Hc + qD], t2 = wE1[(j + 2) * Hc + qD], t3 = wE1[(j + 3) * Hc + qD];
            d0 = fmaf(gv[0], u0, d0); d1 = fmaf(gv[1], u1, d1); d0 = fmaf(gv[2], u2, d0); d1 = fmaf(gv[3], u3, d1);
            c0 = fmaf(mv[0], t0, c0); c1 = fmaf(mv[1], t1, c1); c0 = fmaf(mv[2], t2, c0); c1 = fmaf(mv[3], t3, c1);
        }
        dl = d0 + d1; dcm = c0 + c1;
    }
    dl += __shfl_xor(dl, 1, 64);
    if (sD == 0 && actD) {
        const float t = lhown[iD * Hc + qD];
        a.dlh[(long)(row0 + iD) * H2 + role * Hc + qD] = dl * (1.0f - t * t);
    }
    if (a.dbg == 8) return;
    if (!seam_xcd_barrier(my_flag, xflags, 4u, err, &s_ok)) return;
    if (a.dbg == 9) return;
    // ---- phase E: dcomb = dmu_raw . Wmu + dlh . Wlh for my columns
    if (RXl > 0) {
        const __amdgpu_buffer_rsrc_t rs = ps_rsrc(a.dlh + (long)row0 * H2, (long)RXl * H2 * 4);
        for (int i = tid; i < RXl * q4; i += 256) {
            const int r = i / q4, k4 = i - r * q4;
            *reinterpret_cast<f32x4*>(act + r * LDW + 4 * k4) = ps_load_sc1_x4(rs, (unsigned)((r * H2 + 4 * k4) * 4));
        }
    }
    __syncthreads();
    if (actD) {
        const int kn = H2 >> 1, k0 = sD * kn;
        float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
#pragma unroll 2
        for (int k = k0; k < k0 + kn; k += 8) {
            const f32x4 xa = *reinterpret_cast<const f32x4*>(act + iD * LDW + k), xb = *reinterpret_cast<const f32x4*>(act + iD * LDW + k + 4);
            const float w0 = wE2[k * Hc + qD], w1 = wE2[(k + 1) * Hc + qD], w2 = wE2[(k + 2) * Hc + qD], w3 = wE2[(k + 3) * Hc + qD];
            const float w4_ = wE2[(k + 4) * Hc + qD], w5 = wE2[(k + 5) * Hc + qD], w6 = wE2[(k + 6) * Hc + qD], w7 = wE2[(k + 7) * Hc + qD];
            e0 = fmaf(xa[0], w0, e0); e1 = fmaf(xa[1], w1, e1); e2 = fmaf(xa[2], w2, e2); e3 = fmaf(xa[3], w3, e3);
            e0 = fmaf(xb[0], w4_, e0); e1 = fmaf(xb[1], w5, e1); e2 = fmaf(xb[2], w6, e2); e3 = fmaf(xb[3], w7, e3);
        }
        dcm += (e0 + e1) + (e2 + e3);
    }
    dcm += __shfl_xor(dcm, 1, 64);
    if (sD == 0 && actD) a.dcomb[(long)(row0 + iD) * H2 + role * Hc + qD] = dcm;
}

inline size_t seam_lds_bytes(int H2, int Z) {
    const int Zc = Z >> 5, Hc = H2 >> 5, LDW = H2 + 4, LDZ = Z + 4;
    return sizeof(float) * ((size_t)(Zc + Hc) * LDW + (size_t)Zc * LDW + 2 * (size_t)Z * Hc + (size_t)H2 * Hc + (size_t)SEAM_R * LDW +
                            2 * (size_t)SEAM_R * LDZ + SEAM_R * 8 + (size_t)SEAM_R * Hc);
}
}  // namespace

// 1 where arcvae_enc_seam is to be used for the shape: OPT-IN (ARCVAE_SEAM_FUSED=1), at most 8 rows per XCD (B <= 64), hidden_dim a
// multiple of 64 up to 256, latent_dim a multiple of 32 up to 128, persistent kernels enabled (ARCVAE_PERSIST != 0).
// Parity-green and measured SLOWER than the five launches it replaces (round 4, default shape, tools/r4_seam_alone.py): 30.9 us
// alone against 19.2, step 0.987 against 0.963 ms.  The seam is not launch-bound: the five launches spread each product over the
// whole chip on the matrix cores and a captured launch costs ~4 us; here a CU works its column slice off with FMA chains out of
// LDS.  Stop points of the kernel alone (us, cumulative): launch + role + every weight slice into LDS 6.4, [mu_raw | lh] 11.0,
// flag barrier 13.6, lv_raw + bounds + statistics 17.9, device-wide arrival counter 20.4, scalars + latent gradients 22.6,
// barrier 22.8, dlh 26.1, barrier 26.9, dcomb 30.9 -- ten dependent steps of 2-5 us; before the KL sums were reduced per XCD
// (one f32 add per XCD instead of one per wave and CU on two addresses) the statistics alone took 28 us.
extern "C" int arcvae_enc_seam_ok(int B, int H, int Z) {
    if (arcvae_env_int("ARCVAE_PERSIST", 1) == 0 || arcvae_env_int("ARCVAE_SEAM_FUSED", 0) == 0) return 0;
    if (B < 1 || B > 8 * SEAM_R || H < 64 || H > 256 || (H % 64) != 0 || Z < 32 || Z > 128 || (Z % 32) != 0) return 0;
    return seam_lds_bytes(2 * H, Z) <= 150 * 1024 ? 1 : 0;
}

// models/encoder.py:106-153 + complete_vae_loss.py:45-99 + the heads' backward up to dcomb, as one per-XCD kernel (see above).
// comb [B,2H] complete (arcvae_enc_prologue + the persistent forward sweep); stats [2Z+4] zeroed ahead of the forward part;
// sync_ws: the sweeps' scratch (words [4864, 5696) are the seam's; flags bit 0: they were zeroed ahead of the step by
// arcvae_enc_prologue with n_sync >= 5696).  Outputs: everything arcvae_enc_heads_forward, arcvae_latent_loss(with gradients) and
// arcvae_enc_heads_backward(phase 1) write.  sync_ws[500] != 0 afterwards = a block gave up waiting.
extern "C" int arcvae_enc_seam(const float* comb, const float* Wmu, const float* bmu, const float* Wlh, const float* blh,
                               const float* Wlv, const float* blv, const float* eps, const float* hyper, float* lh, float* mu_raw,
                               float* lv_raw, float* mu, float* logvar, float* z, float* stats, float* scalars, float* dmu_raw,
                               float* dlv_raw, float* dlh, float* dcomb, unsigned* sync_ws, int B, int H, int Z, int T,
                               float free_bits, int phases, int flags, hipStream_t stream) {
    if (!comb || !Wmu || !bmu || !Wlh || !blh || !Wlv || !blv || !eps || !lh || !mu_raw || !lv_raw || !mu || !logvar || !z ||
        !stats || !sync_ws)
        return ARCVAE_ERR_ARG;
    if (phases < 1 || phases > 3 || T <= 0 || !arcvae_enc_seam_ok(B, H, Z)) return ARCVAE_ERR_ARG;
    if ((phases & 2) && (!hyper || !scalars || !dmu_raw || !dlv_raw || !dlh || !dcomb)) return ARCVAE_ERR_ARG;
    const uintptr_t al = reinterpret_cast<uintptr_t>(comb) | reinterpret_cast<uintptr_t>(Wmu) | reinterpret_cast<uintptr_t>(Wlh) |
                         reinterpret_cast<uintptr_t>(Wlv) | reinterpret_cast<uintptr_t>(lh) | reinterpret_cast<uintptr_t>(dmu_raw) |
                         reinterpret_cast<uintptr_t>(dlv_raw) | reinterpret_cast<uintptr_t>(dlh);
    if (al & 15) return ARCVAE_ERR_ARG;
    if (!(flags & 1)) {
        const int rc = arcvae_zero(reinterpret_cast<float*>(sync_ws + PSEAM), 1, PSEAM_WORDS - PSEAM, PSEAM_WORDS - PSEAM, stream);
        if (rc != ARCVAE_OK) return rc;
    }
    SeamArgs a;
    a.comb = comb; a.Wmu = Wmu; a.bmu = bmu; a.Wlh = Wlh; a.blh = blh; a.Wlv = Wlv; a.blv = blv; a.eps = eps; a.hyper = hyper;
    a.lh = lh; a.mu_raw = mu_raw; a.lv_raw = lv_raw; a.mu = mu; a.logvar = logvar; a.z = z; a.stats = stats; a.scalars = scalars;
    a.dmu_raw = dmu_raw; a.dlv_raw = dlv_raw; a.dlh = dlh; a.dcomb = dcomb; a.sync = sync_ws;
    a.B = B; a.H2 = 2 * H; a.Z = Z; a.T = T; a.RX = ceil_div(B, 8); a.phases = phases; a.launch = (phases == 2) ? 1 : 0;
    a.fb_min = free_bits > 0.0f ? free_bits / (float)Z : 0.0f;
    a.dbg = arcvae_env_int("ARCVAE_SEAM_DEBUG", 0);
    const size_t lds = seam_lds_bytes(2 * H, Z);
    (void)hipFuncSetAttribute((const void*)enc_seam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(enc_seam_kernel, dim3(256), dim3(256), lds, stream, a);
    return arcvae_launch_status();
}

// =============================================================================================
// Stand-alone forms of the loss pieces, for the reference's `losses` module API
// (losses/recon.py, losses/kl.py, losses/info.py called on arbitrary tensors).  The training step
// does not use these: it runs the fused kernels above.
// =============================================================================================
namespace {

__global__ __launch_bounds__(256) void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                      const float* __restrict__ eps, float* z, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = mu[i] + eps[i] * expf(0.5f * logvar[i]);
}

// stats (layout at the top of this file) from given mu/logvar; krow[b] (optional) = per-sample free-bits KL
__global__ __launch_bounds__(256) void latent_stats_kernel(const float* __restrict__ mu,
                                                           const float* __restrict__ logvar, float* stats,
                                                           float* krow, int B, int Z, float fb_min) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float klmi = 0.f, klfb = 0.f;
    for (int j = lane; j < Z; j += 64) {
        const long i = (long)b * Z + j;
        const float mc = clipf(mu[i], -3.0f, 3.0f), lc = clipf(logvar[i], -6.0f, 3.0f);
        const float var = expf(lc);
        const float k = -0.5f * (1.0f + lc - mc * mc - var);
        atomicAdd(stats + j, mc);
        atomicAdd(stats + Z + j, var);
        klmi += k;
        float kf = fmaxf(k, 0.0f);
        if (fb_min > 0.0f) kf = fmaxf(kf, fb_min);
        klfb += kf;
    }
    klmi = wave_sum(klmi);
    klfb = wave_sum(klfb);
    if (lane == 0) {
        atomicAdd(stats + 2 * Z, klmi);
        atomicAdd(stats + 2 * Z + 1, klfb);
        atomicAdd(stats + 2 * Z + 2, 1.0f);
        if (krow) krow[b] = klfb;
    }
}

// ce[r] = logsumexp(logits[r]) - logits[r, target[r]]   (one wave per row)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits,
                                                      const int32_t* __restrict__ targets, float* ce, long R, int V) {
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
    if (lane == 0) {
        const int t = min(max(targets[r], 0), V - 1);
        ce[r] = -((x[t] - m) - logf(s));
    }
}

__global__ __launch_bounds__(256) void sum_scaled_kernel(const float* __restrict__ x, long n, float* out, float scale) {
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) s += x[i];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

}  // namespace

// models/encoder.py:147-153 with eps supplied by the caller.
extern "C" int arcvae_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, long n,
                                     hipStream_t stream) {
    if (!mu || !logvar || !eps || !z || n <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mu, logvar, eps, z, n);
    return arcvae_launch_status();
}

// losses/kl.py:39-58 + losses/info.py:27-35 partial sums for given mu/logvar (stats zeroed here).
extern "C" int arcvae_latent_stats(const float* mu, const float* logvar, float* stats, float* krow, int B, int Z,
                                   float free_bits, hipStream_t stream) {
    if (!mu || !logvar || !stats || B <= 0 || Z <= 0) return ARCVAE_ERR_ARG;
    if (arcvae_zero(stats, 1, 2 * Z + 4, 2 * Z + 4, stream) != ARCVAE_OK) return ARCVAE_ERR_LAUNCH;
    const float fb_min = free_bits > 0.0f ? free_bits / (float)Z : 0.0f;
    hipLaunchKernelGGL(latent_stats_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, stream, mu, logvar, stats, krow, B, Z,
                       fb_min);
    return arcvae_launch_status();
}

// losses/recon.py:29-57 per-position cross entropy.
extern "C" int arcvae_ce_rows(const float* logits, const int32_t* targets, float* ce, long R, int V,
                              hipStream_t stream) {
    if (!logits || !targets || !ce || R <= 0 || V <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, logits, targets, ce, R, V);
    return arcvae_launch_status();
}

// out[0] = scale * sum(x)
extern "C" int arcvae_sum(const float* x, long n, float* out, float scale, hipStream_t stream) {
    if (!x || !out || n <= 0) return ARCVAE_ERR_ARG;
    hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(256), 0, stream, x, n, out, scale);
    return arcvae_launch_status();
}
