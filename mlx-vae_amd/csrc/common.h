// Shared device/host helpers for the AR-CVAE gfx950 kernels.
// Wavefront = 64 lanes everywhere in this directory (CDNA4); block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The PUBLIC header is the single source of the C ABI: error codes, flag bits and every extern "C" prototype.  Including it here
// (with the real hipStream_t behind arcvae_stream_t) makes a definition in this directory that drifts from its declaration a
// compile error -- csrc/ used to re-declare flags and prototypes in ops.h, and include/arcvae_hip.h could (and did) go stale.
#define ARCVAE_HIP_BUILD 1
#include "arcvae_hip.h"

#define ARCVAE_MAX_LAYERS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int arcvae_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ARCVAE_OK : ARCVAE_ERR_LAUNCH;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- tuning knobs (read once per process; the captured hipGraph segments keep what they were recorded with) ----
// (defaults: ARCVAE_STEP_PRIO=3, ARCVAE_XCD_REMAP=1, ARCVAE_SIDE_MAX_BLOCKS=0 i.e. no cap)
// ARCVAE_STEP_PRIO      0..3  s_setprio of the LSTM step kernels' waves: they share CUs with the side-stream GEMMs
//                             and every cycle a step wave waits for an issue slot is on the dependent chain.
// ARCVAE_SIDE_MAX_BLOCKS n    cap on resident blocks per CU of the kernels that run BESIDE the chain (weight-gradient
//                             and decoder GEMMs, segment/column sums), enforced by padding their LDS allocation:
//                             a step launch can only start on a CU that still has wave slots and registers free.
#include <stdlib.h>
static inline int arcvae_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
static inline int arcvae_step_prio() {
    static const int v = arcvae_env_int("ARCVAE_STEP_PRIO", 3);
    return v < 0 ? 0 : (v > 3 ? 3 : v);
}
// Slots of the per-layer rings over t (k-chunk-major operand copies, dc and dX slabs): a slab is produced by one launch
// and consumed by the next only, so a short ring keeps them cache-resident instead of streaming [L,T,..] buffers
// through the Infinity Cache.  16, not 2: re-writing a line two launches after other XCDs read it costs +0.3 us per
// BPTT launch (measured: 2-4 slots 5.3 us, >= 8 slots 5.05 us; whole step 1.58 / 1.55 ms).  ARCVAE_RING=0: T slots.
static inline int arcvae_ring_slots(int T) {
    const int v = arcvae_env_int("ARCVAE_RING", 16);
    return (v <= 0 || v > T) ? T : v;
}
static inline int arcvae_xcd_remap() {
    static const int v = arcvae_env_int("ARCVAE_XCD_REMAP", 1);
    return v != 0;
}
// dynamic-LDS pad (bytes) that limits a kernel with `own` bytes of LDS to the configured blocks per CU
static inline unsigned arcvae_side_lds_pad(unsigned own) {
    static const int n = arcvae_env_int("ARCVAE_SIDE_MAX_BLOCKS", 0);
    if (n <= 0) return 0;
    const unsigned cu_lds = 160u * 1024u, cap = 64u * 1024u;
    unsigned want = cu_lds / (unsigned)n;          // n blocks fit, n+1 do not
    want = (want / 512u) * 512u;
    if (want > cap) want = cap;
    if (cu_lds / want != (unsigned)n && want == cap) { /* n < 3 cannot be enforced below the 64 KB block limit */ }
    return want > own + 512u ? want - own - 256u : 0u;
}
__device__ __forceinline__ void arcvae_set_prio(int p) {
    if (p == 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

// Accurate (non fast-math) transcendental forms: parity mode needs ~1-2 ulp expf/tanhf.
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// Activations of the PERSISTENT sweeps, where one wave per layer evaluates them on the dependent chain of every tick.
// Default: the libm forms (~180 VALU instructions per cell: 3 sigmoids of ~30, 2 tanh of ~43).  -DARCVAE_CHAIN_FAST_ACT
// switches to hardware forms (v_exp_f32 / v_rcp_f32 are 1-ulp instructions; sigmoid = rcp(1 + exp2(-x log2 e)), ~4e-7
// relative; tanh = (1 - e) / (1 + e), e = exp(-2|x|), with an odd polynomial below |x| = 0.25 where the quotient loses
// relative accuracy to cancellation: ~3e-7 relative everywhere, parity-green at 1e-4).  Measured on one box, A/B of the
// two libraries: 1.308 vs 1.300 ms and 1.101 vs 1.107 ms per step -- no gain beyond noise, so the accurate forms stay.
#ifndef ARCVAE_CHAIN_FAST_ACT
__device__ __forceinline__ float chain_sigmoid(float x) { return sigmoidf_acc(x); }
__device__ __forceinline__ float chain_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float chain_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float chain_tanh(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * ax);
    const float big = copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
    const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
    return ax < 0.25f ? small : big;
}
#endif

// Wave-level reductions over 64 lanes.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
