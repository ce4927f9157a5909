// Shared device/host helpers for the AR-CVAE gfx950 kernels.
// Wavefront = 64 lanes everywhere in this directory (CDNA4); block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ARCVAE_OK 0
#define ARCVAE_ERR_ARG (-1)      // bad shape / null pointer / unsupported size
#define ARCVAE_ERR_LAUNCH (-2)   // hipGetLastError() after a launch
#define ARCVAE_ERR_DEVICE (-3)   // wrong device / no gfx950 device

#define ARCVAE_MAX_LAYERS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int arcvae_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ARCVAE_OK : ARCVAE_ERR_LAUNCH;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Accurate (non fast-math) transcendental forms: parity mode needs ~1-2 ulp expf/tanhf.
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

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
