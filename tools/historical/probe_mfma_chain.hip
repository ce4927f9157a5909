// Issue rate of v_mfma_f32_4x4x1_16B_f32 (the persistent sweeps' product instruction) as a function of the number of
// INDEPENDENT accumulator chains: cycles per instruction for NCH = 1, 2, 4, 8 chains, 1 wave per SIMD.  tools/bin/probe_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void chain(float* out, long long* cyc, int iters) {
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NCH; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH>
void run(float* out, long long* cyc, int waves) {
    const int iters = 2000;
    chain<NCH><<<1, 64 * waves>>>(out, cyc, iters);
    chain<NCH><<<1, 64 * waves>>>(out, cyc, iters);
    long long h;
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("chains %d, %d wave(s) per block: %.2f clock64 ticks per MFMA (16 per iteration)\n", NCH, waves, (double)h / (iters * 16.0));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    for (int w : {1, 4}) { run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<8>(out, cyc, w); run<16>(out, cyc, w); }
    return 0;
}
