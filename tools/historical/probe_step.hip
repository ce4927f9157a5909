// Diagnostic (not shipped): where does an LSTM step launch spend its time?
// Times chains of dependent launches of cut-down variants of the forward step kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mlx-vae_amd/csrc tools/probe_step.hip -o gpurun_out/probe_step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "skinny.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Job {
    const float* xin; const float* Wx; const float* hprev; const float* Wh; const float* pre;
    const float* cprev; float* h; float* c; float* gates;
};
struct Args { Job job[8]; int B, H; };

// MODE 0 empty, 1 loads only, 2 loads+mfma+lds, 3 full (no token table), 4 full but cheap pointwise
template <int MODE, int CH, bool TILED>
__global__ __launch_bounds__(256) void step_kernel(Args a) {
    if (MODE == 0) return;
    __shared__ float red[4 * 256];
    __shared__ float act[256];
    const Job& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H;
    const int r0 = blockIdx.y * 16, u0 = blockIdx.x * 4;
    const int arow = min(r0 + (lane & 15), B - 1);
    const int jc = lane & 15;
    const long wrow = (long)(jc >> 2) * H + u0 + (jc & 3);
    SkinnyFrag<CH> f1, f2;
    if (TILED) {
        // k-chunk-major layouts: A_t[(k/16)][b][k%16], W_t[(k/16)][permuted row][k%16]: one wave-instruction = 1 KB contiguous
        const int q4 = (lane >> 4) * 4;
        const long wr = (long)blockIdx.x * 16 + jc;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const long kc = wave * CH + c;
            f1.a[c] = *reinterpret_cast<const float4*>(j.xin + (kc * B + arow) * 16 + q4);
            f1.w[c] = *reinterpret_cast<const float4*>(j.Wx + (kc * 4 * H + wr) * 16 + q4);
            f2.a[c] = *reinterpret_cast<const float4*>(j.hprev + (kc * B + arow) * 16 + q4);
            f2.w[c] = *reinterpret_cast<const float4*>(j.Wh + (kc * 4 * H + wr) * 16 + q4);
        }
    } else {
        skinny_load<CH>(f1, j.xin, (long)arow * H, j.Wx, wrow * H, wave, lane);
        skinny_load<CH>(f2, j.hprev, (long)arow * H, j.Wh, wrow * H, wave, lane);
    }
    if (MODE == 1) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) s += f1.a[c].x + f1.w[c].y + f2.a[c].z + f2.w[c].w;
        if (s == 12345.678f) j.h[0] = s;
        return;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    skinny_mfma<CH>(f1, acc0, acc1);
    skinny_mfma<CH>(f2, acc0, acc1);
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    __syncthreads();
    if (MODE == 2) {
        if (tid < 64) j.h[(long)min(r0 + (tid >> 2), B - 1) * H + u0 + (tid & 3)] = skinny_reduced_n<4>(red, tid >> 2, tid & 3);
        return;
    }
    {
        const int row = tid >> 4, col = tid & 15;
        const int b = r0 + row;
        if (b < B) {
            const int gcol = (col >> 2) * H + u0 + (col & 3);
            const float v = skinny_reduced_n<4>(red, row, col) + j.pre[gcol];
            float av;
            if (MODE == 4) av = v * 0.5f;
            else av = ((col >> 2) == 2) ? tanhf(v) : sigmoidf_acc(v);
            act[tid] = av;
            j.gates[(long)b * 4 * H + gcol] = av;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int row = tid >> 2, u = tid & 3;
        const int b = r0 + row;
        if (b < B) {
            const float i = act[row * 16 + u], f = act[row * 16 + 4 + u], g = act[row * 16 + 8 + u], o = act[row * 16 + 12 + u];
            const long hb = (long)b * H + u0 + u;
            const float c = f * j.cprev[hb] + i * g;
            j.h[hb] = (MODE == 4) ? o * c : o * tanhf(c);
            j.c[hb] = c;
        }
    }
}

// TILED + 32 gate columns per block (A fragment reused for two W fragments): 1 block per CU at B=64, 2 jobs
template <int CH>
__global__ __launch_bounds__(256) void step_kernel_bn32(Args a) {
    __shared__ float red[2 * 4 * 256];
    __shared__ float act[512];
    const Job& j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H;
    const int r0 = blockIdx.y * 16, u0 = blockIdx.x * 8;
    const int arow = min(r0 + (lane & 15), B - 1);
    const int jc = lane & 15;
    const int q4 = (lane >> 4) * 4;
    const long wr = (long)blockIdx.x * 32 + jc;
    float4 a1[CH], a2[CH], w1a[CH], w1b[CH], w2a[CH], w2b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const long kc = wave * CH + c;
        a1[c] = *reinterpret_cast<const float4*>(j.xin + (kc * B + arow) * 16 + q4);
        w1a[c] = *reinterpret_cast<const float4*>(j.Wx + (kc * 4 * H + wr) * 16 + q4);
        w1b[c] = *reinterpret_cast<const float4*>(j.Wx + (kc * 4 * H + wr + 16) * 16 + q4);
        a2[c] = *reinterpret_cast<const float4*>(j.hprev + (kc * B + arow) * 16 + q4);
        w2a[c] = *reinterpret_cast<const float4*>(j.Wh + (kc * 4 * H + wr) * 16 + q4);
        w2b[c] = *reinterpret_cast<const float4*>(j.Wh + (kc * 4 * H + wr + 16) * 16 + q4);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, bcc0 = {0.f, 0.f, 0.f, 0.f}, bcc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        { SKINNY_MFMA4(a1[c], w1a[c]) }
        { SKINNY_MFMA4(a2[c], w2a[c]) }
    }
    skinny_store_partial_n(red, acc0, acc1, wave, lane);
    acc0 = bcc0; acc1 = bcc1;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        { SKINNY_MFMA4(a1[c], w1b[c]) }
        { SKINNY_MFMA4(a2[c], w2b[c]) }
    }
    skinny_store_partial_n(red + 1024, acc0, acc1, wave, lane);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = tid >> 4, col = tid & 15;
        const int b = r0 + row;
        const int gcol = (col >> 2) * H + u0 + h * 4 + (col & 3);
        const float v = skinny_reduced_n<4>(red + h * 1024, row, col) + j.pre[gcol];
        const float av = ((col >> 2) == 2) ? tanhf(v) : sigmoidf_acc(v);
        act[h * 256 + tid] = av;
        if (b < B) j.gates[(long)b * 4 * H + gcol] = av;
    }
    __syncthreads();
    if (tid < 128) {
        const int h = tid >> 6, t = tid & 63;
        const int row = t >> 2, u = t & 3;
        const int b = r0 + row;
        if (b < B) {
            const float* ac = act + h * 256 + row * 16;
            const float i = ac[u], f = ac[4 + u], g = ac[8 + u], o = ac[12 + u];
            const long hb = (long)b * H + u0 + h * 4 + u;
            const float c = f * j.cprev[hb] + i * g;
            j.h[hb] = o * tanhf(c);
            j.c[hb] = c;
        }
    }
}

template <int MODE, bool TILED = false>
int run(const char* name, int B, int H, int T, int njobs, int blocks_x_div, float* buf, hipStream_t s) {
    const long sH = (long)B * H, sG = (long)B * 4 * H;
    float* W = buf;                       // 4 weight matrices [4H,H]
    float* bias = W + 4L * 4 * H * H;
    float* hseq = bias + 4 * H;           // [2][T][B][H]
    float* cseq = hseq + 2L * T * sH;
    float* gseq = cseq + 2L * T * sH;     // [2][T][B][4H]
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 1; t < T; ++t) {
        Args a; a.B = B; a.H = H;
        for (int l = 0; l < njobs; ++l) {
            Job& j = a.job[l];
            j.xin = hseq + ((1 - l) * (long)T + t) * sH;  // any [B,H] slab written earlier
            j.Wx = W + (2L * l) * 4 * H * H; j.Wh = W + (2L * l + 1) * 4 * H * H; j.pre = bias;
            j.hprev = hseq + (l * (long)T + t - 1) * sH; j.cprev = cseq + (l * (long)T + t - 1) * sH;
            j.h = hseq + (l * (long)T + t) * sH; j.c = cseq + (l * (long)T + t) * sH;
            j.gates = gseq + (l * (long)T + t) * sG;
        }
        for (int k = njobs; k < 8; ++k) a.job[k] = a.job[0];
        dim3 grid(H / 4 / blocks_x_div, (B + 15) / 16, njobs);
        if (MODE == 9) { grid.x = H / 8; hipLaunchKernelGGL((step_kernel_bn32<4>), grid, dim3(256), 0, s, a); }
        else hipLaunchKernelGGL((step_kernel<(MODE == 9 ? 3 : MODE), 4, TILED>), grid, dim3(256), 0, s, a);
    }
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s B=%4d jobs=%d grid.x=%3d : %7.2f us per launch\n", name, B, njobs, H / 4 / blocks_x_div, 1e3 * ms / reps / (T - 1));
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return 0;
}

int main() {
    const int H = 256, T = 128, BMAX = 256;
    const long n = 4L * 4 * H * H + 4 * H + 2L * T * BMAX * H * 2 + 2L * T * BMAX * 4 * H;
    float* buf; CK(hipMalloc(&buf, n * sizeof(float)));
    std::vector<float> host(n);
    for (long i = 0; i < n; ++i) host[i] = 0.01f * (float)((i * 2654435761u) % 200) - 1.0f;
    CK(hipMemcpy(buf, host.data(), n * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int B : {64, 256}) {
        run<0>("empty kernel", B, H, T, 2, 1, buf, s);
        run<0>("empty kernel, 1/8 grid", B, H, T, 2, 8, buf, s);
        run<1>("loads only", B, H, T, 2, 1, buf, s);
        run<2>("loads + mfma + lds reduce", B, H, T, 2, 1, buf, s);
        run<4>("full, cheap pointwise", B, H, T, 2, 1, buf, s);
        run<3>("full", B, H, T, 2, 1, buf, s);
        run<3>("full, 1 job", B, H, T, 1, 1, buf, s);
        run<1, true>("TILED loads only", B, H, T, 2, 1, buf, s);
        run<2, true>("TILED loads + mfma + lds", B, H, T, 2, 1, buf, s);
        run<3, true>("TILED full", B, H, T, 2, 1, buf, s);
        run<3, true>("TILED full, 1 job", B, H, T, 1, 1, buf, s);
        run<9, true>("TILED BN=32 full", B, H, T, 2, 1, buf, s);
        run<9, true>("TILED BN=32 full, 1 job", B, H, T, 1, 1, buf, s);
    }
    return 0;
}
