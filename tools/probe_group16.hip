// Diagnostic (not shipped; round 4): would a SMALLER exchange domain shorten the persistent sweeps' tick?  Today the 32 CUs of an
// XCD form one group (8 rows, 32 gate columns per CU): every CU reads the same 24 KB per tick (32 readers per line queue at that
// line's L2 channel: 0.80 us from issue to landed) and polls 32 flag words.  Here the XCD's CUs are split into
// 32 / GS groups of GS CUs; a group owns 8 GS / 32 rows and its CUs own 32 x 32 / GS gate columns each -- the same values produced
// and the same matrix work per CU (kept constant on purpose), half the bytes read per CU at GS = 16, GS readers per line, GS flag
// words.  Built from the product's exchange form (sc1 data loads, plain flag store, one vmcnt(0) + block barrier before the flag).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_group16.hip -o gpurun_out/probe_group16 && gpurun_out/probe_group16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, NSRC = 3, NXCD = 8, NCU = 32;
struct Args {
    const float* W;      // [NSRC][k/16][4H][16]
    float* hbuf;         // [2 slots][NSRC][64 rows][H]
    unsigned* flags;     // [NXCD][32]
    unsigned* cnt;       // [NXCD]
    unsigned* err;
    int T;
};
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int GS>
__global__ __launch_bounds__(256) void persist_kernel(Args a) {
    constexpr int ROWS = 8 * GS / 32;            // rows of a group
    constexpr int UNITS = 8 * 32 / GS;           // hidden units a CU produces per row (values per source: ROWS * UNITS = 64 always)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                               // [NSRC][16 chunks][32][16]  = 96 KB (matrix work kept at 32 columns per CU)
    float* red = lds + NSRC * 16 * 32 * 16;
    __shared__ unsigned s_role, s_xcc, s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_xcc = xcc_id(); s_role = atomicAdd(&a.cnt[s_xcc], 1u); s_ok = 1; }
    __syncthreads();
    const unsigned xcc = s_xcc, role = s_role;
    if (xcc >= NXCD || role >= NCU) return;
    const unsigned grp = role / GS, gr = role % GS;          // my group in the XCD, my index in the group
    for (int i = tid; i < NSRC * 16 * 32 * 4; i += 256) {
        const int s = i / (16 * 32 * 4), rem = i % (16 * 32 * 4);
        const int kc = rem / (32 * 4), c4 = rem % (32 * 4);
        reinterpret_cast<float4*>(wl)[i] =
            *reinterpret_cast<const float4*>(a.W + (((long)s * 16 + kc) * 4 * H + role * 32 + (c4 >> 2)) * 16 + (c4 & 3) * 4);
    }
    __syncthreads();
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int arow = xcc * 8 + grp * ROWS + (r % ROWS);       // tile rows repeat the group's rows
    float cst = 0.f;
    const unsigned* gflags = a.flags + xcc * 32 + grp * GS;
    for (int t = 0; t < a.T; ++t) {
        if (t > 0) {
            if (wave == 1) {
                unsigned spins = 0;
                while (true) {
                    unsigned v = (lane < GS) ? __hip_atomic_load(gflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)t;
                    if (__all((int)(v - (unsigned)t) >= 0)) break;
                    if (++spins > 200000u) { if (lane == 0) { atomicAdd(a.err, 1u); s_ok = 0; } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (!s_ok) return;
        }
        const float* hb = a.hbuf + (long)(t & 1) * NSRC * 64 * H;
        f32x4 fa[NSRC][4];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, NSRC * 64 * H * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((((long)s * 64 + arow) * H + (wave * 4 + c) * 16 + q4) * 4), 0, 16);
                fa[s][c] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            }
        f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kc = wave * 4 + c;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (((long)s * 16 + kc) * 32 + 16 * n + r) * 16 + q4);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].x, w.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].y, w.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].z, w.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].w, w.w, acc[n], 0, 0, 0);
                }
            }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[wave * 512 + ((lane >> 4) * 4 + reg) * 32 + 16 * n + r] = acc[n][reg];
        __syncthreads();
        {
            const int row = tid >> 5, col = tid & 31;
            float v = (red[row * 32 + col] + red[512 + row * 32 + col]) + (red[1024 + row * 32 + col] + red[1536 + row * 32 + col]);
            cst = 0.5f * cst + tanhf(v * 0.01f);
            float* ho = a.hbuf + (long)((t + 1) & 1) * NSRC * 64 * H;
            // 64 values per source: the group's ROWS rows x my UNITS units (thread p = tid & 63 of the first wave-pair)
            const int p = tid & 63, prow = p / UNITS, pu = p % UNITS;
            if (tid < 64)
#pragma unroll
                for (int s = 0; s < NSRC; ++s) ho[((long)s * 64 + xcc * 8 + grp * ROWS + prow) * H + gr * UNITS + pu] = cst;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <int GS>
int run(const char* name, Args a, int lds_bytes) {
    CK(hipFuncSetAttribute((const void*)persist_kernel<GS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f; unsigned err = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(a.flags, 0, NXCD * 32 * 4)); CK(hipMemset(a.cnt, 0, NXCD * 4)); CK(hipMemset(a.err, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(persist_kernel<GS>, dim3(256), dim3(256), lds_bytes, 0, a);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        unsigned e; CK(hipMemcpy(&e, a.err, 4, hipMemcpyDeviceToHost)); err += e;
    }
    printf("%-40s %8.1f us total  %6.2f us per tick  (timeouts %u)\n", name, best * 1e3, best * 1e3 / a.T, err);
    return 0;
}
int main() {
    Args a; a.T = 129;
    float* W; CK(hipMalloc(&W, (size_t)NSRC * 4 * H * H * 4));
    std::vector<float> hw((size_t)NSRC * 4 * H * H); for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&a.hbuf, (size_t)2 * NSRC * 64 * H * 4)); CK(hipMemset(a.hbuf, 0, (size_t)2 * NSRC * 64 * H * 4));
    CK(hipMalloc(&a.flags, NXCD * 32 * 4)); CK(hipMalloc(&a.cnt, NXCD * 4)); CK(hipMalloc(&a.err, 4));
    a.W = W;
    const int lds_bytes = (NSRC * 16 * 32 * 16 + 4 * 16 * 32) * 4;
    printf("exchange-domain probe: T = %d ticks, same matrix work and values per CU, groups of GS CUs per XCD\n", a.T);
    if (run<32>("GS = 32 (today: 8 rows per group)", a, lds_bytes)) return 1;
    if (run<16>("GS = 16 (4 rows per group)", a, lds_bytes)) return 1;
    if (run<8>("GS = 8  (2 rows per group)", a, lds_bytes)) return 1;
    if (run<32>("GS = 32 again", a, lds_bytes)) return 1;
    return 0;
}
