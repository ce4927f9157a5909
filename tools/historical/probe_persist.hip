// Diagnostic (not shipped): how fast can ONE persistent launch tick through an LSTM-shaped recurrence when the batch
// rows are partitioned over the 8 XCDs?  Every XCD then owns its rows' recurrence outright: h_{t-1} is produced and
// consumed inside one XCD (one L2: coherent without fences if loads bypass L1), the weights are stationary in LDS, and
// the per-tick barrier spans the 32 CUs of one XCD only (a flag line in that L2), never the chip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_persist.hip -o gpurun_out/probe_persist
// Shape: H = 256, 3 sources per tick (Wh0, Wx1, Wh1 as in the 2-layer forward wavefront), 8 rows per XCD (B = 64),
// 32 gate columns (8 units) per CU.  MODE: 0 barrier only, 1 + A loads, 2 + MFMA (W from LDS), 3 + reduce/cell/stores,
// 4 = the full tick WITHOUT flags: the exchanged values carry their own sequence tag (8-byte {value, tick} stores, the
// consumer re-polls its operand lines until every tag is the tick it expects) -- no store wait, no flag line, no
// per-XCD barrier (DESIGN.md section 9, next lever 1; written at the end of round 1, not yet run on the GPU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 256, NSRC = 3, ROWS = 8, COLS = 32, NXCD = 8, NCU = 32;

struct Args {
    const float* W;      // [NSRC][4H rows][H] k-chunk-major: [src][k/16][4H][16]
    float* hbuf;         // [2 slots][NSRC][64 rows][H]  (A operands of the next tick)
    uint2* hx;           // MODE 4: the same slots as {value bits, tag} pairs; tag = tick that consumes the value
    unsigned* flags;     // [NXCD][32]   (one 128-B line per XCD)
    unsigned* cnt;       // [NXCD] role counters
    unsigned* info;      // [256][2] (xcc, role) per block
    unsigned* err;
    int T;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void persist_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                               // [NSRC][16 chunks][COLS][16]  = 96 KB
    float* red = lds + NSRC * 16 * COLS * 16;      // [4 waves][16][32]
    __shared__ unsigned s_role, s_xcc, s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_xcc = xcc_id();
        s_role = atomicAdd(&a.cnt[s_xcc], 1u);
        a.info[2 * blockIdx.x] = s_xcc; a.info[2 * blockIdx.x + 1] = s_role;
        s_ok = 1;
    }
    __syncthreads();
    const unsigned xcc = s_xcc, role = s_role;
    if (xcc >= NXCD || role >= NCU) return;        // surplus block on this XCD (reported through info)
    // stationary weights: my 32 columns of each source
    for (int i = tid; i < NSRC * 16 * COLS * 4; i += 256) {
        const int s = i / (16 * COLS * 4), rem = i % (16 * COLS * 4);
        const int kc = rem / (COLS * 4), c4 = rem % (COLS * 4);
        const int col = c4 >> 2, q = c4 & 3;
        reinterpret_cast<float4*>(wl)[i] =
            *reinterpret_cast<const float4*>(a.W + (((long)s * 16 + kc) * 4 * H + role * COLS + col) * 16 + q * 4);
    }
    __syncthreads();
    const int r = lane & 15, q4 = (lane >> 4) * 4;
    const int arow = xcc * ROWS + (r & 7);          // rows 8..15 of the MFMA tile repeat rows 0..7 (ignored)
    float cst = 0.f;                                // cell state of my (row, unit) stays in a register
    for (int t = 0; t < a.T; ++t) {
        if (MODE >= 4 && MODE != 7 && MODE != 10 && MODE != 11) {
            // ---- tagged exchange: my A operands ARE the synchronisation.  Lane (r, q4) of wave w needs k = 16c + q4 .. +3
            // of its row for its 4 chunks: 4 pairs = 32 B = two 16-byte loads per chunk and source.
            const uint2* hb = a.hx + (long)(t & 1) * NSRC * 64 * H;
            f32x4 fa[NSRC][4];
            unsigned spins = 0;
            while (true) {
                bool ok = true;
                if (MODE == 9) {   // sentinel: ONE 16-byte load per source (last chunk) until its tags match, THEN everything
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, NSRC * 64 * H * 8, 0x00020000);
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) {
                        const unsigned off = (unsigned)((((long)s * 64 + arow) * H + (wave * 4 + 3) * 16 + q4) * 8);
                        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                        ok = ok && lo.y == (unsigned)t && lo.w == (unsigned)t;
                    }
                    if (!__all(ok)) {
                        if (++spins > 2000000u) { if (lane == 0) atomicAdd(a.err, 1u); return; }
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                }
#pragma unroll
                for (int s = 0; s < NSRC; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x4* p = reinterpret_cast<const u32x4*>(hb + ((long)s * 64 + arow) * H + (wave * 4 + c) * 16 + q4);
                        u32x4 lo, hi;
                        if (MODE == 4) { lo = __builtin_nontemporal_load(p); hi = __builtin_nontemporal_load(p + 1); }
                        else {   // MODE 5, 6, 8: agent-scope (sc1) 16-byte loads: bypass L1 by the memory model, not by a hint
                            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, NSRC * 64 * H * 8, 0x00020000);
                            const unsigned off = (unsigned)((((long)s * 64 + arow) * H + (wave * 4 + c) * 16 + q4) * 8);
                            lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                            hi = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16);
                        }
                        fa[s][c] = f32x4{__uint_as_float(lo.x), __uint_as_float(lo.z), __uint_as_float(hi.x), __uint_as_float(hi.z)};
                        ok = ok && lo.y == (unsigned)t && lo.w == (unsigned)t && hi.y == (unsigned)t && hi.w == (unsigned)t;
                    }
                if (__all(ok)) break;
                if (++spins > 200000u) { if (lane == 0) atomicAdd(a.err, 1u); return; }   // every wave gives up on its own
            }
            f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kc = wave * 4 + c;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (((long)s * 16 + kc) * COLS + 16 * n + r) * 16 + q4);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].x, w.x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].y, w.y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].z, w.z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].w, w.w, acc[n], 0, 0, 0);
                    }
                }
            float* redt = red;
            if (MODE == 8 || MODE == 9) redt = red + (t & 1) * 2048;   // double-buffered reduction tile: ONE block barrier per tick
            else __syncthreads();                  // red of the previous tick has been read by everybody
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) redt[wave * 512 + ((lane >> 4) * 4 + reg) * 32 + 16 * n + r] = acc[n][reg];
            __syncthreads();
            {
                const int row = tid >> 5, col = tid & 31;
                float v = (redt[row * 32 + col] + redt[512 + row * 32 + col]) + (redt[1024 + row * 32 + col] + redt[1536 + row * 32 + col]);
                const float g = tanhf(v * 0.01f);
                cst = 0.5f * cst + g;
                uint2* ho = a.hx + (long)((t + 1) & 1) * NSRC * 64 * H;
#pragma unroll
                for (int s = 0; s < NSRC; ++s)       // one 8-byte store per value: the tag arrives with it or not at all
                    if (col < 8) {
                        uint2* dst = ho + ((long)s * 64 + xcc * ROWS + row) * H + role * 8 + col;
                        const unsigned long long pk = ((unsigned long long)(unsigned)(t + 1) << 32) | __float_as_uint(cst);
                        if (MODE == 6) __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through, drops the L2 line
                        else *dst = make_uint2(__float_as_uint(cst), (unsigned)(t + 1));
                    }
            }
            continue;
        }
        // ---- wait until all 32 CUs of my XCD have published tick t-1
        if (t > 0) {
            if (wave == 0) {
                unsigned spins = 0;
                while (true) {
                    unsigned v = (lane < NCU) ? __hip_atomic_load(a.flags + xcc * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)t;  // sc1 load: served by the XCD's L2
                    if (__all((int)(v - (unsigned)t) >= 0)) break;
                    if (++spins > 200000u) { if (lane == 0) { atomicAdd(a.err, 1u); s_ok = 0; } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (!s_ok) return;
        }
        if (MODE == 0) {
            if (tid == 0) __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store: stays in the XCD's L2
            continue;
        }
        // ---- A operands of this tick: slot t&1, rows of my XCD, loads bypass L1 (served by the XCD's L2)
        const float* hb = a.hbuf + (long)(t & 1) * NSRC * 64 * H;
        f32x4 fa[NSRC][4];
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {         // wave w owns chunks [4w, 4w+4) of the 16 chunks of a source
                if (MODE == 7 || MODE == 10) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, NSRC * 64 * H * 4, 0x00020000);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((((long)s * 64 + arow) * H + (wave * 4 + c) * 16 + q4) * 4), 0, 16);
                    fa[s][c] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
                } else
                fa[s][c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(hb + ((long)s * 64 + arow) * H + (wave * 4 + c) * 16 + q4));
            }
        if (MODE == 1) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) sum += fa[s][c].x;
            if (sum == 1234.5f) a.hbuf[0] = sum;
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store: stays in the XCD's L2
            continue;
        }
        f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kc = wave * 4 + c;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (((long)s * 16 + kc) * COLS + 16 * n + r) * 16 + q4);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].x, w.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].y, w.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].z, w.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][c].w, w.w, acc[n], 0, 0, 0);
                }
            }
        if (MODE == 2) {
            if (acc[0][0] + acc[1][3] == 1234.5f) a.hbuf[1] = acc[0][1];
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store: stays in the XCD's L2
            continue;
        }
        // ---- reduce the 4 waves' partial tiles, cell update for (row = tid>>5 (0..7), col = tid&31), publish h
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[wave * 512 + ((lane >> 4) * 4 + reg) * 32 + 16 * n + r] = acc[n][reg];
        __syncthreads();
        {
            const int row = tid >> 5, col = tid & 31;   // 8 rows x 32 gate columns
            float v = (red[row * 32 + col] + red[512 + row * 32 + col]) + (red[1024 + row * 32 + col] + red[1536 + row * 32 + col]);
            const float g = tanhf(v * 0.01f);
            cst = 0.5f * cst + g;
            // every source slab of the next slot gets this block's columns (stand-in for h0, h0, h1 of the real wavefront)
            float* ho = a.hbuf + (long)((t + 1) & 1) * NSRC * 64 * H;
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
                if (col < 8) ho[((long)s * 64 + xcc * ROWS + row) * H + role * 8 + col] = cst;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my stores have reached L2 (a workgroup-scope release fence emits no vmcnt wait)
        __syncthreads();
        if (tid == 0) {
            if (MODE == 7 || MODE == 11) __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 flag store
            else __hip_atomic_store(a.flags + xcc * 32 + role, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store: stays in the XCD's L2
        }
    }
}

template <int MODE>
int run(const char* name, Args a, int lds_bytes) {
    CK(hipMemset(a.flags, 0, NXCD * 32 * 4)); CK(hipMemset(a.cnt, 0, NXCD * 4)); CK(hipMemset(a.err, 0, 4));
    CK(hipMemset(a.hx, 0, (size_t)2 * NSRC * 64 * H * 8));      // tag 0 = what tick 0 expects
    CK(hipFuncSetAttribute((const void*)persist_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(persist_kernel<MODE>, dim3(256), dim3(256), lds_bytes, 0, a);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipMemset(a.flags, 0, NXCD * 32 * 4)); CK(hipMemset(a.cnt, 0, NXCD * 4));
    CK(hipMemset(a.hx, 0, (size_t)2 * NSRC * 64 * H * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(persist_kernel<MODE>, dim3(256), dim3(256), lds_bytes, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned err, cnt[NXCD]; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cnt, a.cnt, sizeof(cnt), hipMemcpyDeviceToHost));
    printf("%-44s %8.1f us total  %6.2f us per tick  (timeouts %u; blocks per XCD:", name, ms * 1e3, ms * 1e3 / a.T, err);
    for (int i = 0; i < NXCD; ++i) printf(" %u", cnt[i]);
    printf(")\n");
    return 0;
}

int main() {
    Args a; a.T = 129;
    float* W; CK(hipMalloc(&W, (size_t)NSRC * 4 * H * H * 4));
    std::vector<float> hw((size_t)NSRC * 4 * H * H); for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&a.hbuf, (size_t)2 * NSRC * 64 * H * 4)); CK(hipMemset(a.hbuf, 0, (size_t)2 * NSRC * 64 * H * 4));
    CK(hipMalloc(&a.hx, (size_t)2 * NSRC * 64 * H * 8));
    CK(hipMalloc(&a.flags, NXCD * 32 * 4)); CK(hipMalloc(&a.cnt, NXCD * 4)); CK(hipMalloc(&a.info, 256 * 2 * 4)); CK(hipMalloc(&a.err, 4));
    a.W = W;
    const int lds_bytes = (NSRC * 16 * COLS * 16 + 2 * 4 * 16 * 32) * 4;
    printf("persistent per-XCD recurrence probe: T = %d ticks, %d KB LDS per block\n", a.T, lds_bytes / 1024);
    if (run<0>("barrier only (32-CU flag line per XCD)", a, lds_bytes)) return 1;
    if (run<1>("+ A operand loads (L1-bypassing)", a, lds_bytes)) return 1;
    if (run<2>("+ 96 MFMAs per wave, W from LDS", a, lds_bytes)) return 1;
    if (run<3>("+ reduce, cell update, h stores (full tick)", a, lds_bytes)) return 1;
    if (run<7>("full tick, flags: sc1 data loads + agent flag store", a, lds_bytes)) return 1;
    if (run<10>("full tick, flags: sc1 data loads, plain flag store", a, lds_bytes)) return 1;
    if (run<11>("full tick, flags: nt data loads, agent flag store", a, lds_bytes)) return 1;
    if (run<9>("tagged: sentinel poll, then load + verify", a, lds_bytes)) return 1;
    if (run<4>("tagged exchange: nt loads, plain stores", a, lds_bytes)) return 1;
    if (run<5>("tagged exchange: sc1 loads, plain stores", a, lds_bytes)) return 1;
    if (run<6>("tagged exchange: sc1 loads, sc1 stores", a, lds_bytes)) return 1;
    if (run<8>("tagged: sc1 loads, plain stores, 1 barrier", a, lds_bytes)) return 1;
    return 0;
}
