// Diagnostic (not shipped): where do the blocks of a 512-block, 2-per-CU persistent grid land?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_census.hip -o gpurun_out/probe_census
// Every block records (XCC id, HW_ID, arrival order on its XCD, start stamp), then waits until all blocks of the grid
// have arrived (bounded spin) so that all of them are resident at once, as the blocks of the two-group persistent sweeps
// (csrc/lstm.hip, NG = 2) must be.  Printed: blocks per XCD, blocks per (XCD, CU key), whether the first 32 arrivals of
// an XCD sit on 32 different CUs (breadth-first dealing) and what the CU-key parity assignment would give.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Rec { unsigned xcc, hwid, order, pad; unsigned long long t0; unsigned long long t1; };

__global__ __launch_bounds__(256, 2) void census(Rec* rec, unsigned* cnt, unsigned* total, unsigned* err, int nblocks) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        Rec r;
        r.xcc = x & 0xf; r.hwid = h; r.order = atomicAdd(cnt + (x & 7), 1u); r.pad = 0;
        r.t0 = wall_clock64();
        atomicAdd(total, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblocks) {
            if (++spins > 2000000u) { atomicAdd(err, 1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        r.t1 = wall_clock64();
        rec[blockIdx.x] = r;
        lds[0] = 1.f;
    }
    __syncthreads();
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 512;
    const int lds_kb = argc > 2 ? atoi(argv[2]) : 56;
    Rec* rec; unsigned *cnt, *total, *err;
    CK(hipMalloc(&rec, sizeof(Rec) * nblocks));
    CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&total, 4)); CK(hipMalloc(&err, 4));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(cnt, 0, 64)); CK(hipMemset(total, 0, 4)); CK(hipMemset(err, 0, 4));
        hipLaunchKernelGGL(census, dim3(nblocks), dim3(256), (size_t)lds_kb * 1024, 0, rec, cnt, total, err, nblocks);
        CK(hipDeviceSynchronize());
        std::vector<Rec> h(nblocks);
        unsigned herr = 0;
        CK(hipMemcpy(h.data(), rec, sizeof(Rec) * nblocks, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_xcc;
        std::map<std::pair<unsigned, unsigned>, std::vector<unsigned>> per_cu;   // (xcc, key) -> arrival orders
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto& r : h) {
            per_xcc[r.xcc]++;
            per_cu[{r.xcc, (r.hwid >> 8) & 0xff}].push_back(r.order);
            tmin = r.t0 < tmin ? r.t0 : tmin; tmax = r.t0 > tmax ? r.t0 : tmax;
        }
        int hist[8] = {0}, first32_distinct = 0, both_halves = 0;
        for (auto& kv : per_cu) {
            hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
            int lo = 0, hi = 0;
            for (unsigned o : kv.second) (o < (unsigned)(nblocks / 16) ? lo : hi)++;
            if (lo == 1) first32_distinct++;
            if (lo >= 1 && hi >= 1) both_halves++;
        }
        printf("rep %d: %d blocks, lds %d KB, give-ups %u, start spread %.2f us\n", rep, nblocks, lds_kb, herr, (tmax - tmin) / 100.0);
        printf("  blocks per XCC:");
        for (auto& kv : per_xcc) printf(" [%u]=%d", kv.first, kv.second);
        printf("\n  distinct (xcc, cu key): %zu; blocks per key histogram:", per_cu.size());
        for (int i = 1; i < 8; ++i) if (hist[i]) printf(" %dx:%d", i, hist[i]);
        printf("\n  keys holding exactly one early arrival (order < %d): %d; keys holding an early AND a late arrival: %d\n",
               nblocks / 16, first32_distinct, both_halves);
        if (rep == 0) {
            printf("  sample (block: xcc hwid[15:8] se sh cu order):");
            for (int b = 0; b < 24 && b < nblocks; ++b)
                printf(" %d:%u/%02x/%u.%u.%u/%u", b, h[b].xcc, (h[b].hwid >> 8) & 0xff, (h[b].hwid >> 13) & 7, (h[b].hwid >> 12) & 1,
                       (h[b].hwid >> 8) & 0xf, h[b].order);
            printf("\n");
        }
    }
    return 0;
}
