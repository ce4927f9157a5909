// Per-XCD exchange helpers of the persistent kernels (lstm.hip: the sweeps; latent.hip: the fused seam): data produced and
// consumed under ONE XCD's L2, flags on lines of their own, the sticky error word.  Moved here from lstm.hip in round 4.
#pragma once
#include "common.h"

constexpr int PS_ERR = 500;                                  // sticky error word of sync_ws, outside the re-armed ranges
// Fused seam kernel (latent.hip: enc_seam_kernel): its flag lines [8 XCDs][32], role counters of its (up to) two launches per
// step [2][8], and the device-wide arrival counter on a line of its own -- behind the sweeps' words; a step re-arms PSEAM_WORDS
// words of sync_ws (arcvae_enc_prologue n_sync)
constexpr int PSEAM = 4864, PSEAM_CNT = PSEAM + 256, PSEAM_GCNT = PSEAM + 288, PSEAM_PART = PSEAM + 320, PSEAM_WORDS = PSEAM + 832;
// (PSEAM_PART: [8 XCDs][32 CUs][2] floats -- every CU's partial KL sums, reduced per XCD before they reach `stats`)

// "My stores are in the XCD's L2" -- what the flag protocol of the persistent kernels needs before a flag may be raised.
// A workgroup-scope release fence is NOT that: on gfx942/gfx950 (not in tgsplit mode) the compiler emits no vmcnt wait
// for it (waves of a workgroup share their CU's L1), so the flag -- a different L2 channel than the data -- could
// overtake the data it announces: seen as 1e-3-level deviations in the layer-0 gradients of ~1 step in 4 beside busy
// GEMM streams (tools/race_hunt.py).  vmcnt counts a store until the L2 has acknowledged it.
__device__ __forceinline__ void ps_stores_in_l2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Loads of data another CU of the XCD has just written: AGENT-scope loads (`sc1`: served by the L2, never by this CU's
// L1 -- by the ISA's definition of the scope, not by a cache hint).  Raw buffer loads, because the builtin carries the
// cache-policy bits (aux 16 = sc1) and stays inside the compiler's vmcnt bookkeeping.  Measured on tools/probe_persist
// (profiles/r02_probe_persist.txt): 2.72 us per tick against 2.82 with the `nt` loads of round 1.
// What the consumer side relies on, stated once: (1) every load of exchanged bytes is one of these; (2) the producer's
// stores are plain (vector L1 is write-through: they land in the XCD's L2, the coherence point of every CU that can
// read them HERE, because a row's producers and consumers sit on ONE XCD by construction -- ps_xcc_id() -- and a block
// that finds another layout raises PS_ERR); (3) every storing wave drains its stores (ps_stores_in_l2) before the
// workgroup's barrier that precedes the flag; (4) the flag is a plain store to a line of its own, polled with sc1
// loads.  An agent-scope (sc1) flag or data STORE writes through to memory and drops the line from the L2 -- the
// same-XCD reader then misses: +0.55 us per tick (same profile) for bytes no other XCD ever reads.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// `base` and `bytes` are wave-uniform at every call site; readfirstlane says so to the compiler, which otherwise wraps
// each load in a waterfall loop over the descriptor (v_readfirstlane / s_and_saveexec with a vmcnt(0) inside: every
// load a serialised round trip -- seen in the .s of this file before this line was added).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ps_rsrc(const void* base, long bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 ps_load_sc1_x4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16);
    return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ float ps_load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 16));
}

__device__ __forceinline__ unsigned ps_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

