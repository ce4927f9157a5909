/* arcvae_hip.h -- C ABI of libarcvae_hip.so: the MI355X (gfx950) kernels behind the SELFIES
 * AR-CVAE training path of Raiden-Makoto/MLX-VAE.
 *
 * The reference has no FFI boundary of its own (its only boundary is the
 * Python module API on top of the third-party `mlx` runtime), so this header IS the kernel
 * boundary a maintainer would bind: each entry point replaces the MLX graph that one reference
 * function builds, named in the comment above it (file:line under the reference tree).
 *
 * Conventions (all entry points):
 *   - return 0 on success, negative on error (ARCVAE_ERR_*); no exceptions cross the ABI;
 *   - no allocation, no ownership transfer: every buffer is caller-owned DEVICE memory,
 *     contiguous row-major float32 unless stated, parameter tensors 16-byte aligned;
 *   - asynchronous on `stream` (a hipStream_t passed as void*); safe under hipGraph capture
 *     (no sync, no malloc); re-entrant: no global mutable host state -- chunk indices, trace buffers and
 *     sync scratch are arguments; the only process-wide values are the ARCVAE_* tuning knobs of the environment (most
 *     are read once per process; the kernel-family selectors -- ARCVAE_STEP_TILE, ARCVAE_RING, ARCVAE_PERSIST*,
 *     ARCVAE_RS_* -- are read at every call, so a caller that changes them must drop its captured graphs);
 *   - `const float* const*` arguments are HOST arrays of device pointers (one per LSTM layer);
 *   - gradients are accumulated ("+="): zero the flat gradient buffer once per step;
 *   - float32 arithmetic throughout (the reference's dtype), contractions on the exact-f32
 *     MFMA forms; hidden_dim a multiple of 64 and <= 512, num_layers <= 8, 1 <= num_conditions <= 8,
 *     vocab_size <= 255 (anything else is ARCVAE_ERR_ARG, never a silent fallback).
 */
#ifndef ARCVAE_HIP_H
#define ARCVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCVAE_OK 0
#define ARCVAE_ERR_ARG (-1)
#define ARCVAE_ERR_LAUNCH (-2)
#define ARCVAE_ERR_DEVICE (-3)

#define ARCVAE_GEMM_ACCUMULATE 1 /* C += ...                                   */
#define ARCVAE_GEMM_TANH 2       /* C = tanh(...)                              */
#define ARCVAE_GEMM_SPLITK 4     /* allow split-K with f32 atomics             */
#define ARCVAE_GEMM_NO_SKINNY 8
#define ARCVAE_GEMM_TILE64 16   /* force 64x64 tiles (tuning / tests) */
#define ARCVAE_GEMM_TILE128 32  /* force 128x128 tiles */
#define ARCVAE_GEMM_DTANH 64    /* C = (A.B) * (1 - T^2), T = `bias` read as an [M,ldc] matrix (tanh backward) */
#define ARCVAE_GEMM_TILE_WIDE 128 /* split-bf16 TN path: 128-row tile (no persistent sweep resident beside it) */
#define ARCVAE_GEMM_BF16 256      /* throughput mode: operands rounded to bf16 (RNE), f32 accumulate on v_mfma_f32_32x32x16_bf16;
                                   * NOT a parity path (SURVEY.md section 8(d) Config 2 "bf16-in/fp32-acc"); products with M <= 256
                                   * rows on the dependent chain keep the f32 skinny kernel */
#define ARCVAE_GEMM_SPLIT3 512    /* any layout: every f32 operand as three bf16 pieces (8+8+8 bits), the six products of weight
                                   * >= 2^-16 on v_mfma_f32_32x32x16_bf16, f32 accumulate -- the accuracy class of the exact-f32
                                   * kernels (a PARITY path) at 2.7x less matrix-pipe time; for GEMMs beside a persistent sweep */
#define ARCVAE_LSTM_RETILE 1      /* arcvae_enc_lstm_backward flags bit 0: write the BPTT weight layouts first */
#define ARCVAE_LSTM_BF16 2        /* arcvae_enc_lstm_forward / _backward flags bit 1: throughput mode -- bf16 operand copies and
                                   * v_mfma_f32_16x16x32_bf16 products where the shape runs on the register-tiled step kernels
                                   * (the MFMA-bound regime, e.g. H512 L4 bs 512); gates, cell state and accumulators stay f32.
                                   * The k-chunk-major workspaces (hseq_t, dG_t, wt, wT) then hold bf16 in 32-wide chunks --
                                   * same pointers, half the bytes; forward and backward of a step must be given the same flag */
#define ARCVAE_LSTM_SPLIT3 4      /* arcvae_enc_lstm_forward / _backward flags bit 2 (what the engine passes on the PARITY path): where the
                                   * shape runs on the register-tiled step kernels every operand value is three bf16 pieces (hi + mid
                                   * + lo, 8 + 8 + 8 bits), six products on v_mfma_f32_16x16x32_bf16, f32 accumulate -- the accuracy
                                   * class of the exact-f32 kernels.  The operand workspaces (hseq_t, dG_t, wt, wT) then hold THREE
                                   * bf16 planes per value: 6 bytes instead of 4, i.e. 3/2 of the f32 sizes, and where the weight
                                   * gradients read those planes the rings keep all T time slots: size them with
                                   * arcvae_enc_lstm_ws_floats and pass the capacities to every call (ws_floats) */
#define ARCVAE_PERSIST_BF16 2     /* arcvae_enc_lstm_forward_persistent flags bit 1 / arcvae_enc_lstm_backward_persistent_rs flags
                                   * bit 1: throughput mode for the persistent sweeps -- the 4x4 MFMA blocks (H 256, B <= 64
                                   * forward, B <= 128 BPTT) on v_mfma_f32_4x4x4_16b_bf16, weights and h / dG rounded to bf16
                                   * on their way into the instruction */
#define ARCVAE_RS_HALF 16         /* arcvae_enc_lstm_backward_persistent_rs flags bit 4: the half-batch form (129 <= B <= 256 where
                                   * arcvae_enc_lstm_bwd_rs_halves says so): this call sweeps rows 32 x + 16 half + [0, 16) of XCD x
                                   * only -- half = 0, or 1 with ARCVAE_RS_HALF1 -- and a chunk of the BPTT is TWO calls, one per
                                   * half, on the same stream with the same sync_ws / part_ws (a 16-row tick costs 4.0 us, the 32-row
                                   * block's 9.7) */
#define ARCVAE_RS_HALF1 32        /* ... flags bit 5: the second half */
#define ARCVAE_DEC_SPLIT3 512     /* arcvae_dec_forward_dense `mode` bit 9 / arcvae_dec_backward_dense `flags` bit 9: the B*V-row
                                   * products with ARCVAE_GEMM_SPLIT3 (fp32-class accuracy, less matrix-pipe time beside a sweep) */
#define ARCVAE_DEC_NO_GPRE 1024   /* arcvae_dec_forward_dense `mode` bit 10: forward only (sampler, loss-only forward) -- `gpre` is not
                                   * written: a layer's GEMM and zero-state cell run as one kernel where the shape allows */
#define ARCVAE_DEC_PART_HEAD 2048 /* arcvae_dec_forward_dense `mode` / arcvae_dec_backward_dense `flags` bit 11: only the token table and
                                     layer 0 (forward: tableD, hact[0]; backward: from dh_0 in dh[0 .. B*V*H)) */
#define ARCVAE_DEC_PART_TAIL 4096 /* bit 12: only fc_out (forward: logits, lse, nxt from hact[L-1]; backward: dWout, dbout and
                                     dh_top = dlogits . Wout into dh[0 .. B*V*H)); the caller runs layers 1 .. L-1 in between:
                                     arcvae_dense_stack_forward / _backward */
#define ARCVAE_DEC_BF16 256       /* arcvae_dec_forward_dense `mode` bit 8 / arcvae_dec_backward_dense `flags` bit 8: the B*V-row
                                   * products with ARCVAE_GEMM_BF16 */

/* A hipStream_t.  Callers without the HIP headers pass it as void*; the library's own translation units include this header
 * too (csrc/common.h defines ARCVAE_HIP_BUILD behind <hip/hip_runtime.h>), so a declaration here that drifts from its
 * definition in csrc/ is a compile error ("conflicting declaration of C function"), not a silent ABI mismatch. */
#ifdef ARCVAE_HIP_BUILD
typedef hipStream_t arcvae_stream_t;
#else
typedef void* arcvae_stream_t;
#endif

/* ABI version / build probe: returns 1000*major + minor; *arch_gfx950 = 1 when the code object
 * was built for gfx950. */
int arcvae_abi_version(int* arch_gfx950);

/* ---- generic contraction (MLX addmm / matmul, M2: y = x W^T + b) -------------------------
 * C[M,N] (+)= op(A) op(B) (+ bias[N]); transA=0: A [M,K], transA=1: A stored [K,M];
 * transB=0: B [K,N], transB=1: B stored [N,K].  Replaces every nn.Linear / matmul on the path,
 * e.g. models/encoder.py:109,115,117,118 and models/decoder.py:175. */
int arcvae_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, const float* bias, int flags,
                    arcvae_stream_t stream);

/* ---- encoder LSTM stack -------------------------------------------------------------------
 * models/encoder.py:93-101: embedding lookup + L stacked nn.LSTM over the full padded sequence
 * (MLX nn.LSTM semantics M1: gates i,f,g,o; t=0 has no recurrent term and c_0 = i*g).
 * x_tb [T,B] int32 tokens (time-major; arcvae_transpose_tokens makes it from [B,T]);
 * table0 [V,4H] = embedding . Wx_0^T + bias_0 (one arcvae_gemm_f32 call);
 * outputs hseq,cseq [L,T,B,H], gseq [L,T,B,4H] (post-activation gates, saved for BPTT);
 * workspaces: hseq_t [L, slots, B*H*q] -- the k-chunk-major operand copy of h_t in ring slot t % slots -- and wt [(2L-1), 4H*H*q],
 * the k-chunk-major weight copies, where q = 3/2 in the three-piece form (ARCVAE_LSTM_SPLIT3 where the shape runs on the tile
 * kernels: three bf16 planes per value) and 1 otherwise, and slots = arcvae_enc_lstm_operand_slots(B, T, H, L, flags): min(T, 16),
 * or T where the weight gradients read the operand planes.  The library decides q and slots from (B, H, L, flags) and the
 * ARCVAE_* kernel-family knobs at EVERY call, so the caller states what it allocated: ws_floats[4] = capacities in floats of
 * {hseq_t, dG_t, wt, wT} (HOST array; entries of buffers a call does not use are ignored).  A call whose kernel family needs more
 * than that returns ARCVAE_ERR_ARG before launching anything -- never a write past the buffer.  arcvae_enc_lstm_ws_floats
 * returns the sizes needed under the current knobs. */
int arcvae_enc_lstm_ws_floats(int B, int T, int H, int L, int flags, long* floats /* out [4]: hseq_t, dG_t, wt, wT */);
int arcvae_transpose_tokens(const int32_t* src_bt, int32_t* dst_tb, int B, int T, arcvae_stream_t stream);
int arcvae_enc_lstm_forward(const int32_t* x_tb, const float* table0, const float* const* Wx,
                            const float* const* Wh, const float* const* bias, float* hseq, float* hseq_t,
                            float* cseq, float* gseq, float* wt, float* wT_bwd /* optional */, int B, int T, int V,
                            int H, int L, int flags /* ARCVAE_LSTM_SPLIT3, ARCVAE_LSTM_BF16 or 0; the same value must go to the
                            backward */, const long* ws_floats /* [4] capacities: hseq_t, -, wt, wT_bwd */,
                            void* h_oct /* optional, throughput mode: [L,T*B/8,H,8] bf16 "octet-major" copy of hseq
                            (k = t*B + b in groups of 8 per row: the layout an MFMA fragment is loaded in) for
                            arcvae_enc_lstm_wgrad; written when the tiled bf16 kernels run and B % 16 == 0 */,
                            unsigned long long* trace /* optional diagnostic: {start,end} 100 MHz stamps
                            of launch s at trace[2s..2s+1], 2*(T+L-1) u64 */, arcvae_stream_t stream);
/* bit 0 / bit 1: the forward / backward sweep of this shape runs on the register-tiled step kernels, i.e. ARCVAE_LSTM_BF16
 * takes effect there and h_oct / dG_oct get written (pass them to arcvae_enc_lstm_wgrad only when both bits are set). */
int arcvae_enc_lstm_tiled(int B, int H, int L);
/* The same for sweeps called with `flags`: with ARCVAE_LSTM_SPLIT3 (the three-piece form, what the engine passes on the parity path)
 * the tile regime starts at smaller grids -- from ~768 rows per GPU at H 256 / L 2 instead of ~1100. */
int arcvae_enc_lstm_tiled_for(int B, int H, int L, int flags);
/* Time slots of the operand rings hseq_t [L,slots,..] / dG_t [L,slots,..] the sweeps use for this shape with these flags: T where
 * the weight gradients read the operand planes (both sweeps on the three-piece tile kernels, B % 32 == 0: pass the rings to
 * arcvae_enc_lstm_wgrad with parts bit 11), else min(T, 16).  The dc / dX rings always have min(T, 16) slots. */
int arcvae_enc_lstm_operand_slots(int B, int T, int H, int L, int flags);
/* Backward of the above (the part of mx.value_and_grad, trainer.py:292, that walks the encoder
 * LSTM graph).  dh_top [B, ld_dh_top]: gradient w.r.t. the top layer's h at t = T-1, the only
 * position read by models/encoder.py:106.  dG out [L,T,B,4H] (may alias gseq: in-place); dG_t ws [L,slots,B*4H*q] and wT ws
 * [(2L-1),H*4H*q] with slots and q as for the forward (capacities in ws_floats[1] and [3]); dcs, dxs ws [L,RS,B,H], rings over t
 * with RS = min(T,16) slots (ARCVAE_RING) whatever the form: a slab is consumed by the next launch only. */
int arcvae_enc_lstm_backward(const float* const* Wx, const float* const* Wh, const float* cseq,
                             const float* gseq, const float* dh_top, int ld_dh_top, float* dG, float* dG_t,
                             float* dcs, float* dxs, float* wT, int B, int T, int H, int L, int s_begin,
                             int s_end, int flags /* ARCVAE_LSTM_RETILE | ARCVAE_LSTM_SPLIT3 or ARCVAE_LSTM_BF16 */,
                             const long* ws_floats /* [4] capacities: -, dG_t, -, wT */,
                             void* dG_oct /* optional, throughput mode: [L,T*B/8,4H,8] bf16 octet-major copy of dG */,
                             unsigned* start_signal /* optional: += 1 when the first launch of
                             this call starts (all earlier work of the stream is complete) */,
                             unsigned long long* trace /* optional diagnostic: stamps of launch s at trace[2s..2s+1],
                             2*(T+2(L-1)) u64 */, arcvae_stream_t stream);
/* The same forward sweep as ONE persistent launch for the latency regime (H = 128, 256 or 384, L <= 4, B <= 256, weight
 * slices within LDS; arcvae_enc_lstm_persistent_ok says whether a shape qualifies): batch rows partitioned over the 8
 * XCDs, weights stationary in LDS, one flag-line barrier per XCD and tick.  No k-chunk-major h
 * copy is written.  sync_ws: 8192 u32 of scratch (forward: words [0, 272); the BPTT entry points: [512, 848); [500] = the
 * error word of both); sync_ws[500] != 0 afterwards = a block gave up waiting (sticky).
 * start_signal (optional): += 1 when the sweep starts.  The BPTT counterpart covers ticks [s_begin, s_end) of
 * arcvae_enc_lstm_backward's schedule per launch (L <= 2, ceil(B/8) * H/32 <= 64); a sweep uses it for all its
 * chunks or for none (no k-chunk-major dG copy is written).  chunk_index: 0 for the launch with s_begin == 0 (it
 * re-arms sync_ws), then 1, 2, .. (< 8) for the following chunk launches of the same sweep: every chunk launch draws
 * its block roles from its own counters.  trace (optional diagnostic): per-tick {start,end} stamps of one block at
 * trace[2s..2s+1] (forward: s < T+L-1; BPTT: the sweep's global tick index s < T+2(L-1)). */
int arcvae_enc_lstm_persistent_ok(int B, int T, int H, int L);
/* (the sweep reads the row-major Wx / Wh themselves: no k-chunk-major copy.  wT_bwd, optional: also write the BPTT
 * layouts for a launch-based arcvae_enc_lstm_backward(retile = 0) of the same step.  flags bit 0: sync_ws has been
 * re-armed by arcvae_enc_prologue; bit 1: ARCVAE_PERSIST_BF16.) */
int arcvae_enc_lstm_forward_persistent(const int32_t* x_tb, const float* table0, const float* const* Wx,
                                       const float* const* Wh, const float* const* bias, float* hseq, float* cseq,
                                       float* gseq, float* wT_bwd, long wT_bwd_floats /* capacity of wT_bwd in floats (entry [3] of
                                       arcvae_enc_lstm_ws_floats); ignored when wT_bwd is NULL */, float* comb /* optional [B,2H]: its first H columns
                                       receive h_{T-1} of the top layer, the heads' input (models/encoder.py:106) */,
                                       unsigned* sync_ws, unsigned* start_signal, int B, int T, int V, int H, int L,
                                       int flags, unsigned long long* trace, arcvae_stream_t stream);
/* The byte-moving launches in front of the persistent forward sweep as one: x_tb = x_bt^T (arcvae_transpose_tokens),
 * zero_f32[0..n_zero) = 0 (optional: the encoder's gradient buffer), sync_ws[0..n_sync) = 0 (optional: the sweep's
 * re-arm). */
/* cond .. stats (optional, all or none): also comb[:, H:2H] = condition_fc(cond) (models/encoder.py:109-112) and
 * stats[0..n_stats) = 0, for arcvae_enc_heads_forward(comb_ready = 1).  onehot_ws (optional, [T*B, roundup(V,4)]): the
 * one-hot token rows of arcvae_enc_lstm_wgrad (called with parts bit 5 then). */
int arcvae_enc_prologue(const int32_t* x_bt, int32_t* x_tb, float* zero_f32, long n_zero, float* zero2_f32 /* optional second
                        zero fill: the token-table workspaces of arcvae_enc_lstm_wgrad (parts bit 8 then) */, long n_zero2,
                        unsigned* sync_ws, int n_sync, int sync_keep /* word index left alone, or -1: the sweeps' sticky error
                        word 500 when n_sync covers it -- 848 re-arms the forward AND the BPTT sweep of the step */,
                        const float* cond, const float* Wc, const float* bc, float* comb, float* stats, int n_stats,
                        float* onehot_ws, int V, int B, int T, int H, int C, arcvae_stream_t stream);
int arcvae_enc_lstm_bwd_persistent_ok(int B, int T, int H, int L);
int arcvae_enc_lstm_backward_persistent(const float* cseq, const float* gseq, const float* dh_top, int ld_dh_top,
                                        float* dG, float* dcs, float* dxs, const float* wT, unsigned* sync_ws,
                                        unsigned* start_signal, int B, int T, int H, int L, int s_begin, int s_end,
                                        int chunk_index, unsigned long long* trace, arcvae_stream_t stream);
/* Reduce-scatter form of the persistent BPTT sweep (H = 256, L <= 2, B <= 256): a CU keeps the gate gradients of its own
 * 32 gate columns on chip, multiplies them with its 32 rows of the row-major Wh / Wx, and the partial sums are
 * reduce-scattered through the XCD's L2 (part_ws: RG*2*(2L-1)*8*32*32*64 floats, RG = 1 / 2 / 4 groups of 8 rows per
 * XCD for B <= 64 / 128 / 256).  Otherwise as arcvae_enc_lstm_backward_persistent. */
int arcvae_enc_lstm_bwd_rs_ok(int B, int T, int H, int L);
/* Floats of part_ws that sweep uses for this shape under the current knobs (0 where it does not run): pass the capacity the
 * caller allocated as part_ws_floats -- a call that would need more returns ARCVAE_ERR_ARG. */
long arcvae_enc_lstm_bwd_rs_part_floats(int B, int T, int H, int L);
/* 1 where that sweep runs as two half-batch sweeps per chunk (ARCVAE_RS_HALVES=1, 129 <= B <= 256): see ARCVAE_RS_HALF. */
int arcvae_enc_lstm_bwd_rs_halves(int B, int T, int H, int L);
/* 2 where the persistent sweeps run in their TWO-GROUP form (H = 256, L <= 2, 129 <= B <= 256: the 256-row shard of
 * BASELINE.json configs[3]), else 1.  Two groups: 512 blocks, two per CU; every XCD's 17..32 rows are two independent
 * recurrences of up to 16 rows with their own flag lines, and a CU's two blocks serve different groups, so that one group's
 * exchange latency sits under the other's matrix work.  Such a step uses sync_ws words [1024, 4352)
 * as well; a step re-arms 4864 words (arcvae_enc_prologue n_sync: the one-group words [0, 848), the two-group words and the
 * reduce-scatter sweep's "gathered" words [4352, 4864) of its single-buffered exchange); part_ws as for RG = 4. */
int arcvae_enc_lstm_persist_groups(int B, int H, int L);
int arcvae_enc_lstm_backward_persistent_rs(const float* const* Wx, const float* const* Wh, const float* cseq,
                                           const float* gseq, const float* dh_top, int ld_dh_top, float* dG, float* dcs,
                                           float* dxs, float* part_ws, long part_ws_floats, unsigned* sync_ws,
                                           unsigned* start_signal, int B, int T, int H, int L, int s_begin, int s_end, int chunk_index,
                                           int flags /* bit 0: sync_ws was re-armed ahead of the step (arcvae_enc_prologue with
                                           n_sync >= 848); bit 1: ARCVAE_PERSIST_BF16; bits 4, 5: ARCVAE_RS_HALF, ARCVAE_RS_HALF1 */,
                                           unsigned long long* trace,
                                           arcvae_stream_t stream);
/* The reduce-scatter sweep with the stack's weight gradients formed INSIDE the kernel: replaces
 * arcvae_enc_lstm_backward_persistent_rs + the per-layer GEMMs / bias sums / token segment-sum of arcvae_enc_lstm_wgrad
 * for the same ticks.  "+=" into dWh[l], dWx[l] (l >= 1), dbias[l] (l >= 1); the layer-0 input side arrives as
 * dtable_ws [V,4H] (zeroed by the chunk_index == 0 call) for arcvae_table_finalize.  hseq [L,T,B,H] and x_tb [T,B] as
 * written / read by the forward sweep; dWx, dWh, dbias: HOST arrays of device pointers.  Same shape rule as _rs. */
int arcvae_enc_lstm_backward_fused(const float* const* Wx, const float* const* Wh, const float* cseq, const float* gseq,
                                   const float* hseq, const int32_t* x_tb, const float* dh_top, int ld_dh_top,
                                   float* dG, float* dcs, float* dxs, float* part_ws, long part_ws_floats, unsigned* sync_ws,
                                   unsigned* start_signal, float* const* dWx, float* const* dWh, float* const* dbias,
                                   float* dtable_ws, int B, int T, int V, int H, int L, int s_begin, int s_end,
                                   int chunk_index, unsigned long long* trace, arcvae_stream_t stream);
/* (the sweep is T+2(L-1) dependent launches; [s_begin, s_end) selects a sub-range so the caller can interleave
 *  events: after launches [0, s_end) every layer has finished all t >= T - s_end + 2(L-1).)
 * Parameter gradients of the stack from dG over time range [t_lo, t_hi): embedding.weight,
 * lstm_layer_l.{Wx,Wh,bias}; `first` zeroes the token-table workspace, `last` folds it into the
 * embedding / layer-0 gradients.  dtable_ws [V,4H]; onehot_ws [T*B, roundup(V,4)] (one-hot token rows, written
 * when `first`: the token segment-sum runs as OneHot^T . dG_0 on the matrix cores).  `parts` selects disjoint pieces
 * that may run on different streams: bit 0 = per-layer GEMMs and bias sums (= bits 2 | 3), bit 1 = token-table path,
 * bit 2 = dWx_l (l >= 1) and bias sums only, bit 3 = dWh_l only; bit 4 = exact-f32 tile GEMMs instead of the split-bf16
 * kernel; bit 5 = onehot_ws was written by arcvae_enc_prologue; bit 6 = the split-bf16 kernel's 128-row tile (the range
 * runs behind the sweep, no sweep block is resident); bit 8 = dtable_ws was zeroed ahead of the call (arcvae_enc_prologue:
 * the zero-fill launch in front of a `first` range is skipped); bit 10 = the per-layer GEMMs as three-piece tile GEMMs
 * (ARCVAE_GEMM_SPLIT3: fp32-class accuracy, for the MFMA-bound regime beside the tiled sweeps); bit 7 = throughput mode (one bf16 product per GEMM step instead of
 * the six of the split form: not a parity path); bit 11 = h_oct / dG_oct are the sweeps' three-plane operand rings hseq_t /
 * dG_t with ALL T time slots (arcvae_enc_lstm_operand_slots == T): the per-layer GEMMs read the planes directly (transposing
 * LDS reads, no f32 loads, no re-splitting; needs B % 32 == 0).  The token-table path is linear in dtable_ws, so a
 * time range may be given its own workspace and both `first` and `last` (zero, accumulate, fold) on any stream. */
int arcvae_enc_lstm_wgrad(const int32_t* x_tb, const float* emb, const float* Wx0, const float* hseq,
                          const float* dG, float* dtable_ws, float* onehot_ws, float* dEmb, float* const* dWx,
                          float* const* dWh, float* const* dbias, int B, int T, int V, int E, int H, int L,
                          int t_lo, int t_hi, int first, int last, int parts /* 1 layers | 2 token table */,
                          const void* h_oct, const void* dG_oct /* optional (both or none), with parts bit 7: the octet-major
                          bf16 copies the sweeps of this step wrote -- the per-layer GEMMs then read those */,
                          const long* ws_floats /* with parts bit 11 (h_oct / dG_oct are plane rings with all T slots: [L,T,B*H*3/2]
                          and [L,T,B*4H*3/2] floats): [0], [1] = their capacities in floats, checked -- ARCVAE_ERR_ARG when smaller;
                          may be NULL otherwise */, arcvae_stream_t stream);

/* ---- encoder heads + reparameterisation + latent loss ----------------------------------------
 * models/encoder.py:106-130 (condition_fc, fc_mu, fc_logvar_hidden, fc_logvar, tanh bounds),
 * models/encoder.py:147-153 (z = mu + eps*exp(logvar/2), eps injected), and the per-rank halves
 * of losses/kl.py:39-56 and losses/info.py:27-41.  stats [2Z+4] is zeroed and filled here.
 * comb_ready != 0: comb [B,2H] and the zeroed stats were written ahead of the call (arcvae_enc_prologue + the persistent
 * forward sweep): hT, cond, Wc, bc are not read. */
int arcvae_enc_heads_forward(const float* hT, const float* cond, const float* Wc, const float* bc,
                             const float* Wmu, const float* bmu, const float* Wlh, const float* blh,
                             const float* Wlv, const float* blv, const float* eps, float* comb, float* lh,
                             float* mu_raw, float* lv_raw, float* mu, float* logvar, float* z, float* stats,
                             int B, int H, int Z, int C, float free_bits, int comb_ready, arcvae_stream_t stream);
/* The whole SEAM between the two sweeps as ONE per-XCD launch (OPT-IN, ARCVAE_SEAM_FUSED=1: parity-green, measured slower than the
 * five launches it replaces -- csrc/latent.hip; latency regime: B <= 64, hidden_dim a multiple of 64 up to 256, latent_dim a
 * multiple of 32 up to 128; arcvae_enc_seam_ok says whether the engine is to use it): arcvae_enc_heads_forward (comb_ready form) + arcvae_latent_loss (with
 * gradients) + arcvae_enc_heads_backward (phase 1) -- models/encoder.py:106-153, complete_vae_loss.py:45-99 and the heads' backward
 * up to dcomb -- with the batch rows partitioned over the 8 XCDs as in the persistent sweeps, every weight slice requested into LDS
 * when the block starts, activations exchanged through the XCD's L2, and the batch statistics (Q11) as atomics plus one device-wide
 * arrival counter.  phases: 1 = forward part (leaves this process's partial `stats`), 2 = backward part from GLOBAL stats (a
 * data-parallel step all-reduces `stats` between the two launches), 3 = both in one launch.  comb [B,2H] complete; stats [2Z+4]
 * zeroed ahead of the forward part; sync_ws: the sweeps' scratch -- words [4864, 5696) are the seam's (flags bit 0: zeroed ahead of
 * the step by arcvae_enc_prologue with n_sync >= 5696); sync_ws[500] != 0 afterwards = a block gave up waiting (sticky). */
int arcvae_enc_seam_ok(int B, int H, int Z);
int arcvae_enc_seam(const float* comb, const float* Wmu, const float* bmu, const float* Wlh, const float* blh, const float* Wlv,
                    const float* blv, const float* eps, const float* hyper, float* lh, float* mu_raw, float* lv_raw, float* mu,
                    float* logvar, float* z, float* stats, float* scalars, float* dmu_raw, float* dlv_raw, float* dlh, float* dcomb,
                    unsigned* sync_ws, int B, int H, int Z, int T, float free_bits, int phases, int flags, arcvae_stream_t stream);
int arcvae_stats_set_recon(const float* rowloss, int B, float* stats, int Z, arcvae_stream_t stream);
/* complete_vae_loss.py:45-99 (+ losses/kl.py, losses/info.py reductions) from GLOBAL stats.
 * hyper [8] device: beta, lambda_collapse, lambda_mi, target_mi, free_bits; scalars [16] device out:
 * total, recon, kl, beta*kl, collapse, prop(0), lambda_prop*prop(0), mutual_info, mi_penalty.
 * dmu_raw/dlv_raw [B,Z] may be NULL (forward only). */
int arcvae_latent_loss(const float* stats, const float* hyper, const float* mu, const float* logvar,
                       float* scalars, float* dmu_raw, float* dlv_raw, int B, int Z, int T, float free_bits,
                       arcvae_stream_t stream);
/* recon = stats[2Z+3]/(B_global*T) and total, once the decoder's CE row sums are in stats (latent_loss may run
 * before that: the encoder's backward does not depend on the reconstruction term, Q2). */
/* guard_a / guard_b (optional, here and in arcvae_recon_finalize): device error words as in arcvae_adam_update; when
 * either is non-zero the nine loss scalars are set to NaN and scalars[15] = 1 (else scalars[15] = 0), so the host's
 * per-batch read of the loss (trainer.py:366) also carries the "stream order was lost" status. */
int arcvae_loss_finalize(const float* stats, float* scalars, int Z, int T, const unsigned* guard_a,
                         const unsigned* guard_b, arcvae_stream_t stream);
/* arcvae_stats_set_recon + arcvae_loss_finalize in one launch (single-process step: losses/recon.py:59-60 mean over
 * B*T and complete_vae_loss.py:76-82 total). */
int arcvae_recon_finalize(const float* rowloss, int B, float* stats, float* scalars, int Z, int T,
                          const unsigned* guard_a, const unsigned* guard_b, arcvae_stream_t stream);
int arcvae_enc_heads_backward(const float* cond, const float* Wmu, const float* Wlh, const float* Wlv,
                              const float* comb, const float* lh, const float* dmu_raw, const float* dlv_raw,
                              float* dlh, float* dcomb, float* dWc, float* dbc, float* dWmu, float* dbmu,
                              float* dWlh, float* dblh, float* dWlv, float* dblv, int B, int H, int Z, int C,
                              int phase /* 0 both, 1 dcomb chain, 2 parameter gradients */, arcvae_stream_t stream);

/* ---- stand-alone loss pieces (the reference's `losses` module called on arbitrary tensors) ----------
 * models/encoder.py:147-153; losses/kl.py:39-58 + losses/info.py:27-35 partial sums (krow [B] may
 * be NULL: per-sample free-bits KL); losses/recon.py:29-57 per-position CE; out[0] = scale*sum(x). */
int arcvae_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, long n,
                          arcvae_stream_t stream);
int arcvae_latent_stats(const float* mu, const float* logvar, float* stats, float* krow, int B, int Z,
                        float free_bits, arcvae_stream_t stream);
int arcvae_ce_rows(const float* logits, const int32_t* targets, float* ce, long R, int V, arcvae_stream_t stream);
int arcvae_sum(const float* x, long n, float* out, float scale, arcvae_stream_t stream);

/* ---- decoder, vocabulary-dense ------------------------------------------------------------------
 * models/decoder.py:152-175: per step embed(token) ++ conditions -> L zero-state LSTM cells
 * (hidden=None, cell=None every call, Q1) -> fc_out.  logits_t depends only on (token_t, cond_b),
 * so all B*V (row, token) pairs are evaluated at once: logits [B*V,V], lse [B*V], nxt [B*V]
 * (mode 0: first argmax(logits), decoder.py:185; mode 1: first argmax(softmax(logits/temperature)),
 * decoder_sampling.py:110-117; | ARCVAE_DEC_BF16: throughput mode). */
int arcvae_dec_forward_dense(const float* emb, const float* const* Wx, const float* const* bias,
                             const float* Wout, const float* bout, const float* cond, float* tableD,
                             float* hact, float* gpre, float* logits, float* lse, int32_t* nxt, int B, int V,
                             int E, int C, int H, int L, int mode, float temperature, arcvae_stream_t stream);
/* models/decoder.py:146-185 teacher-forcing walk (coins[t] = np.random.rand() < ratio, Q5) fused
 * with losses/recon.py:29-60 row sums: fed [B,T], rowloss [B] = sum_t CE. */
int arcvae_dec_chain_ce(const int32_t* x, const uint8_t* coins, const int32_t* nxt, const float* lse,
                        const float* logits, int32_t* fed, float* rowloss, int B, int T, int V,
                        arcvae_stream_t stream);
/* d(recon)/d(dense logits); inv_count = 1/(B_global*T) (mean over ALL positions, Q3). */
int arcvae_dec_ce_backward(const int32_t* x, const int32_t* fed, const float* logits, const float* lse,
                           float* dlogits, int B, int T, int V, float inv_count, arcvae_stream_t stream);
/* out[b,t,:] = dense[b, fed[b,t], :]: the [B,T,V] logits of models/decoder.py:188. */
int arcvae_dec_gather_logits(const float* dense, const int32_t* fed, float* out, int B, int T, int V,
                             arcvae_stream_t stream);
/* models/decoder_sampling.py:85-123 greedy walk: tokens [B,max_len], first_end [B]. */
int arcvae_dec_sample_chain(const int32_t* nxt, int32_t* tokens, int32_t* first_end, int B, int V,
                            int max_len, int end_token, arcvae_stream_t stream);
/* EXTENSION (not in the reference, whose sampler is greedy: "for now, use argmax", models/decoder_sampling.py:115-116): true
 * categorical sampling, tokens[b,t] ~ Categorical(softmax(dense_logits[b, cur, :] / temperature)) walked from the start token,
 * uniform numbers from a counter-based generator keyed by (seed, row, step): the same seed and batch give the same molecules.  dense_logits [B*V, V] as written by arcvae_dec_forward_dense (mode 0); vocab_size <= 256; first_end as above. */
int arcvae_dec_sample_chain_categorical(const float* dense_logits, int32_t* tokens, int32_t* first_end, int B, int V, int max_len,
                                        int end_token, float temperature, unsigned long long seed, arcvae_stream_t stream);
/* Backward of arcvae_dec_forward_dense: embedding.weight, lstm_layer_l.{Wx,bias}, fc_out.{weight,bias}.
 * ws: dh [2,B*V,H], dG [B*V,4H], dtableD [V,4H], wcpart [V,4H,max(C,1)]. */
int arcvae_dec_backward_dense(const float* emb, const float* const* Wx, const float* const* bias,
                              const float* Wout, const float* cond, const float* tableD, const float* hact,
                              const float* gpre, const float* dlogits, float* dh, float* dG, float* dtableD,
                              float* wcpart, float* dEmb, float* const* dWx, float* const* dbias, float* dWout,
                              float* dbout, int B, int V, int E, int C, int H, int L, int flags /* ARCVAE_DEC_BF16 or 0 */,
                              arcvae_stream_t stream);

/* Layers 1 .. L-1 of the vocabulary-dense decoder -- zero-state cells over R = B*V rows (Q1: hidden = cell = None at every call), an
 * (L-1)-layer stack at T = 1 -- on the three-piece tile kernels of the encoder's sweeps (lstm_fwd_tile_kernel,
 * lstm_bwd_tile_ks3_kernel, wgrad_planes_kernel: three bf16 pieces per operand, six products, fp32-class accuracy), for the
 * MFMA-bound regime where the decoder's exact-f32 GEMMs would run beside the forward sweep.
 *   arcvae_dense_stack_ok: 1 where that path applies (R % 32 == 0, H % 64 == 0, L >= 2, grids that fill the chip);
 *   arcvae_dense_stack_ws_floats: *floats = size of `ws` in floats (operand planes of every layer's h, cell states, gate-gradient planes,
 *   weight planes);  forward: hact [L,R,H] with layer 0 given (arcvae_dec_forward_dense | ARCVAE_DEC_PART_HEAD), layers 1 ..
 *   written; gates [L-1,R,4H] POST-activation (the gpre workspace holds them in this mode);  backward: dh_top [R,H] in, dG [R,4H]
 *   scratch, dh0 [R,H] out (the input of arcvae_dec_backward_dense | ARCVAE_DEC_PART_HEAD), dWx[l] / dbias[l] += for l >= 1
 *   (HOST arrays [L], entry 0 unused), `ws` as the forward left it. */
int arcvae_dense_stack_ok(long R, int H, int L);
int arcvae_dense_stack_ws_floats(long R, int H, int L, long* floats /* out */);
int arcvae_dense_stack_forward(const float* const* Wx, const float* const* bias, float* hact, float* gates, float* ws,
                               long ws_floats /* capacity of ws in floats: >= arcvae_dense_stack_ws_floats, else ARCVAE_ERR_ARG */, long R,
                               int H, int L, int flags /* bit 0: forward only (sampler): nothing the backward needs is written;
                               ARCVAE_LSTM_BF16 (bit 1): throughput mode -- bf16 operand copies, one product (not a parity path) */,
                               arcvae_stream_t stream);
int arcvae_dense_stack_backward(const float* gates, const float* dh_top, float* dG, float* dh0, float* const* dWx,
                                float* const* dbias, float* ws, long ws_floats, long R, int H, int L, int flags /* ARCVAE_LSTM_BF16 or 0: as the
                                forward was called */, arcvae_stream_t stream);

/* ---- optimizer ------------------------------------------------------------------------------------
 * trainer.py:75-76,320,324: MLX optim.Adam, NO bias correction (Q7):
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p = p - lr m / (sqrt(v) + eps), over a flat buffer.
 * guard_a / guard_b (optional): device error words (engine: the gates' ERR word and the persistent sweeps'
 * sync_ws[500]); when either is non-zero at execution time the update is skipped -- gradients formed after a lost
 * stream order must not reach the weights. */
int arcvae_adam_update(float* params, const float* grads, float* m, float* v, long n, double lr,
                       double beta1, double beta2, double eps, const unsigned* guard_a, const unsigned* guard_b,
                       arcvae_stream_t stream);

/* arcvae_recon_finalize + arcvae_adam_update as ONE launch (the exposed tail of the single-process step, trainer.py:320-366):
 * block 0 sums the CE rows into stats[2Z+3] and writes the recon / total scalars (NaN and scalars[15] = 1 when a guard word is set),
 * then the update as above (skipped on a guard). */
int arcvae_adam_update_finalize(float* params, const float* grads, float* m, float* v, long n, double lr, double beta1,
                                double beta2, double eps, const unsigned* guard_a, const unsigned* guard_b,
                                const float* rowloss, int B, float* stats, float* scalars, int Z, int T, arcvae_stream_t stream);

/* ---- small helpers ---------------------------------------------------------------------------------- */
int arcvae_colsum_accum(const float* X, int rows, int cols, int ld, float* out, float scale,
                        arcvae_stream_t stream);
int arcvae_segsum_rows_accum(const float* X, const int32_t* seg, int rows, int nseg, int cols, float* out,
                             arcvae_stream_t stream);
int arcvae_transpose_batched(const float* const* src, float* const* dst, const int* rows, const int* cols,
                             int n, arcvae_stream_t stream);
/* Diagnostic / test entry point: `blocks` workgroups of `threads` threads (a multiple of 64) that hold `lds_bytes` of LDS each
 * and wait `spin_us` microseconds -- a stand-in for a communication kernel that occupies CU resources on another stream
 * beside the persistent sweeps (tests/test_occupancy_gpu.py).  Computes nothing. */
int arcvae_debug_occupy(int blocks, int threads, int lds_bytes, int spin_us, arcvae_stream_t stream);
int arcvae_scale_inplace(float* x, long n, float s, arcvae_stream_t stream);
int arcvae_zero(float* x, int rows, int cols, int ld, arcvae_stream_t stream);
/* Token-table gradient dT [V,4H] of an LSTM layer-0 input projection folded back in one launch (backward of
 * nn.Embedding + the x.Wx^T term of nn.LSTM, models/encoder.py:93,98 and models/decoder.py:154-166):
 * dEmb [V,E] += dT . Wx0[:, :E];  dWx0[:, :E] += dT^T . emb (row stride ldw);  db0 [4H] += colsum(dT). */
int arcvae_table_finalize(const float* dT, const float* Wx0, int ldw, const float* emb, float* dEmb, float* dWx0,
                          float* db0, int V, int E, int G, arcvae_stream_t stream);
/* Device-side gates (no reference counterpart): cross-stream ordering by a one-wave polling kernel instead of an
 * event wait, because a hardware queue blocked on an event slows every dependent dispatch of the chain that is
 * running.  wait: returns once (int)(*flag - ((*steps) * stride + offset)) >= 0 (steps may be
 * NULL: target = offset); advance != 0: ++*steps afterwards; bounded spin of max_polls polls (~1 us each), on expiry
 * *err += 1.  set: *flag = value (add == 0) or *flag += value, ordered after the earlier work of the stream. */
int arcvae_gate_wait(const unsigned* flag, unsigned* steps, unsigned stride, unsigned offset, int advance,
                     unsigned max_polls, unsigned* err, arcvae_stream_t stream);
int arcvae_gate_set(unsigned* flag, unsigned value, int add, arcvae_stream_t stream);
int arcvae_tile_weights(const float* const* src, float* const* dst, const int* cols, const int* mode, int n, int H,
                        arcvae_stream_t stream);
/* n <= 8 small device-to-device copies in ONE launch (a step's inputs -- tokens, conditions, eps, coins -- were four copy
 * launches of ~4 us each in front of every step): dst_i[0..nbytes_i) = src_i[0..nbytes_i).  HOST arrays of pointers. */
int arcvae_copy_buffers(const void* const* src, void* const* dst, const long* nbytes, int n, arcvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCVAE_HIP_H */
