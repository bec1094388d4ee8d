/* rsys_debug.h -- TEST AND PARITY HOOKS of librsys_hip.so.  Not part of the drop-in boundary: a host that trains or serves binds
 * include/rsys.h only.  These entry points exist for the parity suite (tests/), the micro benchmarks (tools/) and
 * bench.py's per-kernel timing: bit-exact read-back of the index paths, raw per-kernel access on caller-provided device buffers,
 * an in-process rank group that lets the multi-rank arithmetic run on a one-GPU box, and a delay kernel for cross-stream ordering
 * tests.  Same conventions as rsys.h (status codes, rsys_last_error, opaque handles). */
#ifndef RSYS_DEBUG_H
#define RSYS_DEBUG_H
#include "rsys.h"

#ifdef __cplusplus
extern "C" {
#endif

/* debug/parity: integer and index paths of the last forward, read back bit-exactly (tests compare them with the
 * reference's mask_tokens, model.py:417-462, and position selection, model.py:501-513).  Keys: "masked.token_mask_ids",
 * "masked.matchedid", "masked.status" (int32 [rows*S]); "masked.rating", "masked.progress" (f32); "masked.<medium>.
 * <watch|rating>.<label|weight|position>"; "idx.<task>" (int32 [mask_topk*rows], task = medium*2 + metric); "npos"
 * (int32 [4]); "tokens.userid", "tokens.token_mask_ids" (int32 [rows*2S], model.py:468-469); "embed.x0" (f32
 * [rows*2S*D]); "table.fused" (f32 [(V+1)*D]); after a training pass also the compact top of the trunk: "top.cap",
 * "top.n" (int32 [1]: capacity / number of selected tokens), "top.sel" (int32 [top.cap], the sorted selected tokens), "top.slot"
 * (int32 [rows*2S], token -> compact row or -1); "host_syncs" (int32 [2]: stream drains and event waits inside the last
 * rsys_forward_backward).  `bytes` must be the exact size of the array. */
int32_t rsys_debug_get(rsys_model* m, const char* key, void* out, int64_t bytes);

/* in-process rank group (tests): `world` ranks of ONE process on one device, each driven by its own host thread; the
 * collectives are device copies between the ranks' buffers.  Two RCCL ranks cannot share a GPU; this lets the multi-rank
 * partition arithmetic run on a one-GPU box with the real kernels. */
int32_t rsys_local_group_create(int32_t world, int32_t device, void** group);
int32_t rsys_local_group_destroy(void* group);
int32_t rsys_comm_init_local(void* group, int32_t rank, rsys_comm** out);
/* tests: occupy the communicator's stream for `microseconds` (<= 2e6) with a spinning kernel -- a collective that starts late; what
 * the cross-stream ordering test of the split table reduce delays (tests/test_gpu_split_table_reduce.py) */
int32_t rsys_comm_debug_delay(rsys_comm* c, int32_t microseconds);

/* per-kernel access for unit tests (device pointers from rsys_dev_alloc) */
int32_t rsys_dev_alloc(void** p, size_t bytes);
int32_t rsys_dev_free(void* p);
int32_t rsys_dev_h2d(void* dst, const void* src, size_t bytes);
int32_t rsys_dev_d2h(void* dst, const void* src, size_t bytes);
int32_t rsys_dev_memset(void* dst, int value, size_t bytes);
/* C[M,N] = sum_k A(m,k)B(n,k); dtype RSYS_DTYPE_*; a_km/b_km: operand stored K-major; a_f32: A is f32 in memory */
int32_t rsys_op_gemm(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                     int64_t lda, int64_t ldb, int64_t ldc, int32_t a_km, int32_t b_km, int32_t a_f32, int32_t c_f32,
                     int32_t splitk);
/* the same product with the row count taken from device memory, as the head GEMMs over the selected positions do
 * (model.py:501-516: only rows with a positive target weight reach the heads): rows >= *rows_dev are not computed
 * (rows up to the end of the last started tile may be written); row-major A, c_f32 / b_km as above; c_f32 == 3: fp32 C ACCUMULATED by
 * split-K atomics (the tied head's dEw = dlogits . F) */
int32_t rsys_op_gemm_rows(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                          int64_t lda, int64_t ldb, int64_t ldc, int32_t b_km, int32_t c_f32, const int32_t* rows_dev);
/* K-major operands (A [K][lda >= M], B [K][ldb >= N]), f32 C stored (accumulate = 0) or added to (1), the reduction limited to the first
 * *k_dev rows of the operands (device memory): the tied head's table gradient dF (+)= dlogits^T Ew over the live selected rows
 * (model.py:153-170 backward); rows >= *k_dev may hold anything */
int32_t rsys_op_gemm_klimit(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                            int64_t lda, int64_t ldb, int64_t ldc, int32_t accumulate, const int32_t* k_dev);
/* attention fwd+bwd on caller-provided device buffers (T-typed): qkv [B*T][(H+2KV)*hd] post-RoPE, dO [B*T][H*hd],
 * uid/tm [B*T] int32 with 0 <= uid < 2^19 and 0 <= tm < 4096, rope tables [T][hd/2] f32; outputs O [B*T][H*hd], lse [B][H][T] f32,
 * dqkv [B*T][(H+2KV)*hd] (gradients w.r.t. the un-rotated q, k and v) */
int32_t rsys_op_attention(int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t KV, int32_t hd, const void* qkv,
                          const int32_t* uid, const int32_t* tm, void* O, float* lse, const void* dO, void* dqkv,
                          const float* rope_cos, const float* rope_sin);
/* embedding-gradient scatter of the backward (nn.Embedding backward, model.py:21) on caller-provided device buffers:
 * gE[id'] += sum over tokens n of gx0[n*ldx .. +D) with id' = m_matchedid[n] (-1 -> row V); matchedid = the raw ids the
 * token index is built from (m_matchedid differs from it only where it is -1).  One writer per table row, fixed summation
 * order: bitwise reproducible.  atomic != 0: the float-atomic form (A/B reference; needs ldx == 2 D). */
int32_t rsys_op_embedding_scatter(const float* gx0, int64_t ldx, const int32_t* matchedid, const int32_t* m_matchedid, int32_t N,
                                  int32_t V, int32_t D, float* gE, int32_t atomic);
/* fp8 trunk (RSYS_DTYPE_FP8: the reference's torchao "tensorwise" float8 linears, transformer.py:671-676), unit-test access on
 * caller-provided device buffers.  fmt: 0 = e4m3, 1 = e5m2.
 * rsys_op_f8_quantize: amax_dev (64 shards of 32 floats; the maximum over the shards of element seg) = max |src| per column segment (layout 0: one; 1: column units of seg_cols, the first seg_rep
 *   units are segment 0 and every further unit its own segment -- q | k | v with grouped-query heads; 2: the [16 a | 16 b] column
 *   blocks of the W13 output, two segments), then dst = sat_rne(src * FMAX / amax) as fp8 bytes (layout 2: columns de-interleaved
 *   to [all a | all b]); src bf16 [rows][cols].  desc_mode 1 / 2 also writes the descales a consumer GEMM takes (1: desc[u] =
 *   1 / (s_src s_w[weight of output unit u]) for n_w weight amaxes, the first w_rep units on weight 0; 2: K segments, desc[0] =
 *   last segment, desc[16 + j] = ratios); desc_dev holds 32 floats.
 * rsys_op_f8_weights: the same for one fp32 weight matrix [rows][cols] (row segments), plus its transposed copy dst_t [cols][ld_t].
 * rsys_op_gemm_f8: C[M,N] = descale * sum_k A8[m][k] B8[n][k] on the 256x256 fp8 pipeline (K % 128 == 0, K >= 256); a_fmt as fmt,
 *   B is e4m3; desc_dev / seg_cols / alt / kb0..kb2 as GemmParams::f8_* (csrc/gemm.hpp); C bf16 or f32. */
int32_t rsys_op_f8_quantize(const void* src, int64_t ld_src, int32_t rows, int32_t cols, int32_t fmt, int32_t layout, int32_t seg_cols,
                            int32_t seg_rep, void* dst, int64_t ld_dst, float* amax_dev, float* desc_dev, const float* wamax_dev,
                            int32_t n_w, int32_t w_rep, int32_t desc_mode);
int32_t rsys_op_f8_weights(const float* src, int64_t ld, int32_t rows, int32_t cols, int32_t layout, int32_t seg_rows, int32_t seg_rep,
                           float* amax_dev, void* dst, void* dst_t, int64_t ld_t);
int32_t rsys_op_gemm_f8(const void* A8, const void* B8, void* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                        int32_t a_fmt, int32_t c_f32, const float* desc_dev, int32_t seg_cols, int32_t alt, int32_t kb0, int32_t kb1,
                        int32_t kb2);

#ifdef __cplusplus
}
#endif
#endif
