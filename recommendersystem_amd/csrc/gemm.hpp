// MFMA GEMM for gfx950: C[M,N] = sum_k A(m,k) * B(n,k) with fused epilogues.
// One kernel family serves every dense contraction of the training step
// (SURVEY.md 2.3 K3/K7/K10/K12/K14/K16): operands may be row-major
// (K contiguous) or K-major (the reduction index is the slow one, as in every
// weight-gradient product); K-major bf16 tiles are consumed with
// ds_read_b64_tr_b16 so no transposed copies are ever written to HBM.
#pragma once
#include "common.hpp"

namespace rsys {

enum GemmEpi : int {
  EPI_STORE = 0,        // C = alpha*acc                      (C: T or f32)
  EPI_ACCUM = 1,        // Cf32 += acc
  EPI_ATOMIC = 2,       // atomicAdd(Cf32, acc)               (split-K weight grads)
  EPI_BIAS = 3,         // C = acc + bias[col]                (C: T or f32)
  EPI_RESIDUAL = 4,     // Cf32 = resid + acc
  EPI_QKV_ROPE = 5,     // C = T(acc) with the q and k column regions rotated (RoPE, interleaved pairs)
  EPI_SWIGLU = 6,       // cols interleaved [16 a | 16 b]: C = ab (T), C2 = silu(a)*b (T)
  EPI_TABLE = 7,        // f = acc + E + bias: Cf32 = f, C2 = T(f)
  EPI_GELU = 8,         // z = acc + bias: C = T(z), C2 = T(gelu(z))
  EPI_SWIGLU_BWD = 9    // acc = dg: C2 = saved [a|b] (interleaved), C = [da|db] same layout (model.py:205-213 backward)
};

struct GemmParams {
  const void* A; const void* B; void* C;
  int M, N, K;
  long long lda, ldb, ldc;
  int epi;
  int c_f32;        // 1: C is float, 0: C is T
  int splitk;       // >=1; >1 requires EPI_ATOMIC
  // optional device-side problem limits (no host sync): tiles whose first row is >= *m_dev are not computed, and the
  // reduction stops at *k_dev rounded up to a tile (rows / k beyond the limit must hold data that contributes zero)
  const int* m_dev; const int* k_dev;
  // host-side estimates of *m_dev / *k_dev (0: none): only the kernel choice and the K-split count look at them -- a product whose
  // capacity is 16 K rows but which will stop at ~3 K should be tiled for 3 K
  int m_expect, k_expect;
  float alpha;
  int accum;        // EPI_STORE / EPI_QKV_ROPE with a T output: C = C + result (LoRA updates)
  const float* bias;
  const float* resid; long long ldr;
  void* C2; long long ldc2;
  // EPI_QKV_ROPE
  const float* rope_cos; const float* rope_sin;   // [pos][hd/2]
  const float* rope_cs;                            // optional [pos][hd/2][2] = (cos, sin) interleaved: the register epilogue of the 256-wide
                                                   // kernels then takes a lane's two pairs in ONE 16-byte load instead of two 8-byte ones
  const int* rope_pos;                             // optional per-row position, else row % T
  int T; int hd; int n_q; int n_k;                 // q cols = [0,n_q), k cols = [n_q,n_q+n_k), v after
  // EPI_TABLE
  const float* E;
  unsigned long long* trace;   // tools/micro/gemm8p_trace.hip only (kernel built with RSYS_8P_TRACE): per-workgroup cycle sums
  // deterministic split-K (Model::deterministic): instead of float atomics into C every K split stores its partial tile into
  // slab[split][M][N] (slab_floats = capacity); the launcher clears the slab, and sums the splits in index order into C afterwards
  float* slab; long long slab_floats;
  int flags;        // bit 0: timing experiment (no allowance for pending stores); bit 1: 256x256 kernel with one workgroup
                    // per tile instead of its persistent grid (used while RCCL kernels share the CUs, see model.hip);
                    // bit 2: start-stagger experiment; bits 3-6: gemm8c walks the tiles in bands of that many tile rows (set by its launcher); bit 7: EPI_ATOMIC row-major product on the 256x256 split-K kernel whatever its tile count; bit 8: gemm8c walks its tile rows from the last to the first (the caller's A was written just before, front to back)
  // fp8 operands (launch_gemm8p_f8, the fp8 trunk of f8.hip): f8 = 1: A is e4m3, 2: A is e5m2; B is always e4m3.  lda / ldb / K
  // count 1-byte elements.  The fp32 accumulators are multiplied by a descale 1 / (scale_A scale_B) before the epilogue:
  int f8;
  const float* f8_desc;   // device floats: [0..15] output descales, [16..18] accumulator ratios at K-segment boundaries
  int f8_seg_cols;        // > 0: output column c takes f8_desc[c / f8_seg_cols] (q | k | v weights of different scale), else f8_desc[0]
  int f8_alt;             // 1: 16-column blocks alternate between f8_desc[0] and f8_desc[1] (the [16 a | 16 b] SwiGLU interleave: W1 | W3)
  int f8_kb[3];           // K is a run of segments quantised with different scales (dq | dk | dv, da | db): f8_kb[j] > 0 = the K tile
                          // (128 elements each) at which segment j + 1 begins; there the accumulators are multiplied by
                          // f8_desc[16 + j]; f8_desc[0] is the last segment's descale
  float* f8_amax_out;     // EPI_SWIGLU: amax |g| -> slot [0]; EPI_SWIGLU_BWD: amax |da|, |db| -> slots [0], [1] (sharded slots, common.hpp f8_amax_note)
  int f8_rseg;            // split-K form (weight gradients dW = dY8^T . X8): output ROW r takes f8_desc[r / f8_rseg] (the rows of dY^T come
                          // in units of different scale: dq | dk | dv, da | db); 0: f8_desc[0]
  int f8_rowmode;         // split-K form: 1 = the rows are the de-interleaved [all da | all db] gradient; row r is added to row
                          // (i >> 4) * 32 + is_b * 16 + (i & 15) of C (W13's [16 w1 | 16 w3] row blocks), i = r - is_b * M / 2
};

// CT = compute type (bf16 -> v_mfma_f32_16x16x32_bf16, float -> v_mfma_f32_16x16x4_f32).
// a_f32/b_f32: operand is float in memory although CT is bf16 (converted while staging).
template <typename CT>
int launch_gemm(const GemmParams& p, bool a_f32, bool b_f32, bool a_km, bool b_km, hipStream_t s);

// floats of slab the split-K launch of this problem needs (0: no split-K / nothing to do); mirrors launch_gemm's dispatch
template <typename CT>
long long gemm_slab_need(const GemmParams& p, bool a_f32, bool b_f32, bool a_km, bool b_km);

// Grouped launch of many K-major split-K products (the weight gradients dW = dY^T X of several layers) on the 256x256 LDS-DMA
// pipeline (gemm8p.hip): a plan deals the products to the XCDs once, launches reuse it.  Products: bf16 K-major operands, fp32
// C, EPI_ATOMIC (C must hold zeros or the sum so far).
struct GemmGroupPlan;
bool gemm8p_group_eligible(const GemmParams& p);   // (p.f8 set: the fp8 split-K form, else the bf16 K-major form; one form per plan)
int gemm8p_group_plan_create(const GemmParams* probs, int n, GemmGroupPlan** out, bool ordered = false);   // ordered: split-K partial tiles to slabs, summed in index order (deterministic mode)
void gemm8p_group_plan_destroy(GemmGroupPlan* pl);
double gemm8p_group_flops(const GemmGroupPlan* pl);
bool gemm8p_group_on_4k(const GemmGroupPlan* pl);
int gemm8p_group_splitk(const GemmGroupPlan* pl);
int launch_gemm8p_group(const GemmGroupPlan* pl, hipStream_t s);

// fp8 row-major operands on the persistent 256x256 pipeline (K tiles of 128 elements, v_mfma_f32_16x16x128_f8f6f4); p.f8 set
// row-major A x K-major B, split-K atomics over device-side live rows (the tied head's dEw); K a multiple of 64
bool gemm8p_mix_eligible(const GemmParams& p);
int launch_gemm8p_mix(const GemmParams& p, hipStream_t s);
bool gemm8p_f8_eligible(const GemmParams& p);
int launch_gemm8p_f8(const GemmParams& p, hipStream_t s);
// the same operands, split-K with fp32 atomics (EPI_ATOMIC): the fp8 trunk's weight gradients on K-contiguous (transposed) copies
bool gemm8p_f8_splitk_eligible(const GemmParams& p);
int launch_gemm8p_f8_splitk(const GemmParams& p, hipStream_t s);

// short name of the kernel launch_gemm picks for this problem ("8c", "8p", "8t", "8s", "nt", "nn", "tn"): timing tags
const char* gemm_kernel_name(const GemmParams& p, bool bf16_mode, bool a_f32, bool b_f32, bool a_km, bool b_km);

}  // namespace rsys
