// Launchers of the non-GEMM kernels of the training step (HBM-bound row kernels,
// attention, optimizer).  T = activation storage type (bf16 in the benchmarked
// mode, float in the 1e-4 parity mode).  All launch on the given stream.
#pragma once
#include "common.hpp"

namespace rsys {

// Device-resident batch record: the 27 arrays of SURVEY 8(a) A1 plus the masked
// copies mask_tokens produces (transformer.model.py:417-462).
struct BatchDev {
  int N, rows, S;
  const int *userid, *tmid, *gender, *source, *matchedid, *status;
  const double* time;
  const float *rating, *progress;
  const float* label[6]; const float* weight[6]; const int* position[6];  // [medium*3 + {watch,rating,status}]
  const unsigned char *watch_mask, *rating_mask;   // optional explicit masks (parity mode)
  const int* rope_pos;                              // optional (inference)
  // outputs of mask_tokens
  int *m_tmid, *m_matchedid, *m_status;
  float *m_rating, *m_progress;
  float* m_label[4]; float* m_weight[4]; int* m_position[4];              // [medium*2 + {watch,rating}]
};

// Deterministic mode (Model::deterministic): while `part` is set, the reduction kernels below write per-workgroup partial sums
// into it instead of issuing float atomics, and their launchers add the partials to the destination in workgroup order
// (launch_reduce_parts).  Thread-local: set by the model around its launches (one host thread drives one model), all on the
// model's stream, so consecutive launches may reuse the scratch from offset 0.
struct DetScratch { float* part = nullptr; long long cap = 0; float* tmp = nullptr; long long tmp_cap = 0; };   // tmp: second stage of long reductions
extern thread_local DetScratch g_det;
// dst[c] += sum_b part[b * stride + c] for c < n, b = 0 .. nparts-1 in that order
int launch_reduce_parts(const float* part, int nparts, long long stride, int n, float* dst, hipStream_t s);

struct SmallParams {  // pointers into the flat fp32 parameter buffer
  const float *per_cos, *per_sin;      // (2),(2)
  const float *status_emb, *gender_emb, *source_emb;  // (10,16),(5,4),(5,4)
  int n_status, n_gender, n_source;    // vocab sizes (mask row index)
  double min_ts, max_ts;
  float rating_mean, rating_std;
};

int launch_mask_tokens(BatchDev b, int finetune, int finetune_metric, float mask_rate,
                       unsigned long long seed, unsigned long long step, hipStream_t s);

template <typename T>
int launch_action_features(const BatchDev& b, const SmallParams& sp, T* feat /*[N][32]*/, hipStream_t s);

// x0[2n] = F32[id'] ; also token-level uid/tm arrays (interleaved, model.py:468-469)
int launch_gather_items(const BatchDev& b, const float* F32, int V, int D, float* x0, int* uid_t, int* tm_t, hipStream_t s);

// rows_dev (optional, device): compact row set -- rows [0, *rows_dev) are computed, rows up to the next multiple of 256 are
// written as zeros, the rest is left alone (compact.hip)
template <typename T>
int launch_rmsnorm_fwd(const float* x, const float* scale, T* y, float* rstd, long long rows, int D, hipStream_t s, const int* rows_dev = nullptr,
                       const int* in_rows = nullptr /* output row r is computed from input row in_rows[r] */,
                       float* f8_amax = nullptr /* fp8 trunk: sharded amax slot of the output (common.hpp f8_amax_add) */);

// dx_out = resid_grad + d/dx rmsnorm ; dscale += column sums (atomic)
// dx_out_t (optional): T-typed copy of dx_out, the A operand of the GEMMs that consume it
template <typename T>
int launch_rmsnorm_bwd(const T* g, const float* x, const float* scale, const float* rstd, const float* resid_grad,
                       float* dx_out, T* dx_out_t, float* dscale, long long rows, int D, hipStream_t s, const int* rows_dev = nullptr,
                       const int* resid_slot = nullptr /* resid_grad is compact: row resid_slot[row] of it, zero where -1 */,
                       const int* io_rows = nullptr /* x is read at, and dx_out / dx_out_t written to, row io_rows[row] (g, rstd, resid_slot: row) */,
                       float* f8_amax = nullptr /* fp8 trunk: sharded amax slot of dx_out_t */);
// same with an f32 incoming gradient (final norm: gy is f32)
template <typename T>
int launch_rmsnorm_bwd_f32(const float* g, const float* x, const float* scale, const float* rstd, const float* resid_grad,
                           float* dx_out, T* dx_out_t, float* dscale, long long rows, int D, hipStream_t s, const int* rows_dev = nullptr,
                           float* f8_amax = nullptr);

// dst = (accumulate ? dst : 0) + src * mask / (1 - p), mask ~ Bernoulli(1-p) from Philox(seed, stream)(element index):
// nn.Dropout of the LoRA input (model.py:238,265,269); the backward pass regenerates the same mask
template <typename T>
int launch_dropout(const T* src, T* dst, long long n, float p, unsigned long long seed, unsigned int stream, int accumulate, hipStream_t s);

int launch_select_positions(const float* w, int N, int topk, int* idx, float* stats /*[2]: wsum_sel, wsum_all*/,
                            int* npos_out /*number of positive-weight rows selected (they come first)*/, hipStream_t s);
// up to four independent selections (the four (medium, metric) tasks) in one launch, one workgroup each
// the same by many workgroups (two small launches + a finishing one); scratch: 12 * 32 words of device memory
int launch_select_positions_chunked(int ntask, const float* const* w, int N, int topk, int* const* idx, float* const* stats, int* const* npos_out,
                                    void* scratch, hipStream_t s);
int launch_select_positions_batch(int ntask, const float* const* w, int N, int topk, int* const* idx, float* const* stats,
                                  int* const* npos_out, hipStream_t s);

template <typename T>
int launch_gather_rows(const T* src, long long ld, const int* idx, int parity, T* dst, int n, int D, hipStream_t s);

int launch_scatter_rows_add(const float* src, const int* idx, int parity, float* dst, long long ld, int n, int D, hipStream_t s,
                            const int* npos = nullptr /* device: only rows r < *npos are added */);

// cross entropy over logits[n][ldl] (valid cols < V): accumulates loss_out[0] += sum ce*label*w ;
// overwrites logits with dlogits = coef*(softmax - onehot), coef = tw*label*w/max(wsum,1e-8); pad cols zeroed
template <typename T>
int launch_ce_fwd_bwd(T* logits, long long ldl, int n, int V, const int* idx, const float* label, const float* weight,
                      const int* position, const float* stats, const int* npos, float task_w, float* loss_out, hipStream_t s);

// rating head tail: pred = hact.w2 + b2 ; losses ; dz = dpred*w2*gelu'(z) (in place over z) ; dw2,db2,db0 accumulated
template <typename T>
int launch_rating_tail(T* z, const T* hact, int n, int D, const float* w2, const float* b2, const int* idx,
                       const float* label, const float* weight, const float* stats, float rating_mean, float task_w,
                       int evaluate, float* loss_out /*[3]*/, float* dw2, float* db2, float* db0, hipStream_t s,
                       const int* npos = nullptr /* device: number of positive-weight rows (they come first); rows beyond the next multiple of 128 are skipped */);

int launch_colsum_add(const float* src, long long ld, long long rows, int cols, float* dst, hipStream_t s);
// dst = bf16(src) and colsum += column sums of src, one pass (D/4 must divide 1024)
int launch_cast_colsum(const float* src, bf16* dst, long long rows, int D, float* colsum, hipStream_t s);
// dstT[c][r] = bf16(src[r][c]) (row stride ldt, rows up to the next multiple of 64 zero-filled) and colsum += column sums, one pass (D % 64 == 0)
int launch_cast_transpose_colsum(const float* src, bf16* dstT, long long rows, int D, long long ldt, float* colsum, hipStream_t s);

int launch_embedding_scatter_add(const float* gx0, const BatchDev& b, int V, int D, float* gE, hipStream_t s);   // float atomics (A/B reference only)

// ---- deterministic embedding-gradient scatter (scatter.hip)
// token index of a batch: tokens sorted by (item id with -1 -> V, token index); keys: workspace of token_index_capacity(N) u64
int token_index_capacity(int N);
int launch_token_index_build(const int* matchedid, int N, int V, unsigned long long* keys, int* skey, int* sidx, hipStream_t s);
size_t seg_scatter_slab_floats(int N, int D);
// gE[id'] += sum over the tokens n (ascending) with masked id' of gx0[n * ldx .. + D); one writer per table row, no atomics
int launch_embedding_scatter_segmented(const float* gx0, long long ldx, const int* m_matchedid, const int* skey, const int* sidx,
                                       int N, int V, int D, float* gE, float* slab, hipStream_t s);

// ---- selected-token (compact) row sets for the top of the trunk (compact.hip)
// union of the tasks' live positions (token 2 idx + (task & 1), rows r < *npos[task]) as a bitmap over the tokens + the number of
// selected tokens before each bitmap word + their total; launch_selected_first turns it into sel[0 .. *nsel) ascending, slot[token] = rank or -1
int launch_token_union(const int* const* idx, const int* const* npos, int ntask, int NT, unsigned int* bits, int* pre, int* nsel, hipStream_t s);
// slot / sel from the bitmap of launch_token_union, and the selected-first token order of every batch row: see compact.hip
int launch_selected_first(const unsigned int* bits, const int* pre, int* slot, int* sel, const int* uid, const int* tm, const int* rope_pos, int B, int T,
                          int* perm, int* uid_p, int* tm_p, int* pos_p, int* slot_p, int* sel_p, int* q_active, hipStream_t s);
// dst rows [0, n) <- src rows sel[r]; dst rows [n, n rounded up to 256) <- 0   (n = *n_dev <= cap)
template <typename T> int launch_gather_rows_sel(const T* src, long long ld, const int* sel, const int* n_dev, int cap, T* dst, int D, hipStream_t s);
// places [0, 64 q_active[b]) of every batch row <- compact row slot_p[place] or zeros; the other places untouched (compact.hip)
template <typename T> int launch_scatter_rows_fill(const T* src, const int* slot_p, const int* q_active, int B, int Tseq, T* dst, long long ld, int D, hipStream_t s);
template <typename T> int launch_scatter_rows_sel(const T* src, const int* sel, const int* n_dev, int cap, T* dst, long long ld, int D, hipStream_t s);
template <typename T> int launch_scatter_rows_map(const T* src, const int* map, int n, T* dst, long long ld, int D, hipStream_t s);   // dst[map[r]] = src[r]
// heads: dst[r] = compact[slot[2 idx[r] + parity]] (zeros where -1), r < n; and compact[slot[..]] += src[r] for r < min(n, *npos)
template <typename T> int launch_gather_rows_slot(const T* compact, const int* slot, const int* idx, int parity, T* dst, int n, int D, hipStream_t s);
int launch_scatter_rows_add_slot(const float* src, const int* slot, const int* idx, int parity, const int* npos, float* compact, int n, int D, hipStream_t s);

// ---- row-sharded item table (shard.hip): exchange plan, rows by id, vocabulary-parallel cross entropy
int launch_plan_unique(const int* skey, const int* sidx, int N, int V, int* slot, int* uniq, int* tok2u, int* plan /*{U, uV}*/, hipStream_t s);
int launch_plan_offsets(const int* uniq, const int* plan, const int* bound_dev, int nb, int* off, hipStream_t s);
int launch_gather_rows_by_id(const float* src, long long ld, const int* ids, int sub, float* dst, int n, int D, hipStream_t s);
int launch_add_rows_by_id(const float* src, const int* ids, int sub, float* dst, long long ld, int n, int D, hipStream_t s);
int launch_add_rows_by_id_counted(const float* src, const int* ids, const int* count_dev, float* dst, long long ld, int cap, int D, hipStream_t s);   // rows [0, min(*count_dev, cap))
int launch_gather_items_remote(const BatchDev& b, const float* Frem, const int* tok2u, const int* plan, int D, float* x0,
                               int* uid_t, int* tm_t, hipStream_t s);
int launch_vp_meta(const int* idx, const float* label, const float* weight, const int* position, const float* stats,
                   const int* npos, float task_w, int KB, int KBmax, float* own /*[KBmax*4 + 4]*/, hipStream_t s);
int launch_add_scalar(float* dst, const float* src, hipStream_t s);
template <typename T>
int launch_vp_compact(const T* EwAll, const float* metaAll, int W, int KB, int D, T* EwC, float* metaC, int* nlive, int* pre /*[W+1]*/, hipStream_t s);
template <typename T>
int launch_vp_stats(const T* logits, long long ldl, int Vloc, int col0, const float* metaC, const int* nlive, float* lmax,
                    float* sums /*[2*cap]*/, int cap, int grid_rows, hipStream_t s);
int launch_vp_rebase(const float* lmax, const float* gmax, float* sums, int n, hipStream_t s);
template <typename T>
int launch_vp_finish(T* logits, long long ldl, int Vloc, int col0, const float* metaC, const float* gmax, const float* sums, int cap,
                     const int* nlive, const int* pre, int rank, float* loss_out, int grid_rows, hipStream_t s);

// sampled softmax over the local classes (shard.hip)
int launch_ss_sample(int len, int n_s, unsigned long long seed, unsigned int stream, int* cols, hipStream_t s);
template <typename T> int launch_gather_rows_plain(const T* src, long long ld, const int* idx, int base, T* dst, int n, int D, hipStream_t s);
int launch_add_rows_plain(const float* src, const int* idx, int base, float* dst, long long ld, int n, int D, hipStream_t s);
template <typename T>
int launch_ss_target_logit(const T* EwC, const T* Floc, int D, int len, int col0, const float* metaC, const int* nlive, float* tl, int grid_rows, hipStream_t s);
template <typename T>
int launch_ss_stats(const T* logits, long long ldl, int n_s, int n_tot, int len, int col0, const int* cols, const float* metaC, const int* nlive,
                    float* lmax, float* lsum, int grid_rows, hipStream_t s);
int launch_ss_targets(const float* metaC, const int* nlive, int cap_rows, int len, int col0, unsigned int* bitmap, int* out, int* count, hipStream_t s);
int launch_ss_drop_hits(int* cols, int n_s, const unsigned int* bitmap, hipStream_t s);
int launch_ss_max_with_target(const float* lmax, const float* tl, float* out, int n, hipStream_t s);
int launch_ss_rebase(const float* lmax, const float* gmax, float* lsum, int n, hipStream_t s);
template <typename T>
int launch_ss_finish(T* logits, long long ldl, int n_s, int n_tot, int len, int col0, const int* cols, const float* metaC, const float* gmax,
                     const float* sneg, const float* tl, const int* nlive, const int* pre, int rank, float* loss_out,
                     float* dt, int grid_rows, hipStream_t s);
template <typename T>
int launch_ss_target_grad(const T* EwC, const T* Floc, int D, int len, int col0, const float* metaC, const float* dt, const int* nlive,
                          float* gE_loc, float* dEwC, int grid_rows, hipStream_t s);

int launch_action_small_bwd(const float* gf /*[N][32]*/, const BatchDev& b, const SmallParams& sp,
                            float* g_per_cos, float* g_per_sin, float* g_status, float* g_gender, float* g_source,
                            hipStream_t s);

// ---- attention (attention.hip)
struct AttnParams {
  int B, T, H, KV, hd;
  int is_bf16;                 // the launches that follow are the bf16 ones (launch_attn_tilemap needs to know: order_k lists kv tile PAIRS for attn_bwd_kv32_kernel)
  const void *q, *k, *v;      // row-major views into qkv [B*T][ld] (q, k already rotated)
  long long ld;                // row stride of q/k/v (elements)
  void* o; long long ldo;     // [B*T][H*hd]
  float* lse;                  // [B][H][T] natural-log sum-exp of the scaled scores
  const int *uid, *tm;         // [B*T]
  // [B][ceil(T/64)] bitmaps: qmap bit j = kv tile j has an allowed pair with this q tile, qmap_full = every pair allowed;
  // kmap / kmap_full: the transposed relation (bit j = q tile j)
  unsigned int *qmap, *kmap, *qmap_full, *kmap_full;
  // [B][ceil(T/64)][4]: the same "some allowed pair" bits per group of 16 queries (qmap16: bit j = kv tile j) and per
  // group of 16 keys (kmap16: bit j = q tile j): a wave owns one such group and skips the tiles it has nothing in
  unsigned int *qmap16, *kmap16;
  // optional [B]: only the first q_active[b] query tiles of row b matter (the compact top, model.hpp: the tokens of the last layer are
  // ordered selected-first).  Forward and dQ skip the other query tiles (their outputs are never read; dQ writes zeros), dK/dV
  // leaves them out of its sums (their dO is identically zero).
  const int* q_active;
  // optional: kmap, kmap_full, qmap_full and kmap16 (the maps the tile-map kernel ORs into) live back to back in one allocation
  // starting here: one zero-fill instead of four
  void* maps_zero_base; size_t maps_zero_bytes;
  // backward
  const void* dO; float* delta;   // delta [B][H][T] = rowsum(dO * O): written by the dQ kernel, read by the dK/dV kernel
  void *dq, *dk, *dv; long long ldg;     // un-rotated gradients, row-major views into dqkv
  const float *rope_cos, *rope_sin; const int* rope_pos;
  // fp8 trunk: sharded amax slots (common.hpp f8_amax_note) of what the kernels write -- forward: |O| -> [0]; backward: |dq| -> [0],
  // |dk| -> [1], |dv| -> [2]; nullptr: off
  float* f8_amax;
  // optional: heaviest-first launch orders, written by launch_attn_tilemap from the maps (attn_order_kernel) and read by the kernels'
  // blockIdx -> work mapping: order_q [B * H * ceil(T/64)] for the forward / dQ kernels (work = kv tiles of the q tile), order_k
  // [B * KV * ceil(T/64)] for the dK/dV kernel (work = q tiles of the kv tile).  nullptr: plain order.
  int *order_q, *order_k;
  // optional: the same for the 128-query forward kernel (attn_fwd32_kernel: slots = (head of the kv group, PAIR of q tiles), work = kv tiles
  // either tile visits): [B * H * ceil(ceil(T/64) / 2)]
  int* order_q2;
  // pair bits of every (q tile, kv tile), written by launch_attn_tilemap (the same compares that build the maps) and read by the three
  // kernels instead of comparing keys per score: qbits [B][nt q][nt kv][64 queries]: bit k = the query may see key k of the kv tile;
  // kbits [B][nt kv][nt q][64 keys]: bit q = the key is seen by query q of the q tile (the transposed matrix).  Required.
  unsigned long long *qbits, *kbits;
};
int launch_attn_tilemap(const AttnParams& p, hipStream_t s);
template <typename T> int launch_attn_fwd(const AttnParams& p, hipStream_t s);
template <typename T> int launch_attn_bwd(const AttnParams& p, hipStream_t s);

// ---- optimizer (optim.hip)
// *out += sum of g[i]^2 in a fixed order (bitwise the same on every rank for the same gradient); part: sumsq_parts() floats of device scratch,
// reused by consecutive calls on one stream
int launch_sumsq(const float* g, long long n, float* out /*device scalar: accumulated, or (overwrite) set*/, float* part, hipStream_t s, bool overwrite = false);
int sumsq_parts();
// replica-consistency checksum of the ranges [ranges[2r], ranges[2r+1]) of a flat fp32 buffer: out (device) = {fp64 sum, fp64 sum of squares,
// low / high 32 bits of a position-weighted wrapping sum of the bit patterns}; part = checksum_scratch_doubles() doubles of device scratch
int launch_checksum(const float* p, const long long* ranges, int n_ranges, double* part, double* out, hipStream_t s);
int checksum_scratch_doubles();
template <typename T>
int launch_adamw(float* p, float* g, float* m, float* v, T* shadow, long long n_decay, long long n_total,
                 float lr, float b1, float b2, float eps, float wd, int step, const float* sumsq, float grad_div,
                 float max_norm, int zero_grad, hipStream_t s, long long sh_skip_lo = 0, long long sh_skip_hi = 0 /* elements [lo, hi) get no shadow copy */);
template <typename T> int launch_cast(const float* src, T* dst, long long n, hipStream_t s);
int launch_add_bias_rows(const float* a, const float* bias, float* dst, long long rows, int D, hipStream_t s);   // dst = a + bias (row-wise)
template <typename T> int launch_widen(const T* src, float* dst, long long n, hipStream_t s);
template <typename T> int launch_rowdot(const T* h, const float* w, const float* b, float* out, int n, int D, hipStream_t s);
// batched 2-D transposes of bf16 matrices (the weight shadows the dx GEMMs read as row-major [in][out] operands)
struct TransposeJob { const void* src; void* dst; int rows, cols; long long ld_src, ld_dst; };
struct TransposeBatch { TransposeJob job[64]; int n; };
int launch_transpose_bf16(const TransposeBatch& b, hipStream_t s);
int launch_scale(float* g, long long n, const float* sumsq, float grad_div, float max_norm, hipStream_t s);
// dst[0, n) = elements [first, first + n) of the N(0, std) stream (seed, stream)
int launch_fill_normal(float* dst, long long n, float std, unsigned long long seed, unsigned int stream, hipStream_t s, long long first = 0);
// rows [row0, row0 + rows) of a table generated row by row
template <typename T>
int launch_fill_normal_t(T* dst, long long rows, int cols, long long ld, float std, unsigned long long seed, hipStream_t s, long long row0 = 0);

// When set, the next launch of rmsnorm_fwd (its T-typed output) / rmsnorm_bwd (its T-typed operand copy of dx) also adds the amax
// of what it writes to this sharded slot (common.hpp f8_amax_note); the launcher clears it.  Same thread as the launch.

// ---- fp8 trunk (f8.hip): torchao's tensor-wise dynamic scaling restated (transformer.py:671-676)
enum { F8_E4M3 = 0, F8_E5M2 = 1 };
enum { F8_LAYOUT_PLAIN = 0, F8_LAYOUT_SEGS = 1, F8_LAYOUT_SWIGLU = 2 };
struct F8Cast {
  const void* src; long long ld_src; int src_f32;   // bf16 (or f32) [rows][cols]
  int rows, cols;
  const int* rows_dev;                    // optional device-side row count (rows up to the next multiple of 256 are zero-filled)
  int fmt, layout, seg_cols, seg_rep;     // SEGS: column units of seg_cols; the first seg_rep units are segment 0, every further unit its own
                                          // segment (q | k | v with grouped-query heads), <= 4 segments, each with its own amax; SWIGLU: [16 a | 16 b] blocks, 2 amaxes
  float* amax;                            // device: consecutive sharded slots, one per segment (common.hpp f8_amax_note: element [shard * 32 + segment],
                                          // 64 shards): zero before the producer / launch_f8_amax adds to it; read by launch_f8_cast
  unsigned char* dst; long long ld_dst;   // [rows][cols] (SWIGLU: columns de-interleaved to [all a | all b])
  // descales of the consumer GEMM (GemmParams::f8_desc), written by the cast: mode 1 = output-column units (this tensor's scale x
  // the weight scale of unit u: the first w_rep units take weight 0, unit u >= w_rep weight u - w_rep + 1; n_w weights), mode 2 = K
  // segments (this tensor's segment j x weight j; desc[0] = last, desc[16 + j] = ratios)
  float* desc; const float* wamax; int n_w; int w_rep; int desc_mode;
  // optional transposed copy dst_t[c'][r] (ld_dst_t >= rows; rows % 128 == 0, cols % 64 == 0): the K-contiguous operand of the fp8
  // weight gradient dW = dY8^T . X8 (K = tokens)
  unsigned char* dst_t; long long ld_dst_t;
  // optional descales of that weight-gradient product (GemmParams::f8_desc of launch_gemm8p_f8_splitk), written by the cast of the
  // GRADIENT operand: desc_dw[u] = 1 / (s_this[segment of row unit u] * s_x), u < dw_units (units as in seg_cols / seg_rep; SWIGLU:
  // two units), s_x from the sharded amax slot xamax of the forward operand
  float* desc_dw; const float* xamax; int dw_units;
};
int launch_f8_amax(const F8Cast& c, hipStream_t s);
// dst[i] += float(bf16(stage[i])); stage[i] = 0   (n % 4 == 0, both 16-byte aligned)
int launch_round_bf16_accum(float* stage, float* dst, long long n, hipStream_t s);
int launch_f8_cast(const F8Cast& c, hipStream_t s);
struct F8WeightJob {
  const float* src; long long ld; int rows, cols;   // fp32 master [rows][cols]
  int layout, seg_rows, seg_rep;                    // SEGS: row units of seg_rows, the first seg_rep units are segment 0 (q | k | v); SWIGLU: [16 w1 | 16 w3] row blocks
  float* amax;                                      // one slot per segment
  unsigned char* dst;                               // e4m3 [rows][cols]
  unsigned char* dst_t; long long ld_t;             // e4m3 [cols][ld_t]: column = row (SWIGLU: de-interleaved to [all w1 | all w3])
};
int launch_f8_weights(const F8WeightJob* jobs_dev, const int* tile_job_dev, const int* tile_first_dev, int ntiles, hipStream_t s);

}  // namespace rsys
