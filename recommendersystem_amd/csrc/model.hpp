// Model state + step orchestration (host side of the HIP path).
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/rsys.h"
#include "../../include/rsys_debug.h"
#include "comm.hpp"
#include "gemm.hpp"
#include "kernels.hpp"

namespace rsys {

enum RowMap { MAP_DIRECT = 0, MAP_W1 = 1, MAP_W3 = 2 };

struct TensorInfo {
  std::string name;      // reference state_dict key
  int64_t rows, cols;    // external (reference) shape; 1-D tensors have rows = 1
  int ndim;
  int64_t off;           // offset (floats) of the internal block inside the flat buffers
  int64_t ld;            // internal row stride
  int map;               // RowMap
  bool trainable;
  bool frozen_table;     // metadata embedding (stored separately, T-typed)
};

struct LayerOff {  // offsets into the flat buffers
  int64_t wqkv, wo, w13, w2, sa, mlp;
  int64_t la, lb;   // finetune: LoRA A = [Aq; Av] (16 x D), B = block matrix (Nqkv x 16): q rows x cols 0-7, v rows x cols 8-15
};

struct PhaseTimer {
  bool enabled = false;
  bool serialize = false;   // run the side-stream GEMMs in line (clean per-kernel attribution)
  std::vector<std::pair<std::string, hipEvent_t>> marks;
  std::map<std::string, double> acc_ms;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  // only call sites whose name contains `filter` are timed ("" = all): bench.py times the dominant kernel family alone inside its timed
  // region (two event records per site cost the step ~4 us each; all ~120 sites of a step: +7 %)
  std::string filter;
  std::vector<char> open;   // per open tic: 1 = recorded, 0 = filtered out (its toc records nothing)
};

struct Model {
  rsys_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream for the weight-gradient GEMMs of the trunk backward: each runs beside the dx GEMM that shares its
  // input (MFMA-bound beside output-bound), forked and joined with events around the pair
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_pending = false;
  hipEvent_t ev_sel = nullptr; bool sel_pending = false;   // position selection + token union of the pass, running on the side stream
  // deferred joins (RSYS_SIDE_STREAM=2): one event per weight-gradient product of a layer (W2, W13, Wo, Wqkv), recorded on the
  // side stream behind it; the main stream waits for a product only where the buffer it reads is overwritten next
  hipEvent_t ev_dw[4] = {nullptr, nullptr, nullptr, nullptr};
  bool dw_pending[4] = {false, false, false, false};
  // DDP-style early gradient reduction (train.py:678-682): when set, the trunk backward hands every finished bucket of
  // per-layer weight gradients [lo, hi) of the flat buffer to this hook (>= 25 MB each, reverse layer order); the hook
  // enqueues its all-reduce on the communicator's stream behind an event and records the range in `reduced`
  std::function<int(int64_t, int64_t)> grad_bucket_hook;
  std::vector<std::pair<int64_t, int64_t>> reduced;
  // Split reduce of the replicated item table's gradient (opt-in, rsys_model_set_split_table_reduce): the head part of dF
  // (complete when heads() returns) is all-reduced OUT OF PLACE into tbl_R while the trunk backward runs and G[E] goes on
  // accumulating the local gradient; the token scatter of that backward goes to a compact list (tok_T rows, u_ids ids, u_plan = {U, uV})
  // and is added to G[E] from there; in the tail only the ranks' lists travel (all-gather), and G[E] = tbl_R + every rank's rows.
  bool split_table = false;          // buffers allocated, the hook may be armed
  bool split_head_reduced = false;   // this backward: the head part is on its way, the token scatter goes through tok_T
  bool split_plan_valid = false;     // u_slot / u_ids / u_plan describe the resident batch (replicated table)
  std::function<int()> table_head_hook;
  hipEvent_t split_head_event = nullptr;   // recorded behind the head part's all-reduce (which reads G[E]); the token-row add into G[E] waits for it
  float *tbl_R = nullptr, *tok_T = nullptr, *tok_Tall = nullptr;
  int *tok_Uall = nullptr, *tok_Pall = nullptr;
  int64_t tok_cap = 0;               // rows of tok_T and ids of u_ids: rows_max * S + 1
  // The tail gathers max_r U_r rows per rank instead of tok_cap (round 6).  A rank knows an upper bound of its own U on the HOST as soon
  // as the batch is staged (distinct matchedid values + the mask row: u_bound_host, counted by batch_stage); when the reduce is armed
  // (rsys_set_grad_sync) the bounds' maximum over the ranks is formed by a one-float all-reduce at the head of the communicator's stream
  // and copied to pinned memory (h_umax[1], event ev_umax): by the time the tail is enqueued it has been there for a whole backward.
  int u_bound_host = 0;
  std::vector<unsigned long long> id_seen;   // batch_stage's bitmap of table rows
  float* h_umax = nullptr; float* d_umax = nullptr; hipEvent_t ev_umax = nullptr; bool umax_pending = false;
  int64_t tok_all_rows = 0;          // rows per rank tok_Tall / tok_Uall are sized for
  int tok_all_world = 0;             // ranks tok_Tall / tok_Uall / tok_Pall are sized for
  int gemm_flags = 0;          // OR-ed into GemmParams.flags: 2 while all-reduce kernels may share the CUs with the backward
  int64_t early_reduced = 0;   // elements the last rsys_allreduce_grads found already reduced (tests)
  // what the last optimizer step's gradient reduction enqueued, in order: {first element, one past the last, phase}; phase 0 = early bucket
  // (from inside the backward), 1 = tail beside the dWp GEMM, 2 = dWp itself, 3 = split table reduce: head part out of place (under the
  // trunk backward), 4 = split table reduce: the ranks' token rows (all-gather; elements = gathered floats).  rsys_grad_sync_schedule reads it.
  struct BucketLog { int64_t lo, hi; int phase; };
  std::vector<BucketLog> bucket_log;
  int bucket_phase = 0;
  bool bf16_mode = false;
  size_t esz = 4;
  // dims
  int L, H, KV, D, I, Ip, S, T, V0, V1, V, M, Mp, K, hd, Nqkv, rows_max;
  int64_t n_decay = 0, n_total = 0;
  int64_t n_opt = 0, n_opt_decay = 0;   // range the optimizer / clip / all-reduce cover (finetune: the LoRA segment only)
  std::vector<TensorInfo> tensors;
  std::map<std::string, int> by_name;
  // flat offsets
  int64_t o_status, o_gender, o_source, o_lin_w, o_E, o_Wp, o_r0w, o_r2w;
  int64_t o_pcos, o_psin, o_lin_b, o_bp, o_norm, o_r0b, o_r2b;
  std::vector<LayerOff> lo;
  // device memory
  float *P = nullptr, *G = nullptr;
  void* Sh = nullptr;        // bf16 shadow of P (bf16 mode); == P in fp32 mode
  void* ShT = nullptr;       // bf16 mode: the trunk weight matrices of Sh, transposed ([in][out]), same offsets: the dx GEMMs then
                             // read row-major operands (256x256 LDS-DMA kernel); rebuilt lazily when wt_dirty
  bool wt_dirty = true;
  // fp8 trunk (cfg.dtype == RSYS_DTYPE_FP8; f8.hip): bf16 arithmetic everywhere except the transformer blocks' linears, whose
  // forward and dx products take tensor-wise scaled e4m3 / e5m2 operands (the reference's torchao recipe, transformer.py:671-676)
  bool fp8 = false;
  int64_t w8_base = 0;            // element offset of the first trunk weight: W8 / W8T hold [w8_base, end of the last layer's W2)
  unsigned char* W8 = nullptr;    // e4m3 copies of the trunk's linear weights, row-major [out][in]: forward B operands
  unsigned char* W8T = nullptr;   // transposed copies [in][out] (W13: K order [all w1 | all w3]): dx B operands
  float* f8_wamax = nullptr;      // [L][8]: amax of q k v o w1 w3 w2 (this step's weights)
  float* f8_aamax = nullptr;      // [L][64 shards][32]: sharded amax slots (common.hpp f8_amax_note) of xn O hn g | dy2 da db dh dq dk dv (this pass)
  float* f8_desc = nullptr;       // [L][8 products][32]: descales the casts write for their consumer GEMMs (GemmParams::f8_desc)
  unsigned char* a8 = nullptr;    // fp8 copy of the current product's A operand
  void* f8_jobs = nullptr; int* f8_tile_job = nullptr; int* f8_tile_first = nullptr; int f8_ntiles = 0;
  bool w8_dirty = true;
  void* sel_scratch = nullptr;    // chunk counts / sums of the multi-workgroup position selection
  float* rope_cs = nullptr;       // [T][hd/2][2]: (cos, sin) interleaved copy of the RoPE tables for the QKV epilogue
  // fp8 weight gradients dW = q_e5m2(dY)^T . q_e4m3(X): every cast also leaves a K-contiguous (transposed, [features][tokens]) copy
  // per layer, consumed by the split-K fp8 form after (or, grouped, at the end of) the backward.  RSYS_F8_DW=0: bf16 operands instead.
  bool f8_dw = false;
  bool f8_tcopies = true;         // this pass is followed by a backward (forward-only passes skip the transposed copies)
  struct F8T { unsigned char *xn, *O, *hn, *g, *gxt, *dab, *dht, *dqkv; };
  std::vector<F8T> f8t;
  float* f8_desc_dw = nullptr;    // [L][4 products: w2 w13 o qkv][32]: row descales of the weight-gradient products
  int64_t f8_ldt = 0;             // row stride of the transposed copies (tokens of a full batch)
  // RSYS_F8_DW_ROUND_BF16=1: every fp8 weight-gradient product is summed (fp32, all its split-K parts) into dw_stage, rounded to
  // bf16 once and only then added to the fp32 gradient -- what autograd does under autocast, where the product's output tensor is
  // bf16 and param.grad is fp32.  Default: the fp32 sum goes to the gradient unrounded.
  float* f8_dw_stage = nullptr;   // [trunk weights of all layers], zero between products
  int64_t f8_dw_stage_base = 0;   // offset (in the flat buffers) of its first element
  std::vector<void*> f8_keep;     // RSYS_F8_DEBUG_KEEP=1 (tests): per layer, copies of the three dx products' outputs (w13_dx, o_dx, qkv_dx)
  bool table_dirty = true;     // the fused item table F / FT must be rebuilt before the next forward: set by everything that changes a
                               // parameter or the metadata, and by the table-gradient pass (it borrows FT); clean between the
                               // micro-steps of one optimizer step, across finetune steps (frozen table) and across inference calls
  void* Meta = nullptr;      // [V+1][Mp] T
  float* F32 = nullptr;      // [V+1][D]
  void* FT = nullptr;        // [V+1][D] T
  // bf16 mode: K-contiguous (transposed) operand copies for the metadata-projection gradient dWp = dF^T Meta, which then
  // runs on the row-major LDS-DMA pipeline: MetaT [Mp][Vp] (built when the table is loaded), dFT [D][Vp] (per step);
  // Vp = V + 1 rounded up to 64, the padding stays zero
  void* MetaT = nullptr; void* dFT = nullptr; int64_t Vp = 0;
  // ---- row-sharded item table (cfg-4, shard.hip): this rank holds table rows [row_lo, row_lo + TR) of E / Meta / F
  bool sharded = false; int sh_rank = 0, sh_world = 1;
  int row_lo = 0, TR = 0;                // TR = V + 1 when the table is replicated
  rsys_comm* shard_comm = nullptr;
  // exchange plan of the resident batch: distinct ids (sorted) and who owns / asks for them
  int *u_slot = nullptr, *u_ids = nullptr, *u_tok = nullptr, *u_plan = nullptr /*{U, uV}*/, *u_bound = nullptr, *u_off = nullptr;
  int U = 0, uV = 0;
  std::vector<long long> need_off, serve_off, need_offD, serve_offD;   // per-rank element offsets (ids; rows of D floats)
  int* req_ids = nullptr; long long req_cap = 0, R = 0;                  // ids the peers ask this rank for, by requester
  float *Frem = nullptr; float* rows_xchg = nullptr; long long xchg_cap = 0;   // [U][D] fetched rows / gradient sums; [R][D] served rows / received gradients
  // vocabulary-parallel head workspaces: gathered selected rows of all ranks, packed live rows, row statistics
  void *EwAll = nullptr, *EwC = nullptr; float *metaOwn = nullptr, *metaAll = nullptr, *metaC = nullptr, *vp_max = nullptr, *vp_lmax = nullptr, *vp_sums = nullptr;
  int *vp_nlive = nullptr, *vp_pre = nullptr; float* dEwC = nullptr;
  // The sizes of the vocabulary-parallel heads' collectives (live rows per rank, in-batch targets of the sampled soft-max) depend on the
  // masked batch only, so they are computed and copied to pinned host memory BEFORE the trunk forward (sharded_counts_early); the
  // heads wait for an event that is a whole trunk forward old instead of draining the stream once per task.
  float* metaAllT[2] = {nullptr, nullptr};   // the gathered row meta of the two watch tasks (kept from the early pass)
  int* h_counts = nullptr;                   // pinned: [2][64] = per watch task pre[0 .. W] and, at [48], the number of in-batch targets
  hipEvent_t ev_counts = nullptr; bool counts_pending = false;
  int host_stream_syncs = 0, host_event_waits = 0;   // blocking host waits inside the last rsys_forward_backward (tests)
  int64_t ldl_loc = 0; float* sumsq_E = nullptr;
  // sampled softmax: sampled local classes, their rows of F, the rows' gradients, target logits and their gradients
  int* ss_cols = nullptr; void* ss_F = nullptr; float *ss_dF = nullptr, *ss_tl = nullptr, *ss_dt = nullptr;
  unsigned int* ss_bitmap = nullptr; int* ss_tcount = nullptr;   // in-batch targets: bitmap over the local classes of a medium, their number
  unsigned long long cur_seed = 0, cur_step = 0;
  float *rope_cos = nullptr, *rope_sin = nullptr;
  int rope_npos = 0;
  std::vector<void*> allocs;
  // batch
  BatchDev bd;
  void* batch_blob = nullptr;
  // one upload = one copy: the batch's arrays are packed back to back (for the rows it has) in a pinned staging buffer and
  // land in `raw_blob` with a single H2D; the BatchDev pointers are re-pointed into it per upload
  unsigned char* raw_blob = nullptr; unsigned char* h_stage = nullptr; size_t raw_bytes = 0;
  // A second (staging, blob) pair for rsys_batch_prefetch: the NEXT batch is checked, packed and copied on its own stream while the
  // current step still runs (the reference's DataLoader + non_blocking to_device overlap, train.py:162-165,178-184); rsys_batch_swap
  // makes it the resident batch.  slot_blob[0] / slot_stage[0] are raw_blob / h_stage.
  unsigned char* slot_blob[2] = {nullptr, nullptr}; unsigned char* slot_stage[2] = {nullptr, nullptr};
  int cur_slot = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copy_done = nullptr;     // the pending batch's H2D copy (copy_stream)
  hipEvent_t ev_blob_free[2] = {nullptr, nullptr};   // recorded on `stream` when slot i stops being the resident batch: its kernels are all enqueued before
  bool blob_free_valid[2] = {false, false};
  struct PendingBatch { bool valid = false; BatchDev bd; int rows = 0; bool has_masks = false, has_rope_pos = false; int distinct_ids = 0;
                        unsigned char *d_wm = nullptr, *d_rm = nullptr; int* d_rope_pos = nullptr; } pending;
  bool tok_index_valid = false;
  // deterministic mode (rsys_model_set_deterministic): every float sum of the step has a fixed order -- split-K partial tiles go
  // to det_slab and are added in split order, the reduction kernels write per-workgroup partials to det_part (kernels.hpp
  // DetScratch) -- so a step is bitwise reproducible (replicated or row-sharded table, full or sampled soft-max)
  bool deterministic = false;
  float* det_slab = nullptr; long long det_slab_floats = 0;
  float* det_part = nullptr; long long det_part_floats = 0;
  float* det_tmp = nullptr; long long det_tmp_floats = 0;   // the token index of the resident batch (scatter / exchange plan) has been built
  bool has_masks = false, has_rope_pos = false;
  int cur_rows = 0;
  // token index of the resident batch (scatter.hip): sorted (item id, token) pairs + the partial-sum slab of the scatter
  unsigned long long* tok_keys = nullptr; int *tok_skey = nullptr, *tok_sidx = nullptr; float* scatter_slab = nullptr;
  unsigned char *d_wm = nullptr, *d_rm = nullptr;
  int* d_rope_pos = nullptr;
  // activations (void* = T-typed)
  size_t maps_zero_bytes = 0;   // kmap | kmap_full | qmap_full | kmap16 are one allocation: bytes of the region the tile-map kernel needs zeroed
  void* feat; float* x0; int *uid_t, *tm_t; unsigned int *qmap, *kmap, *qmap_full, *kmap_full, *qmap16, *kmap16;
  struct LayerAct { float* x; void* xn; void* xnd; void* La; void* qkv; void* O; float* lse; float* rstd1; float* h; void* hn; float* rstd2; void* ab; void* g; };
  std::vector<LayerAct> la;
  float* xL; float* rstdf; void* out;
  // heads
  // losses of finished steps parked on the device (rsys_losses_push / _drain): the loop's per-step read-back -- its only host
  // synchronisation -- becomes one read every `loss_ring_cap` steps; slot = [16 loss sums | 8 stats] of a step
  float* loss_ring = nullptr; int loss_ring_n = 0; static constexpr int loss_ring_cap = 1024;
  int* idx[4]; int* npos; float* stats; void* Ew; void* logits; int64_t ldl; float* dE; void* z; void* hact; float* loss_acc;
  // backward workspaces
  float *gy, *gxa, *gxb, *dh; void *gxa_t, *gxb_t, *dh_t;   // *_t: T-typed operand copies (bf16 mode)
  void *dab, *dhn, *dO, *dqkv; float* delta; float* gf;
  void *dLa, *dxl;   // finetune workspaces
  float* sumsq; float* sumsq_part;   // the gradient norm's scalar and the per-workgroup partial sums of its fixed-order reduction
  // Deferred, grouped weight gradients (bf16 pretraining, products too small for the chip one at a time): the backward keeps
  // every layer's dY operands (gradient of the layer's output, of the SwiGLU pre-activations, of the attention residual, of
  // q/k/v) in per-layer buffers and ONE grouped launch (per gradient bucket when buckets are reduced early) computes the four
  // products of all those layers (gemm8p.hip: gemm8p_group_kernel).  Plans are cached per (first layer, last layer, rows).
  bool defer_dw = false;
  struct DwOperands { void *gxt, *dab, *dht, *dqkv; };
  std::vector<DwOperands> dwb;
  std::map<long long, GemmGroupPlan*> dw_plans;
  // Compact top of the trunk (compact.hip).  A training pass reads the trunk's output only at the positions the heads select, and
  // everything after the last layer's attention is token-local: the last layer's output projection, MLP and the final norm run --
  // forward and backward -- on the sorted set of selected tokens c_sel[0 .. *c_n) (c_slot: token -> row of the compact buffers
  // or -1).  Rows no head reads are not computed; backward rows whose gradient is identically zero are not multiplied.
  bool sparse_top = false;        // this model uses it for its training passes (pretraining, replicated table)
  bool top_is_sparse = false;     // the resident forward ran the compact tail: la[L-1].{h,hn,ab,g}, xL and out are NOT materialised
  int ctop_cap = 0;               // rows of the compact buffers: min(tokens, 4 * mask_topk * rows) rounded up to 256
  int *c_sel = nullptr, *c_slot = nullptr, *c_n = nullptr;
  unsigned int* c_bits = nullptr; int* c_pre = nullptr;   // bitmap of the selected tokens and the selected count before each of its words
  float *c_x = nullptr, *c_h = nullptr, *c_xL = nullptr, *c_rstd2 = nullptr, *c_rstdf = nullptr;
  void *c_O = nullptr, *c_hn = nullptr, *c_ab = nullptr, *c_g = nullptr, *c_out = nullptr;
  float *c_gy = nullptr, *c_gx = nullptr, *c_dh = nullptr;
  bool loss_acc_with_c_gy = false;   // loss_acc = the 256-byte header of c_gy's allocation (one zero fill for both)
  void *c_gx_t = nullptr, *c_dab = nullptr, *c_dhn = nullptr, *c_dh_t = nullptr, *c_dO = nullptr;
  // ... and the last layer runs on a SELECTED-FIRST token order (inside every batch row the selected tokens, then the others; attention
  // is indifferent to token order): its attention kernels then visit only the leading query tiles c_qact[b] of a row.  c_perm: original
  // token of a permuted place; uid_p / tm_p / pos_p: ids and RoPE position per permuted place; c_slot_p / c_sel_p: the compact maps in
  // permuted places; *_p maps: the attention tile maps of that order.  la[L-1].{xn, rstd1, qkv, O, lse} are stored in permuted order.
  int *c_perm = nullptr, *uid_p = nullptr, *tm_p = nullptr, *pos_p = nullptr, *c_slot_p = nullptr, *c_sel_p = nullptr, *c_qact = nullptr;
  unsigned int *qmap_p = nullptr, *kmap_p = nullptr, *qmap_full_p = nullptr, *kmap_full_p = nullptr, *qmap16_p = nullptr, *kmap16_p = nullptr;
  unsigned long long *attn_qbits = nullptr, *attn_kbits = nullptr, *attn_qbits_p = nullptr, *attn_kbits_p = nullptr;   // pair bits of the two map sets (AttnParams::qbits / kbits)
  int *attn_order_q = nullptr, *attn_order_k = nullptr, *attn_order_q_p = nullptr, *attn_order_k_p = nullptr;   // heaviest-first launch orders of the two map sets (AttnParams::order_q / order_k)
  bool table_grads_pending = false;
  // the item-table gradient rows of medium m are known to be zero (just zeroed by zero_grad / AdamW and not written since):
  // the first head GEMM of a step then stores dF instead of reading 245 MB of zeros to add to
  bool gE_clean[2] = {false, false};
  bool drop_active = false; unsigned long long drop_seed = 0, drop_step = 0;   // LoRA dropout of the current pass
  bool last_evaluate = false;
  PhaseTimer timer;
  std::vector<hipEvent_t> step_marks;   // rsys_step_mark: one event per optimizer-step boundary (per-step time distribution)
};

struct Optimizer {
  Model* m;
  int device = 0;   // (kept here: the optimizer may be destroyed after its model)
  float lr, b1, b2, eps, wd;
  int step = 0;
  float *mom = nullptr, *var = nullptr;
  // ZeRO-1 (optimizer_set_zero1): this rank keeps moments for, and updates, only its chunk [z_rank * z_chunk, + z_chunk) of the flat
  // optimized range (the last rank also the < 64 * world elements behind the last chunk); mom / var hold z_chunk + z_tail floats
  bool zero1 = false;
  int z_rank = 0, z_world = 1;
  long long z_chunk = 0, z_tail = 0;
  float* z_tailbuf = nullptr;
};
int optimizer_set_zero1(Optimizer* o, int rank, int world);
int optimizer_step_zero1(Optimizer* o, struct rsys_comm* c, float lr_factor, float clip, float grad_div);

int model_create(const rsys_config* cfg, int device, Model** out);
int model_destroy(Model* m);
int model_init_random(Model* m, uint64_t seed);
int model_load_metadata(Model* m, const float* table, int64_t V, int64_t Mdim);
int model_random_metadata(Model* m, uint64_t seed);
int model_set_rope(Model* m, const float* c, const float* s, int64_t n_pos);
int model_param_io(Model* m, const char* name, float* out, const float* in, int64_t n, int which /*0 P,1 G*/);
int model_refresh_shadow(Model* m);
int model_batch_upload(Model* m, const rsys_batch* b);
int model_forward_backward(Model* m, int evaluate, const float task_w[4], float grad_scale, uint64_t seed, uint64_t step);
int model_infer(Model* m, int task, const int32_t* token_index, int64_t n_tokens, float* out, int64_t n);   // token_index == nullptr: every token
int model_item_table(Model* m, float* out, int64_t n);
int model_materialise_trunk_output(Model* m);   // dense trunk output of the resident forward in m->out (a training pass computes it at the selected tokens only)
int model_finalize_grads(Model* m);
bool model_finalize_splittable(const Model* m);
int model_batch_prefetch(Model* m, const rsys_batch* b);   // stage + copy the NEXT batch beside the running step (replicated table only)
int model_batch_swap(Model* m);                            // the prefetched batch becomes the resident one
int model_split_table_enable(Model* m, int on);
int model_split_table_arm(Model* m, struct rsys_comm* c);   // at the head of the communicator's stream: the ranks' maximum of distinct ids of their resident batches
int model_split_table_tail(Model* m, struct rsys_comm* c, hipStream_t cs, int64_t* rows_out);   // on cs: gather the ranks' token rows (*rows_out per rank), G[E] = tbl_R + rows
int model_finalize_stage(Model* m, int stage /*1: prepare, 2: dWp GEMM*/, int64_t* wp_off, int64_t* wp_n);
int model_clip(Model* m, float max_norm, float* norm_out);
int model_set_deterministic(Model* m, int on);
int optimizer_step(Optimizer* o, float lr_factor, float clip, float grad_div);

}  // namespace rsys
