/* rsys.h -- C ABI of the MI355X-native training hot path (librsys_hip.so).
 *
 * Drop-in boundary for the data-parallel training step of Fro116/RecommenderSystem
 * (reference files: notebooks/Training/transformer.model.py = "model.py",
 * notebooks/Training/transformer.py = "train.py").  The reference has no FFI of its
 * own; each entry point below names the reference
 * interface it replaces.  A Julia host binds these with `ccall`, a Python host with
 * ctypes (recommendersystem_amd/_lib.py); INTEGRATION.md shows both stubs.
 *
 * Conventions: every function returns 0 on success, <0 on error
 * (rsys_last_error gives the thread-local message).  Handles are opaque.  The caller
 * owns every host buffer; the library owns all device memory.  Calls on one handle
 * must be serialised by the caller; one handle set per GPU (one process per GPU as
 * in train.py:582-587).  No callbacks, no exceptions cross the boundary.
 */
#ifndef RSYS_H
#define RSYS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsys_model rsys_model;
typedef struct rsys_optimizer rsys_optimizer;
typedef struct rsys_comm rsys_comm;

/* RSYS_DTYPE_FP8: bf16 arithmetic with the transformer blocks' linears (q k v o w1 w3 w2: forward and input-gradient products) on
 * tensor-wise dynamically scaled fp8 operands -- the reference's pretraining arithmetic (transformer.py:671-676, torchao
 * "tensorwise": e4m3 inputs and weights, e5m2 output gradients); pretraining only, needs embed_dim % 128 == 0 >= 256,
 * (num_kv_heads * head_dim) % 128 == 0, num_heads / num_kv_heads <= 14 (the hidden width is padded to a multiple of 128 inside) */
enum { RSYS_DTYPE_FP32 = 0, RSYS_DTYPE_BF16 = 1, RSYS_DTYPE_FP8 = 2 };

/* mirrors the config dict of train.py:535-560 (+ finetune keys of :520-524) */
typedef struct rsys_config {
  int32_t num_layers, num_heads, num_kv_heads, embed_dim, intermediate_dim;
  int32_t max_sequence_length;          /* S interactions per row -> 2S tokens */
  int32_t vocab_0, vocab_1;             /* vocab_sizes["0_matchedid"], ["1_matchedid"] */
  int32_t vocab_status, vocab_gender, vocab_source;
  int32_t metadata_dim;                 /* metadata_emb_size */
  double min_ts, max_ts;
  float rating_mean, rating_std;
  float mask_rate;
  int32_t mask_topk;
  int32_t finetune;                     /* 0/1 */
  int32_t finetune_metric;              /* 0 = watch, 1 = rating */
  int32_t dtype;                        /* RSYS_DTYPE_* : arithmetic type of the dense contractions */
  int32_t max_rows;                     /* rows (of S interactions) per forward call = local batch size */
  float lora_dropout;                   /* finetune only: nn.Dropout(p) on the LoRA input (model.py:238); 0 disables */
  /* row-sharded item table (beyond the reference): world 0 = replicated table (the reference's scheme,
   * transformer.py:678-682); world >= 1: this rank owns table rows [rank*(V+1)/world, (rank+1)*(V+1)/world) of the item
   * embedding, the metadata table, the fused table and their Adam moments; rsys_model_set_shard_comm gives the communicator */
  int32_t table_shard_rank, table_shard_world;
  /* row-sharded table only: > 0 replaces the full soft-max of the watch heads by a sampled one -- per step and medium every
   * rank draws this many of its local classes (stratified uniform) and the partition function is estimated by importance
   * weighting (log-Q correction), the target class always included.  0 = full soft-max (the reference, model.py:514-519). */
  int32_t sampled_negatives;
} rsys_config;

/* the batch record of train.py:75-98 / transformer.jl:79-142: 27 parallel arrays of
 * rows*S interactions.  Index of label/weight/position: medium*3 + {0 watch,1 rating,2 status}. */
typedef struct rsys_batch {
  int32_t rows;
  const int32_t *userid, *token_mask_ids, *gender, *source, *matchedid, *status;
  const double* time;
  const float *rating, *progress;
  const float* label[6];
  const float* weight[6];
  const int32_t* position[6];
  const uint8_t* watch_mask;   /* optional (parity mode): replaces the random draw of model.py:437-440 */
  const uint8_t* rating_mask;  /* optional, with watch_mask */
  const int32_t* rope_input_pos; /* optional (inference, model.py:470-476) */
} rsys_batch;

const char* rsys_version(void);
size_t rsys_last_error(char* buf, size_t n);
int32_t rsys_device_count(int32_t* n);
int32_t rsys_device_synchronize(void);

/* RecommenderModel(config) -- model.py:346-377.  Parameters are zero until set or
 * rsys_model_init_random (init_weights, model.py:5-12) is called. */
int32_t rsys_model_create(const rsys_config* cfg, int32_t device, rsys_model** out);
int32_t rsys_model_destroy(rsys_model* m);
int32_t rsys_model_init_random(rsys_model* m, uint64_t seed);
/* load_pretrained_embeddings -- model.py:379-389; table is (V, M) row-major f32 */
int32_t rsys_model_load_metadata(rsys_model* m, const float* table, int64_t V, int64_t M);
/* synthetic frozen table generated on the device: N(0,1)/sqrt(M) */
int32_t rsys_model_random_metadata(rsys_model* m, uint64_t seed);
/* RoPE tables precompute_freqs_cis -- model.py:173-179; (n_pos, head_dim/2) f32 each */
int32_t rsys_model_set_rope(rsys_model* m, const float* cos, const float* sin, int64_t n_pos);

/* state_dict interchange -- names are the reference's state_dict keys */
int32_t rsys_param_count(rsys_model* m, int32_t* n);
int32_t rsys_param_info(rsys_model* m, int32_t i, char* name, size_t name_cap, int64_t shape[2], int32_t* ndim,
                        int32_t* trainable);
int32_t rsys_param_get(rsys_model* m, const char* name, float* out, int64_t n);
int32_t rsys_param_set(rsys_model* m, const char* name, const float* in, int64_t n);
int32_t rsys_grad_get(rsys_model* m, const char* name, float* out, int64_t n);
int32_t rsys_zero_grad(rsys_model* m);

/* to_device -- train.py:178-184: copies the batch into device-resident buffers */
int32_t rsys_batch_upload(rsys_model* m, const rsys_batch* b);
/* The reference's loader overlap (DataLoader workers + non_blocking to_device, train.py:162-165,178-184): rsys_batch_prefetch checks
 * and packs the NEXT batch on the calling thread and copies it on a stream of its own while the current step still runs on the
 * device (a second staging buffer / device blob); rsys_batch_swap then makes it the resident batch -- enqueue the step's forward /
 * backward / optimizer first, prefetch, read the step's losses, swap.  The host arrays may be freed when prefetch returns.
 * Replicated item table only (the row-sharded table builds its row-exchange plan inside rsys_batch_upload). */
int32_t rsys_batch_prefetch(rsys_model* m, const rsys_batch* b);
int32_t rsys_batch_swap(rsys_model* m);
/* device-side synthetic batch (bench): fills the resident batch from a counter RNG */

/* model(d, evaluate) + loss.backward() -- model.py:493-529, train.py:259-272.
 * task_w[4] in the order (0,watch),(0,rating),(1,watch),(1,rating); the gradient of
 * sum_i task_w[i]*loss_i*grad_scale is ACCUMULATED into the gradient buffer (skipped
 * when evaluate != 0).  mask_seed/step drive the Philox mask draw when the batch
 * carries no explicit masks.  Asynchronous: results are read with rsys_losses_get. */
int32_t rsys_forward_backward(rsys_model* m, int32_t evaluate, const float task_w[4], float grad_scale,
                              uint64_t mask_seed, uint64_t step);
/* losses_out[12]: per task 3 slots (train: [loss,0,0]; evaluate rating tasks: 3 moments of model.py:395-401);
 * weight_sums_out[4]: d[name.weight].sum() after masking (train.py:261).  Synchronises. */
int32_t rsys_losses_get(rsys_model* m, float losses_out[12], float weight_sums_out[4]);
/* The same without a host wait per step (transformer.py:245-262 adds the losses into device tensors and reads them at the end of the
 * epoch, :279-283): rsys_losses_push parks the finished step's sums on the device (stream-ordered, at most 1024 steps),
 * rsys_losses_drain synchronises once and writes every parked step in order: losses_out[n][12], weight_sums_out[n][4] as
 * rsys_losses_get would have returned them; *n_out = n <= cap. */
int32_t rsys_losses_push(rsys_model* m);
int32_t rsys_losses_drain(rsys_model* m, float* losses_out, float* weight_sums_out, int32_t cap, int32_t* n_out);
/* number of positive-weight positions selected per task in the last forward (they bound the head GEMMs) */
int32_t rsys_head_rows_get(rsys_model* m, int32_t out[4]);
/* ItemEmbedding.forward over all items (model.py:139-145), the table Finetune/register.py:27-33 exports as the watch-head
 * weights of the serving registry: out [V][embed_dim] f32, V = vocab_0 + vocab_1 (manga rows first) */
int32_t rsys_item_table(rsys_model* m, float* out, int64_t n);
/* on != 0: every float sum of the training step gets a fixed order (split-K partial tiles summed in split order, reductions through
 * per-workgroup partials instead of float atomics), so a step -- losses, gradients, updated parameters -- is bitwise reproducible
 * from run to run; costs a few percent of the step.  Replicated or row-sharded table, full or sampled soft-max.  (The reference's CUDA path is not reproducible:
 * its embedding backward and split-K reductions use atomics too.) */
int32_t rsys_model_set_deterministic(rsys_model* m, int32_t on);
/* inference forward -- model.py:531-538; task 0 = retrieval (out: rows*2S*D), 1 = ranking (out: rows*2S) */
int32_t rsys_infer(rsys_model* m, int32_t task, float* out, int64_t n);
/* the same forward, returning only the tokens a server reads (Finetune/embed.py:147-161 takes token 2n of a user for
 * retrieval and the candidates' action tokens for ranking): token_index[n_tokens] = flat token indices in [0, rows*2S);
 * out = n_tokens*D floats (retrieval: the trunk output rows) or n_tokens floats (ranking: the rating head on those rows only) */
int32_t rsys_infer_select(rsys_model* m, int32_t task, const int32_t* token_index, int64_t n_tokens, float* out, int64_t n);
/* debug/parity: trunk output of the last forward (rows*2S*D floats).  A training pass computes it only at the positions the heads
 * select; this call then runs the dense tail of the last layer first (results as model.py:335-343 over every token). */
int32_t rsys_trunk_output_get(rsys_model* m, float* out, int64_t n);

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

/* torch.nn.utils.clip_grad_norm_(params, max_norm) -- train.py:273; norm_out may be NULL */
int32_t rsys_clip_grad_norm(rsys_model* m, float max_norm, float* norm_out);

/* create_optimizer -- train.py:285-298 (AdamW, betas 0.9/0.95, wd on dim>=2) */
int32_t rsys_adamw_create(rsys_model* m, float lr, float beta1, float beta2, float eps, float weight_decay,
                          rsys_optimizer** out);
int32_t rsys_adamw_destroy(rsys_optimizer* o);
/* optimizer.step(); optimizer.zero_grad() with lr = lr0*lr_factor (LambdaLR, train.py:684-689).
 * fused_clip_max_norm > 0 fuses clip_grad_norm_ (global norm over the flat gradient buffer) and the
 * data-parallel mean (grads / grad_div) into the update -- one pass over the parameters. */
int32_t rsys_adamw_step(rsys_optimizer* o, float lr_factor, float fused_clip_max_norm, float grad_div);
/* (beyond the reference, opt-in) ZeRO-1 for the replicated data-parallel model: rsys_adamw_set_zero1, right after the create, keeps
 * AdamW moments for this rank's 1/world of the flat parameter range only; rsys_adamw_step_zero1 then replaces rsys_allreduce_grads +
 * rsys_adamw_step: reduce-scatter of the gradient, global-norm clip from the ranks' partial sums, AdamW on the rank's part,
 * all-gather of the parameters.  No early gradient buckets in this mode (do not arm rsys_set_grad_sync). */
int32_t rsys_adamw_set_zero1(rsys_optimizer* o, int32_t rank, int32_t world);
int32_t rsys_adamw_step_zero1(rsys_optimizer* o, rsys_comm* c, float lr_factor, float fused_clip_max_norm, float grad_div);
int32_t rsys_adamw_state_get(rsys_optimizer* o, const char* name, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step);
int32_t rsys_adamw_state_set(rsys_optimizer* o, const char* name, const float* exp_avg, const float* exp_avg_sq, int64_t n, int32_t step);

/* init_process_group("nccl") + DDP gradient all-reduce -- train.py:582, 678-682, 268-272;
 * RCCL over xGMI, one communicator per process.  id_buf: 128 bytes made on rank 0 and
 * distributed by the host (the reference uses torchrun's store). */
int32_t rsys_comm_unique_id(uint8_t id_buf[128]);
int32_t rsys_comm_init(const uint8_t id_buf[128], int32_t rank, int32_t world, int32_t device, rsys_comm** out);
int32_t rsys_comm_destroy(rsys_comm* c);
/* DDP launches a bucket's all-reduce as soon as its gradients are final (train.py:678-682): call before the backward of
 * the LAST micro-step of an optimizer step; the trunk backward then starts the all-reduce of each finished >= 25 MB run
 * of per-layer weight gradients on the communicator's stream while it continues, and rsys_allreduce_grads reduces the
 * rest.  comm == NULL disarms.  (Micro-steps before the last one accumulate locally: DDP no_sync, train.py:268-271.) */
int32_t rsys_set_grad_sync(rsys_model* m, rsys_comm* c);
/* (beyond the reference, opt-in; replicated table, bf16) split the reduce of the item table's gradient, 80 % of the flat buffer:
 * with this on, an armed rsys_set_grad_sync also starts -- as soon as the heads' part of dF is complete, before the trunk backward --
 * an out-of-place all-reduce of that part; the backward's token scatter is kept as a list of distinct rows, and rsys_allreduce_grads
 * all-gathers the ranks' lists instead of all-reducing the table: G[E] = sum of the head parts + every rank's token rows, added in
 * rank order.  Same gradient up to the order of the additions.  Not combined with ZeRO-1. */
int32_t rsys_model_set_split_table_reduce(rsys_model* m, int32_t on);
/* row-sharded table mode: the communicator the forward / backward use for the row exchange and the vocabulary-parallel
 * cross entropy (world must equal cfg.table_shard_world; NULL only when that is 1) */
int32_t rsys_model_set_shard_comm(rsys_model* m, rsys_comm* c);
/* rows [lo, hi) of the (V + 1)-row item table this model holds (the whole table when it is replicated) */
int32_t rsys_table_rows(rsys_model* m, int64_t* lo, int64_t* hi);
/* in-process rank group (tests): `world` ranks of ONE process on one device, each driven by its own host thread; the
 * collectives are device copies between the ranks' buffers.  Two RCCL ranks cannot share a GPU; this lets the multi-rank
 * partition arithmetic run on a one-GPU box with the real kernels. */
int32_t rsys_local_group_create(int32_t world, int32_t device, void** group);
int32_t rsys_local_group_destroy(void* group);
int32_t rsys_comm_init_local(void* group, int32_t rank, rsys_comm** out);
/* tests: occupy the communicator's stream for `microseconds` (<= 2e6) with a spinning kernel -- a collective that starts late; what
 * the cross-stream ordering test of the split table reduce delays (tests/test_gpu_split_table_reduce.py) */
int32_t rsys_comm_debug_delay(rsys_comm* c, int32_t microseconds);
/* sum-all-reduce of (the rest of) the flat gradient buffer in buckets (the mean is folded into rsys_adamw_step's
 * grad_div); *early_floats (optional query): how many gradient elements the last call found already reduced */
int32_t rsys_allreduce_grads(rsys_model* m, rsys_comm* c);
int32_t rsys_grad_sync_early(rsys_model* m, int64_t* early_floats);
/* what the last optimizer step's gradient reduction enqueued, in enqueue order (the DDP bucket schedule of train.py:678-682 as this
 * library runs it): up to cap triples {first element, one past the last element, phase} of the flat gradient buffer; phase 0 = early
 * bucket from inside the backward, 1 = tail beside the metadata-projection gradient GEMM, 2 = that GEMM's output, 3 / 4 = the two parts of
 * the split table reduce (head part out of place; the ranks' token rows, elements = gathered floats).  *n = entries recorded. */
int32_t rsys_grad_sync_schedule(rsys_model* m, int64_t* triples, int32_t cap, int32_t* n);
/* {rank, world, transport (1 = RCCL, 2 = in-process rank group of the tests), RCCL version code (ncclGetVersion) or 0} */
int32_t rsys_comm_info(rsys_comm* c, int32_t out[4]);
int32_t rsys_allreduce_f64(rsys_comm* c, double* x, int32_t n);   /* reduce_mean, train.py:199-204 */
int32_t rsys_self_test(rsys_comm* c);                              /* hardware_check.py:6-12 */

/* raw views for hosts that run collectives themselves (e.g. torch.distributed on aliased memory) */
int32_t rsys_grad_buffer(rsys_model* m, void** dev_ptr, int64_t* n_floats);
int32_t rsys_param_buffer(rsys_model* m, void** dev_ptr, int64_t* n_floats);
int32_t rsys_refresh_shadow(rsys_model* m);   /* re-derive the bf16 compute copies after external parameter writes */

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
 * (rows up to the end of the last started tile may be written); row-major A, c_f32 / b_km as above */
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
/* per-step time distribution (bench.py): rsys_step_mark records an event on the model's stream at an optimizer-step
 * boundary; rsys_step_marks_get writes the milliseconds between consecutive marks (at most cap) and clears the marks */
int32_t rsys_step_mark(rsys_model* m);
int32_t rsys_step_marks_get(rsys_model* m, float* ms_out, int32_t cap, int32_t* n_out);
int32_t rsys_op_timing(rsys_model* m, int32_t enable);  /* collect per-call-site HIP-event timings; 2: also run the side-stream GEMMs in line;
                                                          * 3: pause -- stop recording without a host wait and keep the recorded spans for rsys_timing_get */
int32_t rsys_timing_get(rsys_model* m, char* buf, size_t cap);
/* time only the call sites whose name contains `substr` (NULL or "": all); cleared by rsys_op_timing(m, 0).  Call after rsys_op_timing(m, 1|2). */
int32_t rsys_op_timing_filter(rsys_model* m, const char* substr);
/* Environment switches (RSYS_*; the table is in DESIGN.md "Environment switches", the fields in csrc/switches.hpp).  The library parses the
 * environment when a model or communicator is created and at the entry of every rsys_op_* operator, never inside a training step;
 * rsys_switches_reload parses it on demand.  rsys_switches_describe writes "NAME=value" of the switches that differ from their defaults
 * (space separated, NUL terminated when it fits) and returns the number of characters that takes. */
int32_t rsys_switches_reload(void);
int32_t rsys_switches_describe(char* buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
