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
/* Replica consistency (the reference's DDP constructor broadcasts rank 0's parameters, train.py:678-682; here every rank builds the
 * same parameters from the same seed / the same checkpoint file and the ranks COMPARE): out = {fp64 sum, fp64 sum of squares, low and
 * high 32 bits of a position-weighted wrapping integer sum of the bit patterns} of this rank's flat fp32 parameter buffer, computed on the
 * device in a fixed order (equal parameters give equal words on every rank; the integer word notices a single differing bit).  A
 * row-sharded model leaves out its own table rows (they differ by construction).  The host gathers the ranks' words through
 * rsys_allreduce_f64 with one slot per rank and every rank checks min == max (recommendersystem_amd/dist.py assert_replicas_equal:
 * after init, after resume, at every epoch end).  Synchronises. */
int32_t rsys_param_checksum(rsys_model* m, double out[4]);

/* raw views for hosts that run collectives themselves (e.g. torch.distributed on aliased memory) */
int32_t rsys_grad_buffer(rsys_model* m, void** dev_ptr, int64_t* n_floats);
int32_t rsys_param_buffer(rsys_model* m, void** dev_ptr, int64_t* n_floats);
int32_t rsys_refresh_shadow(rsys_model* m);   /* re-derive the bf16 compute copies after external parameter writes */

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
