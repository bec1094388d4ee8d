// Training-step orchestration: RecommenderModel.train_forward + backward
// (transformer.model.py:493-529 and its autograd), restructured for MI355X:
//  * the item table is fused once per step, F = E + Meta Wp^T + bp (model.py:120-133
//    does the same at inference), and shared by the token gather and both watch heads;
//    its gradient dF IS the gradient of E, and dWp = dF^T Meta is one GEMM per optimizer step;
//  * q/k/v and w1/w3 projections are single GEMMs (weights stored concatenated /
//    16-row interleaved in the flat parameter buffer) with RoPE and SwiGLU epilogues;
//  * residual stream fp32, GEMM operands T (bf16 or fp32), fp32 accumulation;
//  * every kernel of a step is enqueued on one HIP stream, no host sync inside a step.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "model.hpp"

namespace rsys {

#define RC(expr)                  \
  do {                            \
    int _rc = (expr);             \
    if (_rc != RSYS_OK) return _rc; \
  } while (0)

static inline int64_t pad8(int64_t n) { return (n + 7) / 8 * 8; }

static int dalloc(Model* m, void** p, size_t bytes) {
  bytes = (bytes + 255) / 256 * 256;
  HIP_CHECK(hipMalloc(p, bytes));
  HIP_CHECK(hipMemset(*p, 0, bytes));
  m->allocs.push_back(*p);
  return RSYS_OK;
}
#define DALLOC(ptr, bytes) RC(dalloc(m, (void**)&(ptr), (size_t)(bytes)))

// ------------------------------------------------------------------ timing
static void tic(Model* m, const char* name, double flops = 0.0, hipStream_t st = nullptr) {
  PhaseTimer& t = m->timer;
  if (!t.enabled) return;
  if (t.used + 2 > t.pool.size()) {
    if (t.pool.size() >= 8192) { t.enabled = false; return; }   // (a report is taken every few steps; past this the caller forgot to collect)
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {   // (the runtime hands out a bounded number of timing events: ~10 K; stop measuring, keep running)
        (void)hipGetLastError();
        t.enabled = false;
        return;
      }
      t.pool.push_back(e);
    }
  }
  hipEvent_t e = t.pool[t.used++];
  if (hipEventRecord(e, st ? st : m->stream) != hipSuccess) {   // (seen at the production shape with every step instrumented: the record fails once
    (void)hipGetLastError();                                    //  thousands of events are pending; an unchecked failure surfaced at the next launch check)
    t.enabled = false;
    return;
  }
  t.marks.push_back({std::string(name), e});
  t.acc_ms[std::string("#flops:") + name] += flops;
}
static void toc(Model* m, hipStream_t st = nullptr) {
  PhaseTimer& t = m->timer;
  if (!t.enabled) return;
  hipEvent_t e = t.pool[t.used++];
  if (hipEventRecord(e, st ? st : m->stream) != hipSuccess) {   // (the span that was open stays unpaired and is dropped by rsys_timing_get)
    (void)hipGetLastError();
    t.enabled = false;
    return;
  }
  t.marks.push_back({std::string(""), e});
}

// ------------------------------------------------------------------ layout
static void add_tensor(Model* m, const std::string& name, int64_t rows, int64_t cols, int ndim, int64_t off, int64_t ld,
                       int map, bool trainable = true, bool frozen = false) {
  TensorInfo t{name, rows, cols, ndim, off, ld, map, trainable, frozen};
  m->by_name[name] = (int)m->tensors.size();
  m->tensors.push_back(t);
}

static void build_layout(Model* m) {
  const int D = m->D, Ip = m->Ip, I = m->I, L = m->L, hd = m->hd, H = m->H, KV = m->KV;
  const bool ft = m->cfg.finetune != 0;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += pad8(n); return o; };
  m->lo.resize(L);
  // ---- finetune: the LoRA tensors come first so that optimizer / clip / all-reduce cover one prefix of the buffer
  if (ft) {
    for (int l = 0; l < L; ++l) { m->lo[l].la = take((int64_t)16 * D); m->lo[l].lb = take((int64_t)m->Nqkv * 16); }
  }
  const int64_t n_lora = off;
  // ---- decay group (tensors with dim >= 2, train.py:288)
  m->o_status = take((m->cfg.vocab_status + 1) * 16);
  m->o_gender = take((m->cfg.vocab_gender + 1) * 4);
  m->o_source = take((m->cfg.vocab_source + 1) * 4);
  m->o_lin_w = take((int64_t)D * 32);
  m->o_E = take((int64_t)m->TR * D);
  m->o_Wp = take((int64_t)D * m->Mp);
  for (int l = 0; l < L; ++l) {
    m->lo[l].wqkv = take((int64_t)m->Nqkv * D);
    m->lo[l].wo = take((int64_t)D * D);
    m->lo[l].w13 = take((int64_t)2 * Ip * D);
    m->lo[l].w2 = take((int64_t)D * Ip);
  }
  m->o_r0w = take((int64_t)D * D);
  m->o_r2w = take(D);
  m->n_decay = off;
  // ---- no-decay group
  m->o_pcos = take(2); m->o_psin = take(2);
  m->o_lin_b = take(D); m->o_bp = take(D);
  for (int l = 0; l < L; ++l) { m->lo[l].sa = take(D); m->lo[l].mlp = take(D); }
  m->o_norm = take(D); m->o_r0b = take(D); m->o_r2b = take(1);
  m->n_total = off;
  m->n_opt = ft ? n_lora : m->n_total;
  m->n_opt_decay = ft ? n_lora : m->n_decay;

  const bool tr = !ft;  // finetune freezes everything but LoRA (not built yet): base tensors are non-trainable
  add_tensor(m, "action_embedding.periodic_time_cos", 1, 2, 1, m->o_pcos, 2, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.periodic_time_sin", 1, 2, 1, m->o_psin, 2, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.status_embedding.embedding.weight", m->cfg.vocab_status + 1, 16, 2, m->o_status, 16, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.gender_embedding.embedding.weight", m->cfg.vocab_gender + 1, 4, 2, m->o_gender, 4, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.source_embedding.embedding.weight", m->cfg.vocab_source + 1, 4, 2, m->o_source, 4, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.linear.weight", D, 32, 2, m->o_lin_w, 32, MAP_DIRECT, tr);
  add_tensor(m, "action_embedding.linear.bias", 1, D, 1, m->o_lin_b, D, MAP_DIRECT, tr);
  // (row-sharded table: these two tensors are the rank's rows [row_lo, row_lo + TR) of the (V + 1)-row tables)
  add_tensor(m, "item_embedding.matchedid_embedding.embedding.weight", m->TR, D, 2, m->o_E, D, MAP_DIRECT, tr);
  add_tensor(m, "item_embedding.metadata_embedding.embedding.weight", m->TR, m->M, 2, 0, m->Mp, MAP_DIRECT, false, true);
  add_tensor(m, "item_embedding.projection_layer.weight", D, m->M, 2, m->o_Wp, m->Mp, MAP_DIRECT, tr);
  add_tensor(m, "item_embedding.projection_layer.bias", 1, D, 1, m->o_bp, D, MAP_DIRECT, tr);
  for (int l = 0; l < L; ++l) {
    std::string p = "transformers.layers." + std::to_string(l) + ".";
    add_tensor(m, p + "attn.q_proj.weight", H * hd, D, 2, m->lo[l].wqkv, D, MAP_DIRECT, tr);
    add_tensor(m, p + "attn.k_proj.weight", KV * hd, D, 2, m->lo[l].wqkv + (int64_t)H * hd * D, D, MAP_DIRECT, tr);
    add_tensor(m, p + "attn.v_proj.weight", KV * hd, D, 2, m->lo[l].wqkv + (int64_t)(H + KV) * hd * D, D, MAP_DIRECT, tr);
    add_tensor(m, p + "attn.output_proj.weight", D, H * hd, 2, m->lo[l].wo, D, MAP_DIRECT, tr);
    if (ft) {  // model.py:235-254: rank 8 on q_proj and v_proj
      add_tensor(m, p + "attn.q_proj_lora_A.weight", 8, D, 2, m->lo[l].la, D, MAP_DIRECT, true);
      add_tensor(m, p + "attn.q_proj_lora_B.weight", H * hd, 8, 2, m->lo[l].lb, 16, MAP_DIRECT, true);
      add_tensor(m, p + "attn.v_proj_lora_A.weight", 8, D, 2, m->lo[l].la + (int64_t)8 * D, D, MAP_DIRECT, true);
      add_tensor(m, p + "attn.v_proj_lora_B.weight", KV * hd, 8, 2, m->lo[l].lb + (int64_t)(H + KV) * hd * 16 + 8, 16, MAP_DIRECT, true);
    }
    add_tensor(m, p + "mlp.w1.weight", I, D, 2, m->lo[l].w13, D, MAP_W1, tr);
    add_tensor(m, p + "mlp.w2.weight", D, I, 2, m->lo[l].w2, Ip, MAP_DIRECT, tr);
    add_tensor(m, p + "mlp.w3.weight", I, D, 2, m->lo[l].w13, D, MAP_W3, tr);
    add_tensor(m, p + "sa_norm.scale", 1, D, 1, m->lo[l].sa, D, MAP_DIRECT, tr);
    add_tensor(m, p + "mlp_norm.scale", 1, D, 1, m->lo[l].mlp, D, MAP_DIRECT, tr);
  }
  add_tensor(m, "transformers.norm.scale", 1, D, 1, m->o_norm, D, MAP_DIRECT, tr);
  add_tensor(m, "rating_head.0.weight", D, D, 2, m->o_r0w, D, MAP_DIRECT, tr);
  add_tensor(m, "rating_head.0.bias", 1, D, 1, m->o_r0b, D, MAP_DIRECT, tr);
  add_tensor(m, "rating_head.2.weight", 1, D, 2, m->o_r2w, D, MAP_DIRECT, tr);
  add_tensor(m, "rating_head.2.bias", 1, 1, 1, m->o_r2b, 1, MAP_DIRECT, tr);
}

static inline int64_t internal_row(const TensorInfo& t, int64_t r) {
  if (t.map == MAP_W1) return (r / 16) * 32 + (r % 16);
  if (t.map == MAP_W3) return (r / 16) * 32 + 16 + (r % 16);
  return r;
}

// rows of medium `med` (global ids [0, V0) / [V0, V)) that this rank holds: `len` rows, the first one is id `col0` inside the
// medium and local table row `row` (replicated table: the whole medium)
static void shard_medium_range(const Model* m, int med, int* len, int* col0, int* row) {
  const int s = med == 0 ? 0 : m->V0, e = med == 0 ? m->V0 : m->V;
  const int a = std::max(s, m->row_lo), b = std::min(e, m->row_lo + m->TR);
  *len = std::max(0, b - a); *col0 = a - s; *row = a - m->row_lo;
  if (*len == 0) { *col0 = 0; *row = 0; }
}

int model_create(const rsys_config* cfg, int device, Model** out) {
  ARG_CHECK(cfg != nullptr && out != nullptr, "null argument");
  ARG_CHECK(cfg->embed_dim % cfg->num_heads == 0, "embed_dim % num_heads");
  ARG_CHECK(cfg->num_heads % cfg->num_kv_heads == 0, "num_heads % num_kv_heads");
  const int hd = cfg->embed_dim / cfg->num_heads;
  ARG_CHECK(hd == 16 || hd == 32 || hd == 64 || hd == 128, "head_dim must be 16/32/64/128");
  ARG_CHECK(cfg->embed_dim % 16 == 0 && cfg->embed_dim <= 2048, "embed_dim must be a multiple of 16 and <= 2048");
  ARG_CHECK(cfg->max_sequence_length % 4 == 0 && 2 * cfg->max_sequence_length <= 2048,
            "max_sequence_length must be a multiple of 4 and <= 1024");
  ARG_CHECK(cfg->max_rows >= 1, "max_rows");
  ARG_CHECK(cfg->mask_topk >= 1 && cfg->mask_topk <= cfg->max_sequence_length, "mask_topk");
  ARG_CHECK(cfg->dtype == RSYS_DTYPE_FP32 || cfg->dtype == RSYS_DTYPE_BF16 || cfg->dtype == RSYS_DTYPE_FP8, "dtype");
  ARG_CHECK(cfg->lora_dropout >= 0.f && cfg->lora_dropout < 1.f, "lora_dropout must be in [0,1)");
  ARG_CHECK(cfg->sampled_negatives == 0 || (cfg->sampled_negatives > 0 && cfg->table_shard_world >= 1),
            "sampled_negatives needs the row-sharded table (table_shard_world >= 1)");
  if (cfg->dtype == RSYS_DTYPE_FP8) {
    ARG_CHECK(!cfg->finetune, "dtype fp8: the reference converts the trunk to float8 for pretraining only (transformer.py:671)");
    ARG_CHECK(cfg->embed_dim % 128 == 0 && cfg->embed_dim >= 256 && cfg->intermediate_dim >= 129 && cfg->num_heads > 0 && cfg->num_kv_heads > 0 &&
              (cfg->num_kv_heads * (cfg->embed_dim / cfg->num_heads)) % 128 == 0 && cfg->num_heads / cfg->num_kv_heads <= 14,
              "dtype fp8 needs embed_dim % 128 == 0 (>= 256), intermediate_dim > 128, (num_kv_heads * head_dim) % 128 == 0, num_heads / num_kv_heads <= 14");
  }
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { set_error("no HIP device visible: the HIP path has no CPU fallback"); return RSYS_ERR_HIP; }
  ARG_CHECK(device >= 0 && device < ndev, "device index");
  HIP_CHECK(hipSetDevice(device));
  Model* m = new Model();
  m->cfg = *cfg; m->device = device;
  m->bf16_mode = cfg->dtype == RSYS_DTYPE_BF16 || cfg->dtype == RSYS_DTYPE_FP8;
  m->fp8 = cfg->dtype == RSYS_DTYPE_FP8;
  m->esz = m->bf16_mode ? 2 : 4;
  m->L = cfg->num_layers; m->H = cfg->num_heads; m->KV = cfg->num_kv_heads; m->D = cfg->embed_dim;
  m->I = cfg->intermediate_dim; m->Ip = m->fp8 ? (m->I + 127) / 128 * 128 : (m->I + 15) / 16 * 16;   // (fp8: K tiles of 128 elements; padding rows / columns of W13 / W2 are zero)
  m->S = cfg->max_sequence_length; m->T = 2 * m->S;
  m->V0 = cfg->vocab_0; m->V1 = cfg->vocab_1; m->V = m->V0 + m->V1; m->M = cfg->metadata_dim;
  m->Mp = (m->M + 63) / 64 * 64; m->K = cfg->mask_topk; m->hd = hd;
  m->Nqkv = (m->H + 2 * m->KV) * hd; m->rows_max = cfg->max_rows;
  m->sharded = cfg->table_shard_world >= 1;
  m->sh_world = m->sharded ? cfg->table_shard_world : 1; m->sh_rank = m->sharded ? cfg->table_shard_rank : 0;
  if (m->sh_world > 16 || m->sh_rank < 0 || m->sh_rank >= m->sh_world || (m->sharded && cfg->finetune)) {
    delete m;
    set_error("table_shard: rank must be in [0, world), world <= 16, and finetuning keeps the table replicated (it is frozen)");
    return RSYS_ERR_ARG;
  }
  m->row_lo = (int)((int64_t)m->sh_rank * (m->V + 1) / m->sh_world);
  m->TR = (int)((int64_t)(m->sh_rank + 1) * (m->V + 1) / m->sh_world) - m->row_lo;
  HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  for (int k = 0; k < 4; ++k) HIP_CHECK(hipEventCreateWithFlags(&m->ev_dw[k], hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&m->ev_sel, hipEventDisableTiming));
  build_layout(m);
  const int64_t D = m->D, N = (int64_t)m->rows_max * m->S, NT = 2 * N, KB = (int64_t)m->K * m->rows_max;
  const size_t e = m->esz;
  DALLOC(m->P, m->n_total * 4); DALLOC(m->G, m->n_total * 4);
  if (m->bf16_mode) { DALLOC(m->Sh, m->n_total * 2); DALLOC(m->ShT, m->n_total * 2); } else { m->Sh = m->P; }
  DALLOC(m->Meta, (int64_t)m->TR * m->Mp * e);
  DALLOC(m->F32, (int64_t)m->TR * D * 4); DALLOC(m->FT, (int64_t)m->TR * D * e);
  if (m->bf16_mode) {
    m->Vp = ((int64_t)m->TR + 63) / 64 * 64;
    DALLOC(m->MetaT, (int64_t)m->Mp * m->Vp * 2); DALLOC(m->dFT, (int64_t)D * m->Vp * 2);
    HIP_CHECK(hipMemset(m->MetaT, 0, (size_t)m->Mp * m->Vp * 2)); HIP_CHECK(hipMemset(m->dFT, 0, (size_t)D * m->Vp * 2));
  }
  // RoPE tables (model.py:173-179), fp32 like torch; the host may overwrite them (rsys_model_set_rope)
  {
    const int half = hd / 2;
    std::vector<float> c((size_t)m->T * half), s((size_t)m->T * half);
    for (int k = 0; k < half; ++k) {
      float freq = 1.0f / powf(500000.0f, (float)(2 * k) / (float)hd);
      for (int t = 0; t < m->T; ++t) { float a = (float)t * freq; c[(size_t)t * half + k] = cosf(a); s[(size_t)t * half + k] = sinf(a); }
    }
    DALLOC(m->rope_cos, c.size() * 4); DALLOC(m->rope_sin, s.size() * 4); DALLOC(m->rope_cs, c.size() * 8);
    HIP_CHECK(hipMemcpy(m->rope_cos, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->rope_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> cs(c.size() * 2);
    for (size_t i = 0; i < c.size(); ++i) { cs[2 * i] = c[i]; cs[2 * i + 1] = s[i]; }
    HIP_CHECK(hipMemcpy(m->rope_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    m->rope_npos = m->T;
  }
  // batch blob: 27 raw arrays + masked copies
  {
    size_t per_i = 4 * 6 + 8 + 4 * 2 + 18 * 4 /*raw*/ + 4 * 3 + 4 * 2 + 12 * 4 /*masked*/;
    DALLOC(m->batch_blob, (size_t)N * per_i + 4096);
    unsigned char* p = (unsigned char*)m->batch_blob;
    auto carve = [&](size_t bytes) { void* r = p; p += (bytes + 15) / 16 * 16; return r; };
    BatchDev& b = m->bd;
    b.time = (const double*)carve(N * 8);
    b.userid = (const int*)carve(N * 4); b.tmid = (const int*)carve(N * 4); b.gender = (const int*)carve(N * 4);
    b.source = (const int*)carve(N * 4); b.matchedid = (const int*)carve(N * 4); b.status = (const int*)carve(N * 4);
    b.rating = (const float*)carve(N * 4); b.progress = (const float*)carve(N * 4);
    for (int k = 0; k < 6; ++k) { b.label[k] = (const float*)carve(N * 4); b.weight[k] = (const float*)carve(N * 4); b.position[k] = (const int*)carve(N * 4); }
    b.m_tmid = (int*)carve(N * 4); b.m_matchedid = (int*)carve(N * 4); b.m_status = (int*)carve(N * 4);
    b.m_rating = (float*)carve(N * 4); b.m_progress = (float*)carve(N * 4);
    for (int k = 0; k < 4; ++k) { b.m_label[k] = (float*)carve(N * 4); b.m_weight[k] = (float*)carve(N * 4); b.m_position[k] = (int*)carve(N * 4); }
    b.watch_mask = nullptr; b.rating_mask = nullptr; b.rope_pos = nullptr;
    m->raw_bytes = (size_t)N * (4 * 6 + 8 + 4 * 2 + 18 * 4 + 2 + 8) + 64 * 32;   // raw arrays + the two masks + 2N positions + padding
    DALLOC(m->raw_blob, m->raw_bytes);
    HIP_CHECK(hipHostMalloc((void**)&m->h_stage, m->raw_bytes, hipHostMallocDefault));
    m->slot_blob[0] = m->raw_blob; m->slot_stage[0] = m->h_stage;
    DALLOC(m->slot_blob[1], m->raw_bytes);
    HIP_CHECK(hipHostMalloc((void**)&m->slot_stage[1], m->raw_bytes, hipHostMallocDefault));
    HIP_CHECK(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&m->ev_copy_done, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreateWithFlags(&m->ev_blob_free[i], hipEventDisableTiming));
    m->d_wm = nullptr; m->d_rm = nullptr; m->d_rope_pos = nullptr;   // (placed in raw_blob per upload)
    DALLOC(m->tok_keys, (size_t)token_index_capacity((int)N) * 8); DALLOC(m->tok_skey, N * 4); DALLOC(m->tok_sidx, N * 4);
    DALLOC(m->scatter_slab, seg_scatter_slab_floats((int)N, m->D) * 4);
  }
  DALLOC(m->feat, N * 32 * e); DALLOC(m->x0, NT * D * 4);
  DALLOC(m->uid_t, NT * 4); DALLOC(m->tm_t, NT * 4);
  {   // attention tile maps: the four the tile-map kernel ORs into sit back to back (one zero-fill per step), then the two it stores
    const int64_t mb = ((int64_t)m->rows_max * ((m->T + 63) / 64) * 4 + 255) / 256 * 256;
    unsigned char* base = nullptr;
    DALLOC(base, mb * 7 + mb * 5);
    m->kmap = (unsigned int*)base; m->kmap_full = (unsigned int*)(base + mb); m->qmap_full = (unsigned int*)(base + 2 * mb); m->kmap16 = (unsigned int*)(base + 3 * mb);
    m->qmap = (unsigned int*)(base + 7 * mb); m->qmap16 = (unsigned int*)(base + 8 * mb);
    m->maps_zero_bytes = (size_t)(7 * mb);
    const int64_t nt = (m->T + 63) / 64;
    DALLOC(m->attn_order_q, (int64_t)m->rows_max * m->H * nt * 4); DALLOC(m->attn_order_k, (int64_t)m->rows_max * m->KV * nt * 4);
    DALLOC(m->attn_qbits, (int64_t)m->rows_max * nt * nt * 64 * 8); DALLOC(m->attn_kbits, (int64_t)m->rows_max * nt * nt * 64 * 8);
  }
  m->la.resize(m->L);
  for (int l = 0; l < m->L; ++l) {
    Model::LayerAct& a = m->la[l];
    if (l == 0) a.x = m->x0; else DALLOC(a.x, NT * D * 4);
    DALLOC(a.xn, NT * D * e); DALLOC(a.qkv, NT * m->Nqkv * e);
    a.xnd = a.xn; a.La = nullptr;
    if (cfg->finetune) { DALLOC(a.La, NT * 16 * e); if (cfg->lora_dropout > 0.f) DALLOC(a.xnd, NT * D * e); }
    DALLOC(a.O, NT * D * e); DALLOC(a.lse, (int64_t)m->rows_max * m->H * m->T * 4);
    DALLOC(a.rstd1, NT * 4); DALLOC(a.h, NT * D * 4); DALLOC(a.hn, NT * D * e); DALLOC(a.rstd2, NT * 4);
    DALLOC(a.ab, NT * 2 * m->Ip * e); DALLOC(a.g, NT * m->Ip * e);
  }
  DALLOC(m->xL, NT * D * 4); DALLOC(m->rstdf, NT * 4); DALLOC(m->out, NT * D * e);
  for (int k = 0; k < 4; ++k) DALLOC(m->idx[k], KB * 4);
  DALLOC(m->stats, 8 * 4); DALLOC(m->npos, 4 * 4); DALLOC(m->Ew, KB * D * e);
  m->ldl = pad8(std::max(m->V0, m->V1));
  if (m->sharded) {
    // vocabulary-parallel heads: the selected rows of ALL ranks against this rank's rows of each medium
    const int W = m->sh_world;
    int len_max = 0;
    for (int med = 0; med < 2; ++med) { int len, col0, row; shard_medium_range(m, med, &len, &col0, &row); len_max = std::max(len_max, len); }
    m->ldl_loc = pad8(std::max(len_max, 8));
    const int64_t cap = (int64_t)W * KB;
    // sampled soft-max: the class list of a step = the sampled classes + the in-batch targets (at most one per gathered row)
    const int64_t ss_ns = std::min<int64_t>(cfg->sampled_negatives, std::max(len_max, 1)), ss_tmax = std::min<int64_t>(cap, len_max);
    if (cfg->sampled_negatives > 0) m->ldl_loc = pad8(std::max<int64_t>(m->ldl_loc, ss_ns + ss_tmax));
    DALLOC(m->logits, cap * m->ldl_loc * e); DALLOC(m->dEwC, cap * D * 4); m->dE = m->dEwC;
    DALLOC(m->EwAll, cap * D * e); DALLOC(m->EwC, cap * D * e);
    DALLOC(m->metaOwn, (KB * 4 + 4) * 4); DALLOC(m->metaAll, (int64_t)W * (KB * 4 + 4) * 4); DALLOC(m->metaC, cap * 4 * 4 + 4096);
    DALLOC(m->metaAllT[0], (int64_t)W * (KB * 4 + 4) * 4); DALLOC(m->metaAllT[1], (int64_t)W * (KB * 4 + 4) * 4);
    HIP_CHECK(hipHostMalloc((void**)&m->h_counts, 2 * 64 * sizeof(int), hipHostMallocDefault));
    HIP_CHECK(hipEventCreateWithFlags(&m->ev_counts, hipEventDisableTiming));
    DALLOC(m->vp_max, cap * 4); DALLOC(m->vp_lmax, cap * 4); DALLOC(m->vp_sums, 2 * cap * 4); DALLOC(m->vp_nlive, 64); DALLOC(m->vp_pre, 64 * 4);
    // exchange plan of the resident batch
    DALLOC(m->u_slot, N * 4); DALLOC(m->u_ids, (N + 1) * 4); DALLOC(m->u_tok, N * 4); DALLOC(m->u_plan, 64);
    DALLOC(m->u_bound, (W + 1) * 4); DALLOC(m->u_off, ((W + 1) + W + (int64_t)W * W) * 4 + 64);
    DALLOC(m->Frem, (N + 1) * D * 4); DALLOC(m->sumsq_E, 64);
    if (cfg->sampled_negatives > 0) {
      const int64_t ns = ss_ns + ss_tmax;
      DALLOC(m->ss_cols, ns * 4 + 64); DALLOC(m->ss_F, ns * D * e); DALLOC(m->ss_dF, ns * D * 4);
      DALLOC(m->ss_tl, cap * 4); DALLOC(m->ss_dt, cap * 4);
      DALLOC(m->ss_bitmap, (len_max / 32 + 2) * 4); DALLOC(m->ss_tcount, 64);
    }
    std::vector<int> bound(W + 1);
    for (int r = 0; r <= W; ++r) bound[r] = (int)((int64_t)r * (m->V + 1) / W);
    HIP_CHECK(hipMemcpy(m->u_bound, bound.data(), (W + 1) * 4, hipMemcpyHostToDevice));
    m->need_off.assign(W + 1, 0); m->serve_off.assign(W + 1, 0); m->need_offD.assign(W + 1, 0); m->serve_offD.assign(W + 1, 0);
  } else {
    DALLOC(m->logits, KB * m->ldl * e); DALLOC(m->dE, KB * D * 4);
  }
  DALLOC(m->z, KB * D * e); DALLOC(m->hact, KB * D * e); DALLOC(m->loss_acc, 16 * 4);
  DALLOC(m->gy, NT * D * 4); DALLOC(m->gxa, NT * D * 4); DALLOC(m->gxb, NT * D * 4); DALLOC(m->dh, NT * D * 4);
  if (m->bf16_mode) { DALLOC(m->gxa_t, NT * D * 2); DALLOC(m->gxb_t, NT * D * 2); DALLOC(m->dh_t, NT * D * 2); }
  else { m->gxa_t = m->gxa; m->gxb_t = m->gxb; m->dh_t = m->dh; }
  DALLOC(m->dab, NT * 2 * m->Ip * e); DALLOC(m->dhn, NT * D * e);
  DALLOC(m->dO, NT * D * e); DALLOC(m->dqkv, NT * m->Nqkv * e);
  DALLOC(m->delta, (int64_t)m->rows_max * m->H * m->T * 4); DALLOC(m->gf, N * 32 * 4);
  DALLOC(m->sumsq, 64);
  DALLOC(m->sel_scratch, 12 * 32 * 4);
  m->dLa = nullptr; m->dxl = nullptr;
  if (cfg->finetune) { DALLOC(m->dLa, NT * 16 * e); DALLOC(m->dxl, NT * D * e); }
  {
    // grouped weight gradients where one layer's products are too small for the 256x256 K-major kernel on their own (the
    // dispatcher's threshold: >= 32 output tiles, gemm.hip use_8p_tn): cfg-3 has 22 / 12 / 8 / 4 tiles per product
    const int grp = sw().dw_group;
    const long long t13 = (long long)((2 * m->Ip + 255) / 256) * ((m->D + 255) / 256);
    m->defer_dw = grp != 0 && m->bf16_mode && !cfg->finetune && t13 < 32 && m->D % 8 == 0 && m->Ip % 8 == 0 && m->Nqkv % 8 == 0 && m->L <= 30;
    const int sp = sw().sparse_top;
    // (fp8 trunk: a tensor-wise scale is the amax over ALL tokens of the last layer's activations, so that layer stays dense)
    m->sparse_top = sp != 0 && NT <= (1 << 19) && !m->fp8;   // (also the LoRA finetune: one target per row -- the last layer's tail shrinks to `rows` tokens)
    if (m->defer_dw) {
      m->dwb.resize(m->L);
      for (int l = 0; l < m->L; ++l) {
        m->dwb[l] = Model::DwOperands{nullptr, nullptr, nullptr, nullptr};
        DALLOC(m->dwb[l].dqkv, NT * m->Nqkv * 2);
        if (l == m->L - 1 && m->sparse_top) continue;   // (the top layer's other products run on the compact rows)
        DALLOC(m->dwb[l].gxt, NT * D * 2); DALLOC(m->dwb[l].dab, NT * 2 * m->Ip * 2); DALLOC(m->dwb[l].dht, NT * D * 2);
      }
    }
    if (m->sparse_top) {
      const int64_t cap = (std::min<int64_t>(NT, 4 * KB) + 255) / 256 * 256;
      m->ctop_cap = (int)cap;
      DALLOC(m->c_sel, cap * 4); DALLOC(m->c_slot, NT * 4); DALLOC(m->c_n, 64);
      DALLOC(m->c_bits, (NT / 32 + 2) * 4); DALLOC(m->c_pre, (NT / 32 + 2) * 4);
      DALLOC(m->c_x, cap * D * 4); DALLOC(m->c_h, cap * D * 4); DALLOC(m->c_xL, cap * D * 4); DALLOC(m->c_rstd2, cap * 4); DALLOC(m->c_rstdf, cap * 4);
      DALLOC(m->c_O, cap * D * e); DALLOC(m->c_hn, cap * D * e); DALLOC(m->c_ab, cap * 2 * m->Ip * e); DALLOC(m->c_g, cap * m->Ip * e); DALLOC(m->c_out, cap * D * e);
      DALLOC(m->c_gy, cap * D * 4); DALLOC(m->c_gx, cap * D * 4); DALLOC(m->c_dh, cap * D * 4);
      DALLOC(m->c_dab, cap * 2 * m->Ip * e); DALLOC(m->c_dhn, cap * D * e); DALLOC(m->c_dO, cap * D * e);
      if (m->bf16_mode) { DALLOC(m->c_gx_t, cap * D * 2); DALLOC(m->c_dh_t, cap * D * 2); } else { m->c_gx_t = m->c_gx; m->c_dh_t = m->c_dh; }
      DALLOC(m->c_perm, NT * 4); DALLOC(m->uid_p, NT * 4); DALLOC(m->tm_p, NT * 4); DALLOC(m->pos_p, NT * 4); DALLOC(m->c_slot_p, NT * 4);
      DALLOC(m->c_sel_p, cap * 4); DALLOC(m->c_qact, (int64_t)m->rows_max * 4 + 64);
      const int64_t mb = ((int64_t)m->rows_max * ((m->T + 63) / 64) * 4 + 255) / 256 * 256;
      unsigned char* base = nullptr;
      DALLOC(base, mb * 12);
      m->kmap_p = (unsigned int*)base; m->kmap_full_p = (unsigned int*)(base + mb); m->qmap_full_p = (unsigned int*)(base + 2 * mb); m->kmap16_p = (unsigned int*)(base + 3 * mb);
      m->qmap_p = (unsigned int*)(base + 7 * mb); m->qmap16_p = (unsigned int*)(base + 8 * mb);
      const int64_t nt = (m->T + 63) / 64;
      DALLOC(m->attn_order_q_p, (int64_t)m->rows_max * m->H * nt * 4); DALLOC(m->attn_order_k_p, (int64_t)m->rows_max * m->KV * nt * 4);
      DALLOC(m->attn_qbits_p, (int64_t)m->rows_max * nt * nt * 64 * 8); DALLOC(m->attn_kbits_p, (int64_t)m->rows_max * nt * nt * 64 * 8);
    }
  }
  if (m->fp8) {
    const int L = m->L, Ip = m->Ip;
    const int64_t base = m->lo[0].wqkv, end = m->lo[L - 1].w2 + (int64_t)D * Ip;
    m->w8_base = base;
    DALLOC(m->W8, end - base); DALLOC(m->W8T, end - base);
    DALLOC(m->f8_wamax, L * 8 * 4); DALLOC(m->f8_aamax, (int64_t)L * F8_AMAX_SHARDS * F8_AMAX_SHARD * 4); DALLOC(m->f8_desc, L * 8 * 32 * 4);
    DALLOC(m->a8, NT * std::max<int64_t>(2 * Ip, m->Nqkv));
    std::vector<F8WeightJob> jobs; std::vector<int> tile_job, tile_first;
    bool aligned = true;
    auto add = [&](int l, int64_t off, int rows, int cols, int layout, int seg_rows, int slot) {
      F8WeightJob j{};
      j.src = m->P + off; j.ld = cols; j.rows = rows; j.cols = cols; j.layout = layout; j.seg_rows = seg_rows; j.seg_rep = m->H / m->KV;
      j.amax = m->f8_wamax + l * 8 + slot; j.dst = m->W8 + (off - base); j.dst_t = m->W8T + (off - base); j.ld_t = rows;
      aligned = aligned && (off - base) % 16 == 0;
      tile_first.push_back((int)tile_job.size());
      const int nt = ((rows + 63) / 64) * ((cols + 63) / 64);
      for (int t = 0; t < nt; ++t) tile_job.push_back((int)jobs.size());
      jobs.push_back(j);
    };
    for (int l = 0; l < L; ++l) {
      add(l, m->lo[l].wqkv, m->Nqkv, (int)D, F8_LAYOUT_SEGS, m->KV * hd, 0);   // q | k | v rows: three linears, three scales
      add(l, m->lo[l].wo, (int)D, (int)D, F8_LAYOUT_PLAIN, 0, 3);
      add(l, m->lo[l].w13, 2 * Ip, (int)D, F8_LAYOUT_SWIGLU, 0, 4);         // [16 w1 | 16 w3] row blocks
      add(l, m->lo[l].w2, (int)D, Ip, F8_LAYOUT_PLAIN, 0, 6);
    }
    if (!aligned) { set_error("fp8 trunk: weight offsets are not 16-byte aligned in the fp8 copies"); return RSYS_ERR_ARG; }
    const int f8dw = sw().f8_dw;
    m->f8_dw = f8dw != 0 && (2 * m->S) % 128 == 0;   // (K = tokens in tiles of 128)
    if (m->f8_dw) {
      m->f8_ldt = NT;
      m->f8t.resize(L);
      for (int l = 0; l < L; ++l) {
        Model::F8T& t = m->f8t[l];
        DALLOC(t.xn, D * NT); DALLOC(t.O, D * NT); DALLOC(t.hn, D * NT); DALLOC(t.g, (int64_t)Ip * NT);
        DALLOC(t.gxt, D * NT); DALLOC(t.dab, (int64_t)2 * Ip * NT); DALLOC(t.dht, D * NT); DALLOC(t.dqkv, (int64_t)m->Nqkv * NT);
      }
      DALLOC(m->f8_desc_dw, L * 4 * 32 * 4);
      if (sw().f8_dw_round_bf16 != 0) {
        m->f8_dw_stage_base = m->lo[0].wqkv;
        const int64_t n = m->lo[L - 1].w2 + pad8((int64_t)D * Ip) - m->f8_dw_stage_base;
        DALLOC(m->f8_dw_stage, n * 4);
        HIP_CHECK(hipMemset(m->f8_dw_stage, 0, n * 4));
      }
    }
    if (sw().f8_debug_keep != 0) {   // stage-wise parity tests of the backward products
      m->f8_keep.assign((size_t)L * 3, nullptr);
      for (size_t i = 0; i < m->f8_keep.size(); ++i) DALLOC(m->f8_keep[i], NT * D * 2);
    }
    m->f8_ntiles = (int)tile_job.size();
    DALLOC(m->f8_jobs, jobs.size() * sizeof(F8WeightJob)); DALLOC(m->f8_tile_job, tile_job.size() * 4); DALLOC(m->f8_tile_first, tile_first.size() * 4);
    HIP_CHECK(hipMemcpy(m->f8_jobs, jobs.data(), jobs.size() * sizeof(F8WeightJob), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->f8_tile_job, tile_job.data(), tile_job.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(m->f8_tile_first, tile_first.data(), tile_first.size() * 4, hipMemcpyHostToDevice));
  }
  *out = m;
  return RSYS_OK;
}

int model_destroy(Model* m) {
  if (!m) return RSYS_OK;
  hipSetDevice(m->device);
  hipStreamSynchronize(m->stream);
  hipStreamSynchronize(m->side);
  for (void* p : m->allocs) hipFree(p);
  if (m->copy_stream) { hipStreamSynchronize(m->copy_stream); hipStreamDestroy(m->copy_stream); }
  if (m->h_stage) hipHostFree(m->h_stage);
  if (m->slot_stage[1]) hipHostFree(m->slot_stage[1]);
  if (m->ev_copy_done) hipEventDestroy(m->ev_copy_done);
  for (int i = 0; i < 2; ++i) if (m->ev_blob_free[i]) hipEventDestroy(m->ev_blob_free[i]);
  if (m->h_counts) hipHostFree(m->h_counts);
  if (m->ev_counts) hipEventDestroy(m->ev_counts);
  if (m->det_slab) hipFree(m->det_slab);
  if (m->det_part) hipFree(m->det_part);
  if (m->det_tmp) hipFree(m->det_tmp);
  if (m->req_ids) hipFree(m->req_ids);
  if (m->tok_Tall) hipFree(m->tok_Tall);
  if (m->tok_Uall) hipFree(m->tok_Uall);
  if (m->tok_Pall) hipFree(m->tok_Pall);
  if (m->rows_xchg) hipFree(m->rows_xchg);
  hipEventDestroy(m->ev_fork); hipEventDestroy(m->ev_join); if (m->ev_sel) hipEventDestroy(m->ev_sel); hipStreamDestroy(m->side);
  for (int k = 0; k < 4; ++k) if (m->ev_dw[k]) hipEventDestroy(m->ev_dw[k]);
  for (auto& kv : m->dw_plans) gemm8p_group_plan_destroy(kv.second);
  for (auto e : m->timer.pool) hipEventDestroy(e);
  for (auto e : m->step_marks) hipEventDestroy(e);
  hipStreamDestroy(m->stream);
  delete m;
  return RSYS_OK;
}

int model_refresh_shadow(Model* m) {
  m->wt_dirty = true; m->table_dirty = true; m->w8_dirty = true;
  if (m->bf16_mode) RC(launch_cast<bf16>(m->P, (bf16*)m->Sh, m->n_total, m->stream));
  return RSYS_OK;
}

// init_weights (model.py:5-12): N(0, 0.006) on every Linear/Embedding weight, zero bias, last row of
// every embedding zeroed, norm scales 1, periodic phases 0.  Generated on the device from Philox.
int model_init_random(Model* m, uint64_t seed) {
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipMemsetAsync(m->P, 0, m->n_total * 4, m->stream));
  unsigned int stream_id = 1;
  std::vector<float> ones(m->D, 1.0f);
  for (const TensorInfo& t : m->tensors) {
    if (t.frozen_table) continue;
    const bool is_scale = t.name.size() > 6 && t.name.compare(t.name.size() - 6, 6, ".scale") == 0;
    if (is_scale) { HIP_CHECK(hipMemcpyAsync(m->P + t.off, ones.data(), m->D * 4, hipMemcpyHostToDevice, m->stream)); HIP_CHECK(hipStreamSynchronize(m->stream)); continue; }
    if (t.ndim == 1) continue;  // biases and phases stay zero
    if (t.map == MAP_W3) continue;  // filled together with w1 (same interleaved block)
    if (t.name.find("lora_B") != std::string::npos) continue;   // zeros (model.py:252,254)
    int64_t rows = t.map == MAP_W1 ? 2 * m->Ip : t.rows;
    if (t.off == m->o_E && t.rows == m->TR && t.name.find("matchedid_embedding") != std::string::npos) {
      // the item table: generated per chunk of 4096 GLOBAL rows (one Philox stream each), a rank fills the part it holds
      const int64_t lo = m->row_lo, hi = lo + m->TR;
      for (int64_t r0 = 0; r0 <= m->V; r0 += 4096, ++stream_id) {
        const int64_t a = std::max(r0, lo), b = std::min<int64_t>(std::min<int64_t>(r0 + 4096, (int64_t)m->V + 1), hi);
        if (a < b) RC(launch_fill_normal(m->P + t.off + (a - lo) * t.ld, (b - a) * t.ld, 0.006f, seed, stream_id, m->stream, (a - r0) * t.ld));
      }
      if (hi == (int64_t)m->V + 1) HIP_CHECK(hipMemsetAsync(m->P + t.off + (t.rows - 1) * t.ld, 0, t.cols * 4, m->stream));   // mask row (model.py:9-12)
      continue;
    }
    for (int64_t r0 = 0; r0 < rows; r0 += 4096) {  // rows*ld contiguous block incl. padding (re-zeroed below)
      int64_t nr = std::min<int64_t>(4096, rows - r0);
      RC(launch_fill_normal(m->P + t.off + r0 * t.ld, nr * t.ld, 0.006f, seed, stream_id++, m->stream));
    }
    // zero the padding the block fill touched
    if (t.ld != t.cols) {
      std::vector<float> host((size_t)rows * t.ld);
      HIP_CHECK(hipStreamSynchronize(m->stream));
      HIP_CHECK(hipMemcpy(host.data(), m->P + t.off, host.size() * 4, hipMemcpyDeviceToHost));
      for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = t.cols; c < t.ld; ++c) host[r * t.ld + c] = 0.f;
      HIP_CHECK(hipMemcpy(m->P + t.off, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
    if (t.map == MAP_W1 && m->Ip != m->I) {  // padded w1/w3 rows
      for (int64_t r = m->I; r < m->Ip; ++r) {
        HIP_CHECK(hipMemsetAsync(m->P + t.off + ((r / 16) * 32 + (r % 16)) * t.ld, 0, t.ld * 4, m->stream));
        HIP_CHECK(hipMemsetAsync(m->P + t.off + ((r / 16) * 32 + 16 + (r % 16)) * t.ld, 0, t.ld * 4, m->stream));
      }
    }
    if (t.name.find("embedding.weight") != std::string::npos)
      HIP_CHECK(hipMemsetAsync(m->P + t.off + (t.rows - 1) * t.ld, 0, t.cols * 4, m->stream));
  }
  RC(model_refresh_shadow(m));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  return RSYS_OK;
}

// bf16 mode: MetaT = Meta^T ([Mp][Vp], rows >= M and columns > V stay zero) for the row-major form of dWp = dF^T Meta
static int build_meta_t(Model* m) {
  m->table_dirty = true;   // (called whenever the metadata rows have changed)
  if (!m->bf16_mode) return RSYS_OK;
  TransposeBatch b; b.n = 1;
  b.job[0].src = (const bf16*)m->Meta; b.job[0].dst = (bf16*)m->MetaT; b.job[0].rows = m->TR; b.job[0].cols = m->Mp;
  b.job[0].ld_src = m->Mp; b.job[0].ld_dst = m->Vp;
  RC(launch_transpose_bf16(b, m->stream));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  return RSYS_OK;
}

// rows [0, nrows) of the rank's part of the metadata table <- `rows` (nrows x M floats); the rest (mask row, padding) stays zero
static int meta_write_rows(Model* m, const float* rows, int64_t nrows) {
  const int64_t Mdim = m->M;
  HIP_CHECK(hipSetDevice(m->device));
  const int64_t chunk = 4096;
  std::vector<float> hostf((size_t)chunk * m->Mp);
  std::vector<unsigned short> hosth;
  if (m->bf16_mode) hosth.resize((size_t)chunk * m->Mp);
  HIP_CHECK(hipMemset(m->Meta, 0, (size_t)m->TR * m->Mp * m->esz));  // mask row V and padding stay zero
  for (int64_t r0 = 0; r0 < nrows; r0 += chunk) {
    int64_t nr = std::min(chunk, nrows - r0);
    for (int64_t r = 0; r < nr; ++r) {
      const float* src = rows + (r0 + r) * Mdim;
      if (m->bf16_mode) {
        unsigned short* d = hosth.data() + r * m->Mp;
        for (int64_t c = 0; c < Mdim; ++c) {
          uint32_t u; memcpy(&u, &src[c], 4);
          uint32_t rb = ((u >> 16) & 1u) + 0x7FFFu;   // round to nearest even (inputs are finite table values)
          d[c] = (unsigned short)((u + rb) >> 16);
        }
        for (int64_t c = Mdim; c < m->Mp; ++c) d[c] = 0;
      } else {
        float* d = hostf.data() + r * m->Mp;
        memcpy(d, src, Mdim * 4);
        for (int64_t c = Mdim; c < m->Mp; ++c) d[c] = 0.f;
      }
    }
    void* dst = (unsigned char*)m->Meta + (size_t)r0 * m->Mp * m->esz;
    HIP_CHECK(hipMemcpy(dst, m->bf16_mode ? (void*)hosth.data() : (void*)hostf.data(), (size_t)nr * m->Mp * m->esz, hipMemcpyHostToDevice));
  }
  return build_meta_t(m);
}

// `table`: the FULL (V, M) array of media_embeddings.h5; a rank of a row-sharded table keeps its rows of it
int model_load_metadata(Model* m, const float* table, int64_t V, int64_t Mdim) {
  ARG_CHECK(V == m->V && Mdim == m->M, "metadata table shape must be (V, metadata_dim)");  // model.py:387
  const int64_t nrows = std::max<int64_t>(0, std::min<int64_t>(m->row_lo + m->TR, m->V) - m->row_lo);
  return meta_write_rows(m, table + (int64_t)m->row_lo * Mdim, nrows);
}

int model_random_metadata(Model* m, uint64_t seed) {
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipMemsetAsync(m->Meta, 0, (size_t)m->TR * m->Mp * m->esz, m->stream));
  const float std_ = 1.0f / sqrtf((float)m->M);
  const long long nrows = std::max<long long>(0, std::min<long long>(m->row_lo + m->TR, m->V) - m->row_lo);   // (the mask row stays zero)
  if (m->bf16_mode) RC(launch_fill_normal_t<bf16>((bf16*)m->Meta, nrows, m->M, m->Mp, std_, seed, m->stream, m->row_lo));
  else RC(launch_fill_normal_t<float>((float*)m->Meta, nrows, m->M, m->Mp, std_, seed, m->stream, m->row_lo));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  return build_meta_t(m);
}

int model_set_rope(Model* m, const float* c, const float* s, int64_t n_pos) {
  ARG_CHECK(n_pos == m->T, "rope tables must have 2*max_sequence_length positions");
  HIP_CHECK(hipSetDevice(m->device));
  size_t bytes = (size_t)n_pos * (m->hd / 2) * 4;
  HIP_CHECK(hipMemcpy(m->rope_cos, c, bytes, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(m->rope_sin, s, bytes, hipMemcpyHostToDevice));
  {
    const size_t n = bytes / 4;
    std::vector<float> cs(n * 2);
    for (size_t i = 0; i < n; ++i) { cs[2 * i] = c[i]; cs[2 * i + 1] = s[i]; }
    HIP_CHECK(hipMemcpy(m->rope_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  }
  return RSYS_OK;
}

// state_dict <-> flat compute layout (which == 0: parameters, 1: gradients)
int model_param_io(Model* m, const char* name, float* out, const float* in, int64_t n, int which) {
  auto it = m->by_name.find(name);
  if (it == m->by_name.end()) { set_error(std::string("unknown parameter: ") + name); return RSYS_ERR_ARG; }
  const TensorInfo& t = m->tensors[it->second];
  ARG_CHECK(n == t.rows * t.cols, "element count does not match the parameter's shape");
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  if (t.frozen_table) {
    ARG_CHECK(which == 0, "the metadata table is frozen (model.py:113-114)");
    if (in) {
      // the rows below the mask row come from the caller; the mask row is kept zero (model.py:386)
      return meta_write_rows(m, in, std::max<int64_t>(0, std::min<int64_t>(m->row_lo + m->TR, m->V) - m->row_lo));
    }
    std::vector<unsigned char> host((size_t)t.rows * m->Mp * m->esz);
    HIP_CHECK(hipMemcpy(host.data(), m->Meta, host.size(), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < t.rows; ++r)
      for (int64_t c = 0; c < t.cols; ++c) {
        if (m->bf16_mode) { uint32_t u = (uint32_t)((unsigned short*)host.data())[r * m->Mp + c] << 16; memcpy(&out[r * t.cols + c], &u, 4); }
        else out[r * t.cols + c] = ((float*)host.data())[r * m->Mp + c];
      }
    return RSYS_OK;
  }
  if (which == 1) RC(model_finalize_grads(m));
  float* base = (which == 0 ? m->P : m->G) + t.off;
  const int64_t int_rows = (t.map == MAP_DIRECT) ? t.rows : 2 * m->Ip;
  std::vector<float> host((size_t)int_rows * t.ld);
  HIP_CHECK(hipStreamSynchronize(m->stream));   // (the model's streams are non-blocking: a plain hipMemcpy does not wait for them)
  HIP_CHECK(hipStreamSynchronize(m->side));
  HIP_CHECK(hipMemcpy(host.data(), base, host.size() * 4, hipMemcpyDeviceToHost));
  if (out) {
    for (int64_t r = 0; r < t.rows; ++r) memcpy(out + r * t.cols, host.data() + internal_row(t, r) * t.ld, t.cols * 4);
    return RSYS_OK;
  }
  for (int64_t r = 0; r < t.rows; ++r) memcpy(host.data() + internal_row(t, r) * t.ld, in + r * t.cols, t.cols * 4);
  HIP_CHECK(hipMemcpy(base, host.data(), host.size() * 4, hipMemcpyHostToDevice));
  if (which == 0) { m->wt_dirty = true; m->table_dirty = true; m->w8_dirty = true; }
  if (which == 0 && m->bf16_mode)
    RC(launch_cast<bf16>(base, (bf16*)m->Sh + t.off, (int64_t)host.size(), m->stream));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  return RSYS_OK;
}

// Row-sharded table: which rows of which rank this batch reads.  Depends on the batch only (a watch-masked token reads
// the mask row V instead of its item's row, and V is always part of the plan), so it is built once per upload:
// distinct ids (sorted) -> contiguous runs per owner -> counts all-gathered -> id lists exchanged.  Every rank calls this
// at the same point (it contains collectives).
static int build_exchange_plan(Model* m, int N) {
  const int W = m->sh_world;
  hipStream_t s = m->stream;
  ARG_CHECK(W == 1 || m->shard_comm != nullptr, "row-sharded table: call rsys_model_set_shard_comm before the first batch");
  RC(launch_plan_unique(m->tok_skey, m->tok_sidx, N, m->V, m->u_slot, m->u_ids, m->u_tok, m->u_plan, s));
  int* d_off = m->u_off; int* d_cnt = m->u_off + (W + 1); int* d_all = d_cnt + W;
  RC(launch_plan_offsets(m->u_ids, m->u_plan, m->u_bound, W + 1, d_off, s));
  int plan[2]; std::vector<int> off(W + 1), cnt(W), all((size_t)W * W);
  HIP_CHECK(hipMemcpyAsync(plan, m->u_plan, 8, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(off.data(), d_off, (W + 1) * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  m->U = plan[0]; m->uV = plan[1];
  ARG_CHECK(off[W] == m->U && m->U >= 1 && m->U <= N + 1, "exchange plan: inconsistent unique-id list");
  for (int q = 0; q <= W; ++q) { m->need_off[q] = off[q]; m->need_offD[q] = (long long)off[q] * m->D; }
  for (int q = 0; q < W; ++q) cnt[q] = off[q + 1] - off[q];
  HIP_CHECK(hipMemcpyAsync(d_cnt, cnt.data(), W * 4, hipMemcpyHostToDevice, s));
  RC(comm_all_gather(m->shard_comm, d_cnt, d_all, (size_t)W * 4, s));
  HIP_CHECK(hipMemcpyAsync(all.data(), d_all, (size_t)W * W * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  long long acc = 0;
  for (int q = 0; q < W; ++q) { m->serve_off[q] = acc; m->serve_offD[q] = acc * m->D; acc += all[(size_t)q * W + m->sh_rank]; }
  m->serve_off[W] = acc; m->serve_offD[W] = acc * m->D; m->R = acc;
  if (acc > m->req_cap) {
    if (m->req_ids) HIP_CHECK(hipFree(m->req_ids));
    if (m->rows_xchg) HIP_CHECK(hipFree(m->rows_xchg));
    m->req_cap = acc + acc / 4 + 1024;
    HIP_CHECK(hipMalloc((void**)&m->req_ids, (size_t)m->req_cap * 4));
    HIP_CHECK(hipMalloc((void**)&m->rows_xchg, (size_t)m->req_cap * m->D * 4));
  }
  RC(comm_exchange(m->shard_comm, m->u_ids, m->need_off.data(), m->req_ids, m->serve_off.data(), 4, s));
  HIP_CHECK(hipStreamSynchronize(s));
  return RSYS_OK;
}

// Checks a host batch and packs it into staging buffer `slot` (pinned); `d` and the flag outputs receive the device addresses the
// arrays will have in that slot's blob.  Index paths are checked on the host BEFORE anything is written.
struct StagedBatch { BatchDev bd; bool has_masks = false, has_rope_pos = false; unsigned char *d_wm = nullptr, *d_rm = nullptr; int* d_rope_pos = nullptr; size_t bytes = 0; };
static int batch_stage(Model* m, const rsys_batch* b, int slot, StagedBatch& out) {
  ARG_CHECK(b != nullptr, "null batch");
  ARG_CHECK(b->rows >= 1 && b->rows <= m->rows_max, "batch rows must be in [1, max_rows]");
  const size_t N = (size_t)b->rows * m->S;
  ARG_CHECK(b->userid && b->token_mask_ids && b->gender && b->source && b->matchedid && b->status && b->time && b->rating && b->progress,
            "batch arrays must not be null");
  for (int k = 0; k < 6; ++k) {
    if (k % 3 == 2 && b->label[k] == nullptr) continue;  // status targets are never used by the losses (model.py:448-449)
    ARG_CHECK(b->label[k] && b->weight[k] && b->position[k], "target arrays must not be null");
  }
  ARG_CHECK(b->watch_mask == nullptr || b->rating_mask != nullptr, "watch_mask and rating_mask come together");
  // index paths are checked on the host BEFORE anything is copied: a rejected batch leaves the resident one untouched
  for (size_t i = 0; i < N; ++i) {
    int id = b->matchedid[i];
    ARG_CHECK(id >= -1 && id < m->V, "matchedid out of range");
    // the attention kernels compare tokens through the key userid << 12 | token_mask_ids (attention.hip: mask_tile)
    ARG_CHECK(b->userid[i] >= 0 && b->userid[i] < (1 << 19), "userid must be in [0, 2^19)");
    ARG_CHECK(b->token_mask_ids[i] >= 0 && b->token_mask_ids[i] < 4096, "token_mask_ids must be in [0, 4096)");
    ARG_CHECK(b->status[i] >= -1 && b->status[i] <= m->cfg.vocab_status, "status out of range");
    ARG_CHECK(b->gender[i] >= -1 && b->gender[i] <= m->cfg.vocab_gender, "gender out of range");
    ARG_CHECK(b->source[i] >= -1 && b->source[i] <= m->cfg.vocab_source, "source out of range");
    ARG_CHECK(b->position[0][i] >= 0 && b->position[0][i] < m->V0 && b->position[1][i] >= 0 && b->position[1][i] < m->V0,
              "manga target position out of range");
    ARG_CHECK(b->position[3][i] >= 0 && b->position[3][i] < m->V1 && b->position[4][i] >= 0 && b->position[4][i] < m->V1,
              "anime target position out of range");
  }
  std::vector<int> pos;
  if (b->rope_input_pos != nullptr) {
    // model.py:470-476: token positions 2p, 2p+1
    pos.resize(2 * N);
    for (size_t i = 0; i < N; ++i) { pos[2 * i] = 2 * b->rope_input_pos[i]; pos[2 * i + 1] = 2 * b->rope_input_pos[i] + 1; }
    for (size_t i = 0; i < 2 * N; ++i) ARG_CHECK(pos[i] >= 0 && pos[i] < m->T, "rope_input_pos out of range");
  }
  // pack (pinned staging) -> one H2D -> the device arrays sit back to back in the slot's blob
  unsigned char* blob = m->slot_blob[slot]; unsigned char* stage = m->slot_stage[slot];
  BatchDev d = m->bd;    // (the mask_tokens outputs and the sizes are the model's own; only the input arrays move)
  size_t off = 0;
  auto place = [&](const void* src, size_t bytes) -> void* {
    void* dev = blob + off;
    if (src != nullptr) memcpy(stage + off, src, bytes);
    off += (bytes + 63) / 64 * 64;
    return dev;
  };
  d.time = (const double*)place(b->time, N * 8);
  d.userid = (const int*)place(b->userid, N * 4); d.tmid = (const int*)place(b->token_mask_ids, N * 4);
  d.gender = (const int*)place(b->gender, N * 4); d.source = (const int*)place(b->source, N * 4);
  d.matchedid = (const int*)place(b->matchedid, N * 4); d.status = (const int*)place(b->status, N * 4);
  d.rating = (const float*)place(b->rating, N * 4); d.progress = (const float*)place(b->progress, N * 4);
  for (int k = 0; k < 6; ++k) {   // (status targets may be absent: their slots stay unwritten, nothing reads them)
    d.label[k] = (const float*)place(b->label[k], N * 4); d.weight[k] = (const float*)place(b->weight[k], N * 4);
    d.position[k] = (const int*)place(b->position[k], N * 4);
  }
  out.has_masks = b->watch_mask != nullptr;
  out.d_wm = (unsigned char*)place(b->watch_mask, N); out.d_rm = (unsigned char*)place(b->rating_mask, N);
  out.has_rope_pos = b->rope_input_pos != nullptr;
  out.d_rope_pos = (int*)place(out.has_rope_pos ? pos.data() : nullptr, 2 * N * 4);
  if (off > m->raw_bytes) { set_error("batch upload: staging buffer too small"); return RSYS_ERR_STATE; }
  out.bd = d; out.bytes = off;
  return RSYS_OK;
}

int model_batch_upload(Model* m, const rsys_batch* b) {
  HIP_CHECK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  m->pending.valid = false;   // (an explicit upload supersedes a prefetched batch)
  // No wait before packing: the previous upload synchronised after ITS copy, so the staging buffer is free, and the copy below is
  // ordered on the stream behind the kernels that still read the old batch -- the host packs while the GPU finishes the previous
  // step's optimizer pass.  (A prefetch into this slot's staging is complete too: model_batch_swap waited for its copy.)
  StagedBatch st;
  RC(batch_stage(m, b, m->cur_slot, st));
  const size_t N = (size_t)b->rows * m->S;
  m->cur_rows = 0;   // (a failing copy below must not leave a half-written batch marked as resident)
  m->bd = st.bd;
  BatchDev& d = m->bd;
  m->has_masks = st.has_masks; m->d_wm = st.d_wm; m->d_rm = st.d_rm;
  m->has_rope_pos = st.has_rope_pos; m->d_rope_pos = st.d_rope_pos;
  HIP_CHECK(hipMemcpyAsync(m->slot_blob[m->cur_slot], m->slot_stage[m->cur_slot], st.bytes, hipMemcpyHostToDevice, s));
  // inverted index "table row -> its tokens" for the backward's segmented scatter: depends on the batch only; built here for
  // the row-sharded table (its exchange plan needs it now), else by the first backward over this batch (an inference or
  // evaluation pass never needs it)
  m->tok_index_valid = false;
  m->split_plan_valid = false;
  if (m->sharded) { RC(launch_token_index_build(d.matchedid, (int)N, m->V, m->tok_keys, m->tok_skey, m->tok_sidx, s)); m->tok_index_valid = true; }
  HIP_CHECK(hipStreamSynchronize(s));
  if (m->sharded) RC(build_exchange_plan(m, (int)N));
  m->cur_rows = b->rows;
  return RSYS_OK;
}

// The NEXT batch beside the running step: checked and packed into the other slot's staging buffer by the calling thread while the GPU
// works, copied on a stream of its own.  Nothing of the resident batch changes until model_batch_swap.
int model_batch_prefetch(Model* m, const rsys_batch* b) {
  ARG_CHECK(!m->sharded, "batch prefetch: the row-sharded table builds its exchange plan at upload (use rsys_batch_upload)");
  HIP_CHECK(hipSetDevice(m->device));
  const int slot = m->cur_slot ^ 1;
  if (m->pending.valid) HIP_CHECK(hipEventSynchronize(m->ev_copy_done));   // (a prefetch that was never swapped in: its copy still reads the staging buffer)
  m->pending.valid = false;
  StagedBatch st;
  RC(batch_stage(m, b, slot, st));
  // the slot's blob was the resident batch two swaps ago: the kernels that read it were enqueued before ev_blob_free[slot]
  if (m->blob_free_valid[slot]) HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->ev_blob_free[slot], 0));
  HIP_CHECK(hipMemcpyAsync(m->slot_blob[slot], m->slot_stage[slot], st.bytes, hipMemcpyHostToDevice, m->copy_stream));
  HIP_CHECK(hipEventRecord(m->ev_copy_done, m->copy_stream));
  m->pending.valid = true; m->pending.bd = st.bd; m->pending.rows = b->rows;
  m->pending.has_masks = st.has_masks; m->pending.has_rope_pos = st.has_rope_pos;
  m->pending.d_wm = st.d_wm; m->pending.d_rm = st.d_rm; m->pending.d_rope_pos = st.d_rope_pos;
  return RSYS_OK;
}

int model_batch_swap(Model* m) {
  ARG_CHECK(m->pending.valid, "batch swap: no prefetched batch (rsys_batch_prefetch first)");
  HIP_CHECK(hipSetDevice(m->device));
  // everything enqueued so far may still read the resident batch's blob: the next prefetch into it waits for this point
  HIP_CHECK(hipEventRecord(m->ev_blob_free[m->cur_slot], m->stream));
  m->blob_free_valid[m->cur_slot] = true;
  // The copy is waited for on the HOST as well as on the stream: the next prefetch re-packs this slot's staging buffer without a wait
  // (as uploads always have), and by now -- a whole step after the copy was issued -- the wait returns at once.
  HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_copy_done, 0));
  HIP_CHECK(hipEventSynchronize(m->ev_copy_done));
  m->cur_slot ^= 1;
  m->bd = m->pending.bd;
  m->has_masks = m->pending.has_masks; m->d_wm = m->pending.d_wm; m->d_rm = m->pending.d_rm;
  m->has_rope_pos = m->pending.has_rope_pos; m->d_rope_pos = m->pending.d_rope_pos;
  m->cur_rows = m->pending.rows;
  m->tok_index_valid = false;
  m->split_plan_valid = false;
  m->pending.valid = false;
  return RSYS_OK;
}

// ------------------------------------------------------------------ GEMM helper
// K splits of a weight-gradient GEMM on the 128x128 kernel: the smallest multiple of 8 (one split never straddles XCDs)
// that gives every CU two workgroups.  Measured on the trunk's shapes (tools/scan_splitk.py, K = 65536): two co-resident
// workgroups per CU hide each other's latencies, and beyond that every further split only adds its fixed cost (first
// tiles from HBM + 64 KB of atomics) -- dW13 (88 tiles) 8 splits 690 TFLOP/s vs 632 at 32, dW2 (44) 16: 620 vs 569 at 32,
// dWqkv (32) 16: 627 vs 561 at 32, dWo (16) 32: 464.
static int pick_splitk(int M, int N, int K, int bk) {
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int kt = (K + bk - 1) / bk;
  if (kt < 16) return 1;
  long long s = (512 + tiles - 1) / tiles;
  s = (s + 7) / 8 * 8;
  if (s > 128) s = 128;
  while (s > 8 && s * 4 > kt) s -= 8;   // keep at least 4 K tiles per split
  return (int)s;
}

// deterministic mode: the reduction kernels launched inside the scope write partial sums to the model's scratch (kernels.hpp)
struct DetScope {
  DetScratch saved;
  explicit DetScope(Model* m) : saved(g_det) { if (m->deterministic) { g_det.part = m->det_part; g_det.cap = m->det_part_floats; g_det.tmp = m->det_tmp; g_det.tmp_cap = m->det_tmp_floats; } else g_det = DetScratch(); }
  ~DetScope() { g_det = saved; }
};
static int det_slab_for(Model* m, long long need, GemmParams& p) {
  if (need <= 0) return RSYS_OK;
  if (need > m->det_slab_floats) {
    ++m->host_stream_syncs;
    HIP_CHECK(hipStreamSynchronize(m->stream));
    HIP_CHECK(hipStreamSynchronize(m->side));
    if (m->det_slab) HIP_CHECK(hipFree(m->det_slab));
    m->det_slab = nullptr; m->det_slab_floats = 0;
    HIP_CHECK(hipMalloc((void**)&m->det_slab, (size_t)need * 4));
    m->det_slab_floats = need;
  }
  p.slab = m->det_slab; p.slab_floats = m->det_slab_floats;
  return RSYS_OK;
}
int model_set_deterministic(Model* m, int on) {
  HIP_CHECK(hipSetDevice(m->device));
  if (on && m->det_part == nullptr) {
    const long long KB = (long long)m->K * m->rows_max;
    m->det_part_floats = std::max<long long>(std::max<long long>(2048LL * m->D, 512LL * (2 * m->D + 4)), std::max<long long>(KB * m->sh_world, 4096)) + 1024;   // (sharded: one loss term per gathered row)
    HIP_CHECK(hipMalloc((void**)&m->det_part, (size_t)m->det_part_floats * 4));
    m->det_tmp_floats = m->det_part_floats / 32 + 4096;
    HIP_CHECK(hipMalloc((void**)&m->det_tmp, (size_t)m->det_tmp_floats * 4));
  }
  m->deterministic = on != 0;
  return RSYS_OK;
}

template <typename T>
static int gemm(Model* m, const char* tag, GemmParams p, bool a_f32, bool a_km, bool b_km) {
  if (p.alpha == 0.f) p.alpha = 1.f;
  if (p.epi == EPI_ATOMIC && p.splitk == 0)
    p.splitk = pick_splitk(p.M, p.N, (p.k_dev != nullptr && p.k_expect > 0) ? std::min(p.K, p.k_expect) : p.K, is_bf16<T>::value ? 64 : 32);
  if (p.splitk == 0) p.splitk = 1;
  p.flags |= m->gemm_flags;
  if (m->deterministic && p.epi == EPI_ATOMIC) RC(det_slab_for(m, gemm_slab_need<T>(p, a_f32, false, a_km, b_km), p));
  if (m->timer.enabled) tic(m, (std::string(tag) + "@" + gemm_kernel_name(p, is_bf16<T>::value, a_f32, false, a_km, b_km)).c_str(), 2.0 * p.M * p.N * (double)p.K);
  int rc = launch_gemm<T>(p, a_f32, false, a_km, b_km, m->stream);
  toc(m);
  return rc;
}

// Weight-gradient GEMM, optionally on the side stream beside what the main stream launches next: the side stream first waits for
// everything the main stream has enqueued so far (the operands).  RSYS_SIDE_STREAM=1: the main stream joins right behind the
// paired dx GEMM (join_side).  RSYS_SIDE_STREAM=2: deferred joins -- the four products of a layer queue up on the side stream and
// the main stream waits for product `slot` only where the buffer that product reads is about to be overwritten (join_dw), so the
// MFMA-bound weight gradients run beside the HBM-bound RMSNorm backward and the VALU-bound attention backward.  rsys_op_timing(2)
// (bench.py --detail) runs everything in line instead, so that every kernel's time is measured without a neighbour.
enum { DW_W2 = 0, DW_W13 = 1, DW_O = 2, DW_QKV = 3 };
static int side_mode() {
  const int mode = sw().side_stream;
  return mode;
}
template <typename T>
static int gemm_side(Model* m, const char* tag, GemmParams p, bool a_f32, bool a_km, bool b_km, int slot) {
  const int mode = side_mode();
  if (mode == 0 || m->deterministic || (m->timer.enabled && m->timer.serialize)) return gemm<T>(m, tag, p, a_f32, a_km, b_km);   // (deterministic: one slab, one stream)
  if (p.alpha == 0.f) p.alpha = 1.f;
  if (p.epi == EPI_ATOMIC && p.splitk == 0) p.splitk = pick_splitk(p.M, p.N, p.K, is_bf16<T>::value ? 64 : 32);
  if (p.splitk == 0) p.splitk = 1;
  HIP_CHECK(hipEventRecord(m->ev_fork, m->stream));
  HIP_CHECK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
  if (m->timer.enabled) tic(m, (std::string(tag) + "@" + gemm_kernel_name(p, is_bf16<T>::value, a_f32, false, a_km, b_km)).c_str(), 2.0 * p.M * p.N * (double)p.K, m->side);   // (events on the stream the kernel runs on)
  RC(launch_gemm<T>(p, a_f32, false, a_km, b_km, m->side));
  toc(m, m->side);
  if (mode >= 2) {
    HIP_CHECK(hipEventRecord(m->ev_dw[slot], m->side));
    m->dw_pending[slot] = true;
  } else {
    HIP_CHECK(hipEventRecord(m->ev_join, m->side));
    m->side_pending = true;
  }
  return RSYS_OK;
}
// mode 1: wait for the product launched last
static int join_side(Model* m) {
  if (m->side_pending) { HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_join, 0)); m->side_pending = false; }
  return RSYS_OK;
}
// mode 2: the main stream is about to overwrite what product `slot` reads (the side stream runs in order: earlier products are done too)
static int join_dw(Model* m, int slot) {
  if (m->dw_pending[slot]) { HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_dw[slot], 0)); m->dw_pending[slot] = false; }
  return RSYS_OK;
}
static int join_all(Model* m) {
  RC(join_side(m));
  for (int k = 3; k >= 0; --k) RC(join_dw(m, k));
  return RSYS_OK;
}

template <typename T> static inline T* W(Model* m, int64_t off) { return (T*)m->Sh + off; }
template <typename T> static inline T* AT(void* p) { return (T*)p; }
template <typename T> static inline T* WT(Model* m, int64_t off) { return (T*)m->ShT + off; }

// Fused item table F = E + Meta Wp^T + bp over all V + 1 rows (model.py:120-133): f32 copy for the token gather, T copy as the
// tied watch-head operand.  (At cfg-3 this is 782 x 2 tiles of 256 x 256 = 6.1 per CU; sending the rows beyond whole rounds
// to the 128 x 128 kernel in a second launch was measured: 1.32 -> 1.18 + 0.12 ms, not worth the second code path.)
template <typename T>
static int table_forward(Model* m) {
  GemmParams p{};
  p.A = m->Meta; p.lda = m->Mp; p.B = W<T>(m, m->o_Wp); p.ldb = m->Mp; p.C = m->F32; p.ldc = m->D; p.c_f32 = 1;
  p.M = m->TR; p.N = m->D; p.K = m->Mp; p.epi = EPI_TABLE; p.E = m->P + m->o_E; p.bias = m->P + m->o_bp;
  p.C2 = m->FT; p.ldc2 = m->D;
  return gemm<T>(m, "gemm_table_fwd", p, false, false, false);
}

// bf16 mode: the dx GEMMs of the trunk (dX = dY . W, W stored [out][in]) read W^T as a row-major [in][out] operand
static int ensure_transposes(Model* m) {
  if (!m->bf16_mode || !m->wt_dirty) return RSYS_OK;
  const int D = m->D, Ip = m->Ip;
  TransposeBatch b; b.n = 0;
  auto add = [&](int64_t off, int rows, int cols, long long ld) -> int {
    TransposeJob& j = b.job[b.n++];
    j.src = (const bf16*)m->Sh + off; j.dst = (bf16*)m->ShT + off; j.rows = rows; j.cols = cols; j.ld_src = ld; j.ld_dst = rows;
    if (b.n == 64) { int rc = launch_transpose_bf16(b, m->stream); b.n = 0; return rc; }
    return RSYS_OK;
  };
  for (int l = 0; l < m->L; ++l) {
    RC(add(m->lo[l].wqkv, m->Nqkv, D, D));
    RC(add(m->lo[l].wo, D, D, D));
    RC(add(m->lo[l].w13, 2 * Ip, D, D));
    RC(add(m->lo[l].w2, D, Ip, Ip));
  }
  RC(launch_transpose_bf16(b, m->stream));
  m->wt_dirty = false;
  return RSYS_OK;
}

// ------------------------------------------------------------------ fp8 trunk (f8.hip)
// this step's e4m3 weight copies: amax per linear, then the row-major and the transposed copy
static int ensure_f8_weights(Model* m) {
  if (!m->fp8 || !m->w8_dirty) return RSYS_OK;
  tic(m, "f8_weights");
  HIP_CHECK(hipMemsetAsync(m->f8_wamax, 0, (size_t)m->L * 8 * 4, m->stream));
  RC(launch_f8_weights((const F8WeightJob*)m->f8_jobs, m->f8_tile_job, m->f8_tile_first, m->f8_ntiles, m->stream));
  toc(m);
  m->w8_dirty = false;
  return RSYS_OK;
}

// the eight fp8 products of a layer: amax slot(s) of the A operand (Model::f8_aamax), its column layout and format, the weight
// scale slot(s) (Model::f8_wamax: q k v o w1 w3 w2) and how the descales combine
enum { F8P_QKV = 0, F8P_O = 1, F8P_W13 = 2, F8P_W2 = 3, F8P_W2_DX = 4, F8P_W13_DX = 5, F8P_O_DX = 6, F8P_QKV_DX = 7 };
struct F8Op { int a_slot, layout, fmt, w_slot, n_w, desc_mode; };
static const F8Op kF8Ops[8] = {
  {0, F8_LAYOUT_PLAIN, F8_E4M3, 0, 3, 1},    // xn  . [Wq; Wk; Wv]^T
  {1, F8_LAYOUT_PLAIN, F8_E4M3, 3, 1, 1},    // O   . Wo^T
  {2, F8_LAYOUT_PLAIN, F8_E4M3, 4, 2, 1},    // hn  . [W1; W3]^T
  {3, F8_LAYOUT_PLAIN, F8_E4M3, 6, 1, 1},    // g   . W2^T
  {4, F8_LAYOUT_PLAIN, F8_E5M2, 6, 1, 2},    // dy  . W2
  {5, F8_LAYOUT_SWIGLU, F8_E5M2, 4, 2, 2},   // [da | db] . [W1; W3]   (two gradients, two weights: K segments)
  {7, F8_LAYOUT_PLAIN, F8_E5M2, 3, 1, 2},    // dh  . Wo
  {8, F8_LAYOUT_SEGS, F8_E5M2, 0, 3, 2},     // [dq | dk | dv] . [Wq; Wk; Wv]
};

// sharded amax slot `slot` of layer l (common.hpp f8_amax_note): producers add to it, the cast reads it
static inline float* f8_slot(Model* m, int l, int slot) { return m->f8_aamax + (int64_t)l * F8_AMAX_SHARDS * F8_AMAX_SHARD + slot; }
enum { F8S_XN = 0, F8S_O = 1, F8S_HN = 2, F8S_G = 3, F8S_DY2 = 4, F8S_DAB = 5, F8S_DH = 7, F8S_DQKV = 8 };

// One linear of the fp8 trunk.  `p` is the bf16 call (A = the bf16 operand [M][K], epilogue, outputs); the A operand is quantised
// (its amax first unless the producer already left it in the slot), the product runs on the fp8 pipeline with weight copy `w8`.
static int gemm_f8(Model* m, int l, int which, const char* tag, GemmParams p, const unsigned char* w8, long long ldw, bool amax_done = false) {
  const F8Op& o = kF8Ops[which];
  hipStream_t s = m->stream;
  F8Cast c{};
  c.src = p.A; c.ld_src = p.lda; c.rows = p.M; c.cols = p.K; c.rows_dev = p.m_dev; c.fmt = o.fmt; c.layout = o.layout;
  c.seg_cols = o.layout == F8_LAYOUT_SEGS ? m->KV * m->hd : 0; c.seg_rep = m->H / m->KV;   // (dq | dk | dv: units of one kv group)
  c.amax = f8_slot(m, l, o.a_slot); c.dst = m->a8; c.ld_dst = p.K;
  c.desc = m->f8_desc + (l * 8 + which) * 32; c.wamax = m->f8_wamax + l * 8 + o.w_slot; c.n_w = o.n_w; c.desc_mode = o.desc_mode;
  c.w_rep = which == F8P_QKV ? m->H / m->KV : 1;
  const bool tcopy = m->f8_dw && m->f8_tcopies && p.m_dev == nullptr && p.M % 128 == 0;
  if (tcopy) {   // K-contiguous copy for the weight gradient; the gradient operand's cast also writes that product's descales
    Model::F8T& t = m->f8t[l];
    unsigned char* const dst_t[8] = {t.xn, t.O, t.hn, t.g, t.gxt, t.dab, t.dht, t.dqkv};
    c.dst_t = dst_t[which]; c.ld_dst_t = m->f8_ldt;
    if (which >= F8P_W2_DX) {
      static const int x_slot[4] = {F8S_G, F8S_HN, F8S_O, F8S_XN};   // forward operand of w2, w13, o, qkv
      c.desc_dw = m->f8_desc_dw + (l * 4 + (which - F8P_W2_DX)) * 32;
      c.xamax = f8_slot(m, l, x_slot[which - F8P_W2_DX]);
      c.dw_units = which == F8P_W13_DX ? 2 : (which == F8P_QKV_DX ? m->H / m->KV + 2 : 1);
    }
  }
  tic(m, "hbm_f8_cast", ((amax_done ? 3.0 : 5.0) + (tcopy ? 1.0 : 0.0)) * p.M * (double)p.K);
  if (!amax_done) RC(launch_f8_amax(c, s));
  RC(launch_f8_cast(c, s));
  toc(m);
  p.A = m->a8; p.lda = p.K; p.B = w8; p.ldb = ldw;
  p.f8 = o.fmt == F8_E5M2 ? 2 : 1; p.f8_desc = c.desc;
  if (which == F8P_QKV) p.f8_seg_cols = m->KV * m->hd;
  if (which == F8P_W13) p.f8_alt = 1;
  if (which == F8P_W13_DX) p.f8_kb[0] = m->Ip / 128;
  if (which == F8P_QKV_DX) { p.f8_kb[0] = m->H * m->hd / 128; p.f8_kb[1] = (m->H + m->KV) * m->hd / 128; }
  if (p.alpha == 0.f) p.alpha = 1.f;
  p.splitk = 1;
  p.flags |= m->gemm_flags;
  if (m->timer.enabled) tic(m, (std::string(tag) + "@8f").c_str(), 2.0 * p.M * p.N * (double)p.K);
  const int rc = launch_gemm8p_f8(p, s);
  toc(m);
  return rc;
}
// weight gradient of linear `k` (0 w2, 1 w13, 2 o, 3 qkv) of layer l from the transposed fp8 copies: dW += q(dY)^T . q(X), K = tokens
static GemmParams f8_dw_params(Model* m, int l, int k, int NT) {
  const Model::F8T& t = m->f8t[l];
  const int D = m->D, Ip = m->Ip;
  GemmParams p{};
  p.lda = p.ldb = m->f8_ldt; p.K = NT; p.c_f32 = 1; p.epi = EPI_ATOMIC; p.alpha = 1.f; p.f8 = 2;
  p.f8_desc = m->f8_desc_dw + (l * 4 + k) * 32;
  float* G = m->f8_dw_stage ? m->f8_dw_stage - m->f8_dw_stage_base : m->G;   // (staged: f8_dw_round_accum moves it to the gradient)
  switch (k) {
    case 0: p.A = t.gxt; p.B = t.g; p.C = G + m->lo[l].w2; p.ldc = Ip; p.M = D; p.N = Ip; break;
    case 1: p.A = t.dab; p.B = t.hn; p.C = G + m->lo[l].w13; p.ldc = D; p.M = 2 * Ip; p.N = D; p.f8_rseg = Ip; p.f8_rowmode = 1; break;
    case 2: p.A = t.dht; p.B = t.O; p.C = G + m->lo[l].wo; p.ldc = D; p.M = D; p.N = D; break;
    default: p.A = t.dqkv; p.B = t.xn; p.C = G + m->lo[l].wqkv; p.ldc = D; p.M = m->Nqkv; p.N = D; p.f8_rseg = m->KV * m->hd; break;
  }
  return p;
}
// RSYS_F8_DW_ROUND_BF16: gradient[lo, hi) += bf16(staged product sums), stage back to zero (layers l_lo .. l_hi: their four weight
// tensors are contiguous, layers ascending)
static int f8_dw_round_accum(Model* m, int l_lo, int l_hi) {
  if (!m->f8_dw_stage) return RSYS_OK;
  const int64_t lo = m->lo[l_lo].wqkv, hi = m->lo[l_hi].w2 + pad8((int64_t)m->D * m->Ip);
  return launch_round_bf16_accum(m->f8_dw_stage + (lo - m->f8_dw_stage_base), m->G + lo, hi - lo, m->stream);
}
static inline bool use_f8_dw(const Model* m) { return m->f8_dw && !m->deterministic && m->cur_rows * 2 * m->S % 128 == 0; }
// one product at a time (layers whose products are large enough alone: the production shape)
static int f8_dw_launch(Model* m, int l, int k, const char* tag, int NT) {
  GemmParams p = f8_dw_params(m, l, k, NT);
  if (m->timer.enabled) tic(m, (std::string(tag) + "@8fs").c_str(), 2.0 * p.M * p.N * (double)p.K);
  const int rc = launch_gemm8p_f8_splitk(p, m->stream);
  toc(m);
  return rc;
}
static inline const unsigned char* W8(Model* m, int64_t off) { return m->W8 + (off - m->w8_base); }
static inline const unsigned char* W8T(Model* m, int64_t off) { return m->W8T + (off - m->w8_base); }

static SmallParams small_params(Model* m) {
  SmallParams sp;
  sp.per_cos = m->P + m->o_pcos; sp.per_sin = m->P + m->o_psin;
  sp.status_emb = m->P + m->o_status; sp.gender_emb = m->P + m->o_gender; sp.source_emb = m->P + m->o_source;
  sp.n_status = m->cfg.vocab_status; sp.n_gender = m->cfg.vocab_gender; sp.n_source = m->cfg.vocab_source;
  sp.min_ts = m->cfg.min_ts; sp.max_ts = m->cfg.max_ts;
  sp.rating_mean = m->cfg.rating_mean; sp.rating_std = m->cfg.rating_std;
  return sp;
}

static int select_join(Model* m);   // (position selection runs on the side stream: defined with select_positions_all below)

// token-local tail of layer l (model.py:300-309): h = x + O Wo^T ; out = h + W2 (silu(W1 hn) * W3 hn), hn = RMSNorm(h)
template <typename T>
static int layer_tail_dense(Model* m, int l, const void* O_in = nullptr /* attention output in token order (default: the layer's own) */) {
  const int D = m->D, Ip = m->Ip, NT = 2 * m->cur_rows * m->S;
  hipStream_t s = m->stream;
  Model::LayerAct& a = m->la[l];
  float* xnext = (l + 1 < m->L) ? m->la[l + 1].x : m->xL;
  {
    GemmParams p{};
    p.A = O_in ? O_in : a.O; p.lda = D; p.B = W<T>(m, m->lo[l].wo); p.ldb = D; p.C = a.h; p.ldc = D; p.c_f32 = 1;
    p.M = NT; p.N = D; p.K = D; p.epi = EPI_RESIDUAL; p.resid = a.x; p.ldr = D;
    if (m->fp8) RC(gemm_f8(m, l, F8P_O, "gemm_o_fwd", p, W8(m, m->lo[l].wo), D, true));   // (amax |O| came with the attention kernel)
    else RC(gemm<T>(m, "gemm_o_fwd", p, false, false, false));
  }
  tic(m, "hbm_rmsnorm_fwd", (4.0 + sizeof(T)) * D * NT);
  RC(launch_rmsnorm_fwd<T>(a.h, m->P + m->lo[l].mlp, AT<T>(a.hn), a.rstd2, NT, D, s, nullptr, nullptr, m->fp8 ? f8_slot(m, l, F8S_HN) : nullptr));
  toc(m);
  {
    GemmParams p{};
    p.A = a.hn; p.lda = D; p.B = W<T>(m, m->lo[l].w13); p.ldb = D; p.C = a.ab; p.ldc = 2 * Ip;
    p.M = NT; p.N = 2 * Ip; p.K = D; p.epi = EPI_SWIGLU; p.C2 = a.g; p.ldc2 = Ip;
    if (m->fp8) { p.f8_amax_out = f8_slot(m, l, F8S_G); RC(gemm_f8(m, l, F8P_W13, "gemm_w13_fwd", p, W8(m, m->lo[l].w13), D, true)); }
    else RC(gemm<T>(m, "gemm_w13_fwd", p, false, false, false));
  }
  {
    GemmParams p{};
    p.A = a.g; p.lda = Ip; p.B = W<T>(m, m->lo[l].w2); p.ldb = Ip; p.C = xnext; p.ldc = D; p.c_f32 = 1;
    p.M = NT; p.N = D; p.K = Ip; p.epi = EPI_RESIDUAL; p.resid = a.h; p.ldr = D;
    if (m->fp8) RC(gemm_f8(m, l, F8P_W2, "gemm_w2_fwd", p, W8(m, m->lo[l].w2), Ip, true));   // (amax |g| came with the SwiGLU epilogue)
    else RC(gemm<T>(m, "gemm_w2_fwd", p, false, false, false));
  }
  return RSYS_OK;
}

// how many selected tokens to expect (K splits of the compact weight gradients only; the device-side count decides what is computed.
// The same hint for the row-limited GEMMs' kernel choice -- 128 x 128 tiles for ~3 K rows instead of 12 row tiles of 256 -- was
// measured and is not used: w2_dx 37 -> 72 us, w13_dx 57 -> 62 us): pretraining masks 2 * mask_rate of the interactions and a part of them carries a target; finetuning has one target per row
static int expected_selected(const Model* m) {
  const long long N = (long long)m->cur_rows * m->S;
  const long long e = m->cfg.finetune ? 2LL * m->cur_rows : (long long)(2.0 * m->cfg.mask_rate * (double)N * 0.6);
  return (int)std::max<long long>(256, std::min<long long>(m->ctop_cap, e));
}

// The same tail of the LAST layer plus the final norm on the compact set of selected tokens (Model::sparse_top, compact.hip):
// every buffer has ctop_cap rows, the GEMMs stop at the device-side row count.
template <typename T>
static int top_tail_compact(Model* m) {
  const int D = m->D, Ip = m->Ip, l = m->L - 1, cap = m->ctop_cap;
  hipStream_t s = m->stream;
  Model::LayerAct& a = m->la[l];
  const int* n = m->c_n;
  RC(select_join(m));
  tic(m, "phase_top_compact_fwd");
  RC(launch_gather_rows_sel<T>(AT<T>(a.O), D, m->c_sel_p, n, cap, AT<T>(m->c_O), D, s));   // (the layer's attention ran in selected-first order)
  RC(launch_gather_rows_sel<float>(a.x, D, m->c_sel, n, cap, m->c_x, D, s));
  {
    GemmParams p{};
    p.A = m->c_O; p.lda = D; p.B = W<T>(m, m->lo[l].wo); p.ldb = D; p.C = m->c_h; p.ldc = D; p.c_f32 = 1;
    p.M = cap; p.N = D; p.K = D; p.epi = EPI_RESIDUAL; p.resid = m->c_x; p.ldr = D; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_o_fwd", p, false, false, false));
  }
  RC(launch_rmsnorm_fwd<T>(m->c_h, m->P + m->lo[l].mlp, AT<T>(m->c_hn), m->c_rstd2, cap, D, s, n));
  {
    GemmParams p{};
    p.A = m->c_hn; p.lda = D; p.B = W<T>(m, m->lo[l].w13); p.ldb = D; p.C = m->c_ab; p.ldc = 2 * Ip;
    p.M = cap; p.N = 2 * Ip; p.K = D; p.epi = EPI_SWIGLU; p.C2 = m->c_g; p.ldc2 = Ip; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_w13_fwd", p, false, false, false));
  }
  {
    GemmParams p{};
    p.A = m->c_g; p.lda = Ip; p.B = W<T>(m, m->lo[l].w2); p.ldb = Ip; p.C = m->c_xL; p.ldc = D; p.c_f32 = 1;
    p.M = cap; p.N = D; p.K = Ip; p.epi = EPI_RESIDUAL; p.resid = m->c_h; p.ldr = D; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_w2_fwd", p, false, false, false));
  }
  RC(launch_rmsnorm_fwd<T>(m->c_xL, m->P + m->o_norm, AT<T>(m->c_out), m->c_rstdf, cap, D, s, n));
  toc(m);
  return RSYS_OK;
}

// ------------------------------------------------------------------ forward trunk (model.py:464-491, 335-343)
template <typename T>
static int forward_trunk(Model* m) {
  const int D = m->D, Ip = m->Ip, hd = m->hd, rows = m->cur_rows;
  const int N = rows * m->S, NT = 2 * N;
  hipStream_t s = m->stream;
  BatchDev b = m->bd; b.N = N; b.rows = rows; b.S = m->S;
  b.rope_pos = m->has_rope_pos ? m->d_rope_pos : nullptr;
  const int* rpos = b.rope_pos;
  // fused item table F = E + Meta Wp^T + bp
  tic(m, "phase_embed");
  if (m->fp8) {
    RC(ensure_f8_weights(m));
    HIP_CHECK(hipMemsetAsync(m->f8_aamax, 0, (size_t)m->L * F8_AMAX_SHARDS * F8_AMAX_SHARD * 4, s));   // this pass's activation / gradient amax slots
  }
  if (m->table_dirty) { RC(table_forward<T>(m)); m->table_dirty = false; }
  SmallParams sp = small_params(m);
  RC(launch_action_features<T>(b, sp, AT<T>(m->feat), s));
  {
    GemmParams p{};
    p.A = m->feat; p.lda = 32; p.B = W<T>(m, m->o_lin_w); p.ldb = 32; p.C = m->x0 + D; p.ldc = 2 * D; p.c_f32 = 1;
    p.M = N; p.N = D; p.K = 32; p.epi = EPI_BIAS; p.bias = m->P + m->o_lin_b;
    RC(gemm<T>(m, "gemm_action_fwd", p, false, false, false));
  }
  if (m->sharded) {
    // sparse row exchange: every owner sends the rows of F its peers' batches read (plan of the resident batch), then the
    // token gather reads the fetched rows (one per distinct id)
    tic(m, "shard_row_exchange");
    RC(launch_gather_rows_by_id(m->F32, D, m->req_ids, m->row_lo, m->rows_xchg, (int)m->R, D, s));
    RC(comm_exchange(m->shard_comm, m->rows_xchg, m->serve_offD.data(), m->Frem, m->need_offD.data(), 4, s));
    toc(m);
    tic(m, "hbm_gather", 8.0 * D * N);
    RC(launch_gather_items_remote(b, m->Frem, m->u_tok, m->u_plan, D, m->x0, m->uid_t, m->tm_t, s));
    toc(m);
  } else {
    tic(m, "hbm_gather", 8.0 * D * N);   // bytes: one fused-table row read + one embedding row written per interaction
    RC(launch_gather_items(b, m->F32, m->V, D, m->x0, m->uid_t, m->tm_t, s));
    toc(m);
  }
  AttnParams ap{};
  ap.B = rows; ap.T = m->T; ap.H = m->H; ap.KV = m->KV; ap.hd = hd; ap.is_bf16 = is_bf16<T>::value ? 1 : 0;
  ap.uid = m->uid_t; ap.tm = m->tm_t; ap.qmap = m->qmap; ap.kmap = m->kmap; ap.qmap_full = m->qmap_full; ap.kmap_full = m->kmap_full; ap.qmap16 = m->qmap16; ap.kmap16 = m->kmap16;
  ap.maps_zero_base = m->kmap; ap.maps_zero_bytes = m->maps_zero_bytes;
  ap.order_q = m->attn_order_q; ap.order_k = m->attn_order_k; ap.qbits = m->attn_qbits; ap.kbits = m->attn_kbits;
  RC(launch_attn_tilemap(ap, s));
  toc(m);
  AttnParams ap_top = ap;   // the last layer under the compact top: selected-first token order, its own tile maps, leading query tiles only
  if (m->top_is_sparse) {
    RC(select_join(m));
    RC(launch_selected_first(m->c_bits, m->c_pre, m->c_slot, m->c_sel, m->uid_t, m->tm_t, rpos, rows, m->T, m->c_perm, m->uid_p, m->tm_p, m->pos_p, m->c_slot_p,
                             m->c_sel_p, m->c_qact, s));
    ap_top.uid = m->uid_p; ap_top.tm = m->tm_p;
    ap_top.qmap = m->qmap_p; ap_top.kmap = m->kmap_p; ap_top.qmap_full = m->qmap_full_p; ap_top.kmap_full = m->kmap_full_p;
    ap_top.qmap16 = m->qmap16_p; ap_top.kmap16 = m->kmap16_p;
    ap_top.maps_zero_base = m->kmap_p;
    ap_top.order_q = m->attn_order_q_p; ap_top.order_k = m->attn_order_k_p; ap_top.qbits = m->attn_qbits_p; ap_top.kbits = m->attn_kbits_p;
    ap_top.q_active = m->c_qact;   // (the launch orders put the query tiles beyond it last)
    RC(launch_attn_tilemap(ap_top, s));
  }
  tic(m, "phase_trunk_fwd");
  for (int l = 0; l < m->L; ++l) {
    Model::LayerAct& a = m->la[l];
    const bool top = m->top_is_sparse && l == m->L - 1;   // this layer runs in selected-first token order
    const int* rpos_l = top ? m->pos_p : rpos;
    tic(m, "hbm_rmsnorm_fwd", (4.0 + sizeof(T)) * D * NT);
    RC(launch_rmsnorm_fwd<T>(a.x, m->P + m->lo[l].sa, AT<T>(a.xn), a.rstd1, NT, D, s, nullptr, top ? m->c_perm : nullptr, m->fp8 ? f8_slot(m, l, F8S_XN) : nullptr));
    toc(m);
    const bool ft = m->cfg.finetune != 0;
    T* xnd = AT<T>(a.xn);   // LoRA input: dropout(x) in a training pass (model.py:265,269), else x itself
    if (ft) {
      if (m->drop_active) {
        RC(launch_dropout<T>(AT<T>(a.xn), AT<T>(a.xnd), (long long)NT * D, m->cfg.lora_dropout, m->drop_seed,
                             (unsigned int)(m->drop_step * 64 + l), 0, s));
        xnd = AT<T>(a.xnd);
      }
      GemmParams p{};  // La = drop(xn) . [Aq; Av]^T   (NT x 16)
      p.A = xnd; p.lda = D; p.B = W<T>(m, m->lo[l].la); p.ldb = D; p.C = a.La; p.ldc = 16;
      p.M = NT; p.N = 16; p.K = D; p.epi = EPI_STORE;
      RC(gemm<T>(m, "gemm_lora_a_fwd", p, false, false, false));
    }
    {
      GemmParams p{};
      p.A = a.xn; p.lda = D; p.B = W<T>(m, m->lo[l].wqkv); p.ldb = D; p.C = a.qkv; p.ldc = m->Nqkv;
      p.M = NT; p.N = m->Nqkv; p.K = D; p.epi = EPI_QKV_ROPE;
      p.rope_cos = m->rope_cos; p.rope_sin = m->rope_sin; p.rope_cs = m->rope_cs; p.rope_pos = rpos_l; p.T = m->T; p.hd = hd;
      p.n_q = m->H * hd; p.n_k = m->KV * hd;
      if (m->fp8) RC(gemm_f8(m, l, F8P_QKV, "gemm_qkv_fwd", p, W8(m, m->lo[l].wqkv), D, true));
      else RC(gemm<T>(m, "gemm_qkv_fwd", p, false, false, false));
    }
    if (ft) {
      // q += 2 * La[:, :8] Bq^T, v += 2 * La[:, 8:] Bv^T (lora_scaling = 16/8, model.py:236-237,264-271).  RoPE is linear,
      // so the rotated update is accumulated onto the rotated projection.
      GemmParams p{};
      p.A = a.La; p.lda = 16; p.B = W<T>(m, m->lo[l].lb); p.ldb = 16; p.C = a.qkv; p.ldc = m->Nqkv;
      p.M = NT; p.N = m->Nqkv; p.K = 16; p.epi = EPI_QKV_ROPE; p.alpha = 2.f; p.accum = 1;
      p.rope_cos = m->rope_cos; p.rope_sin = m->rope_sin; p.rope_cs = m->rope_cs; p.rope_pos = rpos_l; p.T = m->T; p.hd = hd;
      p.n_q = m->H * hd; p.n_k = m->KV * hd;
      RC(gemm<T>(m, "gemm_lora_b_fwd", p, false, false, false));
    }
    AttnParams& apl = top ? ap_top : ap;
    apl.q = a.qkv; apl.k = AT<T>(a.qkv) + m->H * hd; apl.v = AT<T>(a.qkv) + (m->H + m->KV) * hd; apl.ld = m->Nqkv;
    apl.o = a.O; apl.ldo = D; apl.lse = a.lse;
    apl.f8_amax = m->fp8 ? f8_slot(m, l, F8S_O) : nullptr;
    tic(m, "attn_fwd");
    RC(launch_attn_fwd<T>(apl, s));
    toc(m);
    if (l == m->L - 1 && m->top_is_sparse) break;   // the tail of the last layer and the final norm run on the selected tokens
    RC(layer_tail_dense<T>(m, l));
  }
  if (m->top_is_sparse) { toc(m); return top_tail_compact<T>(m); }
  tic(m, "hbm_rmsnorm_fwd", (4.0 + sizeof(T)) * D * NT);
  RC(launch_rmsnorm_fwd<T>(m->xL, m->P + m->o_norm, AT<T>(m->out), m->rstdf, NT, D, s));
  toc(m);
  toc(m);
  return RSYS_OK;
}

// The dense trunk output of the resident forward (tests, rsys_trunk_output_get): a training pass with the compact top has not
// computed it; the dense tail of the last layer and the final norm run now, from the saved attention output.
template <typename T>
static int materialise_output_t(Model* m) {
  const int D = m->D, NT = 2 * m->cur_rows * m->S, l = m->L - 1, hd = m->hd;
  // the last layer's attention ran in selected-first order over the leading query tiles only: run all of them, then bring the
  // attention output back to token order (into the free dO buffer of the backward) for the dense tail
  Model::LayerAct& a = m->la[l];
  AttnParams ap{};
  ap.B = m->cur_rows; ap.T = m->T; ap.H = m->H; ap.KV = m->KV; ap.hd = hd; ap.is_bf16 = is_bf16<T>::value ? 1 : 0;
  ap.uid = m->uid_p; ap.tm = m->tm_p; ap.qmap = m->qmap_p; ap.kmap = m->kmap_p; ap.qmap_full = m->qmap_full_p; ap.kmap_full = m->kmap_full_p;
  ap.qmap16 = m->qmap16_p; ap.kmap16 = m->kmap16_p; ap.order_q = m->attn_order_q_p; ap.order_k = m->attn_order_k_p; ap.qbits = m->attn_qbits_p; ap.kbits = m->attn_kbits_p;
  ap.q = a.qkv; ap.k = AT<T>(a.qkv) + m->H * hd; ap.v = AT<T>(a.qkv) + (m->H + m->KV) * hd; ap.ld = m->Nqkv;
  ap.o = a.O; ap.ldo = D; ap.lse = a.lse;
  RC(launch_attn_fwd<T>(ap, m->stream));
  RC(launch_scatter_rows_map<T>(AT<T>(a.O), m->c_perm, NT, AT<T>(m->dO), D, D, m->stream));
  RC(layer_tail_dense<T>(m, l, m->dO));
  RC(launch_rmsnorm_fwd<T>(m->xL, m->P + m->o_norm, AT<T>(m->out), m->rstdf, NT, D, m->stream));
  return RSYS_OK;
}
int model_materialise_trunk_output(Model* m) {
  if (!m->top_is_sparse) return RSYS_OK;
  ARG_CHECK(m->cur_rows > 0, "no batch uploaded");
  HIP_CHECK(hipSetDevice(m->device));
  const bool tim = m->timer.enabled; m->timer.enabled = false;
  const int rc = m->bf16_mode ? materialise_output_t<bf16>(m) : materialise_output_t<float>(m);
  m->timer.enabled = tim;
  return rc;
}

// d(trunk output) += the row gradients of head `ti` (rows r < n of `src` belong to positions idx[ti][r]; item tokens parity 0,
// action tokens parity 1): into the dense buffer, or -- compact top -- into the selected tokens' rows through the slot map
static int head_add_rows(Model* m, const float* src, int ti, int parity, int n) {
  if (m->top_is_sparse) return launch_scatter_rows_add_slot(src, m->c_slot, m->idx[ti], parity, m->npos + ti, m->c_gy, n, m->D, m->stream);
  return launch_scatter_rows_add(src, m->idx[ti], parity, m->gy, m->D, n, m->D, m->stream, m->npos + ti);
}

// ------------------------------------------------------------------ sampled soft-max watch head (cfg-4 option)
// Called by watch_head_sharded after the selected rows of all ranks have been gathered and packed.  Per rank: n_s sampled
// local classes (one per stratum, fresh per step and medium, weighted by the stratum's size) instead of all `len`; see shard.hip.
template <typename T>
static int watch_head_sampled(Model* m, int ti, int medium, bool bwd, int nlive, int npad, int own0, int nown, int len, int col0, int lrow, int n_t) {
  const int D = m->D, W = m->sh_world, cap = W * m->K * m->rows_max;
  hipStream_t s = m->stream;
  rsys_comm* c = m->shard_comm;
  const int n_s = std::min(len, m->cfg.sampled_negatives);
  const int n_tot = n_s + n_t;           // sampled classes, then the in-batch targets (listed by watch_head_sharded)
  const int64_t lds = pad8(std::max(n_tot, 8));
  T* Floc = AT<T>(m->FT) + (int64_t)lrow * D;
  HIP_CHECK(hipMemsetAsync(m->ss_tl, 0, (size_t)nlive * 4, s));
  if (n_s > 0) {
    RC(launch_ss_sample(len, n_s, m->cur_seed ^ (0x5A3Dull + 977ull * (unsigned long long)m->sh_rank), (unsigned int)(m->cur_step * 2 + medium), m->ss_cols, s));
    if (n_t > 0) RC(launch_ss_drop_hits(m->ss_cols, n_s, m->ss_bitmap, s));
    RC(launch_gather_rows_plain<T>(Floc, D, m->ss_cols, 0, AT<T>(m->ss_F), n_tot, D, s));
    GemmParams p{};
    p.A = m->EwC; p.lda = D; p.B = m->ss_F; p.ldb = D; p.C = m->logits; p.ldc = lds;
    p.M = cap; p.N = n_tot; p.K = D; p.epi = EPI_STORE; p.m_dev = m->vp_nlive;
    RC(gemm<T>(m, "gemm_logits", p, false, false, false));
    RC(launch_ss_target_logit<T>(AT<T>(m->EwC), Floc, D, len, col0, m->metaC, m->vp_nlive, m->ss_tl, nlive, s));
  }
  tic(m, "ce");
  RC(launch_ss_stats<T>(AT<T>(m->logits), lds, n_s, n_tot, len, col0, m->ss_cols, m->metaC, m->vp_nlive, m->vp_lmax, m->vp_sums, nlive, s));
  RC(comm_all_reduce_f32(c, m->ss_tl, (size_t)nlive, COMM_SUM, s));            // the target's owner has the only non-zero term
  RC(launch_ss_max_with_target(m->vp_lmax, m->ss_tl, m->vp_max, nlive, s));
  RC(comm_all_reduce_f32(c, m->vp_max, (size_t)nlive, COMM_MAX, s));
  RC(launch_ss_rebase(m->vp_lmax, m->vp_max, m->vp_sums, nlive, s));
  RC(comm_all_reduce_f32(c, m->vp_sums, (size_t)nlive, COMM_SUM, s));
  if (n_s > 0)
    RC(launch_ss_finish<T>(AT<T>(m->logits), lds, n_s, n_tot, len, col0, m->ss_cols, m->metaC, m->vp_max, m->vp_sums, m->ss_tl, m->vp_nlive,
                           m->vp_pre, m->sh_rank, m->loss_acc + 3 * ti, m->ss_dt, npad, s));
  else
    RC(launch_ss_finish<T>(AT<T>(m->logits), 8, 0, 0, len, col0, m->ss_cols, m->metaC, m->vp_max, m->vp_sums, m->ss_tl, m->vp_nlive,
                           m->vp_pre, m->sh_rank, m->loss_acc + 3 * ti, m->ss_dt, npad, s));   // a rank without classes of this medium still owns loss rows
  toc(m);
  if (!bwd) return RSYS_OK;
  HIP_CHECK(hipMemsetAsync(m->dEwC, 0, (size_t)nlive * D * 4, s));
  if (n_s > 0) {
    {
      GemmParams p{};  // d(selected rows) = dlogits . F[sampled rows]
      p.A = m->logits; p.lda = lds; p.B = m->ss_F; p.ldb = D; p.C = m->dEwC; p.ldc = D; p.c_f32 = 1;
      p.M = cap; p.N = D; p.K = n_tot; p.epi = EPI_ATOMIC; p.m_dev = m->vp_nlive;
      RC(gemm<T>(m, "gemm_head_dx", p, false, false, true));
    }
    {
      GemmParams p{};  // dF[sampled rows] = dlogits^T . (selected rows of all ranks), then added to the table gradient rows
      p.A = m->logits; p.lda = lds; p.B = m->EwC; p.ldb = D; p.C = m->ss_dF; p.ldc = D; p.c_f32 = 1;
      p.M = n_tot; p.N = D; p.K = cap; p.epi = EPI_STORE; p.k_dev = m->vp_nlive;
      RC(gemm<T>(m, "gemm_head_dw", p, false, true, true));
      RC(launch_add_rows_plain(m->ss_dF, m->ss_cols, lrow, m->G + m->o_E, D, n_tot, D, s));
    }
    RC(launch_ss_target_grad<T>(AT<T>(m->EwC), Floc, D, len, col0, m->metaC, m->ss_dt, m->vp_nlive, m->G + m->o_E + (int64_t)lrow * D, m->dEwC, nlive, s));
  }
  m->gE_clean[medium] = false;
  RC(comm_all_reduce_f32(c, m->dEwC, (size_t)nlive * D, COMM_SUM, s));
  if (nown > 0) RC(head_add_rows(m, m->dEwC + (size_t)own0 * D, ti, 0, nown));
  m->table_grads_pending = true;
  return RSYS_OK;
}

// ------------------------------------------------------------------ sizes of the sharded heads' collectives, ahead of the trunk
// Row meta (target, label * weight, loss coefficient) of the selected positions depends on the masked batch only: per watch task
// it is gathered over the ranks now, the live-row prefix (and the sampled soft-max's number of in-batch targets among this
// rank's classes) is computed with the kernels the heads use later, and copied to pinned host memory behind one event.  Every
// rank calls this at the same point (it contains collectives).
template <typename T>
static int sharded_counts_early(Model* m, bool train, const float tw[4]) {
  const int D = m->D, rows = m->cur_rows, KB = m->K * rows, W = m->sh_world, KBmax = m->K * m->rows_max;
  hipStream_t s = m->stream;
  ARG_CHECK(W + 1 <= 48, "row-sharded table: at most 47 ranks");
  for (int t2 = 0; t2 < 2; ++t2) {
    const int ti = 2 * t2, medium = t2;
    int len, col0, lrow;
    shard_medium_range(m, medium, &len, &col0, &lrow);
    RC(launch_vp_meta(m->idx[ti], m->bd.m_label[ti], m->bd.m_weight[ti], m->bd.m_position[ti], m->stats + 2 * ti, m->npos + ti,
                      train ? tw[ti] : 0.f, KB, KBmax, m->metaOwn, s));
    RC(comm_all_gather(m->shard_comm, m->metaOwn, m->metaAllT[t2], ((size_t)KBmax * 4 + 4) * 4, s));
    // (the row payload EwAll is not there yet: this pass is for the counts and the packed meta only)
    RC(launch_vp_compact<T>(AT<T>(m->EwAll), m->metaAllT[t2], W, KBmax, D, AT<T>(m->EwC), m->metaC, m->vp_nlive, m->vp_pre, s));
    const bool sampled = m->cfg.sampled_negatives > 0 && train;
    const int ss_ns = sampled ? std::min(len, m->cfg.sampled_negatives) : 0;
    const bool ss_targets = sampled && ss_ns > 0 && ss_ns < len;
    int* h = m->h_counts + 64 * t2;
    h[48] = 0;
    if (ss_targets) {
      RC(launch_ss_targets(m->metaC, m->vp_nlive, W * KBmax, len, col0, m->ss_bitmap, m->ss_cols + ss_ns, m->ss_tcount, s));
      HIP_CHECK(hipMemcpyAsync(h + 48, m->ss_tcount, 4, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipMemcpyAsync(h, m->vp_pre, (W + 1) * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_CHECK(hipEventRecord(m->ev_counts, s));
  m->counts_pending = true;
  return RSYS_OK;
}

// ------------------------------------------------------------------ watch head over a row-sharded table (cfg-4)
// Vocabulary-parallel form of model.py:153-170 + 514-519: the selected rows of ALL ranks against this rank's rows of the
// medium.  all-gather (rows, row meta) -> pack the live rows -> local logits -> all-reduce(max) -> all-reduce(sum-exp,
// target logit) -> loss of the own rows, dlogits of every row over the local columns -> dF of the local rows (complete:
// no all-reduce) and the gradient of the selected rows (partial over the vocabulary: all-reduced, own rows scattered).
template <typename T>
static int watch_head_sharded(Model* m, int ti, int medium, bool train, bool bwd, float tw) {
  const int D = m->D, rows = m->cur_rows, KB = m->K * rows, W = m->sh_world, KBmax = m->K * m->rows_max;
  hipStream_t s = m->stream;
  rsys_comm* c = m->shard_comm;
  int len, col0, lrow;
  shard_medium_range(m, medium, &len, &col0, &lrow);
  T* Fm = AT<T>(m->FT) + (int64_t)lrow * D;
  float* st = m->stats + 2 * ti;
  int* np = m->npos + ti;
  // every rank contributes a block of KBmax rows (ranks may hold batches of different row counts: the tail is dead rows); the
  // rows' meta was gathered ahead of the trunk (sharded_counts_early)
  (void)st; (void)np; (void)KB;
  RC(comm_all_gather(c, m->Ew, m->EwAll, (size_t)KBmax * D * m->esz, s));
  RC(launch_vp_compact<T>(AT<T>(m->EwAll), m->metaAllT[medium], W, KBmax, D, AT<T>(m->EwC), m->metaC, m->vp_nlive, m->vp_pre, s));
  // sampled soft-max (training passes only; an evaluation reports the exact loss): list the in-batch targets among this rank's
  // classes behind the slots of the sampled ones -- unless every class is sampled anyway
  const bool sampled = m->cfg.sampled_negatives > 0 && train;
  const int ss_ns = sampled ? std::min(len, m->cfg.sampled_negatives) : 0;
  const bool ss_targets = sampled && ss_ns > 0 && ss_ns < len;
  if (ss_targets) RC(launch_ss_targets(m->metaC, m->vp_nlive, W * KBmax, len, col0, m->ss_bitmap, m->ss_cols + ss_ns, m->ss_tcount, s));
  // the sizes of the collectives below were copied to the host before the trunk forward: the event is long past by now, so this
  // wait does not drain the stream (one wait per step, the second task finds it done)
  if (m->counts_pending) { HIP_CHECK(hipEventSynchronize(m->ev_counts)); m->counts_pending = false; ++m->host_event_waits; }
  const int* hc = m->h_counts + 64 * medium;
  std::vector<int> pre(hc, hc + W + 1);
  const int n_t = ss_targets ? hc[48] : 0;
  const int nlive = pre[W], cap = W * KBmax, own0 = pre[m->sh_rank], nown = pre[m->sh_rank + 1] - own0;
  if (nlive == 0) return RSYS_OK;
  const int npad = std::min(cap, (nlive + 255) & ~255);
  if (sampled)
    return watch_head_sampled<T>(m, ti, medium, bwd, nlive, npad, own0, nown, len, col0, lrow, n_t);
  if (len > 0) {
    GemmParams p{};
    p.A = m->EwC; p.lda = D; p.B = Fm; p.ldb = D; p.C = m->logits; p.ldc = m->ldl_loc;
    p.M = cap; p.N = len; p.K = D; p.epi = EPI_STORE; p.m_dev = m->vp_nlive;
    RC(gemm<T>(m, "gemm_logits", p, false, false, false));
  }
  tic(m, "ce");
  RC(launch_vp_stats<T>(AT<T>(m->logits), m->ldl_loc, len, col0, m->metaC, m->vp_nlive, m->vp_lmax, m->vp_sums, cap, nlive, s));
  HIP_CHECK(hipMemcpyAsync(m->vp_max, m->vp_lmax, (size_t)nlive * 4, hipMemcpyDeviceToDevice, s));
  RC(comm_all_reduce_f32(c, m->vp_max, (size_t)nlive, COMM_MAX, s));
  RC(launch_vp_rebase(m->vp_lmax, m->vp_max, m->vp_sums, nlive, s));
  RC(comm_all_reduce_f32(c, m->vp_sums, (size_t)nlive, COMM_SUM, s));
  RC(comm_all_reduce_f32(c, m->vp_sums + cap, (size_t)nlive, COMM_SUM, s));
  RC(launch_vp_finish<T>(AT<T>(m->logits), m->ldl_loc, len, col0, m->metaC, m->vp_max, m->vp_sums, cap, m->vp_nlive, m->vp_pre,
                         m->sh_rank, m->loss_acc + 3 * ti, npad, s));
  toc(m);
  if (!bwd) return RSYS_OK;   // (the task weights are the same on every rank: all ranks leave here together)
  HIP_CHECK(hipMemsetAsync(m->dEwC, 0, (size_t)nlive * D * 4, s));
  if (len > 0) {
    GemmParams p{};  // d(selected rows) = dlogits . F[local rows]   (partial over the vocabulary)
    p.A = m->logits; p.lda = m->ldl_loc; p.B = Fm; p.ldb = D; p.C = m->dEwC; p.ldc = D; p.c_f32 = 1;
    p.M = cap; p.N = D; p.K = len; p.epi = EPI_ATOMIC; p.m_dev = m->vp_nlive;
    RC(gemm<T>(m, "gemm_head_dx", p, false, false, true));
  }
  RC(comm_all_reduce_f32(c, m->dEwC, (size_t)nlive * D, COMM_SUM, s));
  if (nown > 0) RC(head_add_rows(m, m->dEwC + (size_t)own0 * D, ti, 0, nown));
  if (len > 0) {
    GemmParams p{};  // dF[local rows of the medium] (+)= dlogits^T . (selected rows of all ranks): complete, no all-reduce
    p.A = m->logits; p.lda = m->ldl_loc; p.B = m->EwC; p.ldb = D; p.C = m->G + m->o_E + (int64_t)lrow * D; p.ldc = D; p.c_f32 = 1;
    p.M = len; p.N = D; p.K = cap; p.epi = m->gE_clean[medium] ? EPI_STORE : EPI_ACCUM; p.k_dev = m->vp_nlive;
    m->gE_clean[medium] = false;
    RC(gemm<T>(m, "gemm_head_dw", p, false, true, true));
  }
  m->table_grads_pending = true;
  return RSYS_OK;
}

// ------------------------------------------------------------------ heads, fwd + bwd fused per task (model.py:501-528)
template <typename T>
static int heads(Model* m, int evaluate, const float tw[4]) {
  const int D = m->D, rows = m->cur_rows, N = rows * m->S, NT = 2 * N, KB = m->K * rows;
  hipStream_t s = m->stream;
  const bool train = !evaluate;
  tic(m, "phase_heads");
  const bool ctop = m->top_is_sparse;   // trunk output and its gradient live in the compact buffers (rows = selected tokens)
  RC(select_join(m));
  HIP_CHECK(hipMemsetAsync(m->loss_acc, 0, 16 * 4, s));
  if (train && !ctop) HIP_CHECK(hipMemsetAsync(m->gy, 0, (size_t)NT * D * 4, s));
  if (train && ctop) HIP_CHECK(hipMemsetAsync(m->c_gy, 0, (size_t)m->ctop_cap * D * 4, s));
  auto add_rows = [&](const float* src, int ti, int parity) -> int { return head_add_rows(m, src, ti, parity, KB); };
  for (int ti = 0; ti < 4; ++ti) {
    const int medium = ti >> 1, metric = ti & 1;
    float* st = m->stats + 2 * ti;
    int* np = m->npos + ti;   // positive-weight rows come first: the head GEMMs and the CE kernel stop there
    if (ctop) RC(launch_gather_rows_slot<T>(AT<T>(m->c_out), m->c_slot, m->idx[ti], metric, AT<T>(m->Ew), KB, D, s));
    else RC(launch_gather_rows<T>(AT<T>(m->out), D, m->idx[ti], metric, AT<T>(m->Ew), KB, D, s));
    const bool bwd = train && tw[ti] != 0.f;
    if (metric == 0 && m->sharded) {
      RC(watch_head_sharded<T>(m, ti, medium, train, bwd, tw[ti]));
    } else if (metric == 0) {
      const int vs = medium == 0 ? 0 : m->V0, Vm = medium == 0 ? m->V0 : m->V1;
      T* Fm = AT<T>(m->FT) + (int64_t)vs * D;
      {
        GemmParams p{};
        p.A = m->Ew; p.lda = D; p.B = Fm; p.ldb = D; p.C = m->logits; p.ldc = m->ldl;
        p.M = KB; p.N = Vm; p.K = D; p.epi = EPI_STORE; p.m_dev = np;
        RC(gemm<T>(m, "gemm_logits", p, false, false, false));
      }
      tic(m, "ce");
      RC(launch_ce_fwd_bwd<T>(AT<T>(m->logits), m->ldl, KB, Vm, m->idx[ti], m->bd.m_label[ti], m->bd.m_weight[ti],
                              m->bd.m_position[ti], st, np, train ? tw[ti] : 0.f, m->loss_acc + 3 * ti, s));
      toc(m);
      if (bwd) {
        {
          GemmParams p{};  // dEw = dlogits . F   (few output tiles, K = V_m: split-K over the vocabulary)
          HIP_CHECK(hipMemsetAsync(m->dE, 0, (size_t)KB * D * 4, s));
          p.A = m->logits; p.lda = m->ldl; p.B = Fm; p.ldb = D; p.C = m->dE; p.ldc = D; p.c_f32 = 1;
          p.M = KB; p.N = D; p.K = Vm; p.epi = EPI_ATOMIC; p.m_dev = np;
          RC(gemm<T>(m, "gemm_head_dx", p, false, false, true));
        }
        RC(add_rows(m->dE, ti, 0));
        if (!m->cfg.finetune) {
          GemmParams p{};  // dF[s:e] += dlogits^T . Ew
          p.A = m->logits; p.lda = m->ldl; p.B = m->Ew; p.ldb = D; p.C = m->G + m->o_E + (int64_t)vs * D; p.ldc = D; p.c_f32 = 1;
          p.M = Vm; p.N = D; p.K = KB; p.epi = m->gE_clean[medium] ? EPI_STORE : EPI_ACCUM; p.k_dev = np;
          m->gE_clean[medium] = false;
          RC(gemm<T>(m, "gemm_head_dw", p, false, true, true));
        }
        if (!m->cfg.finetune) m->table_grads_pending = true;
      }
    } else {
      {
        GemmParams p{};
        p.A = m->Ew; p.lda = D; p.B = W<T>(m, m->o_r0w); p.ldb = D; p.C = m->z; p.ldc = D;
        p.M = KB; p.N = D; p.K = D; p.epi = EPI_GELU; p.bias = m->P + m->o_r0b; p.C2 = m->hact; p.ldc2 = D;
        p.m_dev = np;   // (the rating head too stops at the positive-weight rows: zero-weight padding adds nothing to loss or gradients)
        RC(gemm<T>(m, "gemm_rating_fwd", p, false, false, false));
      }
      RC(launch_rating_tail<T>(AT<T>(m->z), AT<T>(m->hact), KB, D, m->P + m->o_r2w, m->P + m->o_r2b, m->idx[ti],
                               m->bd.m_label[ti], m->bd.m_weight[ti], st, m->cfg.rating_mean, bwd ? tw[ti] : 0.f,
                               bwd ? 0 : 1, m->loss_acc + 3 * ti, m->G + m->o_r2w, m->G + m->o_r2b, m->G + m->o_r0b, s, np));
      if (bwd) {
        if (!m->cfg.finetune) {
          GemmParams p{};  // dW0 += dz^T . Er
          p.A = m->z; p.lda = D; p.B = m->Ew; p.ldb = D; p.C = m->G + m->o_r0w; p.ldc = D; p.c_f32 = 1;
          p.M = D; p.N = D; p.K = KB; p.epi = EPI_ATOMIC; p.k_dev = np;   // (dz of the padding rows up to the next tile is zero: rating_tail)
          RC(gemm<T>(m, "gemm_rating_dw", p, false, true, true));
        }
        {
          GemmParams p{};  // dEr = dz . W0
          p.A = m->z; p.lda = D; p.B = W<T>(m, m->o_r0w); p.ldb = D; p.C = m->dE; p.ldc = D; p.c_f32 = 1;
          p.M = KB; p.N = D; p.K = D; p.epi = EPI_STORE; p.m_dev = np;
          RC(gemm<T>(m, "gemm_rating_dx", p, false, false, true));
        }
        RC(add_rows(m->dE, ti, 1));
      }
    }
  }
  toc(m);
  return RSYS_OK;
}

// position selection of the four (medium, metric) tasks in one launch (model.py:501,509): depends on the masked batch only, so it
// runs before the trunk; with the compact top also the union of the live positions
// Both are one-workgroup kernels (~40 us each) that nothing needs before the last layer's tail; running them on the side stream
// beside the fused-table GEMM was measured and is NOT the default (see below; select_join is the matching wait).
static int select_positions_all(Model* m) {
  const int N = m->cur_rows * m->S, KB = m->K * m->cur_rows;
  const float* ws[4]; int* is[4]; float* sts[4]; int* nps[4];
  for (int ti = 0; ti < 4; ++ti) { ws[ti] = m->bd.m_weight[ti]; is[ti] = m->idx[ti]; sts[ti] = m->stats + 2 * ti; nps[ti] = m->npos + ti; }
  // (RSYS_SELECT_ASIDE=1: measured on one box, alternating, 30 steps each: 24.27 / 24.41 / 24.36 ms in line against 24.44 / 24.61 / 24.38 ms
  // aside -- a 1024-thread workgroup landing on a CU stalls that CU's share of the persistent GEMM's tiles: off by default)
  const bool aside_on = sw().select_aside == 1;
  const bool aside = aside_on && !m->sharded && !(m->timer.enabled && m->timer.serialize);   // (sharded: the early counts need them at once)
  hipStream_t s = aside ? m->side : m->stream;
  if (aside) { HIP_CHECK(hipEventRecord(m->ev_fork, m->stream)); HIP_CHECK(hipStreamWaitEvent(m->side, m->ev_fork, 0)); }
  const bool chunked_on = sw().select_chunked != 0;   // (A/B)
  if (chunked_on && N >= 4096) RC(launch_select_positions_chunked(4, ws, N, KB, is, sts, nps, m->sel_scratch, s));
  else RC(launch_select_positions_batch(4, ws, N, KB, is, sts, nps, s));
  if (m->top_is_sparse) RC(launch_token_union(is, nps, 4, 2 * N, m->c_bits, m->c_pre, m->c_n, s));
  if (aside) { HIP_CHECK(hipEventRecord(m->ev_sel, m->side)); m->sel_pending = true; }
  return RSYS_OK;
}
static int select_join(Model* m) {
  if (m->sel_pending) { HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_sel, 0)); m->sel_pending = false; }
  return RSYS_OK;
}

// Backward of top_tail_compact for the last layer: W2 / SwiGLU / W13 / RMSNorm / Wo on the compact rows (gradients of all other
// tokens are identically zero there), then d(attention output) scattered into the zeroed dense buffer the attention backward reads.
// The weight gradients reduce over the compact rows (k_dev); each has one operand whose rows [n, n rounded up to 256) are zero.
template <typename T>
static int top_tail_compact_bwd(Model* m, bool wt) {
  const int D = m->D, Ip = m->Ip, l = m->L - 1, cap = m->ctop_cap, NT = 2 * m->cur_rows * m->S;
  hipStream_t s = m->stream;
  const int* n = m->c_n;
  const bool cp = m->bf16_mode;
  const bool ft = m->cfg.finetune != 0;   // finetune: the base weights are frozen, only the dx chain runs here (the LoRA tensors sit before the attention)
  tic(m, "phase_top_compact_bwd");
  if (!ft) {
    GemmParams p{};  // dW2 += gx^T . g
    p.A = m->c_gx_t; p.lda = D; p.B = m->c_g; p.ldb = Ip; p.C = m->G + m->lo[l].w2; p.ldc = Ip; p.c_f32 = 1;
    p.M = D; p.N = Ip; p.K = cap; p.epi = EPI_ATOMIC; p.k_dev = n;
    p.k_expect = expected_selected(m);
    RC(gemm<T>(m, "gemm_top_w2_dw", p, false, true, true));
  }
  {
    GemmParams p{};  // dg = gx . W2, fused with the SwiGLU backward
    p.A = m->c_gx_t; p.lda = D; p.B = W<T>(m, m->lo[l].w2); p.ldb = Ip; p.C = m->c_dab; p.ldc = 2 * Ip;
    if (wt) { p.B = WT<T>(m, m->lo[l].w2); p.ldb = D; }
    p.M = cap; p.N = Ip; p.K = D; p.epi = EPI_SWIGLU_BWD; p.C2 = m->c_ab; p.ldc2 = 2 * Ip; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_w2_dx", p, false, false, !wt));
  }
  if (!ft) {
    GemmParams p{};  // dW13 += dab^T . hn
    p.A = m->c_dab; p.lda = 2 * Ip; p.B = m->c_hn; p.ldb = D; p.C = m->G + m->lo[l].w13; p.ldc = D; p.c_f32 = 1;
    p.M = 2 * Ip; p.N = D; p.K = cap; p.epi = EPI_ATOMIC; p.k_dev = n;
    p.k_expect = expected_selected(m);
    RC(gemm<T>(m, "gemm_top_w13_dw", p, false, true, true));
  }
  {
    GemmParams p{};  // dhn = dab . W13
    p.A = m->c_dab; p.lda = 2 * Ip; p.B = W<T>(m, m->lo[l].w13); p.ldb = D; p.C = m->c_dhn; p.ldc = D;
    if (wt) { p.B = WT<T>(m, m->lo[l].w13); p.ldb = 2 * Ip; }
    p.M = cap; p.N = D; p.K = 2 * Ip; p.epi = EPI_STORE; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_w13_dx", p, false, false, !wt));
  }
  RC(launch_rmsnorm_bwd<T>(AT<T>(m->c_dhn), m->c_h, m->P + m->lo[l].mlp, m->c_rstd2, m->c_gx, m->c_dh, cp ? AT<T>(m->c_dh_t) : nullptr,
                           m->G + m->lo[l].mlp, cap, D, s, n));
  if (!ft) {
    GemmParams p{};  // dWo += dh^T . O
    p.A = m->c_dh_t; p.lda = D; p.B = m->c_O; p.ldb = D; p.C = m->G + m->lo[l].wo; p.ldc = D; p.c_f32 = 1;
    p.M = D; p.N = D; p.K = cap; p.epi = EPI_ATOMIC; p.k_dev = n;
    p.k_expect = expected_selected(m);
    RC(gemm<T>(m, "gemm_top_o_dw", p, false, true, true));
  }
  {
    GemmParams p{};  // dO = dh . Wo
    p.A = m->c_dh_t; p.lda = D; p.B = W<T>(m, m->lo[l].wo); p.ldb = D; p.C = m->c_dO; p.ldc = D;
    if (wt) p.B = WT<T>(m, m->lo[l].wo);
    p.M = cap; p.N = D; p.K = D; p.epi = EPI_STORE; p.m_dev = n;
    RC(gemm<T>(m, "gemm_top_o_dx", p, false, false, !wt));
  }
  HIP_CHECK(hipMemsetAsync(m->dO, 0, (size_t)NT * D * sizeof(T), s));
  RC(launch_scatter_rows_sel<T>(AT<T>(m->c_dO), m->c_sel_p, n, cap, AT<T>(m->dO), D, D, s));   // (selected-first order, as the attention backward reads it)
  toc(m);
  return RSYS_OK;
}

// The four weight-gradient products of layers [l_lo, l_hi] from the operands the backward kept (Model::dwb), one grouped launch.
template <typename T>
static int grouped_weight_grads(Model* m, int l_lo, int l_hi) {
  const int D = m->D, Ip = m->Ip, NT = 2 * m->cur_rows * m->S;
  const bool top_compact = m->top_is_sparse;   // the last layer's W2 / W13 / Wo products ran on the compact rows already
  const bool f8 = use_f8_dw(m);                 // fp8 trunk: the products read the transposed fp8 copies instead
  const bool ordered = m->deterministic;        // the K splits' partial tiles to per-product slabs, added in index order
  const long long key = ((long long)ordered << 58) | ((long long)f8 << 57) | ((long long)top_compact << 56) | ((long long)l_lo << 40) | ((long long)l_hi << 32) | (unsigned int)m->cur_rows;
  auto it = m->dw_plans.find(key);
  if (it == m->dw_plans.end()) {
    std::vector<GemmParams> ps;
    for (int l = l_hi; l >= l_lo; --l) {
      const Model::LayerAct& a = m->la[l];
      const Model::DwOperands& o = m->dwb[l];
      auto add = [&](const void* A, long long lda, const void* B, long long ldb, float* C, long long ldc, int M, int N) {
        GemmParams p{};
        p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.c_f32 = 1; p.M = M; p.N = N; p.K = NT; p.epi = EPI_ATOMIC; p.alpha = 1.f;
        ps.push_back(p);
      };
      const bool full = !(top_compact && l == m->L - 1);
      if (f8) { for (int k : {1, 0, 3, 2}) ps.push_back(f8_dw_params(m, l, k, NT)); continue; }
      if (full) add(o.dab, 2 * Ip, a.hn, D, m->G + m->lo[l].w13, D, 2 * Ip, D);       // dW13 += dab^T . hn
      if (full) add(o.gxt, D, a.g, Ip, m->G + m->lo[l].w2, Ip, D, Ip);                 // dW2  += gx^T . g
      add(o.dqkv, m->Nqkv, a.xn, D, m->G + m->lo[l].wqkv, D, m->Nqkv, D);              // dWqkv += dqkv^T . xn
      if (full) add(o.dht, D, a.O, D, m->G + m->lo[l].wo, D, D, D);                    // dWo  += dh^T . O
    }
    GemmGroupPlan* pl = nullptr;
    ++m->host_stream_syncs;   // (first use of this layer range / batch size only)
    HIP_CHECK(hipStreamSynchronize(m->stream));
    if (m->dw_plans.size() >= 8) {   // (a plan bakes the row count in: a loader with many distinct last-batch sizes must not grow this without bound)
      for (auto& kv : m->dw_plans) gemm8p_group_plan_destroy(kv.second);
      m->dw_plans.clear();
    }
    RC(gemm8p_group_plan_create(ps.data(), (int)ps.size(), &pl, ordered));
    it = m->dw_plans.emplace(key, pl).first;
  }
  if (m->timer.enabled) tic(m, f8 ? "gemm_dw_group@8gf" : "gemm_dw_group@8g", gemm8p_group_flops(it->second));
  int rc = launch_gemm8p_group(it->second, m->stream);
  toc(m);
  if (rc == RSYS_OK && f8) rc = f8_dw_round_accum(m, l_lo, l_hi);
  return rc;
}

// ------------------------------------------------------------------ backward trunk + embeddings
template <typename T>
static int backward_trunk(Model* m) {
  const int D = m->D, Ip = m->Ip, hd = m->hd, rows = m->cur_rows;
  const int N = rows * m->S, NT = 2 * N;
  hipStream_t s = m->stream;
  const int* rpos = m->has_rope_pos ? m->d_rope_pos : nullptr;
  float* gx = m->gxa;   // gradient w.r.t. the current layer's output (fp32 residual stream)
  float* gx_other = m->gxb;
  T* gxt = AT<T>(m->gxa_t);      // the same gradient as a GEMM operand (T)
  T* gxt_other = AT<T>(m->gxb_t);
  T* dht = AT<T>(m->dh_t);
  const bool cp = m->bf16_mode;  // fp32 mode: the operand IS the fp32 buffer, no copy
  // deferred weight gradients: the dY operands of layer l live in m->dwb[l] until the grouped launch that consumes them
  // (deterministic mode: the grouped launch in its ordered form -- bf16 products only, and the fp8 weight gradients are off in that mode)
  const bool det_group = sw().det_dw_group != 0;   // A/B: 0 = the per-layer slab path
  const bool defer = m->defer_dw && (!m->deterministic || (det_group && m->bf16_mode)) && side_mode() == 0;
  const bool ctop = m->top_is_sparse;
  if (defer && !ctop) gxt = AT<T>(m->dwb[m->L - 1].gxt);
  RC(ensure_transposes(m));
  const bool wt = m->bf16_mode;   // dx GEMMs: row-major W^T (bf16 mode) or the K-major read of W itself (fp32 parity mode)
  tic(m, "phase_trunk_bwd");
  const double nb_bytes = (sizeof(T) + 4.0 + 4.0 + 4.0 + (cp ? 2.0 : 0.0)) * D * NT;   // g, x, residual gradient in; dx (+ its bf16 operand copy) out
  if (ctop) {   // final norm on the compact rows: c_gx = d(last layer's output) at the selected tokens, zero everywhere else
    RC(launch_rmsnorm_bwd_f32<T>(m->c_gy, m->c_xL, m->P + m->o_norm, m->c_rstdf, nullptr, m->c_gx, cp ? AT<T>(m->c_gx_t) : nullptr,
                                 m->G + m->o_norm, m->ctop_cap, D, s, m->c_n));
  } else {
    tic(m, "hbm_rmsnorm_bwd", (4.0 + 4.0 + 4.0 + (cp ? 2.0 : 0.0)) * D * NT);
    RC(launch_rmsnorm_bwd_f32<T>(m->gy, m->xL, m->P + m->o_norm, m->rstdf, nullptr, gx, cp ? gxt : nullptr, m->G + m->o_norm, NT, D, s, nullptr,
                                 m->fp8 ? f8_slot(m, m->L - 1, F8S_DY2) : nullptr));   // (fp8: the top layer's W2 takes this gradient as its dy)
    toc(m);
  }
  AttnParams ap{};
  ap.B = rows; ap.T = m->T; ap.H = m->H; ap.KV = m->KV; ap.hd = hd; ap.is_bf16 = is_bf16<T>::value ? 1 : 0;
  ap.uid = m->uid_t; ap.tm = m->tm_t; ap.qmap = m->qmap; ap.kmap = m->kmap; ap.qmap_full = m->qmap_full; ap.kmap_full = m->kmap_full; ap.qmap16 = m->qmap16; ap.kmap16 = m->kmap16;
  ap.order_q = m->attn_order_q; ap.order_k = m->attn_order_k; ap.qbits = m->attn_qbits; ap.kbits = m->attn_kbits;
  ap.rope_cos = m->rope_cos; ap.rope_sin = m->rope_sin; ap.rope_pos = rpos;
  int bucket_top = m->L - 1;
  for (int l = m->L - 1; l >= 0; --l) {
    Model::LayerAct& a = m->la[l];
    const bool ft = m->cfg.finetune != 0;   // finetune: base weights are frozen, only the dx chain and the LoRA grads run
    const bool f8dw = use_f8_dw(m);         // fp8 weight gradients: launched behind the cast of their gradient operand (inside the dx product)
    const bool top = ctop && l == m->L - 1;   // this layer's token-local part runs on the compact rows
    void* const dab = (defer && !top) ? m->dwb[l].dab : m->dab;
    void* const dqkv = defer ? m->dwb[l].dqkv : m->dqkv;
    if (defer) { if (!top) dht = AT<T>(m->dwb[l].dht); gxt_other = l > 0 ? AT<T>(m->dwb[l - 1].gxt) : AT<T>(m->gxa_t); }
    if (top) RC(top_tail_compact_bwd<T>(m, wt));
    if (!top) {
    if (!ft && !defer && !f8dw) {
      GemmParams p{};  // dW2 += gx^T . g
      p.A = gxt; p.lda = D; p.B = a.g; p.ldb = Ip; p.C = m->G + m->lo[l].w2; p.ldc = Ip; p.c_f32 = 1;
      p.M = D; p.N = Ip; p.K = NT; p.epi = EPI_ATOMIC;
      RC(gemm_side<T>(m, "gemm_w2_dw", p, false, true, true, DW_W2));
    }
    {
      GemmParams p{};  // dg = gx . W2, fused with the SwiGLU backward: writes [da|db] directly
      p.A = gxt; p.lda = D; p.B = W<T>(m, m->lo[l].w2); p.ldb = Ip; p.C = dab; p.ldc = 2 * Ip;
      if (wt) { p.B = WT<T>(m, m->lo[l].w2); p.ldb = D; }
      p.M = NT; p.N = Ip; p.K = D; p.epi = EPI_SWIGLU_BWD; p.C2 = a.ab; p.ldc2 = 2 * Ip;
      RC(join_dw(m, DW_W13));   // the layer above's dW13 reads dab
      if (m->fp8) { p.f8_amax_out = f8_slot(m, l, F8S_DAB); RC(gemm_f8(m, l, F8P_W2_DX, "gemm_w2_dx", p, W8T(m, m->lo[l].w2), D, true)); }
      else RC(gemm<T>(m, "gemm_w2_dx", p, false, false, !wt));
      if (f8dw && !defer) RC(f8_dw_launch(m, l, 0, "gemm_w2_dw", NT));
      RC(join_side(m));
    }
    if (!ft && !defer && !f8dw) {
      GemmParams p{};  // dW13 += dab^T . hn
      p.A = dab; p.lda = 2 * Ip; p.B = a.hn; p.ldb = D; p.C = m->G + m->lo[l].w13; p.ldc = D; p.c_f32 = 1;
      p.M = 2 * Ip; p.N = D; p.K = NT; p.epi = EPI_ATOMIC;
      RC(gemm_side<T>(m, "gemm_w13_dw", p, false, true, true, DW_W13));
    }
    {
      GemmParams p{};  // dhn = dab . W13
      p.A = dab; p.lda = 2 * Ip; p.B = W<T>(m, m->lo[l].w13); p.ldb = D; p.C = m->dhn; p.ldc = D;
      if (wt) { p.B = WT<T>(m, m->lo[l].w13); p.ldb = 2 * Ip; }
      p.M = NT; p.N = D; p.K = 2 * Ip; p.epi = EPI_STORE;
      if (m->fp8) RC(gemm_f8(m, l, F8P_W13_DX, "gemm_w13_dx", p, W8T(m, m->lo[l].w13), 2 * Ip, true));   // (amax |da|, |db| came with the SwiGLU-backward epilogue)
      else RC(gemm<T>(m, "gemm_w13_dx", p, false, false, !wt));
      if (f8dw && !defer) RC(f8_dw_launch(m, l, 1, "gemm_w13_dw", NT));
      if (!m->f8_keep.empty()) HIP_CHECK(hipMemcpyAsync(m->f8_keep[l * 3 + 0], m->dhn, (size_t)NT * D * 2, hipMemcpyDeviceToDevice, s));
      RC(join_side(m));
    }
    RC(join_dw(m, DW_O));       // the layer above's dWo reads dht
    tic(m, "hbm_rmsnorm_bwd", nb_bytes);
    RC(launch_rmsnorm_bwd<T>(AT<T>(m->dhn), a.h, m->P + m->lo[l].mlp, a.rstd2, gx, m->dh, cp ? dht : nullptr, m->G + m->lo[l].mlp, NT, D, s, nullptr, nullptr, nullptr,
                             m->fp8 ? f8_slot(m, l, F8S_DH) : nullptr));
    toc(m);
    if (!ft && !defer && !f8dw) {
      GemmParams p{};  // dWo += dh^T . O
      p.A = dht; p.lda = D; p.B = a.O; p.ldb = D; p.C = m->G + m->lo[l].wo; p.ldc = D; p.c_f32 = 1;
      p.M = D; p.N = D; p.K = NT; p.epi = EPI_ATOMIC;
      RC(gemm_side<T>(m, "gemm_o_dw", p, false, true, true, DW_O));
    }
    {
      GemmParams p{};  // dO = dh . Wo
      p.A = dht; p.lda = D; p.B = W<T>(m, m->lo[l].wo); p.ldb = D; p.C = m->dO; p.ldc = D;
      if (wt) p.B = WT<T>(m, m->lo[l].wo);
      p.M = NT; p.N = D; p.K = D; p.epi = EPI_STORE;
      if (m->fp8) RC(gemm_f8(m, l, F8P_O_DX, "gemm_o_dx", p, W8T(m, m->lo[l].wo), D, true));
      else RC(gemm<T>(m, "gemm_o_dx", p, false, false, !wt));
      if (f8dw && !defer) RC(f8_dw_launch(m, l, 2, "gemm_o_dw", NT));
      if (!m->f8_keep.empty()) HIP_CHECK(hipMemcpyAsync(m->f8_keep[l * 3 + 1], m->dO, (size_t)NT * D * 2, hipMemcpyDeviceToDevice, s));
      RC(join_side(m));
    }
    }   // !top
    ap.q = a.qkv; ap.k = AT<T>(a.qkv) + m->H * hd; ap.v = AT<T>(a.qkv) + (m->H + m->KV) * hd; ap.ld = m->Nqkv;
    ap.o = a.O; ap.ldo = D; ap.lse = a.lse;
    ap.dO = m->dO; ap.delta = m->delta;
    ap.dq = dqkv; ap.dk = AT<T>(dqkv) + m->H * hd; ap.dv = AT<T>(dqkv) + (m->H + m->KV) * hd; ap.ldg = m->Nqkv;
    RC(join_dw(m, DW_QKV));     // the layer above's dWqkv reads dqkv
    ap.f8_amax = m->fp8 ? f8_slot(m, l, F8S_DQKV) : nullptr;
    tic(m, "attn_bwd");
    if (top) {   // selected-first token order of the last layer: its ids, tile maps, RoPE positions; dO is non-zero in the leading query tiles only
      AttnParams at = ap;
      at.uid = m->uid_p; at.tm = m->tm_p; at.rope_pos = m->pos_p; at.q_active = m->c_qact;
      at.qmap = m->qmap_p; at.kmap = m->kmap_p; at.qmap_full = m->qmap_full_p; at.kmap_full = m->kmap_full_p; at.qmap16 = m->qmap16_p; at.kmap16 = m->kmap16_p;
      at.order_q = m->attn_order_q_p; at.order_k = m->attn_order_k_p; at.qbits = m->attn_qbits_p; at.kbits = m->attn_kbits_p;
      RC(launch_attn_bwd<T>(at, s));
    } else {
      RC(launch_attn_bwd<T>(ap, s));
    }
    toc(m);
    if (!ft && !defer && !f8dw) {
      GemmParams p{};  // dWqkv += dqkv^T . xn
      p.A = dqkv; p.lda = m->Nqkv; p.B = a.xn; p.ldb = D; p.C = m->G + m->lo[l].wqkv; p.ldc = D; p.c_f32 = 1;
      p.M = m->Nqkv; p.N = D; p.K = NT; p.epi = EPI_ATOMIC;
      RC(gemm_side<T>(m, "gemm_qkv_dw", p, false, true, true, DW_QKV));
    }
    {
      GemmParams p{};  // dxn = dqkv . Wqkv
      p.A = dqkv; p.lda = m->Nqkv; p.B = W<T>(m, m->lo[l].wqkv); p.ldb = D; p.C = m->dhn; p.ldc = D;
      if (wt) { p.B = WT<T>(m, m->lo[l].wqkv); p.ldb = m->Nqkv; }
      p.M = NT; p.N = D; p.K = m->Nqkv; p.epi = EPI_STORE;
      if (m->fp8) RC(gemm_f8(m, l, F8P_QKV_DX, "gemm_qkv_dx", p, W8T(m, m->lo[l].wqkv), m->Nqkv, true));
      else RC(gemm<T>(m, "gemm_qkv_dx", p, false, false, !wt));
      if (f8dw && !defer) RC(f8_dw_launch(m, l, 3, "gemm_qkv_dw", NT));
      if (!m->f8_keep.empty()) HIP_CHECK(hipMemcpyAsync(m->f8_keep[l * 3 + 2], m->dhn, (size_t)NT * D * 2, hipMemcpyDeviceToDevice, s));
      RC(join_side(m));
    }
    if (ft) {
      T* xnd = m->drop_active ? AT<T>(a.xnd) : AT<T>(a.xn);
      const int nq = m->H * hd, nv0 = (m->H + m->KV) * hd, nkv = m->KV * hd;
      {
        GemmParams p{};  // dLa = 2 * dqkv . Bcat    (the unused blocks of Bcat are zero)
        p.A = dqkv; p.lda = m->Nqkv; p.B = W<T>(m, m->lo[l].lb); p.ldb = 16; p.C = m->dLa; p.ldc = 16;
        p.M = NT; p.N = 16; p.K = m->Nqkv; p.epi = EPI_STORE; p.alpha = 2.f;
        RC(gemm<T>(m, "gemm_lora_dla", p, false, false, true));
      }
      {
        GemmParams p{};  // dBq += 2 * dq^T . La[:, :8]
        p.A = dqkv; p.lda = m->Nqkv; p.B = a.La; p.ldb = 16; p.C = m->G + m->lo[l].lb; p.ldc = 16; p.c_f32 = 1;
        p.M = nq; p.N = 8; p.K = NT; p.epi = EPI_ATOMIC; p.alpha = 2.f;
        RC(gemm<T>(m, "gemm_lora_db", p, false, true, true));
      }
      {
        GemmParams p{};  // dBv += 2 * dv^T . La[:, 8:]
        p.A = AT<T>(dqkv) + nv0; p.lda = m->Nqkv; p.B = AT<T>(a.La) + 8; p.ldb = 16;
        p.C = m->G + m->lo[l].lb + (int64_t)nv0 * 16 + 8; p.ldc = 16; p.c_f32 = 1;
        p.M = nkv; p.N = 8; p.K = NT; p.epi = EPI_ATOMIC; p.alpha = 2.f;
        RC(gemm<T>(m, "gemm_lora_db", p, false, true, true));
      }
      {
        GemmParams p{};  // d[Aq; Av] += dLa^T . drop(xn)
        p.A = m->dLa; p.lda = 16; p.B = xnd; p.ldb = D; p.C = m->G + m->lo[l].la; p.ldc = D; p.c_f32 = 1;
        p.M = 16; p.N = D; p.K = NT; p.epi = EPI_ATOMIC;
        RC(gemm<T>(m, "gemm_lora_da", p, false, true, true));
      }
      {
        GemmParams p{};  // dxn += dropout'(dLa . [Aq; Av])
        p.A = m->dLa; p.lda = 16; p.B = W<T>(m, m->lo[l].la); p.ldb = D; p.ldc = D;
        p.M = NT; p.N = D; p.K = 16; p.epi = EPI_STORE;
        if (m->drop_active) {
          p.C = m->dxl;
          RC(gemm<T>(m, "gemm_lora_dx", p, false, false, true));
          RC(launch_dropout<T>(AT<T>(m->dxl), AT<T>(m->dhn), (long long)NT * D, m->cfg.lora_dropout, m->drop_seed,
                               (unsigned int)(m->drop_step * 64 + l), 1, s));
        } else {
          p.C = m->dhn; p.accum = 1;
          RC(gemm<T>(m, "gemm_lora_dx", p, false, false, true));
        }
      }
    }
    tic(m, "hbm_rmsnorm_bwd", nb_bytes);
    if (top)   // the residual gradient dh exists at the selected tokens only (compact rows, through the token -> row map)
      // ... and this layer's rows are in selected-first order: x is read at, and dx written to, the original token of each place
      RC(launch_rmsnorm_bwd<T>(AT<T>(m->dhn), a.x, m->P + m->lo[l].sa, a.rstd1, m->c_dh, gx_other, cp ? gxt_other : nullptr, m->G + m->lo[l].sa, NT, D, s,
                               nullptr, m->c_slot_p, m->c_perm));
    else {
      RC(launch_rmsnorm_bwd<T>(AT<T>(m->dhn), a.x, m->P + m->lo[l].sa, a.rstd1, m->dh, gx_other, cp ? gxt_other : nullptr, m->G + m->lo[l].sa, NT, D, s, nullptr, nullptr, nullptr,
                               (m->fp8 && l > 0) ? f8_slot(m, l - 1, F8S_DY2) : nullptr));   // (fp8: the layer below takes this gradient as its W2's dy)
    }
    toc(m);
    std::swap(gx, gx_other);
    std::swap(gxt, gxt_other);
    if (f8dw && !defer && !ft) RC(f8_dw_round_accum(m, l, l));   // (the layer's four products were launched one by one above)
    if (defer && !m->grad_bucket_hook && l == 0) RC(grouped_weight_grads<T>(m, 0, m->L - 1));   // all layers' products in one launch
    if (m->grad_bucket_hook && !ft) {
      // weight gradients of layers l .. bucket_top are final (the four tensors of a layer are contiguous, layers ascending)
      const int64_t lo = m->lo[l].wqkv, hi = m->lo[bucket_top].w2 + pad8((int64_t)D * Ip);
      // DDP's 25 MB buckets; with the grouped weight gradients a bucket is also a launch, and a grouped launch wants several
      // layers' products to fill the chip: two buckets (upper and lower half of the trunk) as long as each has its 25 MB
      const bool boundary = defer ? (l == m->L / 2 && (hi - lo) * 4 >= (25ll << 20)) : (hi - lo) * 4 >= (25ll << 20);
      if (l == 0 || boundary) {
        if (defer) RC(grouped_weight_grads<T>(m, l, bucket_top));   // the bucket's products, then its all-reduce
        RC(join_all(m));
        RC(m->grad_bucket_hook(lo, hi));
        bucket_top = l - 1;
        // From here on all-reduce kernels share the CUs with the backward.  A persistent grid (one workgroup pinned per
        // CU, a fixed share of the tiles each) would stall on every CU a communication kernel holds, so the 256x256
        // GEMMs go back to one workgroup per tile until the reduction is over: the tiles flow to whatever CUs are free.
        m->gemm_flags |= 2;
      }
    }
  }
  RC(join_all(m));   // every weight gradient is final (and the saved activations may be overwritten by the next forward)
  toc(m);
  if (m->cfg.finetune) return RSYS_OK;   // embeddings are frozen (model.py:361-369)
  // gx = gradient w.r.t. the interleaved input embeddings (even rows: items, odd rows: actions)
  tic(m, "phase_embed_bwd");
  BatchDev b = m->bd; b.N = N; b.rows = rows; b.S = m->S;
  tic(m, "hbm_scatter", 12.0 * D * N);   // bytes: one gradient row read + one table-gradient row read-modify-written per interaction
  {
    const bool atomic_ab = sw().scatter_atomic != 0;   // A/B measurement against the float-atomic form only
    if (m->sharded) {
      // one gradient row per distinct id of the batch (keys = the ids' slots in the exchange plan, mask row = slot uV), sent
      // to the rows' owners; an owner adds what it receives requester by requester (the ids of one requester are distinct)
      HIP_CHECK(hipMemsetAsync(m->Frem, 0, (size_t)m->U * D * 4, s));
      RC(launch_embedding_scatter_segmented(gx, 2LL * D, b.m_matchedid, m->u_slot, m->tok_sidx, N, m->uV, D, m->Frem, m->scatter_slab, s));
      RC(comm_exchange(m->shard_comm, m->Frem, m->need_offD.data(), m->rows_xchg, m->serve_offD.data(), 4, s));
      for (int q = 0; q < m->sh_world; ++q) {
        const long long o = m->serve_off[q], n = m->serve_off[q + 1] - o;
        RC(launch_add_rows_by_id(m->rows_xchg + o * D, m->req_ids + o, m->row_lo, m->G + m->o_E, D, (int)n, D, s));
      }
    } else if (atomic_ab) RC(launch_embedding_scatter_add(gx, b, m->V, D, m->G + m->o_E, s));
    else {
      if (!m->tok_index_valid) {   // first backward over this batch: tokens sorted by (item id, position)
        RC(launch_token_index_build(b.matchedid, N, m->V, m->tok_keys, m->tok_skey, m->tok_sidx, s));
        m->tok_index_valid = true;
      }
      if (m->split_head_reduced) {
        // split table reduce: one row per distinct id of the batch in tok_T (the sharded path's compact scatter: keys = ranks of the
        // sorted ids, mask row = slot uV), added to G[E] from there -- the same sums in the same order as the direct scatter
        if (!m->split_plan_valid) {   // once per resident batch (one host sync: the launcher needs U and uV)
          RC(launch_plan_unique(m->tok_skey, m->tok_sidx, N, m->V, m->u_slot, m->u_ids, m->u_tok, m->u_plan, s));
          int plan[2];
          HIP_CHECK(hipMemcpyAsync(plan, m->u_plan, 8, hipMemcpyDeviceToHost, s));
          HIP_CHECK(hipStreamSynchronize(s));
          ARG_CHECK(plan[0] >= 1 && plan[0] <= N + 1 && plan[0] <= m->tok_cap && plan[1] >= 0 && plan[1] < plan[0], "split table reduce: inconsistent list of distinct ids");
          m->U = plan[0]; m->uV = plan[1];
          m->split_plan_valid = true;
        }
        HIP_CHECK(hipMemsetAsync(m->tok_T, 0, (size_t)m->U * D * 4, s));
        RC(launch_embedding_scatter_segmented(gx, 2LL * D, b.m_matchedid, m->u_slot, m->tok_sidx, N, m->uV, D, m->tok_T, m->scatter_slab, s));
        // G[E] is still the SEND buffer of the head part's out-of-place all-reduce on the communicator's stream (capi.hip
        // table_head_hook): the first write to it since then waits for that collective to have read it
        if (m->split_head_event) HIP_CHECK(hipStreamWaitEvent(s, m->split_head_event, 0));
        RC(launch_add_rows_by_id(m->tok_T, m->u_ids, 0, m->G + m->o_E, D, m->U, D, s));
      } else
      RC(launch_embedding_scatter_segmented(gx, 2LL * D, b.m_matchedid, m->tok_skey, m->tok_sidx, N, m->V, D, m->G + m->o_E, m->scatter_slab, s));
    }
  }
  toc(m);
  m->table_grads_pending = true;
  m->gE_clean[0] = m->gE_clean[1] = false;   // the rows now hold token gradients: a later head GEMM must add, not store
  {
    GemmParams p{};  // dWlin += g_act^T . feat
    p.A = gxt + D; p.lda = 2 * D; p.B = m->feat; p.ldb = 32; p.C = m->G + m->o_lin_w; p.ldc = 32; p.c_f32 = 1;
    p.M = D; p.N = 32; p.K = N; p.epi = EPI_ATOMIC;
    RC(gemm<T>(m, "gemm_action_dw", p, false, true, true));
  }
  RC(launch_colsum_add(gx + D, 2 * D, N, D, m->G + m->o_lin_b, s));
  {
    GemmParams p{};  // gf = g_act . Wlin
    p.A = gxt + D; p.lda = 2 * D; p.B = W<T>(m, m->o_lin_w); p.ldb = 32; p.C = m->gf; p.ldc = 32; p.c_f32 = 1;
    p.M = N; p.N = 32; p.K = D; p.epi = EPI_STORE;
    RC(gemm<T>(m, "gemm_action_dx", p, false, false, true));
  }
  SmallParams sp = small_params(m);
  RC(launch_action_small_bwd(m->gf, b, sp, m->G + m->o_pcos, m->G + m->o_psin, m->G + m->o_status, m->G + m->o_gender,
                             m->G + m->o_source, s));
  toc(m);
  return RSYS_OK;
}

// dWp = dF^T Meta and dbp = colsum(dF), once per optimizer step from the accumulated dF (= grad of E).
// stage 1: operand copy of dF + bias gradient (afterwards nothing reads G[E] any more: in bf16 mode the GEMM works on the
// copy, so the gradient all-reduce of the item table can run beside it); stage 2: the GEMM; stage 0: both.
template <typename T>
static int finalize_grads_t(Model* m, int stage) {
  tic(m, stage == 2 ? "phase_table_bwd_gemm" : "phase_table_bwd");
  const bool direct = m->bf16_mode && m->D % 64 == 0;   // fp32 dF -> bf16 dF^T + bias gradient in one pass
  if (stage != 2 && direct) {
    RC(launch_cast_transpose_colsum(m->G + m->o_E, (bf16*)m->dFT, m->TR, m->D, m->Vp, m->G + m->o_bp, m->stream));
  } else if (stage != 2) {
    m->table_dirty = true;   // (FT is borrowed below)
    const bool fused = m->bf16_mode && m->D <= 1024 && 1024 % (m->D >> 2) == 0;
    if (fused) {   // operand copy of dF in the fused-table buffer (dead until the next forward rebuilds it) + bias gradient, one pass
      RC(launch_cast_colsum(m->G + m->o_E, (bf16*)m->FT, m->TR, m->D, m->G + m->o_bp, m->stream));
    } else {
      if (m->bf16_mode) RC(launch_cast<bf16>(m->G + m->o_E, (bf16*)m->FT, (long long)m->TR * m->D, m->stream));
      RC(launch_colsum_add(m->G + m->o_E, m->D, m->TR, m->D, m->G + m->o_bp, m->stream));
    }
    if (m->bf16_mode) {   // K-contiguous copy of dF for the row-major pipeline: dFT[d][v]
      TransposeBatch b; b.n = 1;
      b.job[0].src = (const bf16*)m->FT; b.job[0].dst = (bf16*)m->dFT; b.job[0].rows = m->TR; b.job[0].cols = m->D;
      b.job[0].ld_src = m->D; b.job[0].ld_dst = m->Vp;
      RC(launch_transpose_bf16(b, m->stream));
    }
  }
  if (stage != 1) {
    GemmParams p{};
    p.C = m->G + m->o_Wp; p.ldc = m->Mp; p.c_f32 = 1; p.M = m->D; p.N = m->Mp; p.epi = EPI_ATOMIC;
    if (m->bf16_mode) {   // dWp[d][c] += sum_v dFT[d][v] MetaT[c][v]  (both operands K-contiguous, padding columns are zero)
      p.A = m->dFT; p.lda = m->Vp; p.B = m->MetaT; p.ldb = m->Vp; p.K = (int)m->Vp;
      RC(gemm<T>(m, "gemm_table_dw", p, false, false, false));
    } else {
      p.A = m->G + m->o_E; p.lda = m->D; p.B = m->Meta; p.ldb = m->Mp; p.K = m->TR;
      RC(gemm<T>(m, "gemm_table_dw", p, false, true, true));
    }
  }
  toc(m);
  return RSYS_OK;
}

int model_finalize_grads(Model* m) {
  if (!m->table_grads_pending || m->cfg.finetune) return RSYS_OK;
  DetScope det(m);
  m->table_grads_pending = false;
  return m->bf16_mode ? finalize_grads_t<bf16>(m, 0) : finalize_grads_t<float>(m, 0);
}

// the two halves separately (gradient all-reduce overlap, capi.hip); only when model_finalize_splittable
bool model_finalize_splittable(const Model* m) { return m->table_grads_pending && !m->cfg.finetune && m->bf16_mode; }

// ---- split reduce of the replicated item table's gradient (model.hpp; armed per backward by rsys_set_grad_sync)
int model_split_table_enable(Model* m, int on) {
  if (!on) { m->split_table = false; m->table_head_hook = nullptr; return RSYS_OK; }
  ARG_CHECK(!m->sharded, "split table reduce: the item table is row-sharded (its rows are reduced by their owners already)");
  ARG_CHECK(m->bf16_mode && !m->cfg.finetune, "split table reduce: bf16 training of the full model only (the fp32 mode's metadata-projection gradient reads G[E] itself)");
  HIP_CHECK(hipSetDevice(m->device));
  if (!m->tbl_R) {
    const int64_t N = (int64_t)m->rows_max * m->S;
    m->tok_cap = N + 1;
    DALLOC(m->tbl_R, (int64_t)m->TR * m->D * 4);
    DALLOC(m->tok_T, m->tok_cap * m->D * 4);
    if (!m->u_slot) { DALLOC(m->u_slot, N * 4); DALLOC(m->u_ids, (N + 1) * 4); DALLOC(m->u_tok, N * 4); DALLOC(m->u_plan, 64); }
  }
  m->split_table = true;
  return RSYS_OK;
}

int model_split_table_tail(Model* m, rsys_comm* c, hipStream_t cs) {
  const int W = comm_active(c) ? c->world : 1, D = m->D;
  const int64_t cap = m->tok_cap;
  if (m->tok_all_world < W) {
    for (void* p : {(void*)m->tok_Tall, (void*)m->tok_Uall, (void*)m->tok_Pall}) if (p) HIP_CHECK(hipFree(p));
    m->tok_Tall = nullptr; m->tok_Uall = nullptr; m->tok_Pall = nullptr;
    HIP_CHECK(hipMalloc((void**)&m->tok_Tall, (size_t)W * cap * D * 4));
    HIP_CHECK(hipMalloc((void**)&m->tok_Uall, (size_t)W * cap * 4));
    HIP_CHECK(hipMalloc((void**)&m->tok_Pall, (size_t)W * 64));
    m->tok_all_world = W;
  }
  RC(comm_all_gather(c, m->tok_T, m->tok_Tall, (size_t)cap * D * 4, cs));
  RC(comm_all_gather(c, m->u_ids, m->tok_Uall, (size_t)cap * 4, cs));
  RC(comm_all_gather(c, m->u_plan, m->tok_Pall, 64, cs));
  HIP_CHECK(hipMemcpyAsync(m->G + m->o_E, m->tbl_R, (size_t)m->TR * D * 4, hipMemcpyDeviceToDevice, cs));
  for (int q = 0; q < W; ++q)   // rank order: every rank adds the same rows in the same order
    RC(launch_add_rows_by_id_counted(m->tok_Tall + (size_t)q * cap * D, m->tok_Uall + (size_t)q * cap, m->tok_Pall + (size_t)q * 16,
                                     m->G + m->o_E, D, (int)cap, D, cs));
  return RSYS_OK;
}
int model_finalize_stage(Model* m, int stage, int64_t* wp_off, int64_t* wp_n) {
  if (wp_off) *wp_off = m->o_Wp;
  if (wp_n) *wp_n = (int64_t)m->D * m->Mp;
  if (stage == 2) m->table_grads_pending = false;
  DetScope det(m);
  return finalize_grads_t<bf16>(m, stage);
}

template <typename T>
static int forward_backward_t(Model* m, int evaluate, const float task_w[4], float grad_scale, uint64_t seed, uint64_t step) {
  const int rows = m->cur_rows, N = rows * m->S;
  BatchDev b = m->bd; b.N = N; b.rows = rows; b.S = m->S;
  if (m->has_masks) { b.watch_mask = m->d_wm; b.rating_mask = m->d_rm; }
  m->cur_seed = seed; m->cur_step = step;
  RC(launch_mask_tokens(b, m->cfg.finetune, m->cfg.finetune_metric, m->cfg.mask_rate, seed, step, m->stream));
  m->drop_active = m->cfg.finetune && !evaluate && m->cfg.lora_dropout > 0.f;   // nn.Dropout is active in train() mode only
  m->drop_seed = seed ^ 0xD409ull; m->drop_step = step;
  m->top_is_sparse = m->sparse_top && !evaluate;
  m->f8_tcopies = !evaluate;
  m->host_stream_syncs = 0; m->host_event_waits = 0;
  float tw[4];
  for (int i = 0; i < 4; ++i) tw[i] = task_w ? task_w[i] * grad_scale : 0.f;
  RC(select_positions_all(m));
  if (m->sharded) RC(sharded_counts_early<T>(m, !evaluate, tw));
  RC(forward_trunk<T>(m));
  RC(heads<T>(m, evaluate, tw));
  if (!evaluate) {
    m->split_head_reduced = false;
    m->split_head_event = nullptr;
    if (m->table_head_hook) RC(m->table_head_hook());   // dF's head part is complete: its all-reduce starts under the trunk backward
    RC(backward_trunk<T>(m));
  }
  return RSYS_OK;
}

int model_forward_backward(Model* m, int evaluate, const float task_w[4], float grad_scale, uint64_t seed, uint64_t step) {
  ARG_CHECK(m->cur_rows > 0, "no batch uploaded");
  ARG_CHECK(evaluate || task_w != nullptr, "task weights are required for training");
  HIP_CHECK(hipSetDevice(m->device));
  DetScope det(m);
  m->last_evaluate = evaluate != 0;
  return m->bf16_mode ? forward_backward_t<bf16>(m, evaluate, task_w, grad_scale, seed, step)
                      : forward_backward_t<float>(m, evaluate, task_w, grad_scale, seed, step);
}

// inference forward (model.py:531-538): the batch is used as given (no masking), rope positions optional.  `sel` (n_sel flat
// token indices, host) restricts the output to those tokens: the rows are gathered on the device, the rating head runs on them
// only, and what crosses PCIe is n_sel rows instead of rows * 2S.
template <typename T>
static int infer_t(Model* m, int task, const int32_t* sel, int64_t n_sel, float* out, int64_t n) {
  const int rows = m->cur_rows, N = rows * m->S, NT = 2 * N, D = m->D;
  hipStream_t s = m->stream;
  BatchDev b = m->bd;
  const int KB = m->K * m->rows_max;
  if (sel != nullptr) {
    ARG_CHECK(n_sel <= NT, "selection: more tokens than the batch holds");
    for (int64_t i = 0; i < n_sel; ++i) ARG_CHECK(sel[i] >= 0 && sel[i] < NT, "selection: token index out of range");
  }
  const int64_t ntok = sel != nullptr ? n_sel : NT;
  ARG_CHECK(n == (task == 0 ? ntok * D : ntok), task == 0 ? "retrieval output has tokens*D floats" : "ranking output has one float per token");
  m->f8_tcopies = false;
  HIP_CHECK(hipMemcpyAsync(b.m_tmid, b.tmid, N * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(b.m_matchedid, b.matchedid, N * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(b.m_status, b.status, N * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(b.m_rating, b.rating, N * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(b.m_progress, b.progress, N * 4, hipMemcpyDeviceToDevice, s));
  m->drop_active = false;
  m->top_is_sparse = false;
  RC(forward_trunk<T>(m));
  // the rows to report: all of m->out, or the selected ones gathered into the (free) dx buffer of the backward
  const T* src = AT<T>(m->out);
  if (sel != nullptr) {
    int* d_sel = (int*)m->gf;                       // N * 32 floats: room for NT indices
    HIP_CHECK(hipMemcpyAsync(d_sel, sel, (size_t)n_sel * 4, hipMemcpyHostToDevice, s));
    RC(launch_gather_rows_plain<T>(AT<T>(m->out), D, d_sel, 0, AT<T>(m->dhn), (int)n_sel, D, s));
    src = AT<T>(m->dhn);
  }
  if (task == 0) {
    // float32 on the device (the head-gradient buffer is free in an inference pass), one copy out
    const float* f = (const float*)src;
    if (m->bf16_mode) { RC(launch_widen<T>(src, m->gy, ntok * D, s)); f = m->gy; }
    HIP_CHECK(hipMemcpyAsync(out, f, (size_t)ntok * D * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return RSYS_OK;
  }
  // rating_head (model.py:355-359) in chunks of the head workspace; predictions gather in a device vector
  float* pred = m->delta;   // (rows_max * H * T floats >= NT; only the attention backward uses it otherwise)
  for (int64_t r0 = 0; r0 < ntok; r0 += KB) {
    const int nr = (int)std::min<int64_t>(KB, ntok - r0);
    GemmParams p{};
    p.A = (const unsigned char*)src + (size_t)r0 * D * m->esz; p.lda = D; p.B = W<T>(m, m->o_r0w); p.ldb = D; p.C = m->z; p.ldc = D;
    p.M = nr; p.N = D; p.K = D; p.epi = EPI_GELU; p.bias = m->P + m->o_r0b; p.C2 = m->hact; p.ldc2 = D;
    RC(gemm<T>(m, "gemm_rating_fwd", p, false, false, false));
    RC(launch_rowdot<T>(AT<T>(m->hact), m->P + m->o_r2w, m->P + m->o_r2b, pred + r0, nr, D, s));
  }
  HIP_CHECK(hipMemcpyAsync(out, pred, (size_t)ntok * 4, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  return RSYS_OK;
}

// ItemEmbedding.forward over every item id (model.py:139-145; what register.py:27-29 stores as the watch-head weights):
// F[id] = E[id] + Wp Meta[id] + bp for id in [0, V), fp32.
template <typename T>
static int item_table_t(Model* m, float* out, int64_t n) {
  ARG_CHECK(!m->sharded, "item table: export from a model with a replicated table");
  ARG_CHECK(n == (int64_t)m->V * m->D, "item table: expected V * embed_dim values");
  if (m->table_dirty) { RC(table_forward<T>(m)); m->table_dirty = false; }
  HIP_CHECK(hipStreamSynchronize(m->stream));
  HIP_CHECK(hipMemcpy(out, m->F32, (size_t)n * 4, hipMemcpyDeviceToHost));
  return RSYS_OK;
}
int model_item_table(Model* m, float* out, int64_t n) {
  HIP_CHECK(hipSetDevice(m->device));
  return m->bf16_mode ? item_table_t<bf16>(m, out, n) : item_table_t<float>(m, out, n);
}

int model_infer(Model* m, int task, const int32_t* token_index, int64_t n_tokens, float* out, int64_t n) {
  ARG_CHECK(m->cur_rows > 0, "no batch uploaded");
  ARG_CHECK(task == 0 || task == 1, "task: 0 retrieval, 1 ranking");
  HIP_CHECK(hipSetDevice(m->device));
  return m->bf16_mode ? infer_t<bf16>(m, task, token_index, n_tokens, out, n) : infer_t<float>(m, task, token_index, n_tokens, out, n);
}

// sum of squares of all gradients into m->sumsq.  Row-sharded table: the replicated gradients are identical on every rank
// (after the all-reduce), the table rows differ: their sum of squares is all-reduced and added.
static int grad_sumsq(Model* m) {
  HIP_CHECK(hipMemsetAsync(m->sumsq, 0, 4, m->stream));
  if (!m->sharded || !comm_active(m->shard_comm)) return launch_sumsq(m->G, m->n_opt, m->sumsq, m->stream);
  const int64_t e0 = m->o_E, e1 = m->o_E + (int64_t)m->TR * m->D;
  RC(launch_sumsq(m->G, e0, m->sumsq, m->stream));
  RC(launch_sumsq(m->G + e1, m->n_opt - e1, m->sumsq, m->stream));
  HIP_CHECK(hipMemsetAsync(m->sumsq_E, 0, 4, m->stream));
  RC(launch_sumsq(m->G + e0, e1 - e0, m->sumsq_E, m->stream));
  RC(comm_all_reduce_f32(m->shard_comm, m->sumsq_E, 1, COMM_SUM, m->stream));
  return launch_add_scalar(m->sumsq, m->sumsq_E, m->stream);
}

int model_clip(Model* m, float max_norm, float* norm_out) {
  HIP_CHECK(hipSetDevice(m->device));
  DetScope det(m);
  RC(model_finalize_grads(m));
  RC(grad_sumsq(m));
  RC(launch_scale(m->G, m->n_opt, m->sumsq, 1.0f, max_norm, m->stream));
  if (norm_out) {
    float ss;
    HIP_CHECK(hipMemcpyAsync(&ss, m->sumsq, 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    *norm_out = sqrtf(ss);
  }
  return RSYS_OK;
}

int optimizer_step(Optimizer* o, float lr_factor, float clip, float grad_div) {
  Model* m = o->m;
  HIP_CHECK(hipSetDevice(m->device));
  DetScope det(m);
  RC(model_finalize_grads(m));
  if (grad_div <= 0.f) grad_div = 1.f;
  const float* ss = nullptr;
  if (clip > 0.f) {
    tic(m, "sumsq", 4.0 * m->n_opt);
    RC(grad_sumsq(m));
    toc(m);
    ss = m->sumsq;
  }
  o->step += 1;
  // the bf16 shadow of the item table E is read by no kernel (the fused-table GEMM adds E in fp32): the pass does not write it
  const long long e_lo = m->cfg.finetune ? 0 : m->o_E, e_hi = m->cfg.finetune ? 0 : m->o_E + pad8((int64_t)m->TR * m->D);
  tic(m, "adamw", 32.0 * m->n_opt + (m->bf16_mode ? 2.0 * (m->n_opt - (e_hi - e_lo)) : 0.0));   // p, g, m, v read; p, m, v, zeroed g (+ bf16 shadow) written
  int rc;
  if (!m->cfg.finetune) { m->wt_dirty = true; m->table_dirty = true; m->w8_dirty = true; }   // (finetune: only the LoRA segment moves; base weights, their transposes and the fused table stay)
  if (m->bf16_mode)
    rc = launch_adamw<bf16>(m->P, m->G, o->mom, o->var, (bf16*)m->Sh, m->n_opt_decay, m->n_opt, o->lr * lr_factor, o->b1, o->b2,
                            o->eps, o->wd, o->step, ss, grad_div, clip, 1, m->stream, e_lo, e_hi);
  else
    rc = launch_adamw<float>(m->P, m->G, o->mom, o->var, nullptr, m->n_opt_decay, m->n_opt, o->lr * lr_factor, o->b1, o->b2,
                             o->eps, o->wd, o->step, ss, grad_div, clip, 1, m->stream);
  toc(m);
  if (rc == RSYS_OK && !m->cfg.finetune) m->gE_clean[0] = m->gE_clean[1] = true;   // the kernel zeroed the gradients it consumed
  return rc;
}

// ---------------------------------------------------------------- ZeRO-1 (opt-in)
// Data parallel with a replicated model and a PARTITIONED optimizer: the flat gradient is reduce-scattered instead of all-reduced, a
// rank runs sumsq + AdamW on its 1/world of the parameters with moments for that part only, and the updated parameters are gathered.
// Chunks are whole multiples of 64 elements; the < 64 * world elements behind the last chunk are all-reduced and updated by the last
// rank.  The gradient clip needs the global norm: the ranks' partial sums of squares are summed (one float all-reduce).  The bf16
// shadows of the gathered parameters are recast locally.  Against the all-reduce path: same bytes on the wire (2 (W-1)/W of the
// buffer), optimizer pass and its state 1/W, but no overlap with the backward (the early buckets need the whole gradient reduced
// per bucket) and the gather sits between two steps.
int optimizer_set_zero1(Optimizer* o, int rank, int world) {
  Model* m = o->m;
  ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "zero1: rank / world");
  ARG_CHECK(!m->sharded && !m->cfg.finetune, "zero1: replicated pretraining model only (the row-sharded table already partitions its optimizer state)");
  ARG_CHECK(o->step == 0, "zero1: set before the first step");
  HIP_CHECK(hipSetDevice(m->device));
  const long long chunk = (m->n_opt / world) & ~63LL, tail = m->n_opt - chunk * world;
  ARG_CHECK(chunk > 0, "zero1: fewer than 64 parameters per rank");
  if (o->mom) HIP_CHECK(hipFree(o->mom));
  if (o->var) HIP_CHECK(hipFree(o->var));
  o->mom = o->var = nullptr;
  const size_t n = (size_t)(chunk + tail);
  HIP_CHECK(hipMalloc((void**)&o->mom, n * 4)); HIP_CHECK(hipMalloc((void**)&o->var, n * 4));
  HIP_CHECK(hipMemset(o->mom, 0, n * 4)); HIP_CHECK(hipMemset(o->var, 0, n * 4));
  if (tail > 0) HIP_CHECK(hipMalloc((void**)&o->z_tailbuf, (size_t)tail * 4));
  o->zero1 = true; o->z_rank = rank; o->z_world = world; o->z_chunk = chunk; o->z_tail = tail;
  return RSYS_OK;
}

int optimizer_step_zero1(Optimizer* o, rsys_comm* c, float lr_factor, float clip, float grad_div) {
  Model* m = o->m;
  ARG_CHECK(o->zero1, "zero1: rsys_adamw_set_zero1 first");
  ARG_CHECK(c != nullptr && c->world == o->z_world && c->rank == o->z_rank, "zero1: communicator of another rank / world");
  HIP_CHECK(hipSetDevice(m->device));
  DetScope det(m);
  RC(model_finalize_grads(m));
  hipStream_t s = m->stream;
  const long long chunk = o->z_chunk, tail = o->z_tail, lo = o->z_rank * chunk, tail_lo = chunk * o->z_world;
  const bool last = o->z_rank == o->z_world - 1;
  if (grad_div <= 0.f) grad_div = 1.f;
  RC(comm_reduce_scatter_f32(c, m->G, (size_t)chunk, s));
  if (tail > 0) RC(comm_all_reduce_f32(c, m->G + tail_lo, (size_t)tail, COMM_SUM, s));
  const float* ss = nullptr;
  if (clip > 0.f) {
    HIP_CHECK(hipMemsetAsync(m->sumsq, 0, 4, s));
    RC(launch_sumsq(m->G + lo, chunk, m->sumsq, s));
    if (last && tail > 0) RC(launch_sumsq(m->G + tail_lo, tail, m->sumsq, s));
    RC(comm_all_reduce_f32(c, m->sumsq, 1, COMM_SUM, s));
    ss = m->sumsq;
  }
  o->step += 1;
  const long long e_lo = m->o_E, e_hi = m->o_E + pad8((int64_t)m->TR * m->D);
  auto part = [&](long long at, long long n, float* mom, float* var) -> int {   // AdamW on [at, at + n) of the flat range
    const long long nd = std::min(std::max(m->n_opt_decay - at, 0LL), n);
    if (m->bf16_mode)
      return launch_adamw<bf16>(m->P + at, m->G + at, mom, var, (bf16*)m->Sh + at, nd, n, o->lr * lr_factor, o->b1, o->b2, o->eps, o->wd, o->step, ss,
                                grad_div, clip, 1, s, e_lo - at, e_hi - at);
    return launch_adamw<float>(m->P + at, m->G + at, mom, var, nullptr, nd, n, o->lr * lr_factor, o->b1, o->b2, o->eps, o->wd, o->step, ss, grad_div, clip, 1, s, 0, 0);
  };
  RC(part(lo, chunk, o->mom, o->var));
  if (last && tail > 0) RC(part(tail_lo, tail, o->mom + chunk, o->var + chunk));
  // the gradient of what other ranks own: consumed there, zero here for the next accumulation
  if (lo > 0) HIP_CHECK(hipMemsetAsync(m->G, 0, (size_t)lo * 4, s));
  if (lo + chunk < tail_lo) HIP_CHECK(hipMemsetAsync(m->G + lo + chunk, 0, (size_t)(tail_lo - lo - chunk) * 4, s));
  if (!last && tail > 0) HIP_CHECK(hipMemsetAsync(m->G + tail_lo, 0, (size_t)tail * 4, s));
  // everybody's updated chunk into everybody's parameters; the tail from the last rank (a sum in which the others hold zeros)
  RC(comm_all_gather(c, m->P + lo, m->P, (size_t)chunk * 4, s));
  if (tail > 0) {
    if (last) HIP_CHECK(hipMemcpyAsync(o->z_tailbuf, m->P + tail_lo, (size_t)tail * 4, hipMemcpyDeviceToDevice, s));
    else HIP_CHECK(hipMemsetAsync(o->z_tailbuf, 0, (size_t)tail * 4, s));
    RC(comm_all_reduce_f32(c, o->z_tailbuf, (size_t)tail, COMM_SUM, s));
    HIP_CHECK(hipMemcpyAsync(m->P + tail_lo, o->z_tailbuf, (size_t)tail * 4, hipMemcpyDeviceToDevice, s));
  }
  if (m->bf16_mode) {   // the bf16 shadows of the chunks other ranks updated (the item table has none: the fused-table GEMM reads it in fp32)
    if (e_lo > 0) RC(launch_cast<bf16>(m->P, (bf16*)m->Sh, e_lo, s));
    if (e_hi < m->n_opt) RC(launch_cast<bf16>(m->P + e_hi, (bf16*)m->Sh + e_hi, m->n_opt - e_hi, s));
  }
  m->wt_dirty = true; m->table_dirty = true; m->w8_dirty = true;
  m->gE_clean[0] = m->gE_clean[1] = true;
  return RSYS_OK;
}

}  // namespace rsys
