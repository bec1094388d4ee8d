// Training-step orchestration: RecommenderModel.train_forward + backward
// (transformer.model.py:493-529 and its autograd), restructured for MI355X:
//  * the item table is fused once per step, F = E + Meta Wp^T + bp (model.py:120-133
//    does the same at inference), and shared by the token gather and both watch heads;
//    its gradient dF IS the gradient of E, and dWp = dF^T Meta is one GEMM per optimizer step;
//  * q/k/v and w1/w3 projections are single GEMMs (weights stored concatenated /
//    16-row interleaved in the flat parameter buffer) with RoPE and SwiGLU epilogues;
//  * residual stream fp32, GEMM operands T (bf16 or fp32), fp32 accumulation;
//  * every kernel of a step is enqueued on one HIP stream, no host sync inside a step.
// This file: model set-up, parameter I/O, batch upload, the step's orchestration and inference.  The forward pass is in
// model_forward.hip, the backward pass in model_backward.hip, clipping and AdamW in model_optim.hip (split in round 5, no behaviour change).
#include "model_internal.hpp"

namespace rsys {

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
    DALLOC(m->attn_order_q, (int64_t)m->rows_max * m->H * nt * 4 * 2);   /* second half: the q-tile-pair list (AttnParams::order_q2) */ DALLOC(m->attn_order_k, (int64_t)m->rows_max * m->KV * nt * 4);
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
  DALLOC(m->sumsq, 64); DALLOC(m->sumsq_part, (size_t)sumsq_parts() * 4);
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
      // c_gy sits 256 bytes behind the 16 loss accumulators: heads() zeroes both with one fill
      { float* blk = nullptr; DALLOC(blk, 256 + cap * D * 4); m->loss_acc = blk; m->c_gy = blk + 64; m->loss_acc_with_c_gy = true; }
      DALLOC(m->c_gx, cap * D * 4); DALLOC(m->c_dh, cap * D * 4);
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
      DALLOC(m->attn_order_q_p, (int64_t)m->rows_max * m->H * nt * 4 * 2); DALLOC(m->attn_order_k_p, (int64_t)m->rows_max * m->KV * nt * 4);
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
  if (m->loss_ring) { hipFree(m->loss_ring); m->loss_ring = nullptr; }
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
  if (m->h_umax) hipHostFree(m->h_umax);
  if (m->d_umax) hipFree(m->d_umax);
  if (m->ev_umax) hipEventDestroy(m->ev_umax);
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
struct StagedBatch { BatchDev bd; bool has_masks = false, has_rope_pos = false; unsigned char *d_wm = nullptr, *d_rm = nullptr; int* d_rope_pos = nullptr; size_t bytes = 0;
                     int distinct_ids = 0; };   // split table reduce: distinct matchedid values of the batch + 1 (the mask row): an upper bound of its token-row list
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
  if (m->split_table) {   // (one pass over the ids with a bitmap of the table's rows: ~30 us for 32 K ids, on the thread that packs the batch anyway)
    m->id_seen.assign(((size_t)m->V + 64) / 64, 0ull);
    int distinct = 1;       // the mask row V: mask_tokens sends the masked events there (model.py:451)
    for (size_t i = 0; i < N; ++i) {
      const int id = b->matchedid[i] < 0 ? m->V : b->matchedid[i];
      unsigned long long& wd = m->id_seen[(size_t)id >> 6];
      const unsigned long long bit = 1ull << (id & 63);
      if (!(wd & bit)) { wd |= bit; if (id != m->V) ++distinct; }
    }
    out.distinct_ids = distinct;
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
  // an explicit upload supersedes a prefetched batch.  Its copy may still be reading the OTHER slot's staging buffer on the copy stream:
  // wait for it here, so that the next prefetch (which skips its own wait when nothing is pending) never re-packs a buffer under a copy
  if (m->pending.valid) HIP_CHECK(hipEventSynchronize(m->ev_copy_done));
  m->pending.valid = false;
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
  m->u_bound_host = st.distinct_ids;
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
  m->pending.distinct_ids = st.distinct_ids;
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
  m->u_bound_host = m->pending.distinct_ids;
  m->tok_index_valid = false;
  m->split_plan_valid = false;
  m->pending.valid = false;
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

}  // namespace rsys
