// Forward pass of the training step (transformer.model.py:493-529): fused item table, token embedding, the trunk's layers, the compact
// top, the heads and their losses (and the heads' own backward, which shares their operands).  Split from model.hip in round 5.
#include "model_internal.hpp"

namespace rsys {

// Fused item table F = E + Meta Wp^T + bp over all V + 1 rows (model.py:120-133): f32 copy for the token gather, T copy as the
// tied watch-head operand.  At cfg-3 this is 782 x 2 tiles of 256 x 256 = 6.1 per CU: a persistent grid runs SEVEN rounds, the last
// one on 28 of 256 CUs (13 % of the launch).  Round 5: the rows beyond the whole rounds (3 393 of 200 001) go to the row-major
// split-K form of the same pipeline instead -- their slice of F starts as E + bp (one row kernel), 28 tiles x 8 K splits add their
// products with fp32 atomics, a cast writes the T copy: three small launches (~35 us) for one round of the main launch (~180 us).
// (Sending those rows to the 128 x 128 kernel was measured in round 3: 1.32 -> 1.18 + 0.12 ms, no gain.)  Deterministic mode and
// fp32 keep the single launch (the atomics' order is free).
template <typename T>
int table_forward(Model* m) {
  GemmParams p{};
  p.A = m->Meta; p.lda = m->Mp; p.B = W<T>(m, m->o_Wp); p.ldb = m->Mp; p.C = m->F32; p.ldc = m->D; p.c_f32 = 1;
  p.M = m->TR; p.N = m->D; p.K = m->Mp; p.epi = EPI_TABLE; p.E = m->P + m->o_E; p.bias = m->P + m->o_bp;
  p.C2 = m->FT; p.ldc2 = m->D;
  int rows_main = m->TR;
  if constexpr (is_bf16<T>::value) {
    static int cus = 0;
    if (cus == 0) { int dev = 0, v = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    const int tiles_n = (m->D + 255) / 256, tiles = ((m->TR + 255) / 256) * tiles_n;
    const int rounds = tiles / cus, rem = tiles % cus;
    if (sw().table_tail != 0 && !m->deterministic && rounds >= 3 && rem > 0 && rem * 4 <= cus && (rounds * cus) % tiles_n == 0 && m->D % 256 == 0 &&
        m->Mp % 64 == 0 && m->Mp >= 1024)
      rows_main = (rounds * cus / tiles_n) * 256;
  }
  p.M = rows_main;
  RC(gemm<T>(m, "gemm_table_fwd", p, false, false, false));
  if (rows_main < m->TR) {
    const int64_t r0 = rows_main, nr = m->TR - rows_main;
    float* Ft = (float*)m->F32 + r0 * m->D;
    tic(m, "table_tail_rows");
    RC(launch_add_bias_rows(m->P + m->o_E + r0 * m->D, m->P + m->o_bp, Ft, nr, (int)m->D, m->stream));
    toc(m);
    GemmParams q{};
    q.A = (const T*)m->Meta + r0 * m->Mp; q.lda = m->Mp; q.B = W<T>(m, m->o_Wp); q.ldb = m->Mp; q.C = Ft; q.ldc = m->D; q.c_f32 = 1;
    q.M = (int)nr; q.N = m->D; q.K = m->Mp; q.epi = EPI_ATOMIC; q.flags = 128;
    RC(gemm<T>(m, "gemm_table_fwd_tail", q, false, false, false));
    tic(m, "table_tail_rows");
    RC(launch_cast<T>(Ft, (T*)m->FT + r0 * m->D, nr * m->D, m->stream));
    toc(m);
  }
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

int select_join(Model* m);   // (position selection runs on the side stream: defined with select_positions_all below)

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
int forward_trunk(Model* m) {
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
  ap.order_q = m->attn_order_q; ap.order_q2 = m->attn_order_q + (int64_t)m->rows_max * m->H * ((m->T + 63) / 64); ap.order_k = m->attn_order_k; ap.qbits = m->attn_qbits; ap.kbits = m->attn_kbits;
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
    ap_top.order_q = m->attn_order_q_p; ap_top.order_q2 = m->attn_order_q_p + (int64_t)m->rows_max * m->H * ((m->T + 63) / 64); ap_top.order_k = m->attn_order_k_p; ap_top.qbits = m->attn_qbits_p; ap_top.kbits = m->attn_kbits_p;
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
  ap.qmap16 = m->qmap16_p; ap.kmap16 = m->kmap16_p; ap.order_q = m->attn_order_q_p; ap.order_q2 = m->attn_order_q_p + (int64_t)m->rows_max * m->H * ((m->T + 63) / 64); ap.order_k = m->attn_order_k_p; ap.qbits = m->attn_qbits_p; ap.kbits = m->attn_kbits_p;
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
int sharded_counts_early(Model* m, bool train, const float tw[4]) {
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
int heads(Model* m, int evaluate, const float tw[4]) {
  const int D = m->D, rows = m->cur_rows, N = rows * m->S, NT = 2 * N, KB = m->K * rows;
  hipStream_t s = m->stream;
  const bool train = !evaluate;
  tic(m, "phase_heads");
  const bool ctop = m->top_is_sparse;   // trunk output and its gradient live in the compact buffers (rows = selected tokens)
  RC(select_join(m));
  if (train && ctop && m->loss_acc_with_c_gy) HIP_CHECK(hipMemsetAsync(m->loss_acc, 0, 256 + (size_t)m->ctop_cap * D * 4, s));   // (one allocation: model.hip)
  else {
    HIP_CHECK(hipMemsetAsync(m->loss_acc, 0, 16 * 4, s));
    if (train && !ctop) HIP_CHECK(hipMemsetAsync(m->gy, 0, (size_t)NT * D * 4, s));
    if (train && ctop) HIP_CHECK(hipMemsetAsync(m->c_gy, 0, (size_t)m->ctop_cap * D * 4, s));
  }
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
int select_positions_all(Model* m) {
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
int select_join(Model* m) {
  if (m->sel_pending) { HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_sel, 0)); m->sel_pending = false; }
  return RSYS_OK;
}

template int table_forward<float>(Model*);
template int table_forward<bf16>(Model*);
template int forward_trunk<float>(Model*);
template int forward_trunk<bf16>(Model*);
template int heads<float>(Model*, int, const float*);
template int heads<bf16>(Model*, int, const float*);
template int sharded_counts_early<float>(Model*, bool, const float*);
template int sharded_counts_early<bf16>(Model*, bool, const float*);

}  // namespace rsys
