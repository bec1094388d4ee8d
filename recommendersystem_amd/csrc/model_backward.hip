// Backward pass of the trunk (autograd of transformer.model.py:493-529 restated): compact top, layers in reverse with the bucketed
// gradient hooks, grouped weight gradients, embedding / table gradients and the finalize stages.  Split from model.hip in round 5.
#include "model_internal.hpp"

namespace rsys {

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
  // dO in selected-first order, as the attention backward reads it: the selected places from the compact rows, zeros up to the end of the last
  // query tile the kernels visit (c_qact); nothing is read beyond (round 6: was a 2 NT D-byte zero fill + a scatter)
  RC(launch_scatter_rows_fill<T>(AT<T>(m->c_dO), m->c_slot_p, m->c_qact, m->cur_rows, m->T, AT<T>(m->dO), D, D, s));
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
  if (m->timer.enabled) tic(m, f8 ? "gemm_dw_group@8gf" : (gemm8p_group_on_4k(it->second) ? "gemm_dw_group@4kg" : "gemm_dw_group@8g"), gemm8p_group_flops(it->second));
  int rc = launch_gemm8p_group(it->second, m->stream);
  toc(m);
  if (rc == RSYS_OK && f8) rc = f8_dw_round_accum(m, l_lo, l_hi);
  return rc;
}

// ------------------------------------------------------------------ backward trunk + embeddings
template <typename T>
int backward_trunk(Model* m) {
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
  ap.order_q = m->attn_order_q; ap.order_q2 = m->attn_order_q + (int64_t)m->rows_max * m->H * ((m->T + 63) / 64); ap.order_k = m->attn_order_k; ap.qbits = m->attn_qbits; ap.kbits = m->attn_kbits;
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
      at.order_q = m->attn_order_q_p; at.order_q2 = m->attn_order_q_p + (int64_t)m->rows_max * m->H * ((m->T + 63) / 64); at.order_k = m->attn_order_k_p; at.qbits = m->attn_qbits_p; at.kbits = m->attn_kbits_p;
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

// rows per rank the tail's all-gathers carry: the maximum over the ranks of the batches' distinct-id bounds (model_split_table_arm formed it
// at the head of the communicator's stream a whole backward ago), at least this rank's exact U, at most the lists' capacity
int model_split_table_rows(Model* m, int64_t* rows) {
  int64_t n = m->tok_cap;
  if (m->umax_pending) {
    HIP_CHECK(hipEventSynchronize(m->ev_umax));
    m->umax_pending = false;
    const int64_t got = (int64_t)m->h_umax[1];
    ARG_CHECK(got >= m->U && got <= m->tok_cap, "split table reduce: the ranks' maximum of distinct ids is smaller than this rank's own list");
    n = (got + 63) / 64 * 64;          // (whole 256-byte id blocks per rank)
    if (n > m->tok_cap) n = m->tok_cap;
  }
  *rows = n;
  return RSYS_OK;
}
int model_split_table_arm(Model* m, rsys_comm* c) {
  if (m->h_umax == nullptr) {
    HIP_CHECK(hipHostMalloc((void**)&m->h_umax, 64, hipHostMallocDefault));
    HIP_CHECK(hipMalloc((void**)&m->d_umax, 64));
    HIP_CHECK(hipEventCreateWithFlags(&m->ev_umax, hipEventDisableTiming));
  }
  // (a batch staged before the split reduce was switched on has no count: the lists' capacity then)
  const int64_t bound = (m->u_bound_host >= 1 && m->u_bound_host <= m->tok_cap) ? m->u_bound_host : m->tok_cap;
  if (m->umax_pending) HIP_CHECK(hipEventSynchronize(m->ev_umax));   // (an armed step that never reached its tail)
  m->h_umax[0] = (float)bound;                                         // (< 2^24: exact)
  HIP_CHECK(hipMemcpyAsync(m->d_umax, m->h_umax, 4, hipMemcpyHostToDevice, c->stream));
  RC(comm_all_reduce_f32(c, m->d_umax, 1, COMM_MAX, c->stream));
  HIP_CHECK(hipMemcpyAsync(m->h_umax + 1, m->d_umax, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipEventRecord(m->ev_umax, c->stream));
  m->umax_pending = true;
  return RSYS_OK;
}

int model_split_table_tail(Model* m, rsys_comm* c, hipStream_t cs, int64_t* rows_out) {
  const int W = comm_active(c) ? c->world : 1, D = m->D;
  int64_t cap = 0;
  RC(model_split_table_rows(m, &cap));
  if (rows_out) *rows_out = cap;
  if (m->tok_all_world < W || m->tok_all_rows < cap) {
    for (void* p : {(void*)m->tok_Tall, (void*)m->tok_Uall, (void*)m->tok_Pall}) if (p) HIP_CHECK(hipFree(p));
    m->tok_Tall = nullptr; m->tok_Uall = nullptr; m->tok_Pall = nullptr;
    const int64_t rows = std::min<int64_t>(m->tok_cap, cap + cap / 8);   // (some room: the next batches' maxima differ by a few percent)
    HIP_CHECK(hipMalloc((void**)&m->tok_Tall, (size_t)W * rows * D * 4));
    HIP_CHECK(hipMalloc((void**)&m->tok_Uall, (size_t)W * rows * 4));
    HIP_CHECK(hipMalloc((void**)&m->tok_Pall, (size_t)W * 64));
    m->tok_all_world = W; m->tok_all_rows = rows;
  }
  // rows [U_r, cap) of a rank's list hold whatever an earlier batch left there: they travel, nobody adds them (the counted add stops at the
  // rank's own U from its gathered plan)
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

template int backward_trunk<float>(Model*);
template int backward_trunk<bf16>(Model*);

}  // namespace rsys
