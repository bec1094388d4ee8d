// Shared by the translation units of the training step (model.hip: set-up, batches, orchestration, inference; model_forward.hip;
// model_backward.hip; model_optim.hip -- one file until round 5): the timing brackets, the GEMM wrappers that pick split counts and
// streams, the fp8 product table, and the declarations of what one unit calls in another.  Everything defined here is `static`:
// each unit gets its own copy, none of it owns state.
#pragma once
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
  if (!t.filter.empty() && strstr(name, t.filter.c_str()) == nullptr) { t.open.push_back(0); return; }
  t.open.push_back(1);
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
  if (!t.open.empty()) { const char rec = t.open.back(); t.open.pop_back(); if (!rec) return; }
  hipEvent_t e = t.pool[t.used++];
  if (hipEventRecord(e, st ? st : m->stream) != hipSuccess) {   // (the span that was open stays unpaired and is dropped by rsys_timing_get)
    (void)hipGetLastError();
    t.enabled = false;
    return;
  }
  t.marks.push_back({std::string(""), e});
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

template <typename T>
static int gemm(Model* m, const char* tag, GemmParams p, bool a_f32, bool a_km, bool b_km) {
  if (p.alpha == 0.f) p.alpha = 1.f;
  if (p.epi == EPI_ATOMIC && p.splitk == 0)
    p.splitk = pick_splitk(p.M, p.N, (p.k_dev != nullptr && p.k_expect > 0) ? std::min(p.K, p.k_expect) : p.K, is_bf16<T>::value ? 64 : 32);
  if (p.splitk == 0) p.splitk = 1;
  p.flags |= m->gemm_flags;
  // A consumer that reads a large activation the previous kernel has just written walks its tile rows from the LAST to the first: the rows written
  // last are still in the Infinity Cache, a forward walk meets only the evicted ones (w2_fwd reads h behind w13_fwd's 553 MB of output, w13_dx reads
  // dab behind w2_dx; measured per call site, profiles/r6_ab_reverse_tile_rows.log: -1.7 % and -2.4 %, nothing at the other sites).  launch_gemm8c
  // honours the bit by RSYS_GEMM_REVERSE.
  if (strcmp(tag, "gemm_w2_fwd") == 0 || strcmp(tag, "gemm_w13_dx") == 0) p.flags |= 256;
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

// how many selected tokens to expect (K splits of the compact weight gradients only; the device-side count decides what is computed.
// The same hint for the row-limited GEMMs' kernel choice -- 128 x 128 tiles for ~3 K rows instead of 12 row tiles of 256 -- was
// measured and is not used: w2_dx 37 -> 72 us, w13_dx 57 -> 62 us): pretraining masks 2 * mask_rate of the interactions and a part of them carries a target; finetuning has one target per row
static int expected_selected(const Model* m) {
  const long long N = (long long)m->cur_rows * m->S;
  const long long e = m->cfg.finetune ? 2LL * m->cur_rows : (long long)(2.0 * m->cfg.mask_rate * (double)N * 0.6);
  return (int)std::max<long long>(256, std::min<long long>(m->ctop_cap, e));
}

// rows of medium `med` (global ids [0, V0) / [V0, V)) that this rank holds: `len` rows, the first one is id `col0` inside the
// medium and local table row `row` (replicated table: the whole medium)
static void shard_medium_range(const Model* m, int med, int* len, int* col0, int* row) {
  const int s = med == 0 ? 0 : m->V0, e = med == 0 ? m->V0 : m->V;
  const int a = std::max(s, m->row_lo), b = std::min(e, m->row_lo + m->TR);
  *len = std::max(0, b - a); *col0 = a - s; *row = a - m->row_lo;
  if (*len == 0) { *col0 = 0; *row = 0; }
}

// ---- defined in model_forward.hip
template <typename T> int table_forward(Model* m);
template <typename T> int forward_trunk(Model* m);
template <typename T> int heads(Model* m, int evaluate, const float tw[4]);
template <typename T> int sharded_counts_early(Model* m, bool train, const float tw[4]);
int select_positions_all(Model* m);
int select_join(Model* m);   // (position selection may run on the side stream)
// ---- defined in model_backward.hip
template <typename T> int backward_trunk(Model* m);

}  // namespace rsys
