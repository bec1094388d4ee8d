// extern "C" surface of librsys_hip.so (declared in include/rsys.h).
#include <algorithm>
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <sstream>

#include "comm.hpp"
#include "model.hpp"

namespace rsys {
static thread_local std::string g_err;
#define RC(expr) do { int _rc = (expr); if (_rc != RSYS_OK) return _rc; } while (0)
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace rsys

using namespace rsys;

struct rsys_model { Model* m; };
struct rsys_optimizer { Optimizer o; };

#define CHECK_HANDLE(h) do { if ((h) == nullptr) { set_error("null handle"); return RSYS_ERR_ARG; } } while (0)

extern "C" {

const char* rsys_version(void) { return "recommendersystem_amd 0.1 (gfx950)"; }

size_t rsys_last_error(char* buf, size_t n) {
  if (buf && n) { strncpy(buf, g_err.c_str(), n - 1); buf[n - 1] = 0; }
  return g_err.size();
}

int32_t rsys_device_count(int32_t* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  if (n) *n = c;
  return RSYS_OK;
}
int32_t rsys_device_synchronize(void) { HIP_CHECK(hipDeviceSynchronize()); return RSYS_OK; }

int32_t rsys_model_create(const rsys_config* cfg, int32_t device, rsys_model** out) {
  switches_parse();
  Model* m = nullptr;
  int rc = model_create(cfg, device, &m);
  if (rc) return rc;
  *out = new rsys_model{m};
  return RSYS_OK;
}
int32_t rsys_model_destroy(rsys_model* h) { if (!h) return RSYS_OK; model_destroy(h->m); delete h; return RSYS_OK; }
int32_t rsys_model_init_random(rsys_model* h, uint64_t seed) { CHECK_HANDLE(h); return model_init_random(h->m, seed); }
int32_t rsys_model_load_metadata(rsys_model* h, const float* t, int64_t V, int64_t M) { CHECK_HANDLE(h); ARG_CHECK(t, "null table"); return model_load_metadata(h->m, t, V, M); }
int32_t rsys_model_random_metadata(rsys_model* h, uint64_t seed) { CHECK_HANDLE(h); return model_random_metadata(h->m, seed); }
int32_t rsys_model_set_rope(rsys_model* h, const float* c, const float* s, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(c && s, "null table"); return model_set_rope(h->m, c, s, n); }

int32_t rsys_param_count(rsys_model* h, int32_t* n) { CHECK_HANDLE(h); *n = (int32_t)h->m->tensors.size(); return RSYS_OK; }
int32_t rsys_param_info(rsys_model* h, int32_t i, char* name, size_t cap, int64_t shape[2], int32_t* ndim, int32_t* trainable) {
  CHECK_HANDLE(h);
  ARG_CHECK(i >= 0 && i < (int)h->m->tensors.size(), "parameter index");
  const TensorInfo& t = h->m->tensors[i];
  if (name && cap) { strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (shape) { if (t.ndim == 1) { shape[0] = t.cols; shape[1] = 0; } else { shape[0] = t.rows; shape[1] = t.cols; } }
  if (ndim) *ndim = t.ndim;
  if (trainable) *trainable = t.trainable ? 1 : 0;
  return RSYS_OK;
}
int32_t rsys_param_get(rsys_model* h, const char* name, float* out, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(name && out, "null"); return model_param_io(h->m, name, out, nullptr, n, 0); }
int32_t rsys_param_set(rsys_model* h, const char* name, const float* in, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(name && in, "null"); return model_param_io(h->m, name, nullptr, in, n, 0); }
int32_t rsys_grad_get(rsys_model* h, const char* name, float* out, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(name && out, "null"); return model_param_io(h->m, name, out, nullptr, n, 1); }
int32_t rsys_zero_grad(rsys_model* h) {
  CHECK_HANDLE(h);
  HIP_CHECK(hipSetDevice(h->m->device));
  HIP_CHECK(hipMemsetAsync(h->m->G, 0, h->m->n_total * 4, h->m->stream));
  h->m->table_grads_pending = false;
  h->m->gE_clean[0] = h->m->gE_clean[1] = true;
  return RSYS_OK;
}

int32_t rsys_batch_upload(rsys_model* h, const rsys_batch* b) { CHECK_HANDLE(h); return model_batch_upload(h->m, b); }
int32_t rsys_batch_prefetch(rsys_model* h, const rsys_batch* b) { CHECK_HANDLE(h); return model_batch_prefetch(h->m, b); }
int32_t rsys_batch_swap(rsys_model* h) { CHECK_HANDLE(h); return model_batch_swap(h->m); }

int32_t rsys_forward_backward(rsys_model* h, int32_t evaluate, const float task_w[4], float grad_scale, uint64_t seed, uint64_t step) {
  CHECK_HANDLE(h);
  return model_forward_backward(h->m, evaluate, task_w, grad_scale, seed, step);
}

int32_t rsys_losses_get(rsys_model* h, float losses_out[12], float wsum_out[4]) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  float acc[16], st[8];
  HIP_CHECK(hipMemcpyAsync(acc, m->loss_acc, 16 * 4, hipMemcpyDeviceToHost, m->stream));
  HIP_CHECK(hipMemcpyAsync(st, m->stats, 8 * 4, hipMemcpyDeviceToHost, m->stream));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  for (int ti = 0; ti < 4; ++ti) {
    float ws = fmaxf(st[2 * ti], 1e-8f);  // w_sum over the selected positions, clamp(min=1e-8) (model.py:392,515)
    for (int k = 0; k < 3; ++k) losses_out[3 * ti + k] = acc[3 * ti + k] / ws;
    if (wsum_out) wsum_out[ti] = st[2 * ti + 1];
  }
  return RSYS_OK;
}

// The reference's loop adds each step's losses into device tensors and reads them once per epoch (transformer.py:245-262, 279-283):
// push parks the finished step's loss sums and weight sums in a device ring (a 96-byte copy on the model's stream, no host wait),
// drain reads every parked step back in order with the arithmetic of rsys_losses_get.
int32_t rsys_losses_push(rsys_model* h) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  if (m->loss_ring == nullptr) HIP_CHECK(hipMalloc((void**)&m->loss_ring, (size_t)Model::loss_ring_cap * 24 * 4));
  if (m->loss_ring_n >= Model::loss_ring_cap) { set_error("rsys_losses_push: ring full (rsys_losses_drain first)"); return RSYS_ERR_STATE; }
  float* slot = m->loss_ring + (size_t)m->loss_ring_n * 24;
  HIP_CHECK(hipMemcpyAsync(slot, m->loss_acc, 16 * 4, hipMemcpyDeviceToDevice, m->stream));
  HIP_CHECK(hipMemcpyAsync(slot + 16, m->stats, 8 * 4, hipMemcpyDeviceToDevice, m->stream));
  m->loss_ring_n += 1;
  return RSYS_OK;
}

int32_t rsys_losses_drain(rsys_model* h, float* losses_out, float* wsum_out, int32_t cap, int32_t* n_out) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  ARG_CHECK(losses_out && wsum_out && n_out, "rsys_losses_drain: null output");
  ARG_CHECK(cap >= m->loss_ring_n, "rsys_losses_drain: output holds fewer steps than are parked");
  HIP_CHECK(hipSetDevice(m->device));
  const int n = m->loss_ring_n;
  *n_out = n;
  if (n == 0) return RSYS_OK;
  std::vector<float> host((size_t)n * 24);
  HIP_CHECK(hipMemcpyAsync(host.data(), m->loss_ring, host.size() * 4, hipMemcpyDeviceToHost, m->stream));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  for (int s = 0; s < n; ++s) {
    const float* acc = host.data() + (size_t)s * 24; const float* st = acc + 16;
    for (int ti = 0; ti < 4; ++ti) {
      const float ws = fmaxf(st[2 * ti], 1e-8f);
      for (int k = 0; k < 3; ++k) losses_out[(size_t)s * 12 + 3 * ti + k] = acc[3 * ti + k] / ws;
      wsum_out[(size_t)s * 4 + ti] = st[2 * ti + 1];
    }
  }
  m->loss_ring_n = 0;
  return RSYS_OK;
}

int32_t rsys_head_rows_get(rsys_model* h, int32_t out[4]) {
  CHECK_HANDLE(h);
  HIP_CHECK(hipSetDevice(h->m->device));
  HIP_CHECK(hipStreamSynchronize(h->m->stream));
  HIP_CHECK(hipMemcpy(out, h->m->npos, 16, hipMemcpyDeviceToHost));
  return RSYS_OK;
}

int32_t rsys_item_table(rsys_model* h, float* out, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(out, "null"); return model_item_table(h->m, out, n); }
int32_t rsys_model_set_deterministic(rsys_model* h, int32_t on) { CHECK_HANDLE(h); return model_set_deterministic(h->m, on); }
int32_t rsys_infer(rsys_model* h, int32_t task, float* out, int64_t n) { CHECK_HANDLE(h); ARG_CHECK(out, "null"); return model_infer(h->m, task, nullptr, 0, out, n); }
int32_t rsys_infer_select(rsys_model* h, int32_t task, const int32_t* token_index, int64_t n_tokens, float* out, int64_t n) {
  CHECK_HANDLE(h); ARG_CHECK(out && token_index && n_tokens >= 1, "null or empty selection");
  return model_infer(h->m, task, token_index, n_tokens, out, n);
}

int32_t rsys_trunk_output_get(rsys_model* h, float* out, int64_t n) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  const int64_t cnt = (int64_t)2 * m->cur_rows * m->S * m->D;
  ARG_CHECK(n == cnt, "trunk output has rows*2S*D floats");
  HIP_CHECK(hipSetDevice(m->device));
  { const int rc_ = model_materialise_trunk_output(m); if (rc_ != RSYS_OK) return rc_; }   // (a training pass computed it at the selected tokens only)
  HIP_CHECK(hipStreamSynchronize(m->stream));
  if (!m->bf16_mode) { HIP_CHECK(hipMemcpy(out, m->out, cnt * 4, hipMemcpyDeviceToHost)); return RSYS_OK; }
  std::vector<unsigned short> host(cnt);
  HIP_CHECK(hipMemcpy(host.data(), m->out, cnt * 2, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < cnt; ++i) { uint32_t u = (uint32_t)host[i] << 16; memcpy(&out[i], &u, 4); }
  return RSYS_OK;
}

// Parity read-back of the integer / index paths of the last forward (tests only; synchronises):
//   "masked.token_mask_ids" | "masked.matchedid" | "masked.status"   int32 [rows*S]   mask_tokens outputs (model.py:417-462)
//   "masked.rating" | "masked.progress"                              f32   [rows*S]
//   "masked.<m>.<watch|rating>.<label|weight|position>"              f32 / f32 / int32 [rows*S]
//   "idx.<task>"   int32 [mask_topk*rows]  flat positions chosen for task = medium*2 + metric (model.py:501-513)
//   "npos"         int32 [4]               positive-weight positions per task
//   "tokens.userid" | "tokens.token_mask_ids"   int32 [rows*2S]  interleaved per-token arrays (model.py:468-469)
//   "embed.x0"     f32 [rows*2S*D]         interleaved input embeddings (even rows: gathered item rows)
//   "table.fused"  f32 [(V+1)*D]           the fused item table the gather reads (row V = mask row)
//   "host_syncs" int32 [2]: blocking host waits inside the last rsys_forward_backward {stream drains, event waits}
//   "top.n" int32 [1] | "top.sel" int32 [top.cap] | "top.slot" int32 [rows*2S] | "top.cap" int32 [1]: compact top of the last training
//                  pass (model.hpp sparse_top): number of selected tokens, the sorted tokens, token -> compact row (-1: not selected)
int32_t rsys_debug_get(rsys_model* h, const char* key, void* out, int64_t bytes) {
  CHECK_HANDLE(h);
  ARG_CHECK(key && out, "null");
  Model* m = h->m;
  ARG_CHECK(m->cur_rows > 0, "no batch uploaded");
  HIP_CHECK(hipSetDevice(m->device));
  const int64_t N = (int64_t)m->cur_rows * m->S;
  const std::string k(key);
  const void* src = nullptr; int64_t n = 0;   // n: bytes
  const BatchDev& b = m->bd;
  if (k == "masked.token_mask_ids") { src = b.m_tmid; n = N * 4; }
  else if (k == "masked.matchedid") { src = b.m_matchedid; n = N * 4; }
  else if (k == "masked.status") { src = b.m_status; n = N * 4; }
  else if (k == "masked.rating") { src = b.m_rating; n = N * 4; }
  else if (k == "masked.progress") { src = b.m_progress; n = N * 4; }
  else if (k == "npos") { src = m->npos; n = 16; }
  else if (k == "tokens.userid") { src = m->uid_t; n = 2 * N * 4; }
  else if (k == "tokens.token_mask_ids") { src = m->tm_t; n = 2 * N * 4; }
  else if (k == "embed.x0") { src = m->x0; n = 2 * N * m->D * 4; }
  else if (k == "table.fused") { src = m->F32; n = (int64_t)m->TR * m->D * 4; }
  else if (k == "host_syncs") {   // blocking host waits inside the last rsys_forward_backward: {stream drains, waits on the early-counts event}
    ARG_CHECK(bytes == 8, "host_syncs: two int32"); ((int32_t*)out)[0] = m->host_stream_syncs; ((int32_t*)out)[1] = m->host_event_waits; return RSYS_OK; }
  else if (k.compare(0, 4, "act.") == 0 && k.size() > 6) {   // act.<layer>.<x|xn|qkv|O|h|hn|ab|g>: a saved activation of the last forward (T-typed ones as stored)
    const size_t dot = k.find('.', 4);
    const int l = atoi(k.substr(4, dot - 4).c_str());
    const std::string f = dot == std::string::npos ? "" : k.substr(dot + 1);
    ARG_CHECK(l >= 0 && l < m->L, "act: layer index");
    const Model::LayerAct& a = m->la[l];
    const int64_t NT = 2 * N, e = m->esz;
    if (f == "x") { src = a.x; n = NT * m->D * 4; } else if (f == "h") { src = a.h; n = NT * m->D * 4; }
    else if (f == "xn") { src = a.xn; n = NT * m->D * e; } else if (f == "hn") { src = a.hn; n = NT * m->D * e; }
    else if (f == "qkv") { src = a.qkv; n = NT * m->Nqkv * e; } else if (f == "O") { src = a.O; n = NT * m->D * e; }
    else if (f == "ab") { src = a.ab; n = NT * 2 * m->Ip * e; } else if (f == "g") { src = a.g; n = NT * m->Ip * e; }
  }
  else if (k.compare(0, 3, "dw.") == 0 && !m->dwb.empty()) {   // dw.<layer>.<gxt|dab|dht|dqkv>: the dY operands the deferred weight gradients kept
    const size_t dot = k.find('.', 3);
    const int l = atoi(k.substr(3, dot - 3).c_str());
    const std::string f = dot == std::string::npos ? "" : k.substr(dot + 1);
    ARG_CHECK(l >= 0 && l < m->L, "dw: layer index");
    const Model::DwOperands& o = m->dwb[l];
    const int64_t NT = 2 * N;
    if (f == "gxt" && o.gxt) { src = o.gxt; n = NT * m->D * 2; } else if (f == "dab" && o.dab) { src = o.dab; n = NT * 2 * m->Ip * 2; }
    else if (f == "dht" && o.dht) { src = o.dht; n = NT * m->D * 2; } else if (f == "dqkv" && o.dqkv) { src = o.dqkv; n = NT * m->Nqkv * 2; }
  }
  else if (k.compare(0, 7, "f8keep.") == 0 && !m->f8_keep.empty()) {   // f8keep.<layer>.<0|1|2>: outputs of w13_dx, o_dx, qkv_dx (RSYS_F8_DEBUG_KEEP=1)
    const size_t dot = k.find('.', 7);
    const int l = atoi(k.substr(7, dot - 7).c_str()), j = dot == std::string::npos ? -1 : atoi(k.substr(dot + 1).c_str());
    ARG_CHECK(l >= 0 && l < m->L && j >= 0 && j < 3, "f8keep: layer / product index");
    src = m->f8_keep[(size_t)l * 3 + j]; n = 2 * N * m->D * 2;
  }
  else if (k == "f8.aamax" && m->fp8) { src = m->f8_aamax; n = (int64_t)m->L * F8_AMAX_SHARDS * F8_AMAX_SHARD * 4; }
  else if (k == "f8.wamax" && m->fp8) { src = m->f8_wamax; n = (int64_t)m->L * 8 * 4; }
  else if (k == "f8.desc" && m->fp8) { src = m->f8_desc; n = (int64_t)m->L * 8 * 32 * 4; }
  else if (k == "top.cap") { ARG_CHECK(bytes == 4, "top.cap: one int32"); *(int32_t*)out = m->top_is_sparse ? m->ctop_cap : 0; return RSYS_OK; }
  else if (k == "top.n" && m->top_is_sparse) { src = m->c_n; n = 4; }
  else if (k == "top.sel" && m->top_is_sparse) { src = m->c_sel; n = (int64_t)m->ctop_cap * 4; }
  else if (k == "top.slot" && m->top_is_sparse) { src = m->c_slot; n = 2 * N * 4; }
  else if (k.size() == 5 && k.compare(0, 4, "idx.") == 0 && k[4] >= '0' && k[4] <= '3') { src = m->idx[k[4] - '0']; n = (int64_t)m->K * m->cur_rows * 4; }
  else if (k.compare(0, 7, "masked.") == 0 && k.size() > 9 && (k[7] == '0' || k[7] == '1') && k[8] == '.') {
    const int med = k[7] - '0';
    const std::string rest = k.substr(9);
    const size_t dot = rest.find('.');
    if (dot != std::string::npos) {
      const std::string metric = rest.substr(0, dot), field = rest.substr(dot + 1);
      const int mi = metric == "watch" ? 0 : metric == "rating" ? 1 : -1;
      if (mi >= 0) {
        const int ti = med * 2 + mi;
        if (field == "label") src = b.m_label[ti]; else if (field == "weight") src = b.m_weight[ti]; else if (field == "position") src = b.m_position[ti];
        n = N * 4;
      }
    }
  }
  if (src == nullptr) { set_error(std::string("rsys_debug_get: unknown key ") + key); return RSYS_ERR_ARG; }
  ARG_CHECK(bytes == n, "rsys_debug_get: buffer size does not match the array");
  HIP_CHECK(hipStreamSynchronize(m->stream));
  HIP_CHECK(hipMemcpy(out, src, (size_t)n, hipMemcpyDeviceToHost));
  return RSYS_OK;
}

int32_t rsys_clip_grad_norm(rsys_model* h, float max_norm, float* norm_out) { CHECK_HANDLE(h); return model_clip(h->m, max_norm, norm_out); }

// Replica-consistency words of this rank's parameters (transformer.py:678-682): the flat fp32 parameter buffer -- of a
// row-sharded model everything but its table rows, which differ between the ranks by construction -- as {fp64 sum, fp64 sum of
// squares, low and high 32 bits of a position-weighted integer sum of the bit patterns}.  Synchronises.
int32_t rsys_param_checksum(rsys_model* h, double out[4]) {
  CHECK_HANDLE(h);
  ARG_CHECK(out != nullptr, "rsys_param_checksum: out is NULL");
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  double* scratch = nullptr;
  HIP_CHECK(hipMalloc((void**)&scratch, (size_t)checksum_scratch_doubles() * 8));
  long long ranges[4] = {0, m->n_total, 0, 0};
  int nr = 1;
  if (m->sharded && m->sh_world > 1) {
    ranges[1] = m->o_E; ranges[2] = m->o_E + ((int64_t)m->TR * m->D + 7) / 8 * 8; ranges[3] = m->n_total; nr = 2;
  }
  double* out_dev = scratch + checksum_scratch_doubles() - 4;
  int rc = launch_checksum(m->P, ranges, nr, scratch, out_dev, m->stream);
  if (rc == RSYS_OK) {
    hipError_t e = hipMemcpyAsync(out, out_dev, 32, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) { set_error(std::string("rsys_param_checksum: ") + hipGetErrorString(e)); rc = RSYS_ERR_HIP; }
  }
  (void)hipFree(scratch);
  return rc;
}

int32_t rsys_adamw_create(rsys_model* h, float lr, float b1, float b2, float eps, float wd, rsys_optimizer** out) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  rsys_optimizer* o = new rsys_optimizer();
  o->o.m = m; o->o.device = m->device; o->o.lr = lr; o->o.b1 = b1; o->o.b2 = b2; o->o.eps = eps; o->o.wd = wd;
  HIP_CHECK(hipMalloc((void**)&o->o.mom, m->n_total * 4));
  HIP_CHECK(hipMalloc((void**)&o->o.var, m->n_total * 4));
  HIP_CHECK(hipMemset(o->o.mom, 0, m->n_total * 4));
  HIP_CHECK(hipMemset(o->o.var, 0, m->n_total * 4));
  *out = o;
  return RSYS_OK;
}
int32_t rsys_adamw_destroy(rsys_optimizer* o) {
  if (!o) return RSYS_OK;
  hipSetDevice(o->o.device);
  hipDeviceSynchronize();   // (not the model's stream: the model may already be gone)
  hipFree(o->o.mom); hipFree(o->o.var);
  if (o->o.z_tailbuf) hipFree(o->o.z_tailbuf);
  delete o;
  return RSYS_OK;
}
int32_t rsys_adamw_step(rsys_optimizer* o, float lr_factor, float clip, float grad_div) {
  CHECK_HANDLE(o);
  ARG_CHECK(!o->o.zero1, "this optimizer is partitioned (rsys_adamw_set_zero1): step it with rsys_adamw_step_zero1");
  return optimizer_step(&o->o, lr_factor, clip, grad_div);
}
int32_t rsys_adamw_set_zero1(rsys_optimizer* o, int32_t rank, int32_t world) { CHECK_HANDLE(o); return optimizer_set_zero1(&o->o, rank, world); }
int32_t rsys_adamw_step_zero1(rsys_optimizer* o, rsys_comm* c, float lr_factor, float clip, float grad_div) {
  CHECK_HANDLE(o); CHECK_HANDLE(c);
  Model* m = o->o.m;
  m->grad_bucket_hook = nullptr; m->reduced.clear(); m->early_reduced = 0;   // (no early buckets in this mode: the step reduces the whole gradient)
  m->gemm_flags &= ~2;
  return optimizer_step_zero1(&o->o, c, lr_factor, clip, grad_div);
}

static int adam_state_io(rsys_optimizer* o, const char* name, float* m_out, float* v_out, const float* m_in, const float* v_in, int64_t n) {
  Model* m = o->o.m;
  auto it = m->by_name.find(name);
  if (it == m->by_name.end()) { set_error(std::string("unknown parameter: ") + name); return RSYS_ERR_ARG; }
  const TensorInfo& t = m->tensors[it->second];
  ARG_CHECK(!t.frozen_table, "frozen table has no optimizer state");
  ARG_CHECK(n == t.rows * t.cols, "element count");
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  const int64_t int_rows = (t.map == MAP_DIRECT) ? t.rows : 2 * m->Ip;
  std::vector<float> host((size_t)int_rows * t.ld);
  // ZeRO-1 (optimizer_set_zero1): this rank holds the moments of flat elements [rank * chunk, (rank + 1) * chunk) at local offset 0 and --
  // the last rank -- of the tail [chunk * world, n_opt) behind them.  A read returns the rank's part of the tensor and zeros elsewhere
  // (the parts of all ranks are disjoint: AdamW.state_dict sums them over the ranks' HostGroup); a write keeps the rank's part.
  struct Piece { int64_t g_lo, g_hi, local; };
  std::vector<Piece> pieces;
  const int64_t t_lo = t.off, t_hi = t.off + (int64_t)host.size();
  if (o->o.zero1) {
    const int64_t chunk = o->o.z_chunk, own = o->o.z_rank * chunk, tail_lo = chunk * o->o.z_world;
    pieces.push_back({own, own + chunk, 0});
    if (o->o.z_rank == o->o.z_world - 1 && o->o.z_tail > 0) pieces.push_back({tail_lo, tail_lo + o->o.z_tail, chunk});
  }
  for (int which = 0; which < 2; ++which) {
    float* zbase = which == 0 ? o->o.mom : o->o.var;
    float* base = zbase + t.off;
    float* out = which == 0 ? m_out : v_out;
    const float* in = which == 0 ? m_in : v_in;
    if (o->o.zero1) {
      std::fill(host.begin(), host.end(), 0.f);
      auto irow = [&](int64_t r) { return t.map == MAP_W1 ? (r / 16) * 32 + r % 16 : t.map == MAP_W3 ? (r / 16) * 32 + 16 + r % 16 : r; };
      if (in) for (int64_t r = 0; r < t.rows; ++r) memcpy(host.data() + irow(r) * t.ld, in + r * t.cols, t.cols * 4);
      for (const Piece& pc : pieces) {
        const int64_t a = std::max(pc.g_lo, t_lo), b = std::min(pc.g_hi, t_hi);
        if (a >= b) continue;
        if (in) {
          // (a W1 / W3 tensor shares its interleaved rows with its partner: write this tensor's rows only)
          if (t.map == MAP_DIRECT) HIP_CHECK(hipMemcpy(zbase + pc.local + (a - pc.g_lo), host.data() + (a - t_lo), (size_t)(b - a) * 4, hipMemcpyHostToDevice));
          else
            for (int64_t r = 0; r < t.rows; ++r) {
              const int64_t ra = std::max(a, t_lo + irow(r) * t.ld), rb = std::min(b, t_lo + irow(r) * t.ld + t.cols);
              if (ra < rb) HIP_CHECK(hipMemcpy(zbase + pc.local + (ra - pc.g_lo), host.data() + (ra - t_lo), (size_t)(rb - ra) * 4, hipMemcpyHostToDevice));
            }
        } else {
          HIP_CHECK(hipMemcpy(host.data() + (a - t_lo), zbase + pc.local + (a - pc.g_lo), (size_t)(b - a) * 4, hipMemcpyDeviceToHost));
        }
      }
      if (out) for (int64_t r = 0; r < t.rows; ++r) memcpy(out + r * t.cols, host.data() + irow(r) * t.ld, t.cols * 4);
      continue;
    }
    HIP_CHECK(hipMemcpy(host.data(), base, host.size() * 4, hipMemcpyDeviceToHost));
    auto irow = [&](int64_t r) { return t.map == MAP_W1 ? (r / 16) * 32 + r % 16 : t.map == MAP_W3 ? (r / 16) * 32 + 16 + r % 16 : r; };
    if (out) for (int64_t r = 0; r < t.rows; ++r) memcpy(out + r * t.cols, host.data() + irow(r) * t.ld, t.cols * 4);
    if (in) {
      for (int64_t r = 0; r < t.rows; ++r) memcpy(host.data() + irow(r) * t.ld, in + r * t.cols, t.cols * 4);
      HIP_CHECK(hipMemcpy(base, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
  }
  return RSYS_OK;
}
int32_t rsys_adamw_state_get(rsys_optimizer* o, const char* name, float* m_out, float* v_out, int64_t n, int32_t* step) {
  CHECK_HANDLE(o);
  if (step) *step = o->o.step;
  if (name == nullptr) return RSYS_OK;
  return adam_state_io(o, name, m_out, v_out, nullptr, nullptr, n);
}
int32_t rsys_adamw_state_set(rsys_optimizer* o, const char* name, const float* m_in, const float* v_in, int64_t n, int32_t step) {
  CHECK_HANDLE(o);
  if (step >= 0) o->o.step = step;
  if (name == nullptr) return RSYS_OK;
  return adam_state_io(o, name, nullptr, nullptr, m_in, v_in, n);
}

// ---------------------------------------------------------------- communicator
int32_t rsys_comm_unique_id(uint8_t id_buf[128]) { return comm_unique_id(id_buf); }
int32_t rsys_comm_init(const uint8_t id_buf[128], int32_t rank, int32_t world, int32_t device, rsys_comm** out) {
  switches_parse();
  ARG_CHECK(id_buf && out, "null");
  return comm_init_rccl(id_buf, rank, world, device, out);
}
int32_t rsys_comm_destroy(rsys_comm* c) { return comm_destroy(c); }
int32_t rsys_comm_debug_delay(rsys_comm* c, int32_t microseconds) { CHECK_HANDLE(c); return comm_debug_delay(c, microseconds); }

// in-process rank group (tests of the multi-rank partition arithmetic on one GPU): `world` ranks of THIS process on one
// device, each driven by its own host thread; rsys_comm_init_local gives rank r its communicator
int32_t rsys_local_group_create(int32_t world, int32_t device, void** out) {
  ARG_CHECK(out && world >= 1 && world <= 16, "in-process group: 1..16 ranks");
  LocalGroup* g = new LocalGroup();
  g->world = world; g->device = device; g->slot.resize(world);
  *out = g;
  return RSYS_OK;
}
int32_t rsys_local_group_destroy(void* group) {
  if (!group) return RSYS_OK;
  LocalGroup* g = (LocalGroup*)group;
  { std::lock_guard<std::mutex> lk(g->mu); if (g->refs > 0) { set_error("in-process group: close its communicators first (rsys_comm_destroy)"); return RSYS_ERR_STATE; } }
  delete g;
  return RSYS_OK;
}
int32_t rsys_comm_init_local(void* group, int32_t rank, rsys_comm** out) {
  switches_parse();
  ARG_CHECK(group && out, "null");
  return comm_init_local((LocalGroup*)group, rank, out);
}

// DDP's bucketed gradient all-reduce (train.py:678-682, 272) on the communicator's own stream.
//
// rsys_set_grad_sync(model, comm) before the backward of the last micro-step arms the early buckets: the trunk backward
// hands over each >= 25 MB run of finished per-layer weight gradients (reverse layer order), and its all-reduce is
// enqueued behind an event while the backward of the lower layers is still running.  rsys_allreduce_grads then reduces
// whatever is left (heads, small tensors, the item table and the metadata projection, final only after the token
// scatter) and makes the compute stream wait for the last bucket before the optimizer reads the gradients.
static int reduce_range(Model* m, rsys_comm* c, int64_t lo, int64_t hi) {
  const int64_t bucket = 16 * 1024 * 1024;  // floats
  for (int64_t o = lo; o < hi; o += bucket) {
    const int64_t n = std::min(bucket, hi - o);
    RC(comm_all_reduce_f32(c, m->G + o, (size_t)n, COMM_SUM, c->stream));
  }
  if (hi > lo) m->bucket_log.push_back({lo, hi, m->bucket_phase});
  return RSYS_OK;
}

int32_t rsys_set_grad_sync(rsys_model* h, rsys_comm* c) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  m->reduced.clear();
  m->bucket_log.clear(); m->bucket_phase = 0;
  m->grad_bucket_hook = nullptr;
  m->table_head_hook = nullptr;
  m->gemm_flags &= ~2;
  if (!comm_active(c) || m->cfg.finetune) return RSYS_OK;
  if (m->split_table && m->bf16_mode && !m->sharded) {
    // the head part of the item table's gradient, complete when heads() returns: summed over the ranks into tbl_R while the trunk
    // backward runs (G[E] itself keeps the local gradient: the token scatter and the metadata-projection gradient still need it)
    m->table_head_hook = [m, c]() -> int {
      // first on the communicator's stream, a whole trunk backward before the tail needs it on the host: the ranks' maximum of the
      // distinct ids of their resident batches = the rows per rank the tail's gathers will carry
      RC(model_split_table_arm(m, c));
      HIP_CHECK(hipEventRecord(c->ev_ready, m->stream));
      HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
      const int64_t n = (int64_t)m->TR * m->D, bucket = 16 * 1024 * 1024;
      for (int64_t o = 0; o < n; o += bucket)
        RC(comm_all_reduce_f32_to(c, m->G + m->o_E + o, m->tbl_R + o, (size_t)std::min(bucket, n - o), c->stream));
      // The collective READS G[E] on the communicator's stream; the trunk backward WRITES G[E] on the model's stream (the token rows,
      // model.hip backward_trunk).  That write waits for this event: free when the reduce is done by then, and otherwise what keeps a
      // slow link from folding some ranks' token rows into tbl_R, which the tail would then add a second time.
      HIP_CHECK(hipEventRecord(c->ev_head, c->stream));
      m->bucket_log.push_back({m->o_E, m->o_E + n, 3});
      m->split_head_event = c->ev_head;
      m->split_head_reduced = true;
      return RSYS_OK;
    };
  }
  m->grad_bucket_hook = [m, c](int64_t lo, int64_t hi) -> int {
    hi = std::min(hi, m->n_opt);
    if (lo >= hi) return RSYS_OK;
    HIP_CHECK(hipEventRecord(c->ev_ready, m->stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    int rc = reduce_range(m, c, lo, hi);
    if (rc) return rc;
    m->reduced.emplace_back(lo, hi);
    return RSYS_OK;
  };
  return RSYS_OK;
}

int32_t rsys_model_set_split_table_reduce(rsys_model* h, int32_t on) {
  CHECK_HANDLE(h);
  return model_split_table_enable(h->m, on);
}

int32_t rsys_model_set_shard_comm(rsys_model* h, rsys_comm* c) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  ARG_CHECK(m->sharded, "the model's item table is replicated (rsys_config.table_shard_world == 0)");
  ARG_CHECK((c == nullptr && m->sh_world == 1) || (c != nullptr && c->world == m->sh_world && c->rank == m->sh_rank),
            "the communicator's rank / world must equal the model's table_shard_rank / table_shard_world");
  m->shard_comm = c;
  return RSYS_OK;
}
int32_t rsys_table_rows(rsys_model* h, int64_t* lo, int64_t* hi) {
  CHECK_HANDLE(h);
  if (lo) *lo = h->m->row_lo;
  if (hi) *hi = h->m->row_lo + h->m->TR;
  return RSYS_OK;
}

int32_t rsys_allreduce_grads(rsys_model* h, rsys_comm* c) {
  CHECK_HANDLE(h); CHECK_HANDLE(c);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  if (m->grad_bucket_hook == nullptr && m->table_head_hook == nullptr) m->bucket_log.clear();   // (not armed: nothing of this step is in the log yet)
  m->grad_bucket_hook = nullptr;   // one backward per arming
  m->table_head_hook = nullptr;
  const bool split = m->split_head_reduced && comm_active(c) && model_finalize_splittable(m);   // (else: G[E] holds the whole local gradient, the dense path is right)
  m->split_head_reduced = false;
  m->split_head_event = nullptr;   // (everything below is ordered behind the head reduce on the communicator's stream itself)
  m->gemm_flags &= ~2;             // the optimizer waits for the reduction: nothing after this call overlaps with it
  m->early_reduced = 0;
  for (auto& r : m->reduced) m->early_reduced += r.second - r.first;
  if (!comm_active(c)) { m->reduced.clear(); return model_finalize_grads(m); }
  // what the early buckets have not covered, ascending
  std::vector<std::pair<int64_t, int64_t>> done = m->reduced, rem;
  m->reduced.clear();
  // row-sharded table: a rank's table rows already hold the gradient of EVERY rank's loss (vocabulary-parallel head, row
  // exchange of the token gradients): they are not part of the dense all-reduce
  if (m->sharded) done.emplace_back(m->o_E, m->o_E + (int64_t)m->TR * m->D);
  if (split) {   // split table reduce: tbl_R + the ranks' token rows instead
    done.emplace_back(m->o_E, m->o_E + (int64_t)m->TR * m->D);
    m->early_reduced += (int64_t)m->TR * m->D;
  }
  std::sort(done.begin(), done.end());
  int64_t at = 0;
  for (auto& r : done) { if (r.first > at) rem.emplace_back(at, r.first); at = std::max(at, r.second); }
  if (at < m->n_opt) rem.emplace_back(at, m->n_opt);
  auto reduce_rest = [&](int64_t lo, int64_t hi) -> int {
    for (auto& r : rem) {
      const int64_t a = std::max(r.first, lo), b = std::min(r.second, hi);
      if (a < b) { int rc = reduce_range(m, c, a, b); if (rc) return rc; }
    }
    return RSYS_OK;
  };
  if (model_finalize_splittable(m)) {
    // The last piece of the backward -- the metadata-projection gradient dWp = dF^T Meta, a ~2 ms GEMM on a bf16 copy
    // of dF -- runs on the model's stream while the communication stream already reduces everything else (80 % of the
    // bytes are dF itself, final since the token scatter); dWp follows when the GEMM is done.
    int64_t wo = 0, wn = 0;
    int rc = model_finalize_stage(m, 1, &wo, &wn);
    if (rc) return rc;
    HIP_CHECK(hipEventRecord(c->ev_ready, m->stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    m->bucket_phase = 1;
    if (split) {   // nothing reads the local G[E] any more (stage 1 made the operand copy and the bias gradient): it becomes the sum
      int64_t tail_rows = 0;
      rc = model_split_table_tail(m, c, c->stream, &tail_rows);
      if (rc) return rc;
      m->bucket_log.push_back({0, (int64_t)c->world * tail_rows * (m->D + 1), 4});
    }
    rc = reduce_rest(0, std::min(wo, m->n_opt));
    if (rc) return rc;
    rc = reduce_rest(std::min(wo + wn, m->n_opt), m->n_opt);
    if (rc) return rc;
    rc = model_finalize_stage(m, 2, nullptr, nullptr);
    if (rc) return rc;
    HIP_CHECK(hipEventRecord(c->ev_ready, m->stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    m->bucket_phase = 2;
    rc = reduce_rest(wo, std::min(wo + wn, m->n_opt));
    if (rc) return rc;
  } else {
    m->bucket_phase = 1;
    int rc = model_finalize_grads(m);
    if (rc) return rc;
    HIP_CHECK(hipEventRecord(c->ev_ready, m->stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    rc = reduce_rest(0, m->n_opt);
    if (rc) return rc;
  }
  HIP_CHECK(hipEventRecord(c->ev_done, c->stream));
  HIP_CHECK(hipStreamWaitEvent(m->stream, c->ev_done, 0));
  return RSYS_OK;
}
int32_t rsys_grad_sync_early(rsys_model* h, int64_t* n) { CHECK_HANDLE(h); ARG_CHECK(n, "null"); *n = h->m->early_reduced; return RSYS_OK; }
int32_t rsys_grad_sync_schedule(rsys_model* h, int64_t* out, int32_t cap, int32_t* n_out) {
  CHECK_HANDLE(h); ARG_CHECK(out && n_out && cap >= 0, "null");
  const auto& log = h->m->bucket_log;
  const int n = (int)std::min<size_t>(log.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) { out[3 * i] = log[i].lo; out[3 * i + 1] = log[i].hi; out[3 * i + 2] = log[i].phase; }
  *n_out = (int32_t)log.size();
  return RSYS_OK;
}
int32_t rsys_comm_info(rsys_comm* c, int32_t out[4]) {
  CHECK_HANDLE(c); ARG_CHECK(out, "null");
  out[0] = c->rank; out[1] = c->world;
  out[2] = c->lg ? 2 : (c->comm ? 1 : 0);     // transport: 1 = RCCL, 2 = in-process rank group (tests)
  out[3] = c->comm ? comm_rccl_version() : 0;
  return RSYS_OK;
}
int32_t rsys_allreduce_f64(rsys_comm* c, double* x, int32_t n) {
  CHECK_HANDLE(c);
  ARG_CHECK(n >= 1 && n <= 64 && x, "n in [1,64]");
  if (!comm_active(c)) return RSYS_OK;
  HIP_CHECK(hipSetDevice(c->device));
  HIP_CHECK(hipMemcpyAsync(c->scratch, x, n * 8, hipMemcpyHostToDevice, c->stream));
  RC(comm_all_reduce_f64(c, c->scratch, (size_t)n, c->stream));
  HIP_CHECK(hipMemcpyAsync(x, c->scratch, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  return RSYS_OK;
}
int32_t rsys_self_test(rsys_comm* c) {  // hardware_check.py:8-12
  CHECK_HANDLE(c);
  double one = 1.0;
  int rc = rsys_allreduce_f64(c, &one, 1);
  if (rc) return rc;
  if (one != (double)c->world) { set_error("all-reduce self test: sum of ones != world size"); return RSYS_ERR_COMM; }
  return RSYS_OK;
}

int32_t rsys_grad_buffer(rsys_model* h, void** p, int64_t* n) {
  CHECK_HANDLE(h);
  int rc = model_finalize_grads(h->m);
  if (rc) return rc;
  *p = h->m->G; *n = h->m->n_opt;
  return RSYS_OK;
}
int32_t rsys_param_buffer(rsys_model* h, void** p, int64_t* n) { CHECK_HANDLE(h); *p = h->m->P; *n = h->m->n_total; return RSYS_OK; }
int32_t rsys_refresh_shadow(rsys_model* h) { CHECK_HANDLE(h); return model_refresh_shadow(h->m); }

// ---------------------------------------------------------------- raw device helpers + per-op entry points (tests)
int32_t rsys_dev_alloc(void** p, size_t bytes) { HIP_CHECK(hipMalloc(p, bytes ? bytes : 16)); HIP_CHECK(hipMemset(*p, 0, bytes ? bytes : 16)); return RSYS_OK; }
int32_t rsys_dev_free(void* p) { HIP_CHECK(hipFree(p)); return RSYS_OK; }
int32_t rsys_dev_h2d(void* d, const void* s, size_t n) { HIP_CHECK(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return RSYS_OK; }
int32_t rsys_dev_d2h(void* d, const void* s, size_t n) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return RSYS_OK; }
int32_t rsys_dev_memset(void* d, int v, size_t n) { HIP_CHECK(hipMemset(d, v, n)); return RSYS_OK; }

int32_t rsys_op_gemm(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                     int64_t ldb, int64_t ldc, int32_t a_km, int32_t b_km, int32_t a_f32, int32_t c_f32, int32_t splitk) {
  switches_parse();
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.c_f32 = c_f32; p.splitk = splitk < 1 ? 1 : splitk; p.alpha = 1.f;
  p.epi = p.splitk > 1 ? EPI_ATOMIC : EPI_STORE;
  if (sw().debug_epi >= 0) p.epi = sw().debug_epi;   // timing experiments only (e.g. 99 = no epilogue)
  int rc = dtype == RSYS_DTYPE_BF16 ? launch_gemm<bf16>(p, a_f32 != 0, false, a_km != 0, b_km != 0, nullptr)
                                    : launch_gemm<float>(p, false, false, a_km != 0, b_km != 0, nullptr);
  if (rc) return rc;
  HIP_CHECK(hipDeviceSynchronize());
  return RSYS_OK;
}

int32_t rsys_op_gemm_rows(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                          int64_t ldb, int64_t ldc, int32_t b_km, int32_t c_f32, const int32_t* rows_dev) {
  switches_parse();
  ARG_CHECK(rows_dev != nullptr, "rsys_op_gemm_rows: rows_dev is null");
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.c_f32 = c_f32 != 0; p.splitk = 1; p.alpha = 1.f; p.epi = EPI_STORE; p.m_dev = rows_dev;
  if (c_f32 == 3) { p.epi = EPI_ATOMIC; p.splitk = 8; }   // the head's dEw form: fp32 C accumulated by split-K atomics
  int rc = dtype == RSYS_DTYPE_BF16 ? launch_gemm<bf16>(p, false, false, false, b_km != 0, nullptr)
                                    : launch_gemm<float>(p, false, false, false, b_km != 0, nullptr);
  if (rc) return rc;
  HIP_CHECK(hipDeviceSynchronize());
  return RSYS_OK;
}

int32_t rsys_op_gemm_klimit(int32_t dtype, const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int32_t accumulate, const int32_t* k_dev) {
  switches_parse();
  ARG_CHECK(k_dev != nullptr, "rsys_op_gemm_klimit: k_dev is null");
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.c_f32 = 1; p.splitk = 1; p.alpha = 1.f; p.epi = accumulate ? EPI_ACCUM : EPI_STORE; p.k_dev = k_dev;
  int rc = dtype == RSYS_DTYPE_BF16 ? launch_gemm<bf16>(p, false, false, true, true, nullptr)
                                    : launch_gemm<float>(p, false, false, true, true, nullptr);
  if (rc) return rc;
  HIP_CHECK(hipDeviceSynchronize());
  return RSYS_OK;
}

int32_t rsys_op_f8_quantize(const void* src, int64_t ld_src, int32_t rows, int32_t cols, int32_t fmt, int32_t layout, int32_t seg_cols,
                            int32_t seg_rep, void* dst, int64_t ld_dst, float* amax_dev, float* desc_dev, const float* wamax_dev,
                            int32_t n_w, int32_t w_rep, int32_t desc_mode) {
  switches_parse();
  ARG_CHECK(src && dst && amax_dev, "rsys_op_f8_quantize: null buffer");
  F8Cast c{};
  c.src = src; c.ld_src = ld_src; c.rows = rows; c.cols = cols; c.fmt = fmt; c.layout = layout; c.seg_cols = seg_cols; c.seg_rep = seg_rep < 1 ? 1 : seg_rep;
  c.amax = amax_dev; c.dst = (unsigned char*)dst; c.ld_dst = ld_dst; c.desc = desc_dev; c.wamax = wamax_dev; c.n_w = n_w; c.w_rep = w_rep < 1 ? 1 : w_rep;
  c.desc_mode = desc_mode;
  HIP_CHECK(hipMemsetAsync(amax_dev, 0, (size_t)F8_AMAX_SHARDS * F8_AMAX_SHARD * 4, nullptr));
  int rc = launch_f8_amax(c, nullptr);
  if (!rc) rc = launch_f8_cast(c, nullptr);
  if (rc) return rc;
  HIP_CHECK(hipDeviceSynchronize());
  return RSYS_OK;
}

int32_t rsys_op_f8_weights(const float* src, int64_t ld, int32_t rows, int32_t cols, int32_t layout, int32_t seg_rows, int32_t seg_rep,
                           float* amax_dev, void* dst, void* dst_t, int64_t ld_t) {
  switches_parse();
  ARG_CHECK(src && dst && amax_dev, "rsys_op_f8_weights: null buffer");
  ARG_CHECK(cols % 4 == 0 && rows % 16 == 0, "rsys_op_f8_weights: rows % 16, cols % 4");
  F8WeightJob j{};
  j.src = src; j.ld = ld; j.rows = rows; j.cols = cols; j.layout = layout; j.seg_rows = seg_rows; j.seg_rep = seg_rep < 1 ? 1 : seg_rep; j.amax = amax_dev;
  j.dst = (unsigned char*)dst; j.dst_t = (unsigned char*)dst_t; j.ld_t = ld_t;
  const int ntiles = ((rows + 63) / 64) * ((cols + 63) / 64);
  std::vector<int> tj(ntiles, 0); int first = 0;
  void* dev = nullptr;
  HIP_CHECK(hipMalloc(&dev, sizeof(F8WeightJob) + 256 + (size_t)ntiles * 4 + 64));
  F8WeightJob* dj = (F8WeightJob*)dev; int* dfirst = (int*)((char*)dev + ((sizeof(F8WeightJob) + 63) / 64) * 64); int* dtj = dfirst + 16;
  bool ok = hipMemcpy(dj, &j, sizeof(j), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dfirst, &first, 4, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(dtj, tj.data(), (size_t)ntiles * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemset(amax_dev, 0, 16) == hipSuccess;
  int rc = ok ? launch_f8_weights(dj, dtj, dfirst, ntiles, nullptr) : RSYS_ERR_HIP;
  hipError_t e2 = hipDeviceSynchronize();
  hipFree(dev);
  if (rc) { if (!ok) set_error("rsys_op_f8_weights: job upload failed"); return rc; }
  HIP_CHECK(e2);
  return RSYS_OK;
}

int32_t rsys_op_gemm_f8(const void* A8, const void* B8, void* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                        int32_t a_fmt, int32_t c_f32, const float* desc_dev, int32_t seg_cols, int32_t alt, int32_t kb0, int32_t kb1,
                        int32_t kb2) {
  switches_parse();
  GemmParams p{};
  p.A = A8; p.B = B8; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.c_f32 = c_f32; p.splitk = 1; p.alpha = 1.f; p.epi = EPI_STORE;
  p.f8 = a_fmt == F8_E5M2 ? 2 : 1; p.f8_desc = desc_dev; p.f8_seg_cols = seg_cols; p.f8_alt = alt; p.f8_kb[0] = kb0; p.f8_kb[1] = kb1; p.f8_kb[2] = kb2;
  int rc = launch_gemm8p_f8(p, nullptr);
  if (rc) return rc;
  HIP_CHECK(hipDeviceSynchronize());
  return RSYS_OK;
}

int32_t rsys_op_attention(int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t KV, int32_t hd, const void* qkv,
                          const int32_t* uid, const int32_t* tm, void* O, float* lse, const void* dO, void* dqkv,
                          const float* rope_cos, const float* rope_sin) {
  switches_parse();
  const size_t e = dtype == RSYS_DTYPE_BF16 ? 2 : 4;
  const int nt = (T + 63) / 64;
  AttnParams p{};
  p.B = B; p.T = T; p.H = H; p.KV = KV; p.hd = hd; p.is_bf16 = dtype == RSYS_DTYPE_BF16;
  p.q = qkv; p.k = (const unsigned char*)qkv + (size_t)H * hd * e; p.v = (const unsigned char*)qkv + (size_t)(H + KV) * hd * e;
  p.ld = (long long)(H + 2 * KV) * hd;
  p.o = O; p.ldo = (long long)H * hd; p.lse = lse; p.uid = uid; p.tm = tm;
  unsigned int* maps = nullptr; float* delta = nullptr;
  unsigned long long* pairbits = nullptr;
  HIP_CHECK(hipMalloc((void**)&maps, sizeof(unsigned int) * (12 * B * nt + (size_t)B * (2 * H + KV) * nt)));
  HIP_CHECK(hipMalloc((void**)&pairbits, sizeof(unsigned long long) * 2 * (size_t)B * nt * nt * 64));
  HIP_CHECK(hipMalloc((void**)&delta, sizeof(float) * B * H * T));
  p.qmap = maps; p.kmap = maps + B * nt; p.qmap_full = maps + 2 * B * nt; p.kmap_full = maps + 3 * B * nt; p.delta = delta;
  p.qmap16 = maps + 4 * B * nt; p.kmap16 = maps + 8 * B * nt;
  p.order_q = (int*)(maps + 12 * B * nt); p.order_k = p.order_q + (size_t)B * H * nt; p.order_q2 = p.order_k + (size_t)B * KV * nt;
  p.qbits = pairbits; p.kbits = pairbits + (size_t)B * nt * nt * 64;
  p.dO = dO; p.dq = dqkv; p.dk = (unsigned char*)dqkv + (size_t)H * hd * e;
  p.dv = (unsigned char*)dqkv + (size_t)(H + KV) * hd * e; p.ldg = p.ld;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_pos = nullptr;
  int rc = launch_attn_tilemap(p, nullptr);
  if (!rc) rc = dtype == RSYS_DTYPE_BF16 ? launch_attn_fwd<bf16>(p, nullptr) : launch_attn_fwd<float>(p, nullptr);
  if (!rc && dO) {
    rc = dtype == RSYS_DTYPE_BF16 ? launch_attn_bwd<bf16>(p, nullptr) : launch_attn_bwd<float>(p, nullptr);
  }
  hipError_t e2 = hipDeviceSynchronize();
  hipFree(maps); hipFree(delta); hipFree(pairbits);
  if (rc) return rc;
  HIP_CHECK(e2);
  return RSYS_OK;
}

// the backward's embedding scatter on caller-provided device buffers: gE[id'] += sum_n gx0[n * ldx ..+D) over the tokens whose
// masked id (m_matchedid, -1 -> row V) is id'; the token index is built from the raw ids (matchedid) as at batch upload.
// atomic != 0 runs the float-atomic form instead (A/B reference).
int32_t rsys_op_embedding_scatter(const float* gx0, int64_t ldx, const int32_t* matchedid, const int32_t* m_matchedid, int32_t N,
                                  int32_t V, int32_t D, float* gE, int32_t atomic) {
  switches_parse();
  ARG_CHECK(gx0 && matchedid && m_matchedid && gE && N >= 1 && ldx >= D, "rsys_op_embedding_scatter: arguments");
  if (atomic) {
    ARG_CHECK(ldx == 2LL * D, "the atomic form reads the interleaved layout (row stride 2 D)");
    BatchDev b{}; b.N = N; b.m_matchedid = const_cast<int*>(m_matchedid);
    int rc = launch_embedding_scatter_add(gx0, b, V, D, gE, nullptr);
    if (rc) return rc;
    HIP_CHECK(hipDeviceSynchronize());
    return RSYS_OK;
  }
  unsigned long long* keys = nullptr; int *skey = nullptr, *sidx = nullptr; float* slab = nullptr;
  HIP_CHECK(hipMalloc((void**)&keys, (size_t)token_index_capacity(N) * 8));
  HIP_CHECK(hipMalloc((void**)&skey, (size_t)N * 4));
  HIP_CHECK(hipMalloc((void**)&sidx, (size_t)N * 4));
  HIP_CHECK(hipMalloc((void**)&slab, seg_scatter_slab_floats(N, D) * 4));
  int rc = launch_token_index_build(matchedid, N, V, keys, skey, sidx, nullptr);
  if (!rc) rc = launch_embedding_scatter_segmented(gx0, ldx, m_matchedid, skey, sidx, N, V, D, gE, slab, nullptr);
  hipError_t e = hipDeviceSynchronize();
  hipFree(keys); hipFree(skey); hipFree(sidx); hipFree(slab);
  if (rc) return rc;
  HIP_CHECK(e);
  return RSYS_OK;
}

// step boundaries on the model's stream: rsys_step_mark records an event, rsys_step_marks_get returns the elapsed time
// between consecutive marks (ms) and clears them -- the per-step time distribution without a host sync per step
int32_t rsys_step_mark(rsys_model* h) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventRecord(e, m->stream));
  m->step_marks.push_back(e);
  return RSYS_OK;
}
int32_t rsys_step_marks_get(rsys_model* h, float* ms_out, int32_t cap, int32_t* n_out) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  int n = 0;
  for (size_t i = 1; i < m->step_marks.size() && n < cap; ++i, ++n) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, m->step_marks[i - 1], m->step_marks[i]));
    if (ms_out) ms_out[n] = ms;
  }
  for (hipEvent_t e : m->step_marks) hipEventDestroy(e);
  m->step_marks.clear();
  if (n_out) *n_out = n;
  return RSYS_OK;
}

int32_t rsys_switches_reload(void) { switches_parse(); return RSYS_OK; }

int32_t rsys_switches_describe(char* buf, int32_t cap) { return switches_describe(buf, cap); }

int32_t rsys_op_timing(rsys_model* h, int32_t enable) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  if (enable == 3) { m->timer.enabled = false; return RSYS_OK; }   // pause: stop recording, keep what was recorded, no host wait (read it later with rsys_timing_get)
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  m->timer.enabled = enable != 0;
  m->timer.serialize = enable == 2;
  HIP_CHECK(hipStreamSynchronize(m->side));
  m->timer.marks.clear(); m->timer.acc_ms.clear(); m->timer.used = 0; m->timer.open.clear();
  if (!enable) m->timer.filter.clear();
  return RSYS_OK;
}

int32_t rsys_op_timing_filter(rsys_model* h, const char* substr) {
  CHECK_HANDLE(h);
  h->m->timer.filter = substr ? substr : "";
  return RSYS_OK;
}

// "name ms count flops\n" per timed span since rsys_op_timing(1); nested spans are supported
int32_t rsys_timing_get(rsys_model* h, char* buf, size_t cap) {
  CHECK_HANDLE(h);
  Model* m = h->m;
  HIP_CHECK(hipSetDevice(m->device));
  HIP_CHECK(hipStreamSynchronize(m->stream));
  HIP_CHECK(hipStreamSynchronize(m->side));
  PhaseTimer& t = m->timer;
  std::map<std::string, std::pair<double, long>> agg;
  std::vector<size_t> stack;
  for (size_t i = 0; i < t.marks.size(); ++i) {
    if (!t.marks[i].first.empty()) { stack.push_back(i); continue; }
    if (stack.empty()) continue;
    size_t b = stack.back(); stack.pop_back();
    float ms = 0.f;
    hipEventElapsedTime(&ms, t.marks[b].second, t.marks[i].second);
    auto& a = agg[t.marks[b].first];
    a.first += ms; a.second += 1;
  }
  std::ostringstream os;
  for (auto& kv : agg) {
    double fl = 0.0;
    auto it = t.acc_ms.find(std::string("#flops:") + kv.first);
    if (it != t.acc_ms.end()) fl = it->second;
    os << kv.first << " " << kv.second.first << " " << kv.second.second << " " << fl << "\n";
  }
  std::string s = os.str();
  if (buf && cap) { strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
  t.marks.clear(); t.acc_ms.clear(); t.used = 0; t.open.clear();
  return (int32_t)RSYS_OK;
}

}  // extern "C"
