// Gradient norm, clipping and the fused AdamW step (transformer.py:256-276), replicated and ZeRO-1 forms.  Split from model.hip in round 5.
#include "model_internal.hpp"

namespace rsys {

// sum of squares of all gradients into m->sumsq.  Row-sharded table: the replicated gradients are identical on every rank
// (after the all-reduce), the table rows differ: their sum of squares is all-reduced and added.
static int grad_sumsq(Model* m) {   // (the first launch of a scalar SETS it: no zero fill in front)
  if (!m->sharded || !comm_active(m->shard_comm)) return launch_sumsq(m->G, m->n_opt, m->sumsq, m->sumsq_part, m->stream, true);
  const int64_t e0 = m->o_E, e1 = m->o_E + (int64_t)m->TR * m->D;
  RC(launch_sumsq(m->G, e0, m->sumsq, m->sumsq_part, m->stream, true));
  RC(launch_sumsq(m->G + e1, m->n_opt - e1, m->sumsq, m->sumsq_part, m->stream));
  RC(launch_sumsq(m->G + e0, e1 - e0, m->sumsq_E, m->sumsq_part, m->stream, true));
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
  if (o->z_tailbuf) { HIP_CHECK(hipFree(o->z_tailbuf)); o->z_tailbuf = nullptr; }   // (a repeated call with another world size)
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
    RC(launch_sumsq(m->G + lo, chunk, m->sumsq, m->sumsq_part, s, true));
    if (last && tail > 0) RC(launch_sumsq(m->G + tail_lo, tail, m->sumsq, m->sumsq_part, s));
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
