// Row-sharded item table (an extension beyond the reference, which replicates the table and
// all-reduces its dense gradient, transformer.py:678-682).  Rank r owns the contiguous table rows [lo_r, hi_r) of
// E / Meta / the fused table F and their Adam moments.  Kernels of the three places where a rank meets rows it
// does not own:
//  * token path: the batch's distinct item ids are fetched from their owners once per step (exchange plan built at
//    batch upload from the sorted token index) and the token gradients go back the same way as one row per distinct
//    id -- a sparse row exchange instead of the dense table gradient in the all-reduce;
//  * watch head: vocabulary-parallel cross entropy.  The selected rows of every rank are all-gathered, each rank
//    forms the logits against its own rows of F, and (max, sum-exp, target logit) are all-reduced: the gradient of
//    the local rows of F is complete locally (no all-reduce), the gradient of the selected rows is summed over ranks.
#include "kernels.hpp"

namespace rsys {

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ exchange plan (once per uploaded batch)
// skey sorted ascending (raw ids, -1 -> V).  slot[p] = rank of skey[p] among the distinct keys; uniq[slot] = key;
// tok2u[token] = slot.  The mask row V always gets a slot (watch-masked tokens read it): plan = {U, uV}.
__global__ __launch_bounds__(1024) void plan_unique_kernel(const int* __restrict__ skey, const int* __restrict__ sidx, int N, int V,
                                                           int* slot, int* uniq, int* tok2u, int* plan) {
  __shared__ int wave_tot[16];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int per = (N + 1023) / 1024;
  const int p0 = t * per, p1 = min(N, p0 + per);
  int cnt = 0;
  for (int p = p0; p < p1; ++p) cnt += (p == 0 || skey[p] != skey[p - 1]);
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
  if (l == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, total = 0;
  for (int k = 0; k < 16; ++k) { int v = wave_tot[k]; if (k < wv) base += v; total += v; }
  int s = base + inc - cnt - 1;   // slot of the last head before p0
  for (int p = p0; p < p1; ++p) {
    if (p == 0 || skey[p] != skey[p - 1]) { ++s; uniq[s] = skey[p]; }
    slot[p] = s;
    tok2u[sidx[p]] = s;
  }
  if (t == 0) {
    int U = total, uV = total - 1;
    if (skey[N - 1] != V) { uniq[total] = V; uV = total; U = total + 1; }
    plan[0] = U; plan[1] = uV;
  }
}
int launch_plan_unique(const int* skey, const int* sidx, int N, int V, int* slot, int* uniq, int* tok2u, int* plan, hipStream_t s) {
  ARG_CHECK(N >= 1 && N <= (1 << 20), "exchange plan: token count");
  hipLaunchKernelGGL(plan_unique_kernel, dim3(1), dim3(1024), 0, s, skey, sidx, N, V, slot, uniq, tok2u, plan);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// off[o] = first unique slot whose id is >= bound[o] (bound[world] = V + 1 -> U): ids owned by rank o are slots [off[o], off[o+1])
__global__ void plan_offsets_kernel(const int* __restrict__ uniq, const int* __restrict__ plan, const int* __restrict__ bound, int nb, int* off) {
  const int o = threadIdx.x;
  if (o >= nb) return;
  const int U = plan[0], b = bound[o];
  int lo = 0, hi = U;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (uniq[mid] < b) lo = mid + 1; else hi = mid; }
  off[o] = lo;
}
int launch_plan_offsets(const int* uniq, const int* plan, const int* bound_dev, int nb, int* off, hipStream_t s) {
  ARG_CHECK(nb >= 2 && nb <= 64, "exchange plan: ranks");
  hipLaunchKernelGGL(plan_offsets_kernel, dim3(1), dim3(64), 0, s, uniq, plan, bound_dev, nb, off);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// ------------------------------------------------------------------ rows by id
// dst[j] = src[(ids[j] - sub) * ld ..+D)   (owner side: the rows its peers asked for)
__global__ void gather_rows_by_id_kernel(const float* __restrict__ src, long long ld, const int* __restrict__ ids, int sub, float* dst, int n, int D) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (j >= n) return;
  const float4* s4 = (const float4*)(src + (long long)(ids[j] - sub) * ld);
  float4* d4 = (float4*)(dst + (long long)j * D);
  for (int c = l; c < (D >> 2); c += 64) d4[c] = s4[c];
}
int launch_gather_rows_by_id(const float* src, long long ld, const int* ids, int sub, float* dst, int n, int D, hipStream_t s) {
  if (n == 0) return RSYS_OK;
  hipLaunchKernelGGL(gather_rows_by_id_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, src, ld, ids, sub, dst, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
// dst[(ids[j] - sub)] += src[j]; the ids of one call are distinct (one requester's list): plain read-modify-write
__global__ void add_rows_by_id_kernel(const float* __restrict__ src, const int* __restrict__ ids, int sub, float* dst, long long ld, int n, int D) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (j >= n) return;
  const float4* s4 = (const float4*)(src + (long long)j * D);
  float4* d4 = (float4*)(dst + (long long)(ids[j] - sub) * ld);
  for (int c = l; c < (D >> 2); c += 64) {
    float4 a = d4[c]; const float4 b = s4[c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[c] = a;
  }
}
int launch_add_rows_by_id(const float* src, const int* ids, int sub, float* dst, long long ld, int n, int D, hipStream_t s) {
  if (n == 0) return RSYS_OK;
  hipLaunchKernelGGL(add_rows_by_id_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, src, ids, sub, dst, ld, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// the same with the row count on the device (split table reduce: a peer's list of distinct token rows, its length in plan[0])
__global__ void add_rows_by_id_counted_kernel(const float* __restrict__ src, const int* __restrict__ ids, const int* __restrict__ count,
                                              float* dst, long long ld, int cap, int D) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (j >= cap || j >= count[0]) return;
  const float4* s4 = (const float4*)(src + (long long)j * D);
  float4* d4 = (float4*)(dst + (long long)ids[j] * ld);
  for (int c = l; c < (D >> 2); c += 64) {
    float4 a = d4[c]; const float4 b = s4[c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[c] = a;
  }
}
int launch_add_rows_by_id_counted(const float* src, const int* ids, const int* count_dev, float* dst, long long ld, int cap, int D, hipStream_t s) {
  if (cap == 0) return RSYS_OK;
  hipLaunchKernelGGL(add_rows_by_id_counted_kernel, dim3(div_up(cap, 4)), dim3(256), 0, s, src, ids, count_dev, dst, ld, cap, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// token gather from the fetched rows: x0[2n] = Frem[masked ? uV : tok2u[n]] (model.py:23-24,139-145) + per-token uid / tm
__global__ void gather_items_remote_kernel(BatchDev b, const float* __restrict__ Frem, const int* __restrict__ tok2u,
                                           const int* __restrict__ plan, int D, float* x0, int* uid_t, int* tm_t) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (wave >= b.N) return;
  const int u = b.m_matchedid[wave] == -1 ? plan[1] : tok2u[wave];
  const float4* src = (const float4*)(Frem + (long long)u * D);
  float4* dst = (float4*)(x0 + (long long)(2 * wave) * D);
  for (int c = l; c < (D >> 2); c += 64) dst[c] = src[c];
  if (l == 0) {
    const int uid = b.userid[wave], tmv = b.m_tmid[wave];
    uid_t[2 * wave] = uid; uid_t[2 * wave + 1] = uid;
    tm_t[2 * wave] = tmv; tm_t[2 * wave + 1] = tmv;
  }
}
int launch_gather_items_remote(const BatchDev& b, const float* Frem, const int* tok2u, const int* plan, int D, float* x0,
                               int* uid_t, int* tm_t, hipStream_t s) {
  hipLaunchKernelGGL(gather_items_remote_kernel, dim3(div_up(b.N, 4)), dim3(256), 0, s, b, Frem, tok2u, plan, D, x0, uid_t, tm_t);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

__global__ void add_scalar_kernel(float* dst, const float* src) { if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] += src[0]; }
int launch_add_scalar(float* dst, const float* src, hipStream_t s) {
  hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(64), 0, s, dst, src);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// ------------------------------------------------------------------ vocabulary-parallel cross entropy
// meta of a selected row: {target id inside the medium, label * weight, gradient coefficient tw * label * weight / max(w_sum, 1e-8)}
// own[KBmax][4] floats (target as int bits; rows >= KB, the batch's selected rows, are zero); own[KBmax * 4] = npos (int bits)
__global__ void vp_meta_kernel(const int* __restrict__ idx, const float* __restrict__ label, const float* __restrict__ weight,
                               const int* __restrict__ position, const float* __restrict__ stats, const int* __restrict__ npos,
                               float task_w, int KB, int KBmax, float* own) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) own[(long long)KBmax * 4] = __int_as_float(*npos);
  if (j >= KBmax) return;
  if (j >= KB) { own[4 * j + 0] = 0.f; own[4 * j + 1] = 0.f; own[4 * j + 2] = 0.f; own[4 * j + 3] = 0.f; return; }
  const int i = idx[j];
  const float lw = label[i] * weight[i];
  own[4 * j + 0] = __int_as_float(position[i]);
  own[4 * j + 1] = lw;
  own[4 * j + 2] = task_w * lw / fmaxf(stats[0], 1e-8f);
  own[4 * j + 3] = 0.f;
}
int launch_vp_meta(const int* idx, const float* label, const float* weight, const int* position, const float* stats,
                   const int* npos, float task_w, int KB, int KBmax, float* own, hipStream_t s) {
  hipLaunchKernelGGL(vp_meta_kernel, dim3(div_up(KBmax, 256)), dim3(256), 0, s, idx, label, weight, position, stats, npos, task_w, KB, KBmax, own);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// live rows of every rank, packed: row (q, i) with i < npos_q goes to pre_q + i.  all[q] = one rank's block
// ([KB][D] rows, [KB * 4 + 4] meta floats).  Rows [nlive, nlive rounded up to 256) get zero meta (GEMM tile padding).
template <typename T>
__global__ void vp_compact_kernel(const T* __restrict__ EwAll, const float* __restrict__ metaAll, int W, int KB, int D,
                                  T* EwC, float* metaC, int* nlive_out, int* pre_out) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  const long long mstride = (long long)KB * 4 + 4;
  int pre = 0, q = -1, i = 0, nlive = 0;
  for (int r = 0; r < W; ++r) {
    const int np = __float_as_int(metaAll[r * mstride + (long long)KB * 4]);
    if (q < 0 && row < nlive + np) { q = r; i = row - nlive; pre = nlive; }
    nlive += np;
  }
  if (row == 0 && l == 0) {
    *nlive_out = nlive;
    int acc = 0;
    for (int r = 0; r < W; ++r) { pre_out[r] = acc; acc += __float_as_int(metaAll[r * mstride + (long long)KB * 4]); }
    pre_out[W] = acc;
  }
  const int pad_end = (nlive + 255) & ~255;
  if (row >= pad_end || row >= W * KB) return;
  if (row >= nlive) { if (l < 4) metaC[4LL * row + l] = 0.f; return; }
  constexpr int E = 16 / sizeof(T);
  const uint4* s4 = (const uint4*)(EwAll + ((long long)q * KB + i) * D);
  uint4* d4 = (uint4*)(EwC + (long long)row * D);
  for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
  if (l < 4) metaC[4LL * row + l] = metaAll[q * mstride + 4LL * i + l];
  (void)pre;
}
template <typename T>
int launch_vp_compact(const T* EwAll, const float* metaAll, int W, int KB, int D, T* EwC, float* metaC, int* nlive, int* pre, hipStream_t s) {
  hipLaunchKernelGGL((vp_compact_kernel<T>), dim3(div_up((long long)W * KB, 4)), dim3(256), 0, s, EwAll, metaAll, W, KB, D, EwC, metaC, nlive, pre);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_vp_compact<bf16>(const bf16*, const float*, int, int, int, bf16*, float*, int*, int*, hipStream_t);
template int launch_vp_compact<float>(const float*, const float*, int, int, int, float*, float*, int*, int*, hipStream_t);

// pass 1 (one read of the local logits, 16-byte loads): online local maximum and sum-exp relative to it, the target's
// logit if the target column is local.  lmax[row] (rows >= *nlive / no local columns: -3e38, the neutral element of the max
// all-reduce), sums[row] = local sum-exp, sums[cap + row] = target logit or 0.
template <typename T>
__global__ __launch_bounds__(256) void vp_stats_kernel(const T* __restrict__ logits, long long ldl, int Vloc, int col0,
                                                       const float* __restrict__ metaC, const int* __restrict__ nlive,
                                                       float* lmax, float* sums, int cap) {
  __shared__ float red[16];
  constexpr int E = 16 / sizeof(T);
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= *nlive || Vloc <= 0) { if (t == 0) { lmax[row] = -3.0e38f; sums[row] = 0.f; sums[cap + row] = 0.f; } return; }
  const T* lr = logits + (long long)row * ldl;
  const int nchunks = (Vloc + E - 1) / E;
  float m = -3.0e38f, ssum = 0.f;
  for (int c = t; c < nchunks; c += 256) {
    const uint4 raw = ((const uint4*)lr)[c];
    const T* e = (const T*)&raw;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (c * E + k < Vloc) {
        const float x = to_f32(e[k]);
        if (x > m) { ssum = ssum * __expf(m - x) + 1.f; m = x; }
        else ssum += __expf(x - m);
      }
    }
  }
  const float gm = block_max(m, red);
  ssum = block_sum(ssum * __expf(m - gm), red);
  if (t == 0) {
    const int tgt = __float_as_int(metaC[4LL * row]) - col0;
    lmax[row] = gm; sums[row] = ssum;
    sums[cap + row] = (tgt >= 0 && tgt < Vloc) ? to_f32(lr[tgt]) : 0.f;
  }
}
// between the two all-reduces: the local sum-exp rebased from the local to the global maximum
__global__ void vp_rebase_kernel(const float* __restrict__ lmax, const float* __restrict__ gmax, float* sums, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sums[i] *= __expf(lmax[i] - gmax[i]);
}
// pass 2: own rows add (lse - target logit) * label * weight to the loss; every live row's logits become
// dlogits = coef * (softmax - onehot) over the local columns; padding columns up to ldl are zeroed
template <typename T>
__global__ __launch_bounds__(256) void vp_finish_kernel(T* logits, long long ldl, int Vloc, int col0, const float* __restrict__ metaC,
                                                        const float* __restrict__ gmax, const float* __restrict__ sums, int cap,
                                                        const int* __restrict__ nlive, const int* __restrict__ pre, int rank,
                                                        float* loss_out, float* part) {
  constexpr int E = 16 / sizeof(T);
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= ((*nlive + 255) & ~255)) return;
  T* lr = logits + (long long)row * ldl;
  const int nchunks = (int)(ldl / E);
  if (row >= *nlive) { for (int c = t; c < nchunks; c += 256) ((uint4*)lr)[c] = make_uint4(0, 0, 0, 0); return; }
  const float lse = gmax[row] + logf(sums[row]);
  const float lw = metaC[4LL * row + 1], coef = metaC[4LL * row + 2];
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  if (t == 0 && row >= pre[rank] && row < pre[rank + 1] && lw != 0.f) {
    if (part != nullptr) part[row] = (lse - sums[cap + row]) * lw;   // deterministic mode: the rows' terms are added in row order afterwards
    else atomicAdd(loss_out, (lse - sums[cap + row]) * lw);
  }
  for (int c = t; c < nchunks; c += 256) {
    uint4 raw = ((const uint4*)lr)[c];
    T* e = (T*)&raw;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int col = c * E + k;
      const float g = col < Vloc ? coef * (__expf(to_f32(e[k]) - lse) - (col == tgt ? 1.f : 0.f)) : 0.f;
      e[k] = from_f32<T>(g);
    }
    ((uint4*)lr)[c] = raw;
  }
}
template <typename T>
int launch_vp_stats(const T* logits, long long ldl, int Vloc, int col0, const float* metaC, const int* nlive, float* lmax,
                    float* sums, int cap, int grid_rows, hipStream_t s) {
  ARG_CHECK((ldl * sizeof(T)) % 16 == 0, "vocabulary-parallel ce: logits rows must be 16-byte multiples");
  hipLaunchKernelGGL((vp_stats_kernel<T>), dim3(grid_rows), dim3(256), 0, s, logits, ldl, Vloc, col0, metaC, nlive, lmax, sums, cap);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
int launch_vp_rebase(const float* lmax, const float* gmax, float* sums, int n, hipStream_t s) {
  hipLaunchKernelGGL(vp_rebase_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, lmax, gmax, sums, n);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template <typename T>
int launch_vp_finish(T* logits, long long ldl, int Vloc, int col0, const float* metaC, const float* gmax, const float* sums, int cap,
                     const int* nlive, const int* pre, int rank, float* loss_out, int grid_rows, hipStream_t s) {
  float* part = g_det.part != nullptr && (long long)grid_rows <= g_det.cap ? g_det.part : nullptr;   // deterministic mode (kernels.hpp DetScratch)
  if (g_det.part != nullptr && part == nullptr) { set_error("vocabulary-parallel CE: deterministic scratch too small"); return RSYS_ERR_STATE; }
  if (part != nullptr) HIP_CHECK(hipMemsetAsync(part, 0, (size_t)grid_rows * 4, s));   // (rows without a term write nothing)
  hipLaunchKernelGGL((vp_finish_kernel<T>), dim3(grid_rows), dim3(256), 0, s, logits, ldl, Vloc, col0, metaC, gmax, sums, cap, nlive, pre, rank, loss_out, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, grid_rows, 1, 1, loss_out, s);
  return RSYS_OK;
}
// ------------------------------------------------------------------ sampled softmax (cfg-4 option; no reference counterpart)
// Per (step, medium) a rank draws n_s of its `len` local classes by stratified sampling: stratum j = classes
// [floor(j len / n_s), floor((j + 1) len / n_s)), one class uniformly from each (distinct, sorted).  A class of a stratum of
// s_j classes is included with probability exactly 1 / s_j, so the partition function of a row with target t is estimated by
//     Z ~= exp(l_t) + sum over ranks sum_{j : c_j != t} s_j exp(l_{c_j}),     loss = log Z - l_t
// (importance weighting = the log-Q correction with the TRUE inclusion probability of each sampled class; a sampled class that
// IS the row's target is skipped: the target's term is exact).  The weights must be the strata's own sizes: one global factor
// len / n_s over strata of unequal size leaves the classes of the larger strata under-weighted for good, and a peaked model
// trained that way drifts (tools/converge_sampled.py).
__device__ __forceinline__ int ss_stratum_lo(int j, int len, int n_s) { return (int)((long long)j * len / n_s); }
__global__ void ss_sample_kernel(int len, int n_s, unsigned long long seed, unsigned int stream, int* cols) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_s) return;
  Philox ph(seed);
  uint32_t r[4];
  ph.gen((unsigned long long)j, stream, r);
  const int lo = ss_stratum_lo(j, len, n_s), size = ss_stratum_lo(j + 1, len, n_s) - lo;   // size >= 1 (n_s <= len)
  cols[j] = lo + min(size - 1, (int)(u01(r[0]) * (float)size));
}
int launch_ss_sample(int len, int n_s, unsigned long long seed, unsigned int stream, int* cols, hipStream_t s) {
  ARG_CHECK(n_s >= 1 && n_s <= len, "sampled softmax: 1 <= samples <= local classes");
  hipLaunchKernelGGL(ss_sample_kernel, dim3(div_up(n_s, 256)), dim3(256), 0, s, len, n_s, seed, stream, cols);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// In-batch targets.  The classes that ARE some live row's target this step are the ones that receive a pull-up gradient; with
// uniform sampling alone each of them would be pushed down only every 1/q-th step (with weight 1/q), and on a Zipf-shaped
// vocabulary that variance is what makes a small-sample run drift.  So every local class that is a target of the gathered rows
// joins the class list of the step deterministically (weight 1), behind the n_s sampled ones; a sampled class that is also in
// that set is dropped (index -1: zero row, zero gradient), so the estimator stays unbiased:
//     Z ~= exp(l_t) + sum_{c in T, c != t} exp(l_c) + sum_j s_j exp(l_{c_j}) [c_j not in T].
__global__ void ss_mark_targets_kernel(const float* __restrict__ metaC, const int* __restrict__ nlive, int len, int col0, unsigned int* bitmap) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= *nlive) return;
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  if (tgt >= 0 && tgt < len) atomicOr(&bitmap[tgt >> 5], 1u << (tgt & 31));
}
// ascending list of the marked classes -> out[0 .. *count); one workgroup of 1024 threads, a contiguous run of words each
__global__ __launch_bounds__(1024) void ss_list_targets_kernel(const unsigned int* __restrict__ bitmap, int words, int* out, int* count) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (words + 1023) / 1024, w0 = t * per, w1 = min(words, w0 + per);
  int n = 0;
  for (int w = w0; w < w1; ++w) n += __popc(bitmap[w]);
  part[t] = n;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int at = part[t] - n;
  for (int w = w0; w < w1; ++w) {
    unsigned int b = bitmap[w];
    while (b) { const int bit = __ffs(b) - 1; out[at++] = w * 32 + bit; b &= b - 1; }
  }
  if (t == 1023) *count = part[1023];
}
__global__ void ss_drop_hits_kernel(int* cols, int n_s, const unsigned int* __restrict__ bitmap) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_s) return;
  const int c = cols[j];
  if ((bitmap[c >> 5] >> (c & 31)) & 1u) cols[j] = -1;
}
int launch_ss_targets(const float* metaC, const int* nlive, int cap_rows, int len, int col0, unsigned int* bitmap, int* out, int* count, hipStream_t s) {
  const int words = (len + 31) / 32;
  HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)words * 4, s));
  hipLaunchKernelGGL(ss_mark_targets_kernel, dim3(div_up(cap_rows, 256)), dim3(256), 0, s, metaC, nlive, len, col0, bitmap);
  hipLaunchKernelGGL(ss_list_targets_kernel, dim3(1), dim3(1024), 0, s, bitmap, words, out, count);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
int launch_ss_drop_hits(int* cols, int n_s, const unsigned int* bitmap, hipStream_t s) {
  hipLaunchKernelGGL(ss_drop_hits_kernel, dim3(div_up(n_s, 256)), dim3(256), 0, s, cols, n_s, bitmap);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// dst[j] = src[(base + idx[j]) * ld ..+D)  (T rows, 16-byte copies); idx[j] < 0: zeros
template <typename T>
__global__ void gather_rows_plain_kernel(const T* __restrict__ src, long long ld, const int* __restrict__ idx, int base, T* dst, int n, int D) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (j >= n) return;
  constexpr int E = 16 / sizeof(T);
  uint4* d4 = (uint4*)(dst + (long long)j * D);
  if (idx[j] < 0) { for (int c = l; c < D / E; c += 64) d4[c] = make_uint4(0, 0, 0, 0); return; }   // dropped entry: a zero row
  const uint4* s4 = (const uint4*)(src + (long long)(base + idx[j]) * ld);
  for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
}
template <typename T>
int launch_gather_rows_plain(const T* src, long long ld, const int* idx, int base, T* dst, int n, int D, hipStream_t s) {
  hipLaunchKernelGGL((gather_rows_plain_kernel<T>), dim3(div_up(n, 4)), dim3(256), 0, s, src, ld, idx, base, dst, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
// dst[(base + idx[j])] += src[j]  (f32 rows; the non-negative idx are distinct, negative ones are skipped)
__global__ void add_rows_plain_kernel(const float* __restrict__ src, const int* __restrict__ idx, int base, float* dst, long long ld, int n, int D) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (j >= n || idx[j] < 0) return;
  const float4* s4 = (const float4*)(src + (long long)j * D);
  float4* d4 = (float4*)(dst + (long long)(base + idx[j]) * ld);
  for (int c = l; c < (D >> 2); c += 64) {
    float4 a = d4[c]; const float4 b = s4[c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[c] = a;
  }
}
int launch_add_rows_plain(const float* src, const int* idx, int base, float* dst, long long ld, int n, int D, hipStream_t s) {
  hipLaunchKernelGGL(add_rows_plain_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, src, idx, base, dst, ld, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// tl[row] = <selected row, F[target]> where the target class is local (one owner per row), else 0; one wave per row
template <typename T>
__global__ void ss_target_logit_kernel(const T* __restrict__ EwC, const T* __restrict__ Floc, int D, int len, int col0,
                                       const float* __restrict__ metaC, const int* __restrict__ nlive, float* tl) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= *nlive) return;
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  float acc = 0.f;
  if (tgt >= 0 && tgt < len) {
    const T* a = EwC + (long long)row * D; const T* b = Floc + (long long)tgt * D;
    for (int c = l; c < D; c += 64) acc += to_f32(a[c]) * to_f32(b[c]);
    acc = wave_sum(acc);
  }
  if (l == 0) tl[row] = acc;
}
// local maximum / sum-exp over the sampled classes, each weighted by its stratum's size (l + log s_j), the accidental hit
// (sampled class == the row's target) left out
template <typename T>
__global__ __launch_bounds__(256) void ss_stats_kernel(const T* __restrict__ logits, long long ldl, int n_s, int n_tot, int len, int col0,
                                                       const int* __restrict__ cols, const float* __restrict__ metaC,
                                                       const int* __restrict__ nlive, float* lmax, float* lsum) {
  __shared__ float red[16];
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= *nlive || n_tot <= 0) { if (t == 0) { lmax[row] = -3.0e38f; lsum[row] = 0.f; } return; }
  const T* lr = logits + (long long)row * ldl;
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  float m = -3.0e38f, ssum = 0.f;
  for (int c = t; c < n_tot; c += 256) {
    if (cols[c] == tgt || cols[c] < 0) continue;
    float x = to_f32(lr[c]);
    if (c < n_s) x += __logf((float)(ss_stratum_lo(c + 1, len, n_s) - ss_stratum_lo(c, len, n_s)));   // + log s_j (in-batch targets: weight 1)
    if (x > m) { ssum = ssum * __expf(m - x) + 1.f; m = x; }
    else ssum += __expf(x - m);
  }
  const float gm = block_max(m, red);
  ssum = block_sum(ssum * __expf(m - gm), red);
  if (t == 0) { lmax[row] = gm; lsum[row] = ssum; }
}
// lmax2 = max(lmax, tl): what the max all-reduce starts from (tl has been all-reduced: every rank knows the target logit)
__global__ void ss_max_with_target_kernel(const float* __restrict__ lmax, const float* __restrict__ tl, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fmaxf(lmax[i], tl[i]);
}
// sneg = lsum * exp(lmax - gmax): this rank's share of the negatives' (weighted) partition sum (before the sum all-reduce)
__global__ void ss_rebase_kernel(const float* __restrict__ lmax, const float* __restrict__ gmax, float* lsum, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lsum[i] = lsum[i] > 0.f ? lsum[i] * __expf(lmax[i] - gmax[i]) : 0.f;
}
// lse = gmax + log(exp(tl - gmax) + sneg); own rows add (lse - tl) lw to the loss; dlogits over the sampled classes
// (coef s_j exp(l - lse), 0 at an accidental hit); dt[row] = coef (exp(tl - lse) - 1) for the target's owner
template <typename T>
__global__ __launch_bounds__(256) void ss_finish_kernel(T* logits, long long ldl, int n_s, int n_tot, int len, int col0, const int* __restrict__ cols,
                                                        const float* __restrict__ metaC, const float* __restrict__ gmax,
                                                        const float* __restrict__ sneg, const float* __restrict__ tl,
                                                        const int* __restrict__ nlive, const int* __restrict__ pre, int rank,
                                                        float* loss_out, float* dt, float* part) {
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= ((*nlive + 255) & ~255)) return;
  T* lr = logits + (long long)row * ldl;
  if (row >= *nlive) { for (int c = t; c < (int)ldl; c += 256) lr[c] = from_f32<T>(0.f); if (t == 0) dt[row] = 0.f; return; }
  const float gm = gmax[row];
  const float lse = gm + logf(__expf(tl[row] - gm) + sneg[row]);
  const float lw = metaC[4LL * row + 1], coef = metaC[4LL * row + 2];
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  if (t == 0) {
    dt[row] = coef * (__expf(tl[row] - lse) - 1.f);
    if (row >= pre[rank] && row < pre[rank + 1] && lw != 0.f) {
      if (part != nullptr) part[row] = (lse - tl[row]) * lw;   // deterministic mode: the rows' terms are added in row order afterwards
      else atomicAdd(loss_out, (lse - tl[row]) * lw);
    }
  }
  for (int c = t; c < (int)ldl; c += 256) {
    float g = 0.f;
    if (c < n_tot && cols[c] != tgt && cols[c] >= 0)
      g = coef * __expf(to_f32(lr[c]) - lse) * (c < n_s ? (float)(ss_stratum_lo(c + 1, len, n_s) - ss_stratum_lo(c, len, n_s)) : 1.f);
    lr[c] = from_f32<T>(g);
  }
}
// target class gradients by the target's owner: dF[target] += dt * (selected row)  (rows may share a target: float atomics),
// d(selected row) += dt * F[target]  (one owner per row: plain).  One wave per live row.
template <typename T>
__global__ void ss_target_grad_kernel(const T* __restrict__ EwC, const T* __restrict__ Floc, int D, int len, int col0,
                                      const float* __restrict__ metaC, const float* __restrict__ dt, const int* __restrict__ nlive,
                                      float* gE_loc, float* dEwC) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= *nlive) return;
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  const float g = dt[row];
  if (tgt < 0 || tgt >= len || g == 0.f) return;
  const T* a = EwC + (long long)row * D; const T* b = Floc + (long long)tgt * D;
  float* gf = gE_loc + (long long)tgt * D; float* ge = dEwC + (long long)row * D;
  for (int c = l; c < D; c += 64) { atomicAdd(&gf[c], g * to_f32(a[c])); ge[c] += g * to_f32(b[c]); }
}

// The same without atomics (deterministic mode): the FIRST live row of a target class adds the terms of all rows that share it, in
// row order, and is the only writer of that class's gradient row; every row still adds its own d(selected row).  One wave per live
// row; the lanes sweep the row list 64 at a time (a compare and a ballot per 64 rows), D <= 64 * DMAX.
template <typename T, int DMAX>
__global__ void ss_target_grad_ordered_kernel(const T* __restrict__ EwC, const T* __restrict__ Floc, int D, int len, int col0,
                                              const float* __restrict__ metaC, const float* __restrict__ dt, const int* __restrict__ nlive,
                                              float* gE_loc, float* dEwC) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  const int n = *nlive;
  if (row >= n) return;
  const int tgt = __float_as_int(metaC[4LL * row]) - col0;
  if (tgt < 0 || tgt >= len) return;
  const float g = dt[row];
  const T* b = Floc + (long long)tgt * D;
  if (g != 0.f) { float* ge = dEwC + (long long)row * D; for (int c = l; c < D; c += 64) ge[c] += g * to_f32(b[c]); }
  for (int r0 = 0; r0 < row; r0 += 64) {   // an earlier row with this target: that one adds for all of them
    const int r = r0 + l;
    if (__ballot(r < row && __float_as_int(metaC[4LL * r]) - col0 == tgt) != 0ull) return;
  }
  float acc[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) acc[k] = 0.f;
  for (int r0 = row & ~63; r0 < n; r0 += 64) {
    const int r = r0 + l;
    unsigned long long hit = __ballot(r >= row && r < n && __float_as_int(metaC[4LL * r]) - col0 == tgt);
    while (hit != 0ull) {
      const int rr = r0 + __ffsll((long long)hit) - 1;
      hit &= hit - 1;
      const float gr = dt[rr];
      const T* a = EwC + (long long)rr * D;
#pragma unroll
      for (int k = 0; k < DMAX; ++k) { const int c = l + 64 * k; if (c < D) acc[k] += gr * to_f32(a[c]); }
    }
  }
  float* gf = gE_loc + (long long)tgt * D;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) { const int c = l + 64 * k; if (c < D) gf[c] += acc[k]; }
}

template <typename T>
int launch_ss_target_logit(const T* EwC, const T* Floc, int D, int len, int col0, const float* metaC, const int* nlive, float* tl, int grid_rows, hipStream_t s) {
  hipLaunchKernelGGL((ss_target_logit_kernel<T>), dim3(div_up(grid_rows, 4)), dim3(256), 0, s, EwC, Floc, D, len, col0, metaC, nlive, tl);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template <typename T>
int launch_ss_stats(const T* logits, long long ldl, int n_s, int n_tot, int len, int col0, const int* cols, const float* metaC, const int* nlive,
                    float* lmax, float* lsum, int grid_rows, hipStream_t s) {
  hipLaunchKernelGGL((ss_stats_kernel<T>), dim3(grid_rows), dim3(256), 0, s, logits, ldl, n_s, n_tot, len, col0, cols, metaC, nlive, lmax, lsum);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
int launch_ss_max_with_target(const float* lmax, const float* tl, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(ss_max_with_target_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, lmax, tl, out, n);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
int launch_ss_rebase(const float* lmax, const float* gmax, float* lsum, int n, hipStream_t s) {
  hipLaunchKernelGGL(ss_rebase_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, lmax, gmax, lsum, n);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template <typename T>
int launch_ss_finish(T* logits, long long ldl, int n_s, int n_tot, int len, int col0, const int* cols, const float* metaC, const float* gmax,
                     const float* sneg, const float* tl, const int* nlive, const int* pre, int rank, float* loss_out,
                     float* dt, int grid_rows, hipStream_t s) {
  float* part = g_det.part != nullptr && (long long)grid_rows <= g_det.cap ? g_det.part : nullptr;   // deterministic mode (kernels.hpp DetScratch)
  if (g_det.part != nullptr && part == nullptr) { set_error("sampled soft-max: deterministic scratch too small"); return RSYS_ERR_STATE; }
  if (part != nullptr) HIP_CHECK(hipMemsetAsync(part, 0, (size_t)grid_rows * 4, s));   // (rows without a term write nothing)
  hipLaunchKernelGGL((ss_finish_kernel<T>), dim3(grid_rows), dim3(256), 0, s, logits, ldl, n_s, n_tot, len, col0, cols, metaC, gmax, sneg, tl,
                     nlive, pre, rank, loss_out, dt, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, grid_rows, 1, 1, loss_out, s);
  return RSYS_OK;
}
template <typename T>
int launch_ss_target_grad(const T* EwC, const T* Floc, int D, int len, int col0, const float* metaC, const float* dt, const int* nlive,
                          float* gE_loc, float* dEwC, int grid_rows, hipStream_t s) {
  if (g_det.part != nullptr) {   // deterministic mode
    ARG_CHECK(D <= 64 * 32, "sampled soft-max, deterministic mode: embed_dim <= 2048");
    if (D <= 64 * 8) hipLaunchKernelGGL((ss_target_grad_ordered_kernel<T, 8>), dim3(div_up(grid_rows, 4)), dim3(256), 0, s, EwC, Floc, D, len, col0, metaC, dt, nlive, gE_loc, dEwC);
    else hipLaunchKernelGGL((ss_target_grad_ordered_kernel<T, 32>), dim3(div_up(grid_rows, 4)), dim3(256), 0, s, EwC, Floc, D, len, col0, metaC, dt, nlive, gE_loc, dEwC);
  } else
    hipLaunchKernelGGL((ss_target_grad_kernel<T>), dim3(div_up(grid_rows, 4)), dim3(256), 0, s, EwC, Floc, D, len, col0, metaC, dt, nlive, gE_loc, dEwC);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

#define INST(T)                                                                                                          \
  template int launch_vp_stats<T>(const T*, long long, int, int, const float*, const int*, float*, float*, int, int, hipStream_t); \
  template int launch_vp_finish<T>(T*, long long, int, int, const float*, const float*, const float*, int, const int*, const int*, int, float*, int, hipStream_t);
INST(bf16)
INST(float)
#undef INST
#define INST(T)                                                                                                          \
  template int launch_gather_rows_plain<T>(const T*, long long, const int*, int, T*, int, int, hipStream_t);               \
  template int launch_ss_target_logit<T>(const T*, const T*, int, int, int, const float*, const int*, float*, int, hipStream_t); \
  template int launch_ss_stats<T>(const T*, long long, int, int, int, int, const int*, const float*, const int*, float*, float*, int, hipStream_t); \
  template int launch_ss_finish<T>(T*, long long, int, int, int, int, const int*, const float*, const float*, const float*, const float*, const int*, const int*, int, float*, float*, int, hipStream_t); \
  template int launch_ss_target_grad<T>(const T*, const T*, int, int, int, const float*, const float*, const int*, float*, float*, int, hipStream_t);
INST(bf16)
INST(float)
#undef INST

}  // namespace rsys
