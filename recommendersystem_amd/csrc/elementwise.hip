// HBM-bound row kernels of the training step (gfx950).  One wave (64 lanes) per
// row with 16-byte accesses wherever a row is contiguous; per-column reductions
// keep partial sums in registers across a grid-stride row loop, combine through
// LDS and finish with one float atomic per column per workgroup.
#include "kernels.hpp"

namespace rsys {

thread_local DetScratch g_det;

__global__ void reduce_parts_kernel(const float* __restrict__ part, int nparts, long long stride, int n, float* dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float acc = 0.f;
  for (int b = 0; b < nparts; ++b) acc += part[(long long)b * stride + c];
  dst[c] += acc;
}
// first stage of a long reduction: out[chunk][c] = sum of the partial rows [32 chunk, 32 chunk + 32) in row order
__global__ void reduce_parts_stage_kernel(const float* __restrict__ part, int nparts, long long stride, int n, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, chunk = blockIdx.y;
  if (c >= n) return;
  const int b0 = chunk * 32, b1 = min(nparts, b0 + 32);
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) acc += part[(long long)b * stride + c];
  out[(long long)chunk * n + c] = acc;
}
int launch_reduce_parts(const float* part, int nparts, long long stride, int n, float* dst, hipStream_t s) {
  if (nparts <= 0 || n <= 0) return RSYS_OK;
  if (nparts > 64 && g_det.tmp != nullptr) {   // two stages, both with a fixed order: 32-row chunks, then the chunks
    const int chunks = (nparts + 31) / 32;
    if ((long long)chunks * n <= g_det.tmp_cap) {
      hipLaunchKernelGGL(reduce_parts_stage_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, s, part, nparts, stride, n, g_det.tmp);
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, g_det.tmp, chunks, (long long)n, n, dst);
      HIP_CHECK(hipGetLastError());
      return RSYS_OK;
    }
  }
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, nparts, stride, n, dst);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
// scratch of the deterministic mode for `floats` partial sums, or nullptr (atomics) when the mode is off
static inline float* det_part(long long floats) {
  if (g_det.part == nullptr) return nullptr;
  return floats <= g_det.cap ? g_det.part : nullptr;
}

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// --------------------------------------------------------------------- mask_tokens
// transformer.model.py:417-462
__global__ void mask_tokens_kernel(BatchDev b, int finetune, int ft_metric, float rate,
                                   unsigned long long seed, unsigned long long step) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.N) return;
  bool wm, rm;
  if (finetune) {
    bool w = b.weight[0 * 3 + ft_metric][i] > 0.f || b.weight[1 * 3 + ft_metric][i] > 0.f;
    wm = (ft_metric == 0) && w;
    rm = (ft_metric == 1) && w;
  } else if (b.watch_mask != nullptr) {
    wm = b.watch_mask[i] != 0;
    rm = b.rating_mask[i] != 0;
  } else {
    Philox ph(seed);
    uint32_t r[4];
    ph.gen((unsigned long long)i, (uint32_t)step, r);
    float u = u01(r[0]);
    wm = u < rate;
    rm = (u >= rate) && (u < 2.f * rate);
  }
  const bool any = wm || rm;
  b.m_tmid[i] = rm ? b.tmid[i] : 0;
  b.m_matchedid[i] = wm ? -1 : b.matchedid[i];
  b.m_status[i] = any ? -1 : b.status[i];
  b.m_rating[i] = any ? 0.f : b.rating[i];
  b.m_progress[i] = any ? 0.f : b.progress[i];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    b.m_label[m * 2 + 0][i] = wm ? b.label[m * 3 + 0][i] : 0.f;
    b.m_weight[m * 2 + 0][i] = wm ? b.weight[m * 3 + 0][i] : 0.f;
    b.m_position[m * 2 + 0][i] = wm ? b.position[m * 3 + 0][i] : 0;
    b.m_label[m * 2 + 1][i] = rm ? b.label[m * 3 + 1][i] : 0.f;
    b.m_weight[m * 2 + 1][i] = rm ? b.weight[m * 3 + 1][i] : 0.f;
    b.m_position[m * 2 + 1][i] = rm ? b.position[m * 3 + 1][i] : 0;
  }
}

int launch_mask_tokens(BatchDev b, int finetune, int finetune_metric, float mask_rate,
                       unsigned long long seed, unsigned long long step, hipStream_t s) {
  hipLaunchKernelGGL(mask_tokens_kernel, dim3(div_up(b.N, 256)), dim3(256), 0, s, b, finetune, finetune_metric,
                     mask_rate, seed, step);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// --------------------------------------------------------------------- action features
// transformer.model.py:51-93.  The periodic argument is formed in fp64, cast to
// fp32, and the learned phase is added IN fp32 (|arg| ~ 1e5): that rounding is
// part of the reference's arithmetic and is reproduced here.
__device__ __forceinline__ void periodic_args(double t, double min_ts, float& p0, float& p1, double& ts_out) {
  double ts = t < min_ts ? min_ts : t;
  p0 = (float)((6.283185307179586 * ts) / 86400.0);
  p1 = (float)((6.283185307179586 * ts) / 604800.0);
  ts_out = ts;
}

template <typename T>
__global__ void action_features_kernel(BatchDev b, SmallParams sp, T* feat) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.N) return;
  float f[32];
  float p0, p1; double ts;
  periodic_args(b.time[i], sp.min_ts, p0, p1, ts);
  f[0] = (float)((ts - sp.min_ts) / (sp.max_ts - sp.min_ts));
  f[1] = cosf(p0 + sp.per_cos[0]);
  f[2] = cosf(p1 + sp.per_cos[1]);
  f[3] = sinf(p0 + sp.per_sin[0]);
  f[4] = sinf(p1 + sp.per_sin[1]);
  int gi = b.gender[i]; gi = gi == -1 ? sp.n_gender : gi;
  int si = b.source[i]; si = si == -1 ? sp.n_source : si;
  int st = b.m_status[i]; st = st == -1 ? sp.n_status : st;
#pragma unroll
  for (int k = 0; k < 4; ++k) f[5 + k] = sp.gender_emb[gi * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) f[9 + k] = sp.source_emb[si * 4 + k];
  float rating = b.m_rating[i];
  float has = rating != 0.f ? 1.f : 0.f;
  f[13] = has;
  f[14] = has * ((rating - sp.rating_mean) / sp.rating_std);
#pragma unroll
  for (int k = 0; k < 16; ++k) f[15 + k] = sp.status_emb[st * 16 + k];
  f[31] = b.m_progress[i];
  T* o = feat + (long long)i * 32;
#pragma unroll
  for (int k = 0; k < 32; ++k) o[k] = from_f32<T>(f[k]);
}

template <typename T>
int launch_action_features(const BatchDev& b, const SmallParams& sp, T* feat, hipStream_t s) {
  hipLaunchKernelGGL((action_features_kernel<T>), dim3(div_up(b.N, 128)), dim3(128), 0, s, b, sp, feat);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_action_features<bf16>(const BatchDev&, const SmallParams&, bf16*, hipStream_t);
template int launch_action_features<float>(const BatchDev&, const SmallParams&, float*, hipStream_t);

// --------------------------------------------------------------------- item gather (K1/K3, fused table)
// transformer.model.py:23-24,139-145 with the fused table of :120-133; one wave per row, 16 B per lane.
__global__ void gather_items_kernel(BatchDev b, const float* __restrict__ F32, int V, int D, float* x0, int* uid_t, int* tm_t) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (wave >= b.N) return;
  int id = b.m_matchedid[wave];
  id = id == -1 ? V : id;
  const float4* src = (const float4*)(F32 + (long long)id * D);
  float4* dst = (float4*)(x0 + (long long)(2 * wave) * D);
  for (int c = l; c < (D >> 2); c += 64) dst[c] = src[c];
  if (l == 0) {
    int u = b.userid[wave], tmv = b.m_tmid[wave];
    uid_t[2 * wave] = u; uid_t[2 * wave + 1] = u;
    tm_t[2 * wave] = tmv; tm_t[2 * wave + 1] = tmv;
  }
}

int launch_gather_items(const BatchDev& b, const float* F32, int V, int D, float* x0, int* uid_t, int* tm_t, hipStream_t s) {
  hipLaunchKernelGGL(gather_items_kernel, dim3(div_up(b.N, 4)), dim3(256), 0, s, b, F32, V, D, x0, uid_t, tm_t);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// --------------------------------------------------------------------- RMSNorm (K6), transformer.model.py:193-202
// One wave per row, 4 consecutive columns per lane and 256-column group; the number of groups NJ is a template
// parameter (EXACT: D == 256 NJ, no column guards), so a row lives in NJ float4 registers, is read once, and the
// outputs leave as 16-byte (f32) / 8-byte (bf16) vector stores.  HBM-bound: 6 B per element forward, 14 B backward.
constexpr int NORM_MAXJ = 8;  // D <= 64 lanes * 4 * 8 = 2048

template <typename T> __device__ __forceinline__ void store4(T* dst, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* dst, float a, float b, float c, float d) { *(float4*)dst = make_float4(a, b, c, d); }
template <> __device__ __forceinline__ void store4<bf16>(bf16* dst, float a, float b, float c, float d) {
  bf16x4 v; v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  *(bf16x4*)dst = v;
}
template <typename T> __device__ __forceinline__ float4 load4(const T* src);
template <> __device__ __forceinline__ float4 load4<float>(const float* src) { return *(const float4*)src; }
template <> __device__ __forceinline__ float4 load4<bf16>(const bf16* src) {
  const bf16x4 v = *(const bf16x4*)src;
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}


template <typename T, int NJ, bool EXACT>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, T* y, float* rstd,
                                                          long long rows, int D, const int* __restrict__ rows_dev, const int* __restrict__ in_rows,
                                                          float* amax) {
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int l = threadIdx.x & 63;
  if (rows_dev != nullptr) {   // compact row set (compact.hip): rows [0, n) are live, rows [n, n rounded up to 256) are written as zeros
    const long long n = *rows_dev;
    if (row >= min(rows, (n + 255) & ~255LL)) return;
    if (row >= n) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) { const int c = (j * 64 + l) * 4; if (EXACT || c < D) store4<T>(y + row * D + c, 0.f, 0.f, 0.f, 0.f); }
      if (l == 0 && rstd) rstd[row] = 0.f;
      return;
    }
  }
  if (row >= rows) return;
  const long long xrow = in_rows != nullptr ? (long long)in_rows[row] : row;
  float4 v[NJ];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + l) * 4;
    v[j] = (EXACT || c < D) ? *(const float4*)(x + xrow * D + c) : make_float4(0, 0, 0, 0);
    ss += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)D + 1e-5f);
  if (l == 0 && rstd) rstd[row] = r;
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + l) * 4;
    if (EXACT || c < D) {
      const float4 sc = *(const float4*)(scale + c);
      const float o0 = v[j].x * r * sc.x, o1 = v[j].y * r * sc.y, o2 = v[j].z * r * sc.z, o3 = v[j].w * r * sc.w;
      store4<T>(y + row * D + c, o0, o1, o2, o3);
      am = fmaxf(am, fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3))));
    }
  }
  if (amax != nullptr) {   // fp8 trunk: amax of the (bf16) output while it is written -- the rounded amax is the amax of the rounded values
    am = wave_max(am);     // (one unconditional atomic per wave, nothing waits for it: a read-and-compare first cost 10 us per launch)
    if (l == 0) f8_amax_add(amax, (float)from_f32<T>(am));
  }
}

template <typename T>
int launch_rmsnorm_fwd(const float* x, const float* scale, T* y, float* rstd, long long rows, int D, hipStream_t s, const int* rows_dev, const int* in_rows,
                       float* f8_amax) {
  float* const amax = f8_amax;
  ARG_CHECK(D % 4 == 0 && D <= 64 * 4 * NORM_MAXJ, "rmsnorm: D must be a multiple of 4 and <= 2048");
  const dim3 grid(div_up(rows, 4)), block(256);
#define RSYS_NORM_FWD(NJ, EX) hipLaunchKernelGGL((rmsnorm_fwd_kernel<T, NJ, EX>), grid, block, 0, s, x, scale, y, rstd, rows, D, rows_dev, in_rows, amax)
  if (D == 256) RSYS_NORM_FWD(1, true);
  else if (D == 512) RSYS_NORM_FWD(2, true);
  else if (D == 1024) RSYS_NORM_FWD(4, true);
  else if (D == 2048) RSYS_NORM_FWD(8, true);
  else if (D < 256) RSYS_NORM_FWD(1, false);
  else if (D < 512) RSYS_NORM_FWD(2, false);
  else if (D < 1024) RSYS_NORM_FWD(4, false);
  else RSYS_NORM_FWD(8, false);
#undef RSYS_NORM_FWD
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_rmsnorm_fwd<bf16>(const float*, const float*, bf16*, float*, long long, int, hipStream_t, const int*, const int*, float*);
template int launch_rmsnorm_fwd<float>(const float*, const float*, float*, float*, long long, int, hipStream_t, const int*, const int*, float*);

// backward: dx = r*g*s - x*r^3*sum(g*s*x)/D (+ residual gradient) ; dscale += g*x*r
// Workgroups of 16 waves (round 6).  Every workgroup ends with D float atomics onto the same D scale gradients, and the workgroups of a
// launch all finish together: with 1024 workgroups of 4 waves that tail was a fixed ~29 us per launch at cfg-2 AND cfg-3 (two-point fit of
// 45 us for 134 MB and 93 us for 537 MB: the streaming part alone runs at 8.4 TB/s; the grid scan in rmsnorm_bwd_any prices an atomic
// tail at ~23 ns per workgroup).  The same 16 waves per CU as one workgroup: a quarter of the atomics, same bytes in flight.
// (a row's three operands are 12 NJ registers per lane: 16 waves per workgroup = 128 registers per lane up to NJ = 2 (D <= 512), 8 waves up to
// NJ = 4, 4 waves at NJ = 8 -- with 16 waves everywhere the D = 2048 kernel spilled and the production shape's norms ran at half their rate)
template <typename TG, typename TO, int NJ, bool EXACT>
__global__ __launch_bounds__(NJ <= 2 ? 1024 : NJ <= 4 ? 512 : 256) void rmsnorm_bwd_kernel(const TG* __restrict__ g, const float* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ rstd,
                                                          const float* resid, float* dx_out, TO* dx_out_t, float* dscale,
                                                          float* part, long long rows, int D, const int* __restrict__ rows_dev,
                                                          const int* __restrict__ resid_slot, const int* __restrict__ io_rows, float* amax) {
  extern __shared__ __attribute__((aligned(16))) float sds[];  // D floats (waves per workgroup * D in deterministic mode: `part` set)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long long wave0 = (long long)blockIdx.x * nw + w, nwaves = (long long)gridDim.x * nw;
  float4 acc[NJ], sc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + l) * 4;
    acc[j] = make_float4(0, 0, 0, 0);
    sc[j] = (EXACT || c < D) ? *(const float4*)(scale + c) : make_float4(0, 0, 0, 0);
  }
  for (int c = threadIdx.x; c < D; c += blockDim.x) sds[c] = 0.f;
  __syncthreads();
  // compact row set (compact.hip): rows [0, n) live, rows [n, n rounded up to 256) get zero outputs
  const long long n_live = rows_dev != nullptr ? (long long)*rows_dev : rows;
  if (rows_dev != nullptr) rows = min(rows, (n_live + 255) & ~255LL);
  float am = 0.f;
  for (long long row = wave0; row < rows; row += nwaves) {
    if (row >= n_live) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = (j * 64 + l) * 4;
        if (EXACT || c < D) { *(float4*)(dx_out + row * D + c) = make_float4(0, 0, 0, 0); if (dx_out_t) store4<TO>(dx_out_t + row * D + c, 0.f, 0.f, 0.f, 0.f); }
      }
      continue;
    }
    const float r = rstd[row];
    // residual gradient: dense rows, or (resid_slot) the compact row resid_slot[row] of `resid`, zero where that is -1
    const long long rrow = resid_slot != nullptr ? (long long)resid_slot[row] : row;
    const long long xrow = io_rows != nullptr ? (long long)io_rows[row] : row;   // x is read at, dx written to, this row
    float4 gv[NJ], xv[NJ], rv[NJ];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = (j * 64 + l) * 4;
      const bool ok = EXACT || c < D;
      const float4 g4 = ok ? load4<TG>(g + row * D + c) : make_float4(0, 0, 0, 0);
      xv[j] = ok ? *(const float4*)(x + xrow * D + c) : make_float4(0, 0, 0, 0);
      rv[j] = (ok && resid && rrow >= 0) ? *(const float4*)(resid + rrow * D + c) : make_float4(0, 0, 0, 0);
      gv[j] = make_float4(g4.x * sc[j].x, g4.y * sc[j].y, g4.z * sc[j].z, g4.w * sc[j].w);
      dot += gv[j].x * xv[j].x + gv[j].y * xv[j].y + gv[j].z * xv[j].z + gv[j].w * xv[j].w;
      // dscale uses the un-scaled g
      acc[j].x += g4.x * xv[j].x * r; acc[j].y += g4.y * xv[j].y * r;
      acc[j].z += g4.z * xv[j].z * r; acc[j].w += g4.w * xv[j].w * r;
    }
    dot = wave_sum(dot);
    const float k = r * r * r * dot / (float)D;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = (j * 64 + l) * 4;
      if (EXACT || c < D) {
        float4 o;
        o.x = r * gv[j].x - xv[j].x * k + rv[j].x; o.y = r * gv[j].y - xv[j].y * k + rv[j].y;
        o.z = r * gv[j].z - xv[j].z * k + rv[j].z; o.w = r * gv[j].w - xv[j].w * k + rv[j].w;
        *(float4*)(dx_out + xrow * D + c) = o;
        if (dx_out_t) store4<TO>(dx_out_t + xrow * D + c, o.x, o.y, o.z, o.w);   // operand copy for the next GEMMs (bf16 mode)
        am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      }
    }
  }
  if (amax != nullptr) {   // fp8 trunk: amax of the operand copy (see rmsnorm_fwd_kernel)
    am = wave_max(am);
    if (l == 0) f8_amax_add(amax, (float)from_f32<TO>(am));
  }
  if (part != nullptr) {
    // deterministic: the waves' sums side by side in LDS ([waves][D], launcher), added in wave order, one partial row per workgroup
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = (j * 64 + l) * 4;
      if (EXACT || c < D) *(float4*)&sds[w * D + c] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      float t = sds[c];
      for (int k = 1; k < nw; ++k) t += sds[k * D + c];
      part[(long long)blockIdx.x * D + c] = t;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + l) * 4;
    if (EXACT || c < D) {
      atomicAdd(&sds[c], acc[j].x); atomicAdd(&sds[c + 1], acc[j].y);
      atomicAdd(&sds[c + 2], acc[j].z); atomicAdd(&sds[c + 3], acc[j].w);
    }
  }
  __syncthreads();
  if (wave0 - w >= n_live) return;   // (a compact row set: workgroups without a live row have nothing to add -- uniform: wave0 - w = first row of the workgroup)
  for (int c = threadIdx.x; c < D; c += blockDim.x) atomicAdd(&dscale[c], sds[c]);
}

template <typename TG, typename TO>
static int rmsnorm_bwd_any(const TG* g, const float* x, const float* scale, const float* rstd, const float* resid,
                           float* dx_out, TO* dx_out_t, float* dscale, long long rows, int D, hipStream_t s, const int* rows_dev,
                           const int* resid_slot, const int* io_rows = nullptr, float* f8_amax = nullptr) {
  float* const amax = f8_amax;
  ARG_CHECK(D % 4 == 0 && D <= 64 * 4 * NORM_MAXJ, "rmsnorm_bwd: D must be a multiple of 4 and <= 2048");
  // (every workgroup ends with D atomics onto the same D scale gradients: with 4 waves per workgroup, 512-1024 workgroups 1.39-1.45 ms per step
  // at cfg-3, 2048: 1.49, 4096: 1.97, 8192: 3.37.)  RSYS_DEBUG_NORM_BWD_GRID caps the number of WAVES / 4 (the unit of those scans).
  const int grid_cap = sw().debug_norm_bwd_grid;
  const int wpb_max = D <= 512 ? 16 : D <= 1024 ? 8 : 4;   // = the kernel's launch bounds for this D
  const int wpb = sw().debug_norm_bwd_waves > 0 ? std::min(sw().debug_norm_bwd_waves, wpb_max) : wpb_max;   // waves per workgroup: the kernel's launch bounds (deterministic mode keeps wpb * D floats in LDS)
  const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((rows + wpb - 1) / wpb, (long long)grid_cap * 4 / wpb))), block(64 * wpb);
  float* part = det_part((long long)grid.x * D);
#define RSYS_NORM_BWD(NJ, EX) hipLaunchKernelGGL((rmsnorm_bwd_kernel<TG, TO, NJ, EX>), grid, block, (part ? wpb : 1) * D * sizeof(float), s, g, x, scale, \
                                                 rstd, resid, dx_out, dx_out_t, dscale, part, rows, D, rows_dev, resid_slot, io_rows, amax)
  if (D == 256) RSYS_NORM_BWD(1, true);
  else if (D == 512) RSYS_NORM_BWD(2, true);
  else if (D == 1024) RSYS_NORM_BWD(4, true);
  else if (D == 2048) RSYS_NORM_BWD(8, true);
  else if (D < 256) RSYS_NORM_BWD(1, false);
  else if (D < 512) RSYS_NORM_BWD(2, false);
  else if (D < 1024) RSYS_NORM_BWD(4, false);
  else RSYS_NORM_BWD(8, false);
#undef RSYS_NORM_BWD
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, (int)grid.x, D, D, dscale, s);
  return RSYS_OK;
}
template <typename T>
int launch_rmsnorm_bwd(const T* g, const float* x, const float* scale, const float* rstd, const float* resid_grad,
                       float* dx_out, T* dx_out_t, float* dscale, long long rows, int D, hipStream_t s, const int* rows_dev, const int* resid_slot,
                       const int* io_rows, float* f8_amax) {
  return rmsnorm_bwd_any<T, T>(g, x, scale, rstd, resid_grad, dx_out, dx_out_t, dscale, rows, D, s, rows_dev, resid_slot, io_rows, f8_amax);
}
template int launch_rmsnorm_bwd<bf16>(const bf16*, const float*, const float*, const float*, const float*, float*, bf16*, float*, long long, int, hipStream_t, const int*, const int*, const int*, float*);
template int launch_rmsnorm_bwd<float>(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, long long, int, hipStream_t, const int*, const int*, const int*, float*);
template <typename T>
int launch_rmsnorm_bwd_f32(const float* g, const float* x, const float* scale, const float* rstd, const float* resid_grad,
                           float* dx_out, T* dx_out_t, float* dscale, long long rows, int D, hipStream_t s, const int* rows_dev, float* f8_amax) {
  return rmsnorm_bwd_any<float, T>(g, x, scale, rstd, resid_grad, dx_out, dx_out_t, dscale, rows, D, s, rows_dev, nullptr, nullptr, f8_amax);
}
template int launch_rmsnorm_bwd_f32<bf16>(const float*, const float*, const float*, const float*, const float*, float*, bf16*, float*, long long, int, hipStream_t, const int*, float*);
template int launch_rmsnorm_bwd_f32<float>(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, long long, int, hipStream_t, const int*, float*);

// --------------------------------------------------------------------- dropout (LoRA input, finetune)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ src, T* dst, long long n, float p, unsigned long long seed,
                               unsigned int stream, int accumulate) {
  Philox ph(seed);
  const float keep = 1.f / (1.f - p);
  const long long n4 = (n + 3) >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    uint32_t r[4];
    ph.gen((unsigned long long)i, stream, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long e = i * 4 + k;
      if (e < n) {
        const float v = u01(r[k]) >= p ? to_f32(src[e]) * keep : 0.f;
        dst[e] = from_f32<T>(accumulate ? to_f32(dst[e]) + v : v);
      }
    }
  }
}
template <typename T>
int launch_dropout(const T* src, T* dst, long long n, float p, unsigned long long seed, unsigned int stream, int accumulate, hipStream_t s) {
  int grid = (int)std::min<long long>((((n + 3) >> 2) + 255) / 256, 8192);
  hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid), dim3(256), 0, s, src, dst, n, p, seed, stream, accumulate);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_dropout<bf16>(const bf16*, bf16*, long long, float, unsigned long long, unsigned int, int, hipStream_t);
template int launch_dropout<float>(const float*, float*, long long, float, unsigned long long, unsigned int, int, hipStream_t);

// --------------------------------------------------------------------- position selection (K11), model.py:501-513
// Deterministic replacement of torch.topk on 0/1 weights: positive-weight positions in
// ascending flat index, then zero-weight positions ascending.  One
// 1024-thread workgroup per task; two-level exclusive scan.
struct SelectBatch { const float* w[4]; int* idx[4]; float* stats[4]; int* npos[4]; };
__global__ __launch_bounds__(1024) void select_positions_kernel(SelectBatch sb, int N, int topk) {
  const float* __restrict__ w = sb.w[blockIdx.x];
  int* idx = sb.idx[blockIdx.x]; float* stats = sb.stats[blockIdx.x]; int* npos_out = sb.npos[blockIdx.x];
  __shared__ int wave_tot[16];
  __shared__ float red[16];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int per = (N + 1023) / 1024;
  const int i0 = t * per, i1 = min(N, i0 + per);
  int cnt = 0; float wall = 0.f;
  for (int i = i0; i < i1; ++i) { float x = w[i]; cnt += x > 0.f; wall += x; }
  // exclusive scan of cnt over 1024 threads
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
  if (l == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, npos = 0;
  for (int k = 0; k < 16; ++k) { int v = wave_tot[k]; if (k < wv) base += v; npos += v; }
  int rank = base + inc - cnt;  // positives before i0
  float wsel = 0.f;
  for (int i = i0; i < i1; ++i) {
    float x = w[i];
    int slot = x > 0.f ? rank : npos + (i - rank);
    if (slot < topk) { idx[slot] = i; wsel += x; }
    rank += x > 0.f;
  }
  wsel = block_sum(wsel, red);
  wall = block_sum(wall, red);
  if (t == 0) { stats[0] = wsel; stats[1] = wall; if (npos_out) *npos_out = min(npos, topk); }
}
// The same selection by many workgroups: SEL_CHUNKS chunks per task count their positive weights (first kernel), every chunk then
// ranks its positions behind the chunks before it (second kernel; the SEL_CHUNKS counts are summed by each workgroup itself).
// One 1024-thread workgroup per task took 44 us at 32 768 positions (two strided passes); this takes two launches of ~4 us.
constexpr int SEL_CHUNKS = 32;
__global__ __launch_bounds__(256) void select_count_kernel(SelectBatch sb, int N, int* __restrict__ counts, float* __restrict__ wsums) {
  __shared__ float red[16];
  const int task = blockIdx.x / SEL_CHUNKS, ch = blockIdx.x % SEL_CHUNKS;
  const float* __restrict__ w = sb.w[task];
  const int per = (N + SEL_CHUNKS - 1) / SEL_CHUNKS, i0 = ch * per, i1 = min(N, i0 + per);
  float cnt = 0.f, wall = 0.f;
  for (int i = i0 + threadIdx.x; i < i1; i += 256) { const float x = w[i]; cnt += x > 0.f ? 1.f : 0.f; wall += x; }
  cnt = block_sum(cnt, red);       // (counts <= 2^24: exact in float)
  wall = block_sum(wall, red);
  if (threadIdx.x == 0) { counts[blockIdx.x] = (int)cnt; wsums[blockIdx.x] = wall; }
}
__global__ __launch_bounds__(256) void select_place_kernel(SelectBatch sb, int N, int topk, const int* __restrict__ counts, const float* __restrict__ wsums,
                                                           float* __restrict__ wsel_part) {
  __shared__ int wave_tot[4];
  __shared__ float red[16];
  const int task = blockIdx.x / SEL_CHUNKS, ch = blockIdx.x % SEL_CHUNKS;
  const float* __restrict__ w = sb.w[task];
  int* idx = sb.idx[task];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  int before = 0, npos = 0;
  for (int k = 0; k < SEL_CHUNKS; ++k) { const int v = counts[task * SEL_CHUNKS + k]; if (k < ch) before += v; npos += v; }
  const int per = (N + SEL_CHUNKS - 1) / SEL_CHUNKS, i0 = ch * per, i1 = min(N, i0 + per);
  float wsel = 0.f;
  int rank0 = before;   // positives before the current 256-position stripe
  for (int j0 = i0; j0 < i1; j0 += 256) {
    const int i = j0 + t;
    const float x = i < i1 ? w[i] : 0.f;
    const int pos = x > 0.f ? 1 : 0;
    int inc = pos;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
    __syncthreads();
    if (l == 63) wave_tot[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int v = wave_tot[k]; if (k < wv) base += v; tot += v; }
    const int rank = rank0 + base + inc - pos;   // positives before position i
    if (i < i1) {
      const int slot = pos ? rank : npos + (i - rank);
      if (slot < topk) { idx[slot] = i; wsel += x; }
    }
    rank0 += tot;
  }
  wsel = block_sum(wsel, red);
  if (t == 0) wsel_part[blockIdx.x] = wsel;
}
__global__ void select_finish_kernel(SelectBatch sb, int ntask, int topk, const int* __restrict__ counts, const float* __restrict__ wsums,
                                     const float* __restrict__ wsel_part) {
  const int task = threadIdx.x;
  if (task >= ntask) return;
  int npos = 0; float wall = 0.f, wsel = 0.f;
  for (int k = 0; k < SEL_CHUNKS; ++k) { npos += counts[task * SEL_CHUNKS + k]; wall += wsums[task * SEL_CHUNKS + k]; wsel += wsel_part[task * SEL_CHUNKS + k]; }
  sb.stats[task][0] = wsel; sb.stats[task][1] = wall;
  if (sb.npos[task]) *sb.npos[task] = min(npos, topk);
}
// scratch: 3 * 4 * SEL_CHUNKS words
int launch_select_positions_chunked(int ntask, const float* const* w, int N, int topk, int* const* idx, float* const* stats, int* const* npos_out,
                                    void* scratch, hipStream_t s) {
  ARG_CHECK(topk <= N && ntask >= 1 && ntask <= 4 && scratch != nullptr, "select_positions: topk > N or more than 4 tasks");
  SelectBatch sb{};
  for (int i = 0; i < ntask; ++i) { sb.w[i] = w[i]; sb.idx[i] = idx[i]; sb.stats[i] = stats[i]; sb.npos[i] = npos_out[i]; }
  int* counts = (int*)scratch; float* wsums = (float*)scratch + 4 * SEL_CHUNKS; float* wsel = (float*)scratch + 8 * SEL_CHUNKS;
  hipLaunchKernelGGL(select_count_kernel, dim3(ntask * SEL_CHUNKS), dim3(256), 0, s, sb, N, counts, wsums);
  hipLaunchKernelGGL(select_place_kernel, dim3(ntask * SEL_CHUNKS), dim3(256), 0, s, sb, N, topk, counts, wsums, wsel);
  hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(64), 0, s, sb, ntask, topk, counts, wsums, wsel);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

int launch_select_positions(const float* w, int N, int topk, int* idx, float* stats, int* npos_out, hipStream_t s) {
  const float* ws[1] = {w}; int* is[1] = {idx}; float* st[1] = {stats}; int* np[1] = {npos_out};
  return launch_select_positions_batch(1, ws, N, topk, is, st, np, s);
}
int launch_select_positions_batch(int ntask, const float* const* w, int N, int topk, int* const* idx, float* const* stats,
                                  int* const* npos_out, hipStream_t s) {
  ARG_CHECK(topk <= N && ntask >= 1 && ntask <= 4, "select_positions: topk > N or more than 4 tasks");
  SelectBatch sb{};
  for (int i = 0; i < ntask; ++i) { sb.w[i] = w[i]; sb.idx[i] = idx[i]; sb.stats[i] = stats[i]; sb.npos[i] = npos_out[i]; }
  hipLaunchKernelGGL(select_positions_kernel, dim3(ntask), dim3(1024), 0, s, sb, N, topk);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// --------------------------------------------------------------------- row gather / scatter for the heads
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, long long ld, const int* __restrict__ idx, int parity, T* dst, int n, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n) return;
  constexpr int E = 16 / sizeof(T);
  const uint4* s4 = (const uint4*)(src + (2LL * idx[row] + parity) * ld);
  uint4* d4 = (uint4*)(dst + (long long)row * D);
  for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
}
template <typename T>
int launch_gather_rows(const T* src, long long ld, const int* idx, int parity, T* dst, int n, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0 && (ld * sizeof(T)) % 16 == 0, "gather_rows: rows must be 16-byte multiples");
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(div_up(n, 4)), dim3(256), 0, s, src, ld, idx, parity, dst, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_gather_rows<bf16>(const bf16*, long long, const int*, int, bf16*, int, int, hipStream_t);
template int launch_gather_rows<float>(const float*, long long, const int*, int, float*, int, int, hipStream_t);

__global__ void scatter_rows_add_kernel(const float* __restrict__ src, const int* __restrict__ idx, int parity, float* dst, long long ld, int n, int D,
                                        const int* __restrict__ npos) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n || (npos != nullptr && row >= *npos)) return;   // (rows from *npos on are zero-weight padding: their gradient is zero)
  const float4* s4 = (const float4*)(src + (long long)row * D);
  float4* d4 = (float4*)(dst + (2LL * idx[row] + parity) * ld);
  for (int c = l; c < (D >> 2); c += 64) {
    float4 a = d4[c], b = s4[c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[c] = a;
  }
}
int launch_scatter_rows_add(const float* src, const int* idx, int parity, float* dst, long long ld, int n, int D, hipStream_t s, const int* npos) {
  hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, src, idx, parity, dst, ld, n, D, npos);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// --------------------------------------------------------------------- cross entropy (K13), model.py:514-519
// One workgroup per selected position.  Rows with zero coefficient only clear their dlogits.
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(T* logits, long long ldl, int V, const int* __restrict__ idx,
                                                 const float* __restrict__ label, const float* __restrict__ weight,
                                                 const int* __restrict__ position, const float* __restrict__ stats,
                                                 const int* __restrict__ npos, float task_w, float* loss_out, float* part) {
  __shared__ float red[16];
  constexpr int E = 16 / sizeof(T);
  const int row = blockIdx.x, t = threadIdx.x;
  // zero-weight padding rows beyond the last 128-row GEMM tile that holds a positive row are never read by the
  // (row-limited) head GEMMs; padding rows inside that tile are cleared below (lw == 0)
  if (npos != nullptr && row >= ((*npos + 127) & ~127)) return;
  T* lr = logits + (long long)row * ldl;
  const int i = idx[row];
  const float w = weight[i], lab = label[i];
  const int nchunks = (int)(ldl / E);
  const float lw = lab * w;
  if (lw == 0.f) {
    for (int c = t; c < nchunks; c += 256) ((uint4*)lr)[c] = make_uint4(0, 0, 0, 0);
    return;
  }
  // online max / sum-exp
  float m = -3.0e38f, ssum = 0.f;
  for (int c = t; c < nchunks; c += 256) {
    uint4 raw = ((const uint4*)lr)[c];
    const T* e = (const T*)&raw;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (c * E + k < V) {
        float x = to_f32(e[k]);
        if (x > m) { ssum = ssum * __expf(m - x) + 1.f; m = x; }
        else ssum += __expf(x - m);
      }
    }
  }
  const float gm = block_max(m, red);
  ssum = block_sum(ssum * __expf(m - gm), red);
  const float lse = gm + logf(ssum);
  const int tgt = position[i];
  const float xt = to_f32(lr[tgt]);
  __syncthreads();
  if (t == 0) {
    if (part != nullptr) part[row] = (lse - xt) * lw;   // deterministic mode: the rows' terms are added in row order afterwards
    else atomicAdd(loss_out, (lse - xt) * lw);
  }
  const float coef = task_w * lw / fmaxf(stats[0], 1e-8f);
  for (int c = t; c < nchunks; c += 256) {
    uint4 raw = ((const uint4*)lr)[c];
    T* e = (T*)&raw;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      int col = c * E + k;
      float x = to_f32(e[k]);
      float gl = col < V ? coef * (__expf(x - lse) - (col == tgt ? 1.f : 0.f)) : 0.f;
      e[k] = from_f32<T>(gl);
    }
    ((uint4*)lr)[c] = raw;
  }
}
template <typename T>
int launch_ce_fwd_bwd(T* logits, long long ldl, int n, int V, const int* idx, const float* label, const float* weight,
                      const int* position, const float* stats, const int* npos, float task_w, float* loss_out, hipStream_t s) {
  ARG_CHECK((ldl * sizeof(T)) % 16 == 0 && ldl >= V, "ce: ldl");
  float* part = det_part(n);
  if (part != nullptr) HIP_CHECK(hipMemsetAsync(part, 0, (size_t)n * 4, s));   // (rows without a term write nothing)
  hipLaunchKernelGGL((ce_kernel<T>), dim3(n), dim3(256), 0, s, logits, ldl, V, idx, label, weight, position, stats, npos, task_w, loss_out, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, n, 1, 1, loss_out, s);
  return RSYS_OK;
}
template int launch_ce_fwd_bwd<bf16>(bf16*, long long, int, int, const int*, const float*, const float*, const int*, const float*, const int*, float, float*, hipStream_t);
template int launch_ce_fwd_bwd<float>(float*, long long, int, int, const int*, const float*, const float*, const int*, const float*, const int*, float, float*, hipStream_t);

// --------------------------------------------------------------------- rating head tail (K14), model.py:355-359,391-401,520-526
template <typename T>
__global__ __launch_bounds__(256) void rating_tail_kernel(T* z, const T* __restrict__ hact, int n, int D,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const int* __restrict__ idx, const float* __restrict__ label,
                                                          const float* __restrict__ weight, const float* __restrict__ stats,
                                                          float rating_mean, float task_w, int evaluate, float* loss_out,
                                                          float* dw2, float* db2, float* db0, float* part, const int* __restrict__ npos) {
  extern __shared__ __attribute__((aligned(16))) float sds[];  // 2*D floats: dw2 | db0  (deterministic mode: one such pair per wave + 4 scalars per wave)
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nacc = part != nullptr ? 4 * 2 * D + 16 : 2 * D;
  for (int c = threadIdx.x; c < nacc; c += 256) sds[c] = 0.f;
  __syncthreads();
  float* const mine = part != nullptr ? sds + w * 2 * D : sds;   // deterministic: a wave adds into its own copy (its lanes own distinct columns)
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, gb2 = 0.f;
  const float inv_ws = 1.f / fmaxf(stats[0], 1e-8f);
  // the positive-weight rows come first (select_positions): rows from *npos on are zero-weight padding, of which only those up to
  // the next multiple of 128 are touched (their dz = 0 is what the row- / K-limited GEMMs around this kernel read)
  if (npos != nullptr) n = min(n, (*npos + 127) & ~127);
  for (int row = blockIdx.x * 4 + w; row < n; row += gridDim.x * 4) {
    const int i = idx[row];
    const float wt = weight[i], tgt = label[i] - rating_mean;
    const T* hr = hact + (long long)row * D;
    T* zr = z + (long long)row * D;
    float acc = 0.f;
    for (int c = l; c < D; c += 64) acc += to_f32(hr[c]) * w2[c];
    acc = wave_sum(acc);
    const float pred = acc + b2[0];
    const float e1 = pred - tgt;
    l0 += e1 * e1 * wt; l1 += tgt * tgt * wt; l2 += (-pred - tgt) * (-pred - tgt) * wt;
    const float dpred = evaluate ? 0.f : task_w * 2.f * e1 * wt * inv_ws;
    gb2 += dpred;
    for (int c = l; c < D; c += 64) {
      float zz = to_f32(zr[c]);
      float gp = 0.5f * (1.f + erff(zz * 0.70710678118654752f)) + zz * __expf(-0.5f * zz * zz) * 0.3989422804014327f;
      float dz = dpred * w2[c] * gp;
      zr[c] = from_f32<T>(dz);
      if (dpred != 0.f) {
        if (part != nullptr) { mine[c] += dpred * to_f32(hr[c]); mine[D + c] += dz; }
        else { atomicAdd(&sds[c], dpred * to_f32(hr[c])); atomicAdd(&sds[D + c], dz); }
      }
    }
  }
  if (part != nullptr) {
    // one partial row per workgroup: [dw2 (D) | db0 (D) | db2 | loss 0..2], the waves' copies added in wave order
    float* sc = sds + 4 * 2 * D;
    if (l == 0) { sc[w * 4 + 0] = gb2; sc[w * 4 + 1] = l0; sc[w * 4 + 2] = l1; sc[w * 4 + 3] = l2; }
    __syncthreads();
    float* pr = part + (long long)blockIdx.x * (2 * D + 4);
    for (int c = threadIdx.x; c < 2 * D; c += 256) pr[c] = ((sds[c] + sds[2 * D + c]) + sds[4 * D + c]) + sds[6 * D + c];
    if (threadIdx.x < 4) pr[2 * D + threadIdx.x] = ((sc[threadIdx.x] + sc[4 + threadIdx.x]) + sc[8 + threadIdx.x]) + sc[12 + threadIdx.x];
    return;
  }
  __syncthreads();
  if (!evaluate) {
    for (int c = threadIdx.x; c < D; c += 256) {
      if (sds[c] != 0.f) atomicAdd(&dw2[c], sds[c]);
      if (sds[D + c] != 0.f) atomicAdd(&db0[c], sds[D + c]);
    }
    if (l == 0 && gb2 != 0.f) atomicAdd(db2, gb2);
  }
  if (l == 0) {
    if (l0 != 0.f) atomicAdd(&loss_out[0], l0);
    if (l1 != 0.f) atomicAdd(&loss_out[1], l1);
    if (l2 != 0.f) atomicAdd(&loss_out[2], l2);
  }
}
template <typename T>
int launch_rating_tail(T* z, const T* hact, int n, int D, const float* w2, const float* b2, const int* idx,
                       const float* label, const float* weight, const float* stats, float rating_mean, float task_w,
                       int evaluate, float* loss_out, float* dw2, float* db2, float* db0, hipStream_t s, const int* npos) {
  int grid = std::min(div_up(n, 4), 512);
  const long long prow = 2LL * D + 4;
  float* part = det_part(grid * prow);
  hipLaunchKernelGGL((rating_tail_kernel<T>), dim3(grid), dim3(256), (part ? 8 * D + 16 : 2 * D) * sizeof(float), s, z, hact, n, D, w2, b2, idx,
                     label, weight, stats, rating_mean, task_w, evaluate, loss_out, dw2, db2, db0, part, npos);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) {
    int rc = RSYS_OK;
    if (!evaluate) {
      rc = launch_reduce_parts(part, grid, prow, D, dw2, s);
      if (rc == RSYS_OK) rc = launch_reduce_parts(part + D, grid, prow, D, db0, s);
      if (rc == RSYS_OK) rc = launch_reduce_parts(part + 2 * D, grid, prow, 1, db2, s);
    }
    return rc == RSYS_OK ? launch_reduce_parts(part + 2 * D + 1, grid, prow, 3, loss_out, s) : rc;
  }
  return RSYS_OK;
}
template int launch_rating_tail<bf16>(bf16*, const bf16*, int, int, const float*, const float*, const int*, const float*, const float*, const float*, float, float, int, float*, float*, float*, float*, hipStream_t, const int*);
template int launch_rating_tail<float>(float*, const float*, int, int, const float*, const float*, const int*, const float*, const float*, const float*, float, float, int, float*, float*, float*, float*, hipStream_t, const int*);

// --------------------------------------------------------------------- inference outputs (model.py:531-538)
// dst[i] = (float)src[i]: the trunk output leaves the device as float32 in one copy
template <typename T>
__global__ void widen_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = to_f32(src[i]);
}
template <typename T>
int launch_widen(const T* src, float* dst, long long n, hipStream_t s) {
  const int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL((widen_kernel<T>), dim3(blocks), dim3(256), 0, s, src, dst, n);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_widen<bf16>(const bf16*, float*, long long, hipStream_t);
template int launch_widen<float>(const float*, float*, long long, hipStream_t);
// out[r] = <h[r, :], w> + b[0]  (second layer of the rating head over every token); one wave per row
template <typename T>
__global__ void rowdot_kernel(const T* __restrict__ h, const float* __restrict__ w, const float* __restrict__ b, float* out, int n, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n) return;
  const T* hr = h + (long long)row * D;
  float acc = 0.f;
  for (int c = l; c < D; c += 64) acc += to_f32(hr[c]) * w[c];
  acc = wave_sum(acc);
  if (l == 0) out[row] = acc + b[0];
}
template <typename T>
int launch_rowdot(const T* h, const float* w, const float* b, float* out, int n, int D, hipStream_t s) {
  hipLaunchKernelGGL((rowdot_kernel<T>), dim3(div_up(n, 4)), dim3(256), 0, s, h, w, b, out, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_rowdot<bf16>(const bf16*, const float*, const float*, float*, int, int, hipStream_t);
template int launch_rowdot<float>(const float*, const float*, const float*, float*, int, int, hipStream_t);

// --------------------------------------------------------------------- column sums
template <typename TS>
__global__ void colsum_kernel(const TS* __restrict__ src, long long ld, long long rows, int cols, float* dst, int rows_per_block, float* part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc = 0.f;
  for (long long r = r0; r < r1; ++r) acc += to_f32(src[r * ld + c]);
  if (part != nullptr) part[(long long)blockIdx.y * cols + c] = acc;
  else if (acc != 0.f) atomicAdd(&dst[c], acc);
}
template <typename TS>
static int colsum_any(const TS* src, long long ld, long long rows, int cols, float* dst, hipStream_t s) {
  int rpb = (int)std::max<long long>(64, (rows + 511) / 512);
  dim3 grid(div_up(cols, 256), div_up(rows, rpb));
  float* part = det_part((long long)grid.y * cols);
  hipLaunchKernelGGL((colsum_kernel<TS>), grid, dim3(256), 0, s, src, ld, rows, cols, dst, rpb, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, (int)grid.y, cols, cols, dst, s);
  return RSYS_OK;
}
int launch_colsum_add(const float* src, long long ld, long long rows, int cols, float* dst, hipStream_t s) {
  return colsum_any<float>(src, ld, rows, cols, dst, s);
}

// dst (bf16) = src (f32) and colsum[c] += sum_r src[r][c] in ONE pass over src: the operand copy of dF and the
// projection-bias gradient of the table backward (model.hip finalize) both stream the same 410 MB.
__global__ __launch_bounds__(1024) void cast_colsum_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long rows, int D,
                                                           float* colsum, int rows_per_block, float* part) {
  __shared__ float red[2048];                    // [row lane][D] partial sums (launcher: 1024 / (D/4) row lanes, D <= 1024... see launcher)
  const int cg = D >> 2;                         // 4-column groups per row (launcher: D % 4 == 0, 1024 % cg == 0)
  const int c = (threadIdx.x % cg) * 4, lane_r = threadIdx.x / cg, nr = 1024 / cg;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float4 acc = make_float4(0, 0, 0, 0);
  long long r = r0 + lane_r;
  for (; r + 3 * nr < r1; r += 4 * nr) {   // four independent rows in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(src + (r + u * nr) * D + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bf16x4 o; o[0] = (bf16)v[u].x; o[1] = (bf16)v[u].y; o[2] = (bf16)v[u].z; o[3] = (bf16)v[u].w;
      *(bf16x4*)(dst + (r + u * nr) * D + c) = o;
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
  }
  for (; r < r1; r += nr) {
    const float4 v = *(const float4*)(src + r * D + c);
    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *(bf16x4*)(dst + r * D + c) = o;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  // row lanes -> one partial sum per column and workgroup (few hundred atomics per address in total)
  if (lane_r < 2) *(float4*)&red[lane_r * D + c] = acc;
  __syncthreads();
  for (int base = 2; base < nr; base += 2) {
    if (lane_r >= base && lane_r < base + 2) {
      float4 o = *(float4*)&red[(lane_r - base) * D + c];
      o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
      *(float4*)&red[(lane_r - base) * D + c] = o;
    }
    __syncthreads();
  }
  for (int cc = threadIdx.x; cc < D; cc += 1024) {
    const float v = red[cc] + (nr > 1 ? red[D + cc] : 0.f);
    if (part != nullptr) part[(long long)blockIdx.x * D + cc] = v;
    else if (v != 0.f) atomicAdd(&colsum[cc], v);
  }
}
int launch_cast_colsum(const float* src, bf16* dst, long long rows, int D, float* colsum, hipStream_t s) {
  ARG_CHECK(D % 4 == 0 && D <= 1024 && 1024 % (D >> 2) == 0, "cast_colsum: D/4 must divide 1024, D <= 1024");
  const int rpb = (int)std::max<long long>(32, (rows + 511) / 512);
  const int grid = div_up(rows, rpb);
  float* part = det_part((long long)grid * D);
  hipLaunchKernelGGL(cast_colsum_kernel, dim3(grid), dim3(1024), 0, s, src, dst, rows, D, colsum, rpb, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, grid, D, D, colsum, s);
  return RSYS_OK;
}

// dstT[c][r] (bf16, row stride ldt) = src[r][c] (f32) and colsum[c] += sum_r src[r][c], one pass over src: the K-contiguous operand
// copy dF^T of the metadata-projection gradient dWp = dF^T Meta (model.hip finalize) straight from the fp32 table gradient -- it used
// to be a row-major bf16 copy (cast_colsum) plus a 2 x 205 MB transpose of that copy.  A workgroup owns `rows_per_block` rows
// (a multiple of 64) and walks them in 64 x 64 tiles through LDS; rows in [rows, rows rounded up to 64) are written as zeros
// (the GEMM's K padding).  Column sums: registers -> LDS -> one partial per column and workgroup.
__global__ __launch_bounds__(256) void cast_transpose_colsum_kernel(const float* __restrict__ src, bf16* __restrict__ dstT, long long rows, int D,
                                                                    long long ldt, float* colsum, int rows_per_block, float* part) {
  __shared__ unsigned short tile[64][72];          // [column][row] of the current 64 x 64 tile (row pitch 144 B: 16-byte aligned reads)
  __shared__ float red[16][64];
  const int t = threadIdx.x;
  const int lr = t >> 4, c4 = (t & 15) * 4;         // loads: rows 4 lr .. 4 lr + 3, columns c4 .. c4 + 3 of the tile (a 4 x 4 block per thread)
  const int oc = t >> 2, or8 = (t & 3) * 16;        // stores: tile column oc, rows or8 .. or8 + 15
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long rend = min((rows + 63) & ~63LL, r0 + rows_per_block);
  for (int ct = 0; ct < D / 64; ++ct) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (long long rb = r0; rb < rend; rb += 64) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long r = rb + 4 * lr + i;
        v[i] = r < rows ? *(const float4*)(src + r * D + ct * 64 + c4) : make_float4(0, 0, 0, 0);
        acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
      }
      {   // the block transposed in registers: one 8-byte LDS store (4 consecutive rows) per column
        bf16x4 o;
        o[0] = (bf16)v[0].x; o[1] = (bf16)v[1].x; o[2] = (bf16)v[2].x; o[3] = (bf16)v[3].x; *(bf16x4*)&tile[c4 + 0][4 * lr] = o;
        o[0] = (bf16)v[0].y; o[1] = (bf16)v[1].y; o[2] = (bf16)v[2].y; o[3] = (bf16)v[3].y; *(bf16x4*)&tile[c4 + 1][4 * lr] = o;
        o[0] = (bf16)v[0].z; o[1] = (bf16)v[1].z; o[2] = (bf16)v[2].z; o[3] = (bf16)v[3].z; *(bf16x4*)&tile[c4 + 2][4 * lr] = o;
        o[0] = (bf16)v[0].w; o[1] = (bf16)v[1].w; o[2] = (bf16)v[2].w; o[3] = (bf16)v[3].w; *(bf16x4*)&tile[c4 + 3][4 * lr] = o;
      }
      __syncthreads();
      {
        const uint4 q0 = *(const uint4*)&tile[oc][or8], q1 = *(const uint4*)&tile[oc][or8 + 8];
        uint4* d = (uint4*)(dstT + (long long)(ct * 64 + oc) * ldt + rb + or8);
        d[0] = q0; d[1] = q1;
      }
      __syncthreads();
    }
    *(float4*)&red[lr][c4] = acc;
    __syncthreads();
    if (t < 64) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += red[k][t];
      if (part != nullptr) part[(long long)blockIdx.x * D + ct * 64 + t] = v;
      else if (v != 0.f) atomicAdd(&colsum[ct * 64 + t], v);
    }
    __syncthreads();
  }
}
int launch_cast_transpose_colsum(const float* src, bf16* dstT, long long rows, int D, long long ldt, float* colsum, hipStream_t s) {
  ARG_CHECK(D % 64 == 0 && ldt % 8 == 0 && ldt >= ((rows + 63) & ~63LL), "cast_transpose_colsum: D % 64, ldt % 8, ldt >= rows rounded up to 64");
  const long long chunks = (rows + 63) / 64;
  const int cpb = (int)std::max<long long>(1, (chunks + 2047) / 2048);
  const int grid = (int)((chunks + cpb - 1) / cpb);
  float* part = det_part((long long)grid * D);
  hipLaunchKernelGGL(cast_transpose_colsum_kernel, dim3(grid), dim3(256), 0, s, src, dstT, rows, D, ldt, colsum, cpb * 64, part);
  HIP_CHECK(hipGetLastError());
  if (part != nullptr) return launch_reduce_parts(part, grid, D, D, colsum, s);
  return RSYS_OK;
}

// --------------------------------------------------------------------- embedding scatter-add (K16)
// gE[id'] += gx0[2n]; one wave per interaction, 256 contiguous bytes per atomic wave-instruction
// (MI355X_MICROARCH.md "Global float atomics": full rate for this shape).
__global__ void embedding_scatter_kernel(const float* __restrict__ gx0, BatchDev b, int V, int D, float* gE) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (n >= b.N) return;
  int id = b.m_matchedid[n];
  id = id == -1 ? V : id;
  const float* src = gx0 + (long long)(2 * n) * D;
  float* dst = gE + (long long)id * D;
  for (int c = l; c < D; c += 64) atomicAdd(&dst[c], src[c]);
}
int launch_embedding_scatter_add(const float* gx0, const BatchDev& b, int V, int D, float* gE, hipStream_t s) {
  hipLaunchKernelGGL(embedding_scatter_kernel, dim3(div_up(b.N, 4)), dim3(256), 0, s, gx0, b, V, D, gE);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// --------------------------------------------------------------------- action embedding: small-table gradients
__global__ __launch_bounds__(256) void action_small_bwd_kernel(const float* __restrict__ gf, BatchDev b, SmallParams sp,
                                                               float* g_cos, float* g_sin, float* g_status, float* g_gender,
                                                               float* g_source) {
  // LDS accumulators: cos(2) sin(2) gender(5*4) source(5*4) status(10*16) -> sized for vocab <= 15 each
  __shared__ float acc[4 + 16 * 4 + 16 * 4 + 16 * 16];
  const int NG = (sp.n_gender + 1) * 4, NS = (sp.n_source + 1) * 4, NST = (sp.n_status + 1) * 16;
  float* a_g = acc + 4; float* a_s = a_g + 64; float* a_st = a_s + 64;
  for (int c = threadIdx.x; c < 4 + 64 + 64 + 256; c += 256) acc[c] = 0.f;
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < b.N; i += gridDim.x * blockDim.x) {
    const float* g = gf + (long long)i * 32;
    float p0, p1; double ts;
    periodic_args(b.time[i], sp.min_ts, p0, p1, ts);
    atomicAdd(&acc[0], -sinf(p0 + sp.per_cos[0]) * g[1]);
    atomicAdd(&acc[1], -sinf(p1 + sp.per_cos[1]) * g[2]);
    atomicAdd(&acc[2], cosf(p0 + sp.per_sin[0]) * g[3]);
    atomicAdd(&acc[3], cosf(p1 + sp.per_sin[1]) * g[4]);
    int gi = b.gender[i]; gi = gi == -1 ? sp.n_gender : gi;
    int si = b.source[i]; si = si == -1 ? sp.n_source : si;
    int st = b.m_status[i]; st = st == -1 ? sp.n_status : st;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&a_g[gi * 4 + k], g[5 + k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&a_s[si * 4 + k], g[9 + k]);
#pragma unroll
    for (int k = 0; k < 16; ++k) atomicAdd(&a_st[st * 16 + k], g[15 + k]);
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 2) atomicAdd(&g_cos[t], acc[t]);
  else if (t < 4) atomicAdd(&g_sin[t - 2], acc[t]);
  for (int c = t; c < NG; c += 256) atomicAdd(&g_gender[c], a_g[c]);
  for (int c = t; c < NS; c += 256) atomicAdd(&g_source[c], a_s[c]);
  for (int c = t; c < NST; c += 256) atomicAdd(&g_status[c], a_st[c]);
}
// deterministic form: one workgroup per output cell; a thread sums the interactions i = t, t + 256, ... that belong to the cell,
// the workgroup adds its threads in a fixed tree, and the cell has this one contributor
__global__ __launch_bounds__(256) void action_small_bwd_det_kernel(const float* __restrict__ gf, BatchDev b, SmallParams sp,
                                                                   float* g_cos, float* g_sin, float* g_status, float* g_gender,
                                                                   float* g_source) {
  __shared__ float red[16];
  const int NG = (sp.n_gender + 1) * 4, NS = (sp.n_source + 1) * 4;
  int cell = blockIdx.x;
  float acc = 0.f;
  float* dst;
  if (cell < 4) {
    dst = cell < 2 ? g_cos + cell : g_sin + (cell - 2);
    for (int i = threadIdx.x; i < b.N; i += 256) {
      float p0, p1; double ts;
      periodic_args(b.time[i], sp.min_ts, p0, p1, ts);
      const float* g = gf + (long long)i * 32;
      acc += cell == 0 ? -sinf(p0 + sp.per_cos[0]) * g[1] : cell == 1 ? -sinf(p1 + sp.per_cos[1]) * g[2]
           : cell == 2 ? cosf(p0 + sp.per_sin[0]) * g[3] : cosf(p1 + sp.per_sin[1]) * g[4];
    }
  } else {
    cell -= 4;
    const int* index; int mask_row, width, col0;
    if (cell < NG) { index = b.gender; mask_row = sp.n_gender; width = 4; col0 = 5; dst = g_gender + cell; }
    else if (cell < NG + NS) { cell -= NG; index = b.source; mask_row = sp.n_source; width = 4; col0 = 9; dst = g_source + cell; }
    else { cell -= NG + NS; index = b.m_status; mask_row = sp.n_status; width = 16; col0 = 15; dst = g_status + cell; }
    const int row = cell / width, k = cell % width;
    for (int i = threadIdx.x; i < b.N; i += 256) {
      int r = index[i]; r = r == -1 ? mask_row : r;
      if (r == row) acc += gf[(long long)i * 32 + col0 + k];
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) *dst += acc;
}
int launch_action_small_bwd(const float* gf, const BatchDev& b, const SmallParams& sp, float* g_per_cos, float* g_per_sin,
                            float* g_status, float* g_gender, float* g_source, hipStream_t s) {
  ARG_CHECK(sp.n_gender < 16 && sp.n_source < 16 && sp.n_status < 16, "action_small_bwd: small vocab sizes must be < 16");
  if (g_det.part != nullptr) {
    const int cells = 4 + (sp.n_gender + 1) * 4 + (sp.n_source + 1) * 4 + (sp.n_status + 1) * 16;
    hipLaunchKernelGGL(action_small_bwd_det_kernel, dim3(cells), dim3(256), 0, s, gf, b, sp, g_per_cos, g_per_sin, g_status, g_gender, g_source);
    HIP_CHECK(hipGetLastError());
    return RSYS_OK;
  }
  int grid = std::min(div_up(b.N, 256), 256);
  hipLaunchKernelGGL(action_small_bwd_kernel, dim3(grid), dim3(256), 0, s, gf, b, sp, g_per_cos, g_per_sin, g_status,
                     g_gender, g_source);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
