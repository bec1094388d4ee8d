// Multi-tensor optimizer kernels over the flat fp32 parameter / gradient buffers
// (HBM-bound; 16-byte accesses, grid-stride).  transformer.py:273 (clip_grad_norm_)
// and :285-298 (AdamW, decoupled decay 0.1 on tensors with dim>=2, betas 0.9/0.95).
#include "kernels.hpp"
#include "switches.hpp"

namespace rsys {

// Sum of squares of a flat fp32 range, ALWAYS in a fixed order (round 6): per-workgroup partial sums (fixed grid for a given n, fixed
// in-block tree) and one small workgroup that adds them in index order into `out`.  The float-atomic form it replaces gave every rank of
// a data-parallel job its own rounding of the SAME all-reduced gradient's norm, hence its own clip coefficient, and the replicas drifted
// apart by an ulp per step -- found by the replica-consistency check (dist.assert_replicas_equal) on three in-process ranks.  Four
// independent 16-byte loads per thread per trip (the one-load loop ran at 0.33 of the HBM peak on cfg-2's 133 MB).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const float4* g4 = (const float4*)g;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];   // (nontemporal loads here: +0.1 ms on the step, AdamW re-reads the buffer)
    a0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
    a1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    a2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
    a3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
  }
  for (; i < n4; i += stride) {
    const float4 v = g4[i];
    a0 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long long k = (n4 << 2) + threadIdx.x; k < n; k += blockDim.x) a1 += g[k] * g[k];
  const float acc = block_sum((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* out, int overwrite) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) *out = overwrite ? a : *out + a;
}

int sumsq_parts() { return 2048; }
int launch_sumsq(const float* g, long long n, float* out, float* part, hipStream_t s, bool overwrite) {
  ARG_CHECK(part != nullptr, "sumsq: no partial-sum buffer");
  if (n <= 0) { if (overwrite) HIP_CHECK(hipMemsetAsync(out, 0, 4, s)); return RSYS_OK; }
  ARG_CHECK(((uintptr_t)g % 16) == 0, "sumsq: the range must start on a 16-byte boundary");
  const int grid = (int)std::min<long long>(((n >> 2) + 1023) / 1024 + 1, 2048);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, g, n, part);
  HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, part, grid, out, overwrite ? 1 : 0);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// Replica-consistency checksum of a flat fp32 range (DDP's constructor broadcasts rank 0's parameters,
// transformer.py:678-682; here every rank initialises from the same seed and the ranks COMPARE): fp64 sum, fp64 sum of squares and the
// wrapping 64-bit sum of the floats' bit patterns.  Fixed grid, fixed in-block tree, partials added in workgroup order by one
// workgroup: the same values give the same three words on every rank, and the integer word notices any differing bit that the two
// float sums could round away.  acc[3 * gridDim.x] holds the partials; out = {sum, sumsq, bits low 32, bits high 32} (exact in doubles).
static constexpr int CHECKSUM_GRID = 1024;
__global__ __launch_bounds__(256) void checksum_part_kernel(const float* __restrict__ p, long long n, double* part, int add) {
  __shared__ double s0[256], s1[256];
  __shared__ unsigned long long s2[256];
  double a = 0.0, b = 0.0;
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = p[i];
    a += (double)v; b += (double)v * (double)v; c += (unsigned long long)__float_as_uint(v) * (unsigned long long)((i & 1023) + 1);
  }
  s0[threadIdx.x] = a; s1[threadIdx.x] = b; s2[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; s2[threadIdx.x] += s2[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* q = part + 3 * blockIdx.x;
    unsigned long long prev = add ? (unsigned long long)__double_as_longlong(q[2]) : 0ull;
    q[0] = (add ? q[0] : 0.0) + s0[0]; q[1] = (add ? q[1] : 0.0) + s1[0];
    q[2] = __longlong_as_double((long long)(prev + s2[0]));   // (the integer word travels in a double-sized slot)
  }
}
__global__ __launch_bounds__(64) void checksum_final_kernel(const double* part, int nparts, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0.0, b = 0.0; unsigned long long c = 0;
  for (int i = 0; i < nparts; ++i) { a += part[3 * i]; b += part[3 * i + 1]; c += (unsigned long long)__double_as_longlong(part[3 * i + 2]); }
  out[0] = a; out[1] = b; out[2] = (double)(unsigned int)(c & 0xffffffffull); out[3] = (double)(unsigned int)(c >> 32);
}
// ranges: n_ranges pairs {first element, one past the last}; part: 3 * CHECKSUM_GRID doubles; out: 4 doubles (device)
int launch_checksum(const float* p, const long long* ranges, int n_ranges, double* part, double* out, hipStream_t s) {
  for (int r = 0; r < n_ranges; ++r) {
    const long long lo = ranges[2 * r], hi = ranges[2 * r + 1];
    ARG_CHECK(hi >= lo, "checksum: empty or reversed range");
    hipLaunchKernelGGL(checksum_part_kernel, dim3(CHECKSUM_GRID), dim3(256), 0, s, p + lo, hi - lo, part, r > 0 ? 1 : 0);
    HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(checksum_final_kernel, dim3(1), dim3(64), 0, s, part, CHECKSUM_GRID, out);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
int checksum_scratch_doubles() { return 3 * CHECKSUM_GRID + 4; }

// clip coefficient of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (norm + 1e-6)); grads are first
// divided by grad_div (data-parallel mean, folded in here instead of a separate pass)
__device__ __forceinline__ float grad_coef(const float* sumsq, float grad_div, float max_norm) {
  float inv = 1.f / grad_div;
  if (sumsq == nullptr || max_norm <= 0.f) return inv;
  float norm = sqrtf(*sumsq) * inv;
  float c = max_norm / (norm + 1e-6f);
  return inv * (c < 1.f ? c : 1.f);
}

template <typename T, bool NT = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* p, float* g, float* m, float* v, T* shadow, long long n_decay,
                                                    long long n_total, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, const float* sumsq, float grad_div,
                                                    float max_norm, int zero_grad, long long sh_skip_lo, long long sh_skip_hi) {
  const float coef = grad_coef(sumsq, grad_div, max_norm);
  const long long n4 = n_total >> 2;  // n_decay and n_total are multiples of 4 (the flat layout pads every tensor)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pv, gv, mv, vv;
    if constexpr (NT) {   // (experiment: every byte is touched once per step and the buffers are 16x the Infinity Cache)
      typedef __attribute__((ext_vector_type(4))) float nt_f32x4;
      auto ld = [](const float* q) { const nt_f32x4 t = __builtin_nontemporal_load((const nt_f32x4*)q); return make_float4(t[0], t[1], t[2], t[3]); };
      pv = ld(p + 4 * i); gv = ld(g + 4 * i); mv = ld(m + 4 * i); vv = ld(v + 4 * i);
    } else { pv = ((float4*)p)[i]; gv = ((float4*)g)[i]; mv = ((float4*)m)[i]; vv = ((float4*)v)[i]; }
    const float decay = (i * 4 < n_decay) ? (1.f - lr * wd) : 1.f;
    float* pp = (float*)&pv; float* gg = (float*)&gv; float* mm = (float*)&mv; float* vvp = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = gg[k] * coef;
      float pk = pp[k] * decay;
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vvp[k] = b2 * vvp[k] + (1.f - b2) * gk * gk;
      float denom = sqrtf(vvp[k]) / bc2_sqrt + eps;
      pp[k] = pk - (lr / bc1) * (mm[k] / denom);
    }
    if constexpr (NT) {
      typedef __attribute__((ext_vector_type(4))) float nt_f32x4;
      auto st = [](float* q, const float4& x) { const nt_f32x4 t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, (nt_f32x4*)q); };
      st(p + 4 * i, pv); st(m + 4 * i, mv); st(v + 4 * i, vv);
      if (zero_grad) st(g + 4 * i, make_float4(0, 0, 0, 0));
    } else {
      ((float4*)p)[i] = pv; ((float4*)m)[i] = mv; ((float4*)v)[i] = vv;
      if (zero_grad) ((float4*)g)[i] = make_float4(0, 0, 0, 0);
    }
    if constexpr (is_bf16<T>::value) {
      if (shadow && !(i * 4 >= sh_skip_lo && i * 4 < sh_skip_hi)) {   // (no bf16 copy where no kernel reads one: the item table E)
        bf16x4 sv; sv[0] = (bf16)pp[0]; sv[1] = (bf16)pp[1]; sv[2] = (bf16)pp[2]; sv[3] = (bf16)pp[3];
        ((bf16x4*)shadow)[i] = sv;
      }
    }
  }
}

template <typename T>
int launch_adamw(float* p, float* g, float* m, float* v, T* shadow, long long n_decay, long long n_total, float lr,
                 float b1, float b2, float eps, float wd, int step, const float* sumsq, float grad_div, float max_norm,
                 int zero_grad, hipStream_t s, long long sh_skip_lo, long long sh_skip_hi) {
  ARG_CHECK(n_decay % 4 == 0 && n_total % 4 == 0 && sh_skip_lo % 4 == 0 && sh_skip_hi % 4 == 0, "adamw: flat sizes must be multiples of 4");
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  // Nontemporal loads and stores (round 6): every byte of the four flat buffers is touched once per step and together they are 16 x the
  // Infinity Cache at cfg-3 -- 0.77 -> 0.66 ms (6.3 TB/s), the step -0.15 ms on a same-box A/B, neutral at cfg-2 (profiles/r6_ab_adamw_nontemporal.log).
  // RSYS_DEBUG_ADAMW: bit 0 = plain loads / stores (the A/B partner), bit 1 = 8192 workgroups
  const int mode = sw().debug_adamw;
  int grid = (int)std::min<long long>(((n_total >> 2) + 255) / 256, (mode & 2) ? 8192 : 4096);
  if (!(mode & 1))
    hipLaunchKernelGGL((adamw_kernel<T, true>), dim3(grid), dim3(256), 0, s, p, g, m, v, shadow, n_decay, n_total, lr, b1, b2, eps,
                       wd, bc1, bc2s, sumsq, grad_div, max_norm, zero_grad, sh_skip_lo, sh_skip_hi);
  else
  hipLaunchKernelGGL((adamw_kernel<T>), dim3(grid), dim3(256), 0, s, p, g, m, v, shadow, n_decay, n_total, lr, b1, b2, eps,
                     wd, bc1, bc2s, sumsq, grad_div, max_norm, zero_grad, sh_skip_lo, sh_skip_hi);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_adamw<bf16>(float*, float*, float*, float*, bf16*, long long, long long, float, float, float, float, float, int, const float*, float, float, int, hipStream_t, long long, long long);
template int launch_adamw<float>(float*, float*, float*, float*, float*, long long, long long, float, float, float, float, float, int, const float*, float, float, int, hipStream_t, long long, long long);

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(src[i]);
}
template <typename T>
int launch_cast(const float* src, T* dst, long long n, hipStream_t s) {
  int grid = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL((cast_kernel<T>), dim3(grid), dim3(256), 0, s, src, dst, n);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_cast<bf16>(const float*, bf16*, long long, hipStream_t);
template int launch_cast<float>(const float*, float*, long long, hipStream_t);

// dst[r][c] = a[r][c] + bias[c]: the start value of the split-K tail of the fused item table (model_forward.hip table_forward)
__global__ void add_bias_rows_kernel(const float* __restrict__ a, const float* __restrict__ bias, float* __restrict__ dst, long long n, int D) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 x = *(const float4*)(a + i), b = *(const float4*)(bias + (int)(i % D));
    *(float4*)(dst + i) = make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w);
  }
}
int launch_add_bias_rows(const float* a, const float* bias, float* dst, long long rows, int D, hipStream_t s) {
  ARG_CHECK(D % 4 == 0, "add_bias_rows: D % 4");
  const long long n = rows * D;
  int grid = (int)std::min<long long>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(add_bias_rows_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, a, bias, dst, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

__global__ void scale_kernel(float* g, long long n, const float* sumsq, float grad_div, float max_norm) {
  const float coef = grad_coef(sumsq, grad_div, max_norm);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) g[i] *= coef;
}
int launch_scale(float* g, long long n, const float* sumsq, float grad_div, float max_norm, hipStream_t s) {
  int grid = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, s, g, n, sumsq, grad_div, max_norm);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// ---- device-side N(0, std) fill (Box-Muller on Philox), used by random init and the synthetic metadata table
__device__ __forceinline__ void normal4(const Philox& ph, unsigned long long ctr, unsigned int stream, float out[4]) {
  uint32_t r[4];
  ph.gen(ctr, stream, r);
  float u0 = fmaxf(u01(r[0]), 5.9604645e-8f), u1 = u01(r[1]);
  float u2 = fmaxf(u01(r[2]), 5.9604645e-8f), u3 = u01(r[3]);
  float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
  out[0] = ra * cosf(6.2831853f * u1); out[1] = ra * sinf(6.2831853f * u1);
  out[2] = rb * cosf(6.2831853f * u3); out[3] = rb * sinf(6.2831853f * u3);
}

// dst[0, n) = elements [first, first + n) of the normal stream (seed, stream); first % 4 == 0
__global__ void fill_normal_kernel(float* dst, long long n, float std, unsigned long long seed, unsigned int stream, long long first4) {
  Philox ph(seed);
  const long long n4 = (n + 3) >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float z[4];
    normal4(ph, (unsigned long long)(i + first4), stream, z);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < n) dst[i * 4 + k] = z[k] * std;
  }
}
int launch_fill_normal(float* dst, long long n, float std, unsigned long long seed, unsigned int stream, hipStream_t s, long long first) {
  ARG_CHECK(first % 4 == 0, "fill_normal: the start offset must be a multiple of 4");
  int grid = (int)std::min<long long>((((n + 3) >> 2) + 255) / 256, 8192);
  hipLaunchKernelGGL(fill_normal_kernel, dim3(grid), dim3(256), 0, s, dst, n, std, seed, stream, first >> 2);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename T>
__global__ void fill_normal_t_kernel(T* dst, long long rows, int cols, long long ld, float std, unsigned long long seed, long long row0) {
  Philox ph(seed);
  const int c4 = (cols + 3) >> 2;
  const long long total = rows * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i / c4; int c = (int)(i % c4) * 4;
    float z[4];
    normal4(ph, (unsigned long long)(i + row0 * c4), 7u, z);   // (row0: the rows are rows [row0, row0 + rows) of a larger table)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c + k < cols) dst[r * ld + c + k] = from_f32<T>(z[k] * std);
  }
}
template <typename T>
int launch_fill_normal_t(T* dst, long long rows, int cols, long long ld, float std, unsigned long long seed, hipStream_t s, long long row0) {
  if (rows <= 0) return RSYS_OK;
  long long total = rows * ((cols + 3) >> 2);
  int grid = (int)std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL((fill_normal_t_kernel<T>), dim3(grid), dim3(256), 0, s, dst, rows, cols, ld, std, seed, row0);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_fill_normal_t<bf16>(bf16*, long long, int, long long, float, unsigned long long, hipStream_t, long long);
template int launch_fill_normal_t<float>(float*, long long, int, long long, float, unsigned long long, hipStream_t, long long);


// dst[c][r] = src[r][c] for a batch of bf16 matrices: 64x64 tiles through LDS, 16-byte global accesses on both sides
__global__ __launch_bounds__(256) void transpose_bf16_kernel(TransposeBatch b) {
  __shared__ unsigned short tile[64][66];
  const TransposeJob j = b.job[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (r0 >= j.rows || c0 >= j.cols) return;
  const unsigned short* src = (const unsigned short*)j.src;
  unsigned short* dst = (unsigned short*)j.dst;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = (t >> 3) + 32 * it, c = (t & 7) * 8;
    if (r0 + r < j.rows) {
      if (c0 + c + 8 <= j.cols) {
        const uint4 q = *(const uint4*)(src + (long long)(r0 + r) * j.ld_src + c0 + c);
        const unsigned short* e = (const unsigned short*)&q;
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[r][c + k] = e[k];
      } else {
        for (int k = 0; k < 8; ++k) tile[r][c + k] = (c0 + c + k < j.cols) ? src[(long long)(r0 + r) * j.ld_src + c0 + c + k] : (unsigned short)0;
      }
    } else {
      for (int k = 0; k < 8; ++k) tile[r][c + k] = 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = (t >> 3) + 32 * it, r = (t & 7) * 8;   // output row c0+c, output columns r0+r..+8
    if (c0 + c < j.cols) {
      __attribute__((aligned(16))) unsigned short e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = tile[r + k][c];
      if (r0 + r + 8 <= j.rows) *(uint4*)(dst + (long long)(c0 + c) * j.ld_dst + r0 + r) = *(const uint4*)e;
      else for (int k = 0; k < 8; ++k) if (r0 + r + k < j.rows) dst[(long long)(c0 + c) * j.ld_dst + r0 + r + k] = e[k];
    }
  }
}

int launch_transpose_bf16(const TransposeBatch& b, hipStream_t s) {
  if (b.n <= 0) return RSYS_OK;
  int mr = 0, mc = 0;
  for (int i = 0; i < b.n; ++i) { mr = b.job[i].rows > mr ? b.job[i].rows : mr; mc = b.job[i].cols > mc ? b.job[i].cols : mc; }
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((mc + 63) / 64, (mr + 63) / 64, b.n), dim3(256), 0, s, b);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
