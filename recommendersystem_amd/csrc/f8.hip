// fp8 trunk: tensor-wise dynamically scaled e4m3 / e5m2 operands for the transformer blocks' linears.
//
// The reference converts every nn.Linear under `transformers.` with torchao's "tensorwise" recipe before pretraining
// (transformer.py:671-676: convert_to_float8_training(model, Float8LinearConfig.from_recipe_name("tensorwise"),
// module_filter_fn = fqn.startswith("transformers."))).  torchao is not in this image, so what follows is a restatement
// of its published recipe; oracle/model_np.py restates the same recipe in numpy and is pinned on the CPU to torch's float8 casts
// and torch._scaled_mm (the primitives torchao's Float8Linear calls); torchao's amax -> scale formula itself is restated, not pinned:
//   forward   y  = (q_e4m3(x sx) . q_e4m3(W sw)^T) / (sx sw)      sx = 448 / amax|x|, sw = 448 / amax|W|   (per TENSOR, this step's values)
//   backward  dx = (q_e5m2(dy sg) . q_e4m3(W sw))   / (sg sw)      sg = 57344 / amax|dy|
//             dW = (q_e5m2(dy sg)^T . q_e4m3(x sx)) / (sg sx)
// scale = float32(float64(FMAX) / max(float64(amax), 1e-12)); q = round-to-nearest-even, saturating; products accumulate in fp32;
// the descale multiplies the fp32 sum before the output is rounded to bf16.  Every one of q, k, v, o, w1, w3, w2 is its own Linear
// with its own weight scale and its own gradient scale; the fused QKV / W13 GEMMs therefore carry per-segment scales
// (GemmParams::f8_*).
//
// This file: amax of a bf16 / f32 tensor (column segments), the cast to fp8 with the consumer's descales written beside it, and
// the per-step refresh of the fp8 weight copies (row-major for the forward, transposed with the gradient's K order for dx).
#include "kernels.hpp"

namespace rsys {

namespace {

__device__ __forceinline__ float f8_fmax(int fmt) { return fmt == F8_E5M2 ? 57344.f : 448.f; }

// torchao.float8 amax_to_scale: float64 division, amax clamped at 1e-12, result kept in float32
__device__ __forceinline__ float f8_scale_of(float amax, int fmt) {
  const double a = (double)amax;
  return (float)((double)f8_fmax(fmt) / (a > 1e-12 ? a : 1e-12));
}

// four floats (already scaled) -> four fp8 bytes, round to nearest even, saturating at +-FMAX
template <int FMT>
__device__ __forceinline__ unsigned int f8_pack4(float a, float b, float c, float d) {
  const float m = FMT == F8_E5M2 ? 57344.f : 448.f;
  a = fminf(fmaxf(a, -m), m); b = fminf(fmaxf(b, -m), m); c = fminf(fmaxf(c, -m), m); d = fminf(fmaxf(d, -m), m);
  int v = 0;
  if constexpr (FMT == F8_E5M2) {
    v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
  } else {
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  }
  return (unsigned int)v;
}

// column segment of a 16-column group starting at column c0
__device__ __forceinline__ int f8_seg_of(int c0, int layout, int seg_cols, int seg_rep) {
  if (layout == F8_LAYOUT_SEGS) { const int u = c0 / seg_cols; return u < seg_rep ? 0 : u - seg_rep + 1; }
  if (layout == F8_LAYOUT_SWIGLU) return (c0 >> 4) & 1;
  return 0;
}
__host__ __device__ __forceinline__ int f8_nseg(int n, int layout, int seg, int seg_rep) {
  return layout == F8_LAYOUT_SEGS ? n / seg - seg_rep + 1 : (layout == F8_LAYOUT_SWIGLU ? 2 : 1);
}

// ---- amax over column segments: thread = one group of 16 columns of one row per trip
template <typename T>
__global__ __launch_bounds__(256) void f8_amax_kernel(const T* __restrict__ src, long long ld, int rows, int cols, const int* __restrict__ rows_dev,
                                                      int layout, int seg_cols, int seg_rep, float* __restrict__ amax) {
  __shared__ float red[16];
  const int R = rows_dev ? min(*rows_dev, rows) : rows;
  const int cg = cols >> 4;
  const long long total = (long long)R * cg;
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cg), g = (int)(i - (long long)r * cg);
    const T* p = src + (long long)r * ld + g * 16;
    float m = 0.f;
    if constexpr (sizeof(T) == 2) {
      const bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) m = fmaxf(m, fmaxf(fabsf((float)a[k]), fabsf((float)b[k])));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float4 a = *(const float4*)(p + 4 * q); m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)))); }
    }
    const int sg = f8_seg_of(g * 16, layout, seg_cols, seg_rep);
    mx[0] = sg == 0 ? fmaxf(mx[0], m) : mx[0];
    mx[1] = sg == 1 ? fmaxf(mx[1], m) : mx[1];
    mx[2] = sg == 2 ? fmaxf(mx[2], m) : mx[2];
    mx[3] = sg == 3 ? fmaxf(mx[3], m) : mx[3];
  }
  const int nseg = f8_nseg(cols, layout, seg_cols, seg_rep);
  for (int sgi = 0; sgi < nseg; ++sgi) {
    const float v = block_max(mx[sgi], red);
    if (threadIdx.x == 0 && v > 0.f) f8_amax_note(amax + sgi, v);
  }
}

// ---- cast: dst[r][c'] = q(src[r][c] * scale[seg(c)]); block 0 also writes the consumer's descales.
// SWIGLU layout: source columns are [16 a | 16 b] blocks (the W13 interleave); destination columns are [all a | all b].
template <typename T, int FMT>
__global__ __launch_bounds__(256) void f8_cast_kernel(F8Cast c) {
  const int R = c.rows_dev ? min(*c.rows_dev, c.rows) : c.rows;
  const int nseg = f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep);
  __shared__ float s_amax[4];
  if (threadIdx.x < 64) {   // an amax slot is 64 shards (common.hpp f8_amax_note): lane j reads shard j
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = wave_max(i < nseg ? c.amax[threadIdx.x * F8_AMAX_SHARD + i] : 0.f);
      if (threadIdx.x == 0) s_amax[i] = v;
    }
  }
  __syncthreads();
  float sc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sc[i] = i < nseg ? f8_scale_of(s_amax[i], FMT) : 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0 && c.desc != nullptr) {
    if (c.desc_mode == 1) {          // output-column units: one activation scale, n_w weight scales
      const float ia = 1.0f / sc[0];
      const int units = c.n_w - 1 + c.w_rep;
      for (int u = 0; u < units; ++u) c.desc[u] = ia * (1.0f / f8_scale_of(c.wamax[u < c.w_rep ? 0 : u - c.w_rep + 1], F8_E4M3));
    } else if (c.desc_mode == 2) {   // K segments: segment j = (gradient scale j, weight scale j)
      float cj[4];
      for (int j = 0; j < nseg; ++j) cj[j] = (1.0f / sc[j]) * (1.0f / f8_scale_of(c.wamax[j], F8_E4M3));
      c.desc[0] = cj[nseg - 1];
      for (int j = 0; j + 1 < nseg; ++j) c.desc[16 + j] = cj[j] / cj[j + 1];
    }
  }
  const int cg = c.cols >> 4;
  // rows [R, rows_pad) of the copy are zero-filled (K tails / row tiles of a consumer that runs on whole tiles)
  const int Rz = c.rows_dev ? min(c.rows, (R + 255) / 256 * 256) : R;
  const long long total = (long long)Rz * cg;
  const T* src = (const T*)c.src;
  const int half = c.cols >> 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cg), g = (int)(i - (long long)r * cg);
    const int sg = f8_seg_of(g * 16, c.layout, c.seg_cols, c.seg_rep);
    const int dcol = c.layout == F8_LAYOUT_SWIGLU ? (sg ? half : 0) + (g >> 1) * 16 : g * 16;
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    if (r < R) {
      const float s = sg == 0 ? sc[0] : (sg == 1 ? sc[1] : (sg == 2 ? sc[2] : sc[3]));
      const T* p = src + (long long)r * c.ld_src + g * 16;
      float v[16];
      if constexpr (sizeof(T) == 2) {
        const bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (float)a[k] * s; v[8 + k] = (float)b[k] * s; }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 a = *(const float4*)(p + 4 * q); v[4 * q] = a.x * s; v[4 * q + 1] = a.y * s; v[4 * q + 2] = a.z * s; v[4 * q + 3] = a.w * s; }
      }
      out.x = f8_pack4<FMT>(v[0], v[1], v[2], v[3]); out.y = f8_pack4<FMT>(v[4], v[5], v[6], v[7]);
      out.z = f8_pack4<FMT>(v[8], v[9], v[10], v[11]); out.w = f8_pack4<FMT>(v[12], v[13], v[14], v[15]);
    }
    *(uint4*)(c.dst + (long long)r * c.ld_dst + dcol) = out;
  }
}

// The same cast for a tensor whose transposed copy is wanted too.  One wave per 128-row x 128-column tile, lane = a 16 x 16 block
// (row block l >> 3, column block l & 7): the lane converts its block four rows at a time -- 16-byte row-major stores -- and
// transposes the 4 x 4 byte blocks in registers (v_perm_b32), so that after sixteen rows it holds, per column, the 16 bytes of its
// rows and stores them as the transposed copy.  No LDS; a row is read in 256 contiguous bytes, both copies leave in 128-byte runs.
__device__ __forceinline__ void f8_transpose4x4(unsigned int x0, unsigned int x1, unsigned int x2, unsigned int x3, unsigned int (&t)[4]) {
  const unsigned int u0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), u1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
  const unsigned int v0 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), v1 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
  t[0] = __builtin_amdgcn_perm(v0, u0, 0x05040100u); t[1] = __builtin_amdgcn_perm(v0, u0, 0x07060302u);
  t[2] = __builtin_amdgcn_perm(v1, u1, 0x05040100u); t[3] = __builtin_amdgcn_perm(v1, u1, 0x07060302u);
}

template <typename T, int FMT>
__global__ __launch_bounds__(256) void f8_cast_t_kernel(F8Cast c) {
  __shared__ float s_amax[4];
  __shared__ float s_xamax;
  const int nseg = f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep);
  const int t = threadIdx.x;
  if (t < 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = wave_max(i < nseg ? c.amax[t * F8_AMAX_SHARD + i] : 0.f);
      if (t == 0) s_amax[i] = v;
    }
    if (c.desc_dw != nullptr) { const float v = wave_max(c.xamax[t * F8_AMAX_SHARD]); if (t == 0) s_xamax = v; }
  }
  __syncthreads();
  float sc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sc[i] = i < nseg ? f8_scale_of(s_amax[i], FMT) : 0.f;
  if (blockIdx.x == 0 && t == 0) {
    if (c.desc != nullptr && c.desc_mode == 1) {
      const float ia = 1.0f / sc[0];
      const int units = c.n_w - 1 + c.w_rep;
      for (int u = 0; u < units; ++u) c.desc[u] = ia * (1.0f / f8_scale_of(c.wamax[u < c.w_rep ? 0 : u - c.w_rep + 1], F8_E4M3));
    } else if (c.desc != nullptr && c.desc_mode == 2) {
      float cj[4];
      for (int j = 0; j < nseg; ++j) cj[j] = (1.0f / sc[j]) * (1.0f / f8_scale_of(c.wamax[j], F8_E4M3));
      c.desc[0] = cj[nseg - 1];
      for (int j = 0; j + 1 < nseg; ++j) c.desc[16 + j] = cj[j] / cj[j + 1];
    }
    if (c.desc_dw != nullptr) {   // weight-gradient product: row unit u of dY^T against the forward operand's scale
      const float ix = 1.0f / f8_scale_of(s_xamax, F8_E4M3);
      for (int u = 0; u < c.dw_units; ++u) {
        const int sg = c.layout == F8_LAYOUT_SEGS ? (u < c.seg_rep ? 0 : u - c.seg_rep + 1) : (c.layout == F8_LAYOUT_SWIGLU ? u : 0);
        c.desc_dw[u] = (1.0f / sc[sg]) * ix;
      }
    }
  }
  const int ct = (c.cols + 127) >> 7;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (t >> 6);
  if (tile >= (c.rows >> 7) * ct) return;
  const int l = t & 63;
  const int r0 = (tile / ct) * 128 + (l >> 3) * 16, col = (tile % ct) * 128 + (l & 7) * 16;
  if (col >= c.cols) return;
  const int sg = f8_seg_of(col, c.layout, c.seg_cols, c.seg_rep);
  const int dcol = c.layout == F8_LAYOUT_SWIGLU ? (sg ? (c.cols >> 1) : 0) + (col >> 5) * 16 : col;
  const float s = sg == 0 ? sc[0] : (sg == 1 ? sc[1] : (sg == 2 ? sc[2] : sc[3]));
  const T* src = (const T*)c.src + (long long)r0 * c.ld_src + col;
  unsigned char* dst = c.dst + (long long)r0 * c.ld_dst + dcol;
  unsigned int acc[16][4];   // [column of the block][4 rows per dword]
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    unsigned int rw[4][4];   // [row of the chunk][4 columns per dword]
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const T* p = src + (long long)(ch * 4 + rr) * c.ld_src;
      float v[16];
      if constexpr (sizeof(T) == 2) {
        const bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[q] = (float)a[q] * s; v[8 + q] = (float)b[q] * s; }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 a = *(const float4*)(p + 4 * q); v[4 * q] = a.x * s; v[4 * q + 1] = a.y * s; v[4 * q + 2] = a.z * s; v[4 * q + 3] = a.w * s; }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) rw[rr][g] = f8_pack4<FMT>(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
      *(uint4*)(dst + (long long)(ch * 4 + rr) * c.ld_dst) = make_uint4(rw[rr][0], rw[rr][1], rw[rr][2], rw[rr][3]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned int tq[4];
      f8_transpose4x4(rw[0][g], rw[1][g], rw[2][g], rw[3][g], tq);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j][ch] = tq[j];
    }
  }
  unsigned char* dt = c.dst_t + (long long)dcol * c.ld_dst_t + r0;
#pragma unroll
  for (int j = 0; j < 16; ++j) *(uint4*)(dt + (long long)j * c.ld_dst_t) = make_uint4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// ---- weights: fp32 master [rows][cols] -> e4m3 row-major copy + transposed copy, one 64 x 64 tile per workgroup, job table.
// Row segments: qkv rows / seg_rows -> q | k | v; W13 rows in [16 w1 | 16 w3] blocks.
__device__ __forceinline__ int f8w_seg(const F8WeightJob& j, int r) {
  if (j.layout == F8_LAYOUT_SEGS) { const int u = r / j.seg_rows; return u < j.seg_rep ? 0 : u - j.seg_rep + 1; }
  if (j.layout == F8_LAYOUT_SWIGLU) return (r >> 4) & 1;
  return 0;
}

__global__ __launch_bounds__(256) void f8_weight_amax_kernel(const F8WeightJob* __restrict__ jobs, const int* __restrict__ tile_job, const int* __restrict__ tile_first) {
  __shared__ float red[16];
  const F8WeightJob j = jobs[tile_job[blockIdx.x]];
  const int tl = blockIdx.x - tile_first[tile_job[blockIdx.x]];
  const int tcn = (j.cols + 63) >> 6;
  const int r0 = (tl / tcn) * 64, c0 = (tl % tcn) * 64;
  const int t = threadIdx.x, cc = c0 + (t & 15) * 4;
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + (t >> 4) + 16 * k;
    float m = 0.f;
    if (r < j.rows && cc < j.cols) {   // (cols % 4 == 0)
      const float4 a = *(const float4*)(j.src + (long long)r * j.ld + cc);
      m = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    const int sg = r < j.rows ? f8w_seg(j, r) : 0;
    mx[0] = sg == 0 ? fmaxf(mx[0], m) : mx[0];
    mx[1] = sg == 1 ? fmaxf(mx[1], m) : mx[1];
    mx[2] = sg == 2 ? fmaxf(mx[2], m) : mx[2];
  }
  const int nseg = f8_nseg(j.rows, j.layout, j.seg_rows, j.seg_rep);
  for (int sgi = 0; sgi < nseg; ++sgi) {
    const float v = block_max(mx[sgi], red);
    if (t == 0 && v > 0.f) atomicMax((int*)(j.amax + sgi), __float_as_int(v));   // (v >= 0: the bit patterns of non-negative floats order like ints)
  }
}

__global__ __launch_bounds__(256) void f8_weight_cast_kernel(const F8WeightJob* __restrict__ jobs, const int* __restrict__ tile_job, const int* __restrict__ tile_first) {
  __shared__ unsigned char tile[64][68];
  const F8WeightJob j = jobs[tile_job[blockIdx.x]];
  const int tl = blockIdx.x - tile_first[tile_job[blockIdx.x]];
  const int tcn = (j.cols + 63) >> 6;
  const int r0 = (tl / tcn) * 64, c0 = (tl % tcn) * 64;
  const int t = threadIdx.x, cl = (t & 15) * 4, cc = c0 + cl;
  const int nseg = f8_nseg(j.rows, j.layout, j.seg_rows, j.seg_rep);
  float sc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) sc[i] = i < nseg ? f8_scale_of(j.amax[i], F8_E4M3) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rl = (t >> 4) + 16 * k, r = r0 + rl;
    unsigned int q = 0u;
    if (r < j.rows && cc < j.cols) {
      const int sg = f8w_seg(j, r);
      const float s = sg == 0 ? sc[0] : (sg == 1 ? sc[1] : sc[2]);
      const float4 a = *(const float4*)(j.src + (long long)r * j.ld + cc);
      q = f8_pack4<F8_E4M3>(a.x * s, a.y * s, a.z * s, a.w * s);
      *(unsigned int*)(j.dst + (long long)r * j.cols + cc) = q;
    }
    *(unsigned int*)&tile[rl][cl] = q;
  }
  if (j.dst_t == nullptr) return;
  __syncthreads();
  // transposed copy: dst_t[c][tcol(r)]; SWIGLU rows de-interleave to [all w1 | all w3] (the K order of the de-interleaved gradient)
  const int c = t >> 2, rq = (t & 3) * 16;   // thread: column c0 + c, 16 consecutive rows
  if (c0 + c < j.cols) {
    unsigned int w[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      unsigned int x = 0u;
#pragma unroll
      for (int b = 0; b < 4; ++b) x |= (unsigned int)tile[rq + q4 * 4 + b][c] << (8 * b);
      w[q4] = x;
    }
    const int r = r0 + rq;   // rows r .. r+15: one 16-row block (a 16-block is all w1 or all w3)
    if (r < j.rows) {        // (rows % 16 == 0)
      const int tcol = j.layout == F8_LAYOUT_SWIGLU ? (((r >> 4) & 1) ? (j.rows >> 1) : 0) + (r >> 5) * 16 : r;
      *(uint4*)(j.dst_t + (long long)(c0 + c) * j.ld_t + tcol) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

}  // namespace

int launch_f8_amax(const F8Cast& c, hipStream_t s) {
  ARG_CHECK(c.cols % 16 == 0 && c.rows > 0, "fp8 amax: columns in groups of 16");
  ARG_CHECK(c.layout != F8_LAYOUT_SEGS || (c.seg_cols % 16 == 0 && c.cols % c.seg_cols == 0 && c.seg_rep >= 1 && f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep) >= 1 && f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep) <= 4), "fp8 amax: at most 4 column segments");
  const long long total = (long long)c.rows * (c.cols >> 4);
  const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
  if (c.src_f32) hipLaunchKernelGGL(f8_amax_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)c.src, c.ld_src, c.rows, c.cols, c.rows_dev, c.layout, c.seg_cols, c.seg_rep, c.amax);
  else hipLaunchKernelGGL(f8_amax_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)c.src, c.ld_src, c.rows, c.cols, c.rows_dev, c.layout, c.seg_cols, c.seg_rep, c.amax);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

__global__ __launch_bounds__(256) void round_bf16_accum_kernel(float4* __restrict__ stage, float4* __restrict__ dst, long long n4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = stage[i];
    float4 d = dst[i];
    d.x += (float)(bf16)v.x; d.y += (float)(bf16)v.y; d.z += (float)(bf16)v.z; d.w += (float)(bf16)v.w;
    dst[i] = d;
    stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
int launch_round_bf16_accum(float* stage, float* dst, long long n, hipStream_t s) {
  if (n <= 0) return RSYS_OK;
  ARG_CHECK(n % 4 == 0 && ((uintptr_t)stage | (uintptr_t)dst) % 16 == 0, "round_bf16_accum: n % 4 == 0 and 16-byte aligned buffers");
  const long long n4 = n / 4;
  hipLaunchKernelGGL(round_bf16_accum_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, (float4*)stage, (float4*)dst, n4);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

int launch_f8_cast(const F8Cast& c, hipStream_t s) {
  ARG_CHECK(c.cols % 16 == 0 && c.rows > 0 && c.ld_dst % 16 == 0, "fp8 cast: columns in groups of 16");
  ARG_CHECK(c.layout != F8_LAYOUT_SEGS || (c.seg_cols % 16 == 0 && c.cols % c.seg_cols == 0 && c.seg_rep >= 1 && f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep) >= 1 && f8_nseg(c.cols, c.layout, c.seg_cols, c.seg_rep) <= 4), "fp8 cast: at most 4 column segments");
  ARG_CHECK(c.layout != F8_LAYOUT_SWIGLU || c.cols % 32 == 0, "fp8 cast: [16 a | 16 b] column blocks");
  ARG_CHECK(c.desc_mode == 0 || (c.desc != nullptr && c.wamax != nullptr && c.n_w >= 1 && c.n_w <= 4), "fp8 cast: descale job");
  ARG_CHECK(c.desc_mode != 1 || (c.w_rep >= 1 && c.n_w - 1 + c.w_rep <= 16), "fp8 cast: at most 16 output-column units");
  const bool e5 = c.fmt == F8_E5M2;
  if (c.dst_t != nullptr) {
    ARG_CHECK(c.rows % 128 == 0 && c.cols % 64 == 0 && c.rows_dev == nullptr && c.ld_dst_t % 16 == 0 && c.ld_dst_t >= c.rows, "fp8 cast with a transposed copy: rows % 128, cols % 64");
    ARG_CHECK(c.desc_dw == nullptr || (c.xamax != nullptr && c.dw_units >= 1 && c.dw_units <= 16), "fp8 cast: weight-gradient descale job");
    const int wpb = std::min(4, std::max(1, sw().debug_f8_cast_waves));   // waves (tiles) per workgroup: A/B; 1..4 (__launch_bounds__(256))
    const int ntile_t = (c.rows / 128) * ((c.cols + 127) / 128);
    const int grid_t = (ntile_t + wpb - 1) / wpb;
    if (c.src_f32) { if (e5) hipLaunchKernelGGL((f8_cast_t_kernel<float, F8_E5M2>), dim3(grid_t), dim3(64 * wpb), 0, s, c); else hipLaunchKernelGGL((f8_cast_t_kernel<float, F8_E4M3>), dim3(grid_t), dim3(64 * wpb), 0, s, c); }
    else { if (e5) hipLaunchKernelGGL((f8_cast_t_kernel<bf16, F8_E5M2>), dim3(grid_t), dim3(64 * wpb), 0, s, c); else hipLaunchKernelGGL((f8_cast_t_kernel<bf16, F8_E4M3>), dim3(grid_t), dim3(64 * wpb), 0, s, c); }
    HIP_CHECK(hipGetLastError());
    return RSYS_OK;
  }
  const long long total = (long long)c.rows * (c.cols >> 4);
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  if (c.src_f32) { if (e5) hipLaunchKernelGGL((f8_cast_kernel<float, F8_E5M2>), dim3(grid), dim3(256), 0, s, c); else hipLaunchKernelGGL((f8_cast_kernel<float, F8_E4M3>), dim3(grid), dim3(256), 0, s, c); }
  else { if (e5) hipLaunchKernelGGL((f8_cast_kernel<bf16, F8_E5M2>), dim3(grid), dim3(256), 0, s, c); else hipLaunchKernelGGL((f8_cast_kernel<bf16, F8_E4M3>), dim3(grid), dim3(256), 0, s, c); }
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// jobs / tile_job / tile_first are device arrays built once (model_create); ntiles = total 64 x 64 tiles; amax slots zeroed by the caller
int launch_f8_weights(const F8WeightJob* jobs_dev, const int* tile_job_dev, const int* tile_first_dev, int ntiles, hipStream_t s) {
  if (ntiles <= 0) return RSYS_OK;
  hipLaunchKernelGGL(f8_weight_amax_kernel, dim3(ntiles), dim3(256), 0, s, jobs_dev, tile_job_dev, tile_first_dev);
  hipLaunchKernelGGL(f8_weight_cast_kernel, dim3(ntiles), dim3(256), 0, s, jobs_dev, tile_job_dev, tile_first_dev);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
