// Selected-token (compact) row sets for the top of the trunk.
//
// A training pass reads the trunk's output only at the positions the four heads select (transformer.model.py:501-513: at most
// mask_topk * rows positions per (medium, metric), the positive-weight ones of them carry loss), and everything downstream of
// the LAST layer's attention is token-local (output projection, residual, RMSNorm, SwiGLU, final RMSNorm, transformer.model.py:
// 297-309,335-343).  So the last layer's tail and its backward run on the compact set of selected tokens: forward rows that no
// head reads are never computed, backward rows whose gradient is identically zero are never multiplied.  This file holds the
// index kernels: the union of the heads' live positions as a sorted token list + its inverse map, and row gathers / scatters
// through them.  Compact buffers are valid for rows [0, n) and ZERO for rows [n, n rounded up to 256): the GEMMs that consume
// them stop at a device-side row / reduction limit rounded up to a tile (gemm.hpp m_dev / k_dev).
#include <stdlib.h>

#include "kernels.hpp"

namespace rsys {

namespace {

struct UnionLists { const int* idx[4]; const int* npos[4]; };

// One workgroup: bitmap of the selected tokens in LDS and a two-level scan over its words; out: the bitmap (bits_out[w]), the number
// of selected tokens before word w (pre_out[w]) and their total.  slot[tok] = rank among the selected (or -1) and sel[rank] = tok are
// written by the many workgroups of selected_first_kernel from these two arrays (one workgroup writing 64 K slots took 42 us).
// Token of position i of task ti: 2 i + (ti & 1) (even tokens: item -> watch heads, odd: action -> rating heads).
__global__ __launch_bounds__(1024) void token_union_kernel(UnionLists ul, int NT, unsigned int* __restrict__ bits_out, int* __restrict__ pre_out, int* nsel) {
  extern __shared__ unsigned int bits[];   // ceil(NT / 32) words
  __shared__ int wave_tot[16];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int nw = (NT + 31) >> 5;
  for (int i = t; i < nw; i += 1024) bits[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ul.idx[ti] == nullptr) continue;
    const int n = *ul.npos[ti];
    for (int r = t; r < n; r += 1024) {
      const int tok = 2 * ul.idx[ti][r] + (ti & 1);
      atomicOr(&bits[tok >> 5], 1u << (tok & 31));
    }
  }
  __syncthreads();
  const int per = (nw + 1023) / 1024;
  const int w0 = t * per, w1 = min(nw, w0 + per);
  int cnt = 0;
  for (int w = w0; w < w1; ++w) cnt += __popc(bits[w]);
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
  if (l == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, total = 0;
  for (int k = 0; k < 16; ++k) { int v = wave_tot[k]; if (k < wv) base += v; total += v; }
  int rank = base + inc - cnt;
  for (int w = w0; w < w1; ++w) {
    const unsigned int b = bits[w];
    bits_out[w] = b; pre_out[w] = rank;
    rank += __popc(b);
  }
  if (t == 0) *nsel = total;
}

// Token order of the last layer under the compact top: inside every batch row the selected tokens come first (in their original
// order), then the others.  Attention does not care about the order of the tokens it is given (positions enter through RoPE,
// applied per token from pos_p; the mask compares per-token keys), so the last layer runs on this order and its attention
// kernels only visit the leading query tiles (q_active).  One workgroup per batch row; T <= 2048 tokens.
//   perm[b T + p]  = original (global) token at permuted place p        uid_p / tm_p / pos_p: that token's ids and position
//   slot_p[b T + p] = compact row of that token or -1                    sel_p[r] = permuted place (global) of compact row r
__global__ __launch_bounds__(1024) void selected_first_kernel(const unsigned int* __restrict__ bits, const int* __restrict__ pre, int* __restrict__ slot,
                                                              int* __restrict__ sel, const int* __restrict__ uid, const int* __restrict__ tm,
                                                              const int* __restrict__ rope_pos, int T, int* __restrict__ perm, int* __restrict__ uid_p,
                                                              int* __restrict__ tm_p, int* __restrict__ pos_p, int* __restrict__ slot_p,
                                                              int* __restrict__ sel_p, int* __restrict__ q_active, int identity) {
  __shared__ int wave_tot[16];
  const int b = blockIdx.x, t = threadIdx.x, l = t & 63, wv = t >> 6;
  const long long base = (long long)b * T;
  const int per = (T + 1023) / 1024;               // 1 or 2 consecutive tokens per thread
  const int j0 = t * per, j1 = min(T, j0 + per);
  int sl[2] = {-1, -1}, cnt = 0;
  for (int j = j0; j < j1; ++j) {   // compact row of token base + j: selected tokens before it in its bitmap word + before that word
    const long long tok = base + j;
    const unsigned int b = bits[tok >> 5];
    const int bit = (int)(tok & 31);
    const int s = ((b >> bit) & 1u) ? pre[tok >> 5] + __popc(b & ((1u << bit) - 1u)) : -1;
    sl[j - j0] = s; cnt += s >= 0;
    slot[tok] = s;
    if (s >= 0) sel[s] = (int)tok;
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
  if (l == 63) wave_tot[wv] = inc;
  __syncthreads();
  int before = 0, n_sel = 0;
  for (int k = 0; k < 16; ++k) { int v = wave_tot[k]; if (k < wv) before += v; n_sel += v; }
  int rank = before + inc - cnt;                   // selected tokens of this row before j0
  for (int j = j0; j < j1; ++j) {
    const int s = sl[j - j0];
    const int p = identity ? j : (s >= 0 ? rank : n_sel + (j - rank));   // (identity: A/B measurement of what the order buys)
    perm[base + p] = (int)(base + j);
    uid_p[base + p] = uid[base + j]; tm_p[base + p] = tm[base + j];
    pos_p[base + p] = rope_pos != nullptr ? rope_pos[base + j] : j;
    slot_p[base + p] = s;
    if (s >= 0) { sel_p[s] = (int)(base + p); ++rank; }
  }
  if (t == 0) q_active[b] = identity ? (T + 63) >> 6 : (n_sel + 63) >> 6;
}

// rows [0, n) <- src rows sel[r]; rows [n, pad256(n)) <- 0.  One wave per row.
template <typename T>
__global__ void gather_rows_sel_kernel(const T* __restrict__ src, long long ld, const int* __restrict__ sel, const int* __restrict__ n_dev, int cap,
                                       T* dst, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  const int n = *n_dev, npad = min(cap, (n + 255) & ~255);
  if (row >= npad) return;
  constexpr int E = 16 / sizeof(T);
  uint4* d4 = (uint4*)(dst + (long long)row * D);
  if (row < n) {
    const uint4* s4 = (const uint4*)(src + (long long)sel[row] * ld);
    for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
  } else {
    for (int c = l; c < D / E; c += 64) d4[c] = make_uint4(0, 0, 0, 0);
  }
}

// dst rows sel[r] <- src rows r, r < n
template <typename T>
__global__ void scatter_rows_sel_kernel(const T* __restrict__ src, const int* __restrict__ sel, const int* __restrict__ n_dev, T* dst, long long ld, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= *n_dev) return;
  constexpr int E = 16 / sizeof(T);
  const uint4* s4 = (const uint4*)(src + (long long)row * D);
  uint4* d4 = (uint4*)(dst + (long long)sel[row] * ld);
  for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
}

// The compact top's dO in selected-first order: place p of batch row b (p < 64 q_active[b]: the query tiles the last layer's attention backward visits)
// <- compact row slot_p[b T + p], or zeros where that place holds no selected token (the ragged end of the last visited tile; every place when the
// order is the identity).  Places beyond the visited tiles are never read and keep what they held.  One wave per place.
template <typename T>
__global__ void scatter_rows_fill_kernel(const T* __restrict__ src, const int* __restrict__ slot_p, const int* __restrict__ q_active, int Tseq, long long places,
                                         T* dst, long long ld, int D) {
  const long long place = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int l = threadIdx.x & 63;
  if (place >= places) return;
  const int b = (int)(place / Tseq), pp = (int)(place - (long long)b * Tseq);
  if (pp >= 64 * q_active[b]) return;
  constexpr int E = 16 / sizeof(T);
  const int sl = slot_p[place];
  uint4* d4 = (uint4*)(dst + place * ld);
  if (sl >= 0) {
    const uint4* s4 = (const uint4*)(src + (long long)sl * D);
    for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
  } else {
    for (int c = l; c < D / E; c += 64) d4[c] = make_uint4(0, 0, 0, 0);
  }
}

// dst rows map[r] <- src rows r, r < n (host count)
template <typename T>
__global__ void scatter_rows_map_kernel(const T* __restrict__ src, const int* __restrict__ map, int n, T* dst, long long ld, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n) return;
  constexpr int E = 16 / sizeof(T);
  const uint4* s4 = (const uint4*)(src + (long long)row * D);
  uint4* d4 = (uint4*)(dst + (long long)map[row] * ld);
  for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
}

// heads: dst[r] = compact[slot[2 idx[r] + parity]] (zeros where that token is not in the set: zero-weight padding positions)
template <typename T>
__global__ void gather_rows_slot_kernel(const T* __restrict__ compact, const int* __restrict__ slot, const int* __restrict__ idx, int parity, T* dst,
                                        int n, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n) return;
  constexpr int E = 16 / sizeof(T);
  const int s = slot[2 * idx[row] + parity];
  uint4* d4 = (uint4*)(dst + (long long)row * D);
  if (s >= 0) {
    const uint4* s4 = (const uint4*)(compact + (long long)s * D);
    for (int c = l; c < D / E; c += 64) d4[c] = s4[c];
  } else {
    for (int c = l; c < D / E; c += 64) d4[c] = make_uint4(0, 0, 0, 0);
  }
}

// heads backward: compact[slot[2 idx[r] + parity]] += src[r] for the live rows r < *npos (one task at a time: no two rows of a
// task share a token)
__global__ void scatter_rows_add_slot_kernel(const float* __restrict__ src, const int* __restrict__ slot, const int* __restrict__ idx, int parity,
                                             const int* __restrict__ npos, float* compact, int n, int D) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  if (row >= n || row >= *npos) return;
  const int s = slot[2 * idx[row] + parity];
  if (s < 0) return;
  const float4* s4 = (const float4*)(src + (long long)row * D);
  float4* d4 = (float4*)(compact + (long long)s * D);
  for (int c = l; c < (D >> 2); c += 64) {
    float4 a = d4[c], b = s4[c];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[c] = a;
  }
}

}  // namespace

int launch_token_union(const int* const* idx, const int* const* npos, int ntask, int NT, unsigned int* bits, int* pre, int* nsel, hipStream_t s) {
  ARG_CHECK(ntask >= 1 && ntask <= 4 && NT >= 1 && NT <= (1 << 19), "token union: 1..4 tasks, at most 2^19 tokens (64 KB bitmap in LDS)");
  UnionLists ul{};
  for (int i = 0; i < ntask; ++i) { ul.idx[i] = idx[i]; ul.npos[i] = npos[i]; }
  hipLaunchKernelGGL(token_union_kernel, dim3(1), dim3(1024), (size_t)((NT + 31) / 32) * 4, s, ul, NT, bits, pre, nsel);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

int launch_selected_first(const unsigned int* bits, const int* pre, int* slot, int* sel, const int* uid, const int* tm, const int* rope_pos, int B, int T,
                          int* perm, int* uid_p, int* tm_p, int* pos_p, int* slot_p, int* sel_p, int* q_active, hipStream_t s) {
  ARG_CHECK(T >= 1 && T <= 2048, "selected-first order: at most 2048 tokens per row");
  const int identity = sw().top_order == 0 ? 1 : 0;   // RSYS_TOP_ORDER=0: keep the token order
  hipLaunchKernelGGL(selected_first_kernel, dim3(B), dim3(1024), 0, s, bits, pre, slot, sel, uid, tm, rope_pos, T, perm, uid_p, tm_p, pos_p, slot_p, sel_p, q_active, identity);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename T>
int launch_gather_rows_sel(const T* src, long long ld, const int* sel, const int* n_dev, int cap, T* dst, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0 && (ld * sizeof(T)) % 16 == 0, "gather_rows_sel: rows must be 16-byte multiples");
  hipLaunchKernelGGL((gather_rows_sel_kernel<T>), dim3((cap + 3) / 4), dim3(256), 0, s, src, ld, sel, n_dev, cap, dst, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_gather_rows_sel<bf16>(const bf16*, long long, const int*, const int*, int, bf16*, int, hipStream_t);
template int launch_gather_rows_sel<float>(const float*, long long, const int*, const int*, int, float*, int, hipStream_t);

template <typename T>
int launch_scatter_rows_sel(const T* src, const int* sel, const int* n_dev, int cap, T* dst, long long ld, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0 && (ld * sizeof(T)) % 16 == 0, "scatter_rows_sel: rows must be 16-byte multiples");
  hipLaunchKernelGGL((scatter_rows_sel_kernel<T>), dim3((cap + 3) / 4), dim3(256), 0, s, src, sel, n_dev, dst, ld, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template <typename T>
int launch_scatter_rows_fill(const T* src, const int* slot_p, const int* q_active, int B, int Tseq, T* dst, long long ld, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0 && (ld * sizeof(T)) % 16 == 0, "scatter_rows_fill: rows must be 16-byte multiples");
  const long long places = (long long)B * Tseq;
  hipLaunchKernelGGL((scatter_rows_fill_kernel<T>), dim3((unsigned)((places + 3) / 4)), dim3(256), 0, s, src, slot_p, q_active, Tseq, places, dst, ld, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_scatter_rows_fill<bf16>(const bf16*, const int*, const int*, int, int, bf16*, long long, int, hipStream_t);
template int launch_scatter_rows_fill<float>(const float*, const int*, const int*, int, int, float*, long long, int, hipStream_t);
template int launch_scatter_rows_sel<bf16>(const bf16*, const int*, const int*, int, bf16*, long long, int, hipStream_t);
template int launch_scatter_rows_sel<float>(const float*, const int*, const int*, int, float*, long long, int, hipStream_t);

template <typename T>
int launch_scatter_rows_map(const T* src, const int* map, int n, T* dst, long long ld, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0 && (ld * sizeof(T)) % 16 == 0, "scatter_rows_map: rows must be 16-byte multiples");
  hipLaunchKernelGGL((scatter_rows_map_kernel<T>), dim3((n + 3) / 4), dim3(256), 0, s, src, map, n, dst, ld, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_scatter_rows_map<bf16>(const bf16*, const int*, int, bf16*, long long, int, hipStream_t);
template int launch_scatter_rows_map<float>(const float*, const int*, int, float*, long long, int, hipStream_t);

template <typename T>
int launch_gather_rows_slot(const T* compact, const int* slot, const int* idx, int parity, T* dst, int n, int D, hipStream_t s) {
  ARG_CHECK((D * sizeof(T)) % 16 == 0, "gather_rows_slot: rows must be 16-byte multiples");
  hipLaunchKernelGGL((gather_rows_slot_kernel<T>), dim3((n + 3) / 4), dim3(256), 0, s, compact, slot, idx, parity, dst, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_gather_rows_slot<bf16>(const bf16*, const int*, const int*, int, bf16*, int, int, hipStream_t);
template int launch_gather_rows_slot<float>(const float*, const int*, const int*, int, float*, int, int, hipStream_t);

int launch_scatter_rows_add_slot(const float* src, const int* slot, const int* idx, int parity, const int* npos, float* compact, int n, int D,
                                 hipStream_t s) {
  hipLaunchKernelGGL(scatter_rows_add_slot_kernel, dim3((n + 3) / 4), dim3(256), 0, s, src, slot, idx, parity, npos, compact, n, D);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
