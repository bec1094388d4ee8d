// Deterministic embedding-gradient scatter (K16): dE[id'] += sum of the token gradients that gathered row id'.
//
// The reference's nn.Embedding backward (transformer.model.py:21, embedding_dense_backward) sorts the indices and
// reduces per segment.  Here:
//  * at batch upload the tokens are sorted once by (item id, token index) -- the inverted index "per table row, the
//    tokens that read it", in a fixed order.  64-bit composite keys are unique, so any sorting network gives the same
//    result; a bitonic network over at most 65 536 keys runs in LDS for all strides below 1024.
//  * per step, one wave per tile of SEG_TILE consecutive sorted positions adds its tokens' gradient rows in order and
//    writes each finished table row ONCE with a plain read-modify-write (one writer per row: no atomics, the sum is
//    bitwise reproducible).  A row whose tokens straddle tiles leaves per-tile partial sums in a slab; a second
//    launch adds the partials of each such row in a fixed tree (four contiguous quarters of the tile list, one wave
//    each, then the four sums in order).
//  * the mask row V is the one row whose member set changes every step (mask_tokens redirects ~mask_rate of the
//    tokens to it, transformer.model.py:437-462): its tokens are summed by position -- one partial per 256 tokens,
//    then the partials in the same tree.  Watch-masked tokens are skipped in their own item's segment.
// HBM-bound: algorithmic bytes = one gradient row read per token + one table row read and written per distinct id.
#include "kernels.hpp"

namespace rsys {

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int SEG_TILE = 8;      // sorted positions per wave
constexpr int MASK_CHUNK = 256;  // tokens per mask-row partial

// ------------------------------------------------------------------ token index (once per uploaded batch)
__global__ void tokidx_init_kernel(const int* __restrict__ matchedid, int N, int V, unsigned long long* keys, int Np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  if (i < N) {
    int id = matchedid[i];
    id = id == -1 ? V : id;
    keys[i] = ((unsigned long long)(unsigned int)id << 32) | (unsigned int)i;
  } else {
    keys[i] = ~0ull;   // padding sorts behind every token
  }
}

// every compare-exchange stage with stride j <= 512 of the levels k_first .. k_last (k doubling), on the 1024 keys of
// this block held in LDS; the direction of a pair comes from the GLOBAL index (bit k)
__global__ __launch_bounds__(512) void bitonic_local_kernel(unsigned long long* keys, int k_first, int k_last) {
  __shared__ unsigned long long sh[1024];
  const int t = threadIdx.x, base = blockIdx.x * 1024;
  sh[t] = keys[base + t]; sh[t + 512] = keys[base + t + 512];
  __syncthreads();
  for (int k = k_first; k <= k_last; k <<= 1) {
    for (int j = min(k >> 1, 512); j >= 1; j >>= 1) {
      const int i = 2 * j * (t / j) + (t % j);
      const unsigned long long a = sh[i], b = sh[i + j];
      const bool up = ((base + i) & k) == 0;
      if ((a > b) == up) { sh[i] = b; sh[i + j] = a; }
      __syncthreads();
    }
  }
  keys[base + t] = sh[t]; keys[base + t + 512] = sh[t + 512];
}

__global__ void bitonic_global_kernel(unsigned long long* keys, int k, int j, int Np) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (Np >> 1)) return;
  const int i = 2 * j * (t / j) + (t % j);
  const unsigned long long a = keys[i], b = keys[i + j];
  const bool up = (i & k) == 0;
  if ((a > b) == up) { keys[i] = b; keys[i + j] = a; }
}

__global__ void tokidx_extract_kernel(const unsigned long long* __restrict__ keys, int N, int* skey, int* sidx) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const unsigned long long k = keys[p];
  skey[p] = (int)(k >> 32);
  sidx[p] = (int)(k & 0xFFFFFFFFull);
}

int token_index_capacity(int N) {
  int np = 1024;
  while (np < N) np <<= 1;
  return np;
}

int launch_token_index_build(const int* matchedid, int N, int V, unsigned long long* keys, int* skey, int* sidx, hipStream_t s) {
  ARG_CHECK(N >= 1 && N <= (1 << 24), "token index: token count out of range");
  const int Np = token_index_capacity(N);
  hipLaunchKernelGGL(tokidx_init_kernel, dim3(div_up(Np, 256)), dim3(256), 0, s, matchedid, N, V, keys, Np);
  hipLaunchKernelGGL(bitonic_local_kernel, dim3(Np / 1024), dim3(512), 0, s, keys, 2, 1024);
  for (int k = 2048; k <= Np; k <<= 1) {
    for (int j = k >> 1; j >= 1024; j >>= 1)
      hipLaunchKernelGGL(bitonic_global_kernel, dim3(div_up(Np / 2, 256)), dim3(256), 0, s, keys, k, j, Np);
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(Np / 1024), dim3(512), 0, s, keys, k, k);
  }
  hipLaunchKernelGGL(tokidx_extract_kernel, dim3(div_up(N, 256)), dim3(256), 0, s, keys, N, skey, sidx);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// ------------------------------------------------------------------ per step: segmented sums
// A lane owns float4 groups c = l + 64 j (j < NJ) of a row; D <= 1024 NJ / 4... (D/4 float4 per row, D <= 256 NJ).
template <int NJ>
struct RowAcc {
  float4 v[NJ];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void add(const float4* r) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) { v[j].x += r[j].x; v[j].y += r[j].y; v[j].z += r[j].z; v[j].w += r[j].w; }
  }
};

template <int NJ>
__device__ __forceinline__ void load_row(const float* __restrict__ src, int D4, int l, float4* r) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = l + 64 * j;
    r[j] = c < D4 ? ((const float4*)src)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NJ>
__device__ __forceinline__ void store_row(float* dst, int D4, int l, const RowAcc<NJ>& a) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) { const int c = l + 64 * j; if (c < D4) ((float4*)dst)[c] = a.v[j]; }
}
template <int NJ>
__device__ __forceinline__ void add_to_row(float* dst, int D4, int l, const RowAcc<NJ>& a) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = l + 64 * j;
    if (c < D4) {
      float4 x = ((float4*)dst)[c];
      x.x += a.v[j].x; x.y += a.v[j].y; x.z += a.v[j].z; x.w += a.v[j].w;
      ((float4*)dst)[c] = x;
    }
  }
}

// waves [0, NT): sorted tiles; waves [NT, NT + NC): mask-row chunks by token position
template <int NJ>
__global__ __launch_bounds__(256) void seg_scatter_kernel(const float* __restrict__ gx0, long long ldx,
                                                          const int* __restrict__ m_id, const int* __restrict__ skey,
                                                          const int* __restrict__ sidx, int N, int V, int D, float* gE,
                                                          float* slab, float* mslab, int NT, int NC) {
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), l = threadIdx.x & 63;
  const int D4 = D >> 2;
  constexpr int G = NJ <= 2 ? 8 : (NJ <= 4 ? 4 : 2);   // rows in flight per wave (<= 16 float4 per lane)
  RowAcc<NJ> acc;
  if (wave < NT) {
    const int p0 = wave * SEG_TILE, p1 = min(N, p0 + SEG_TILE);
    int key[SEG_TILE], row[SEG_TILE];
#pragma unroll
    for (int k = 0; k < SEG_TILE; ++k) {
      const int p = p0 + k;
      key[k] = -2; row[k] = -1;
      if (p < p1) {
        key[k] = skey[p];
        const int i = sidx[p];
        row[k] = (key[k] != V && m_id[i] != -1) ? i : -1;   // watch-masked tokens belong to the mask row this step
      }
    }
    const int prev = p0 > 0 ? skey[p0 - 1] : -3;
    const int next = p1 < N ? skey[p1] : -4;
    int cur = key[0];
    bool first = true;
    acc.zero();
    auto flush = [&](bool open_tail) {
      if (cur == V) return;
      const bool open_head = first && cur == prev;
      if (open_head) store_row<NJ>(slab + (long long)(2 * wave) * D, D4, l, acc);
      else if (open_tail) store_row<NJ>(slab + (long long)(2 * wave + 1) * D, D4, l, acc);
      else add_to_row<NJ>(gE + (long long)cur * D, D4, l, acc);
    };
#pragma unroll
    for (int g0 = 0; g0 < SEG_TILE; g0 += G) {
      float4 r[G][NJ];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (row[g0 + g] >= 0) load_row<NJ>(gx0 + (long long)row[g0 + g] * ldx, D4, l, r[g]);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int k = g0 + g;
        if (key[k] == -2) continue;
        if (key[k] != cur) { flush(false); cur = key[k]; first = false; acc.zero(); }
        if (row[k] >= 0) acc.add(r[g]);
      }
    }
    flush(cur == next);
    return;
  }
  const int c = wave - NT;
  if (c >= NC) return;
  acc.zero();
  for (int t0 = c * MASK_CHUNK; t0 < min(N, (c + 1) * MASK_CHUNK); t0 += 64) {
    const int i = t0 + l;
    unsigned long long live = __ballot(i < N && m_id[i] == -1);
    while (live) {
      int idx[G];
      float4 r[G][NJ];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        idx[g] = -1;
        if (live) { const int b = __ffsll((long long)live) - 1; live &= live - 1; idx[g] = t0 + b; }
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (idx[g] >= 0) load_row<NJ>(gx0 + (long long)idx[g] * ldx, D4, l, r[g]);
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (idx[g] >= 0) acc.add(r[g]);
    }
  }
  store_row<NJ>(mslab + (long long)c * D, D4, l, acc);
}

// block t < NT: if a straddling row's tokens BEGIN in tile t, the block adds that row's per-tile partial sums -- its four
// waves a contiguous quarter of the list each, in tile order, then the four quarter sums in order -- and writes the row;
// the last block does the same with the mask-row partials into row V
template <int NJ>
__global__ __launch_bounds__(256) void seg_fixup_kernel(const int* __restrict__ skey, int N, int V, int D, float* gE,
                                                        const float* __restrict__ slab, const float* __restrict__ mslab,
                                                        int NT, int NC) {
  __shared__ float part[4][NJ * 256];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int D4 = D >> 2;
  constexpr int G = NJ <= 2 ? 8 : (NJ <= 4 ? 4 : 2);
  const int t = blockIdx.x;
  int L, dest;                       // partial rows to add, destination table row
  const float* first; const float* rest;   // partial 0; partial i >= 1 at rest + (i - 1) * rest_stride
  long long rest_stride;
  if (t == NT) {
    L = NC; dest = V; first = mslab; rest = mslab + D; rest_stride = D;
  } else {
    const int p0 = t * SEG_TILE, p1 = p0 + SEG_TILE;
    if (p1 >= N) return;
    const int last = skey[p1 - 1];
    if (last == V || skey[p1] != last) return;                          // no row leaves this tile open
    if (skey[p0] == last && p0 > 0 && skey[p0 - 1] == last) return;     // the row began in an earlier tile: not the owner
    int lo = p1, hi = N;                                                // first sorted position with a larger key
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (skey[mid] <= last) lo = mid + 1; else hi = mid; }
    const int t_end = (lo - 1) / SEG_TILE;                              // the row's last tile: its partial is a head partial
    L = 1 + (t_end - t); dest = last;
    first = slab + (long long)(2 * t + 1) * D; rest = slab + (long long)(2 * (t + 1)) * D; rest_stride = 2LL * D;
  }
  const int q = (L + 3) / 4, i0 = wv * q, i1 = min(L, i0 + q);
  RowAcc<NJ> acc;
  acc.zero();
  for (int ib = i0; ib < i1; ib += G) {
    float4 r[G][NJ];
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (ib + g < i1) load_row<NJ>(ib + g == 0 ? first : rest + (long long)(ib + g - 1) * rest_stride, D4, l, r[g]);
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (ib + g < i1) acc.add(r[g]);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) ((float4*)part[wv])[l + 64 * j] = acc.v[j];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      float4 r[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) r[j] = ((const float4*)part[w])[l + 64 * j];
      acc.add(r);
    }
    add_to_row<NJ>(gE + (long long)dest * D, D4, l, acc);
  }
}

size_t seg_scatter_slab_floats(int N, int D) {
  const long long NT = (N + SEG_TILE - 1) / SEG_TILE, NC = (N + MASK_CHUNK - 1) / MASK_CHUNK;
  return (size_t)((2 * NT + NC) * D);
}

int launch_embedding_scatter_segmented(const float* gx0, long long ldx, const int* m_matchedid, const int* skey, const int* sidx,
                                       int N, int V, int D, float* gE, float* slab, hipStream_t s) {
  ARG_CHECK(D % 4 == 0 && D <= 2048, "segmented scatter: embed_dim must be a multiple of 4 and <= 2048");
  const int NT = div_up(N, SEG_TILE), NC = div_up(N, MASK_CHUNK);
  float* mslab = slab + (size_t)2 * NT * D;
  const int nj = (D / 4 + 63) / 64;
  const dim3 gridA(div_up(NT + NC, 4)), gridB(NT + 1), blk(256);
#define LAUNCH_SEG(NJ)                                                                                                  \
  do {                                                                                                                  \
    hipLaunchKernelGGL((seg_scatter_kernel<NJ>), gridA, blk, 0, s, gx0, ldx, m_matchedid, skey, sidx, N, V, D, gE, slab, mslab, NT, NC); \
    hipLaunchKernelGGL((seg_fixup_kernel<NJ>), gridB, blk, 0, s, skey, N, V, D, gE, slab, mslab, NT, NC);                 \
  } while (0)
  if (nj <= 1) LAUNCH_SEG(1); else if (nj <= 2) LAUNCH_SEG(2); else if (nj <= 4) LAUNCH_SEG(4); else LAUNCH_SEG(8);
#undef LAUNCH_SEG
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
