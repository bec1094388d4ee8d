// Block-sparse bidirectional attention with the reference's two-predicate mask
// (transformer.model.py:479-487): allowed(q,kv) = userid[q]==userid[kv] AND
// (tmid[kv]==0 OR tmid[q]==tmid[kv]); GQA (q head h -> kv head h/(H/KV)); scores
// scaled by 1/sqrt(hd) (flex_attention default, model.py:278-285).
//
// gfx950 design.  64x64 (q x kv) tiles, 4 waves per workgroup, 16 q (or kv) per wave ON THE LANES:
// every first-stage product is computed transposed (S^T = K Q^T: rows = kv in the accumulator registers,
// column = q on the lane), so
//   * softmax statistics of a query are lane-local (16 values + two cross-group shuffles), and
//   * the probabilities never leave the registers: a 16x16 accumulator tile IS the B operand of the next
//     MFMA (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"); the
//     matching A operand (V^T, K^T, Q^T, dO^T fragments with the same k-slot order) is read from the
//     row-major LDS tile with ds_read_b64_tr_b16 -- no transposed copies in HBM, no P round trip through LDS.
// Tile pairs with no allowed element are skipped, pairs where every element is allowed skip the mask
// arithmetic (two bitmaps per row built once per step; the reference rebuilds a dense block mask every step,
// model.py:488-490).  K/V (resp. Q/dO) tiles are double-buffered in LDS, one barrier per tile.
// Backward is two kernels (dK/dV per kv tile, dQ per q tile): no float atomics, bitwise reproducible.
#include "kernels.hpp"

namespace rsys {

#define SENT_Q (-2147483647)
#define SENT_K (-2147483646)
// token keys (uid << 12 | tm) of tokens past the end of a row: never equal to each other or to a real key (uid < 2^19)
#define KEY_NO_Q ((int)0xFFFFE000u)
#define KEY_NO_K ((int)0xFFFFFFFFu)
constexpr float LOG2E = 1.4426950408889634f;
// v_exp_f32 directly (exp2f adds range scaling the softmax arguments never need: they are <= 0 or hugely negative)
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum over the four lanes that share a query (lane, lane ^ 16, lane ^ 32, lane ^ 48): two row swaps in the vector
// ALU (v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips through the LDS pipeline, which put
// ~5 index instructions and an LDS latency each into the soft-max's dependency chain.
__device__ __forceinline__ float quad_rows_max(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned int)a[0]), __builtin_bit_cast(float, (unsigned int)a[1]));
  const unsigned int w = __builtin_bit_cast(unsigned int, v);
  auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)b[0]), __builtin_bit_cast(float, (unsigned int)b[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned int)a[0]) + __builtin_bit_cast(float, (unsigned int)a[1]);
  const unsigned int w = __builtin_bit_cast(unsigned int, v);
  auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __builtin_bit_cast(float, (unsigned int)b[0]) + __builtin_bit_cast(float, (unsigned int)b[1]);
}

template <typename T> struct AMma;
template <> struct AMma<bf16> {
  static constexpr int KS = 32;   // contraction per MFMA
  // (register-staged kernels; every head size: rows of 96 / 96 / 160 / 288 bytes for head_dim 16 / 32 / 64 / 128 are all free of conflicts)
  // rows of 64 + 16 elements = 160 bytes: by the guide's lane groups the natural ds_read_b128 fragments and both ds_read_b64_tr_b16
  // reads of a transposed fragment are then free of bank conflicts (144-byte rows: 2x the cycles on both; 224-byte rows are free of
  // them too but cost a workgroup per CU).  Measured on one box: dK/dV 311 -> 302 us, forward and dQ unchanged -- the kernels wait on
  // dependencies, not on the LDS array (profiles/r4_ab_attn_lds_row_padding.log).
  static constexpr int PAD = 16;
  using Frag = bf16x8;
  static __device__ __forceinline__ Frag zero() { Frag f; for (int i = 0; i < 8; ++i) f[i] = (bf16)0.f; return f; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AMma<float> {
  static constexpr int KS = 4;
  static constexpr int PAD = 4;
  using Frag = float;
  static __device__ __forceinline__ Frag zero() { return 0.f; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

template <typename T, int HD> struct ACfg {
  static constexpr int KS = AMma<T>::KS;
  static constexpr int HDP = HD > KS ? HD : KS;      // contraction over d padded to one k-step
  static constexpr int LDD = HDP + AMma<T>::PAD;      // [64 tokens][HDP] tiles
  static constexpr int E = 16 / sizeof(T);
  static constexpr int NDS = HDP / KS;                // k-steps of a contraction over d
  static constexpr int TILE = 64 * LDD;               // elements per staged tile
};

// ---- fragments --------------------------------------------------------------------------------------------
// natural-order fragment of a row-major tile: row = row0 + (l&15), k = k0 + [8*(l>>4) .. +7] (bf16) / k0 + (l>>4) (f32)
template <typename T>
__device__ __forceinline__ typename AMma<T>::Frag frag_rows(const T* tile, int ld, int row0, int k0, int l) {
  if constexpr (is_bf16<T>::value) return *(const bf16x8*)(tile + (row0 + (l & 15)) * ld + k0 + 8 * (l >> 4));
  else return tile[(row0 + (l & 15)) * ld + k0 + (l >> 4)];
}
template <typename T>
__device__ __forceinline__ typename AMma<T>::Frag frag_global(const T* rowptr, int k0, int kmax, int l) {
  if constexpr (is_bf16<T>::value) {
    const int k = k0 + 8 * (l >> 4);
    if (k < kmax) return *(const bf16x8*)(rowptr + k);
    return AMma<bf16>::zero();
  } else {
    const int k = k0 + (l >> 4);
    return k < kmax ? rowptr[k] : 0.f;
  }
}
__device__ __forceinline__ float frag_dot(bf16x8 a, bf16x8 b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
  return s;
}
__device__ __forceinline__ float frag_dot(float a, float b) { return a * b; }
// bf16: transposed fragment of a row-major [token][d] tile for the contraction over 32 tokens tok0..tok0+31 in the
// ACCUMULATOR slot order (slot j<4 -> token tok0 + 4g + j, j>=4 -> tok0 + 16 + 4g + j-4); MFMA row = d0 + (l&15).
__device__ __forceinline__ bf16x8 frag_tr(const bf16* tile, int ld, int tok0, int d0, int l) {
  const int g = l >> 4, i = l & 15;
  const bf16* a = tile + (tok0 + 4 * g + (i >> 2)) * ld + d0 + 4 * (i & 3);
  bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)a);
  bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a + 16 * ld));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// The same fragments from an UNPADDED bf16 tile of 64 columns whose 16-byte chunks are XOR-swizzled (chunk c of row r in slot c ^ (r & 7)):
// free of bank conflicts for both kinds of read, and what one LDS-DMA instruction can fill (attn_*_dma_kernel below).
__device__ __forceinline__ bf16x8 frag_rows_sw(const bf16* tile, int row0, int k0, int l) {
  const int row = row0 + (l & 15), ch = (k0 >> 3) + (l >> 4);
  return *(const bf16x8*)(tile + row * 64 + ((ch ^ (row & 7)) << 3));
}
__device__ __forceinline__ bf16x8 frag_tr_sw(const bf16* tile, int tok0, int d0, int l) {
  const int g = l >> 4, i = l & 15;
  const int row = tok0 + 4 * g + (i >> 2), col = d0 + 4 * (i & 3);
  const bf16* a = tile + row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 4);
  bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)a);
  bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a + 16 * 64));   // (row + 16: the same slot permutation)
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <typename T, bool SW>
__device__ __forceinline__ typename AMma<T>::Frag frag_rows_x(const T* tile, int ld, int row0, int k0, int l) {
  if constexpr (SW) return frag_rows_sw(tile, row0, k0, l); else return frag_rows<T>(tile, ld, row0, k0, l);
}
template <bool SW>
__device__ __forceinline__ bf16x8 frag_tr_x(const bf16* tile, int ld, int tok0, int d0, int l) {
  if constexpr (SW) return frag_tr_sw(tile, tok0, d0, l); else return frag_tr(tile, ld, tok0, d0, l);
}
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  bf16x8 o;
  o[0] = (bf16)lo[0]; o[1] = (bf16)lo[1]; o[2] = (bf16)lo[2]; o[3] = (bf16)lo[3];
  o[4] = (bf16)hi[0]; o[5] = (bf16)hi[1]; o[6] = (bf16)hi[2]; o[7] = (bf16)hi[3];
  return o;
}

// acc[jd] (rows d, col = lane) += X^T[d][tok] * P[tok][lane] for the 64 tokens of a tile, where P[i] are the four
// 16-token accumulator tiles (rows = tokens) and X is the row-major LDS tile [token][d].
template <typename T, int HD>
__device__ __forceinline__ void acc_second_stage(f32x4 (&acc)[HD / 16], const f32x4 (&P)[4], const T* X, int l) {
  using C = ACfg<T, HD>;
  if constexpr (is_bf16<T>::value) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const bf16x8 pf = pack8(P[2 * t2], P[2 * t2 + 1]);
#pragma unroll
      for (int jd = 0; jd < HD / 16; ++jd) acc[jd] = AMma<bf16>::mma(frag_tr(X, C::LDD, 32 * t2, 16 * jd, l), pf, acc[jd]);
    }
  } else {
    const int g = l >> 4, fr = l & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* xr = X + (16 * i + 4 * g + r) * C::LDD + fr;
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd) acc[jd] = AMma<float>::mma(xr[16 * jd], P[i][r], acc[jd]);
      }
  }
}

// first stage: S[i] (rows = 16 tokens of tile i, col = lane) = X[tok][:] . f[:]  with f = register fragments of the lane's vector
template <typename T, int HD>
__device__ __forceinline__ void first_stage(f32x4 (&S)[4], const T* X, const typename AMma<T>::Frag (&f)[ACfg<T, HD>::NDS], int l) {
  using C = ACfg<T, HD>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    S[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) S[i] = AMma<T>::mma(frag_rows<T>(X, C::LDD, 16 * i, s * C::KS, l), f[s], S[i]);
  }
}

// The same two stages for R query heads that share one kv head (grouped-query attention) and the same 16 tokens per wave: the
// K (V^T, ...) fragment of the staged tile is read from LDS ONCE and feeds R MFMAs.
// (init != nullptr: the chains of block i start from init[i] instead of zero -- the mask as an additive 0 / -1e30, shared by the heads)
template <typename T, int HD, int R, bool SW = false>
__device__ __forceinline__ void first_stage_r(f32x4 (&S)[R][4], const T* X, const typename AMma<T>::Frag (&f)[R][ACfg<T, HD>::NDS], int l,
                                              const f32x4* init = nullptr) {
  using C = ACfg<T, HD>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < R; ++r) S[r][i] = init != nullptr ? init[i] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) {
      const typename AMma<T>::Frag x = frag_rows_x<T, SW>(X, C::LDD, 16 * i, s * C::KS, l);
#pragma unroll
      for (int r = 0; r < R; ++r) S[r][i] = AMma<T>::mma(x, f[r][s], S[r][i]);
    }
  }
}
template <typename T, int HD, int R, bool SW = false>
__device__ __forceinline__ void acc_second_stage_r(f32x4 (&acc)[R][HD / 16], const f32x4 (&P)[R][4], const T* X, int l) {
  using C = ACfg<T, HD>;
  if constexpr (is_bf16<T>::value) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      bf16x8 pf[R];
#pragma unroll
      for (int r = 0; r < R; ++r) pf[r] = pack8(P[r][2 * t2], P[r][2 * t2 + 1]);
#pragma unroll
      for (int jd = 0; jd < HD / 16; ++jd) {
        const bf16x8 x = frag_tr_x<SW>(X, C::LDD, 32 * t2, 16 * jd, l);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][jd] = AMma<bf16>::mma(x, pf[r], acc[r][jd]);
      }
    }
  } else {
    const int g = l >> 4, fr = l & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float* xr = X + (16 * i + 4 * g + rr) * C::LDD + fr;
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd) {
          const float x = xr[16 * jd];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][jd] = AMma<float>::mma(x, P[r][i][rr], acc[r][jd]);
        }
      }
  }
}

// One 16-token block (i) of the first stage / one 32-token half (t2 = blocks 2 t2, 2 t2 + 1) of the second stage: the backward
// kernels need no row maximum (they exponentiate against the stored log-sum-exp), so they run the two stages per half and keep
// only two score blocks per head alive instead of four.
// (S enters with its initial value: zero, or a row / lane constant that the chain then carries -- the dP chains start from -delta, so
// dS = P * (dP - delta) needs no subtraction)
template <typename T, int HD, int R, bool SW = false>
__device__ __forceinline__ void first_stage_block_r(f32x4 (&S)[R], const T* X, const typename AMma<T>::Frag (&f)[R][ACfg<T, HD>::NDS], int i, int l) {
  using C = ACfg<T, HD>;
#pragma unroll
  for (int s = 0; s < C::NDS; ++s) {
    const typename AMma<T>::Frag x = frag_rows_x<T, SW>(X, C::LDD, 16 * i, s * C::KS, l);
#pragma unroll
    for (int r = 0; r < R; ++r) S[r] = AMma<T>::mma(x, f[r][s], S[r]);
  }
}
template <typename T, int HD, int R, bool SW = false>
__device__ __forceinline__ void acc_second_stage_half_r(f32x4 (&acc)[R][HD / 16], const f32x4 (&P)[R][2], const T* X, int t2, int l) {
  using C = ACfg<T, HD>;
  if constexpr (is_bf16<T>::value) {
    bf16x8 pf[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pf[r] = pack8(P[r][0], P[r][1]);
#pragma unroll
    for (int jd = 0; jd < HD / 16; ++jd) {
      const bf16x8 x = frag_tr_x<SW>(X, C::LDD, 32 * t2, 16 * jd, l);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][jd] = AMma<bf16>::mma(x, pf[r], acc[r][jd]);
    }
  } else {
    const int g = l >> 4, fr = l & 15;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float* xr = X + (16 * (2 * t2 + ii) + 4 * g + rr) * C::LDD + fr;
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd) {
          const float x = xr[16 * jd];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][jd] = AMma<float>::mma(x, P[r][ii][rr], acc[r][jd]);
        }
      }
  }
}

// ---- staging ----------------------------------------------------------------------------------------------
template <typename T, int HD> struct TileRegs { uint4 v[(64 * HD * sizeof(T) / 16 + 255) / 256]; };

template <typename T, int HD>
__device__ __forceinline__ void tile_store(const TileRegs<T, HD>& r, T* dst, int t) {
  using C = ACfg<T, HD>;
  constexpr int CPR = HD / C::E, N = 64 * CPR;
#pragma unroll
  for (int k = 0; k < (N + 255) / 256; ++k) {
    const int c = t + 256 * k;
    if (c < N) *(uint4*)(dst + (c / CPR) * C::LDD + (c % CPR) * C::E) = r.v[k];
  }
}
// The same tile through a buffer descriptor: the per-thread byte offsets are loop invariant (TileOffs, computed once per kernel), the
// tile's origin is a scalar offset, and rows past the end of the sequence come back as zeros from the descriptor's bounds check --
// no per-load address arithmetic, predicate or zero fill in the staging loops, which are bound by instruction issue
// (profiles/r4_attn_dkv_item_loop_phases.log).
typedef __attribute__((ext_vector_type(4))) int at_i32x4;
extern "C" __device__ at_i32x4 rsys_at_buffer_load_b128(at_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
extern "C" __device__ int rsys_at_buffer_load_b32(at_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
typedef __attribute__((ext_vector_type(2))) int at_i32x2;
extern "C" __device__ at_i32x2 rsys_at_buffer_load_b64(at_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
// base .. base + bytes is what a load may touch (wave-uniform); bytes < 2^31
__device__ __forceinline__ at_i32x4 at_rsrc(const void* base, long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  at_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)((a >> 32) & 0xFFFFu));   // stride 0
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;                                                               // raw buffer, 32-bit data format
  return r;
}
template <typename T, int HD> struct TileOffs { int v[(64 * HD * sizeof(T) / 16 + 255) / 256]; };
template <typename T, int HD>
__device__ __forceinline__ void tile_offsets(TileOffs<T, HD>& o, long long ld, int t) {
  using C = ACfg<T, HD>;
  constexpr int CPR = HD / C::E, N = 64 * CPR;
#pragma unroll
  for (int k = 0; k < (N + 255) / 256; ++k) {
    const int c = t + 256 * k;
    o.v[k] = c < N ? (int)(((c / CPR) * ld + (c % CPR) * C::E) * (long long)sizeof(T)) : 0x7FFFFFF0;   // (no chunk: out of every range, reads zeros)
  }
}
template <typename T, int HD>
__device__ __forceinline__ void tile_load_buf(TileRegs<T, HD>& r, at_i32x4 rsrc, const TileOffs<T, HD>& o, int soffset) {
  using C = ACfg<T, HD>;
  constexpr int N = 64 * (HD / C::E);
#pragma unroll
  for (int k = 0; k < (N + 255) / 256; ++k) r.v[k] = __builtin_bit_cast(uint4, rsys_at_buffer_load_b128(rsrc, o.v[k], soffset, 0));
}
template <typename T, int HD>
__device__ __forceinline__ void zero_pad_cols(T* dst, int t) {
  using C = ACfg<T, HD>;
  if constexpr (C::HDP > HD) {
    for (int c = t; c < 64 * (C::HDP - HD); c += 256) dst[(c / (C::HDP - HD)) * C::LDD + HD + c % (C::HDP - HD)] = from_f32<T>(0.f);
  }
}
__device__ __forceinline__ int next_bit(unsigned int bits, int from) {  // lowest set bit index >= from, or 32
  const unsigned int m = from >= 32 ? 0u : (bits >> from) << from;
  return m ? __ffs(m) - 1 : 32;
}

// ------------------------------------------------------------------------ tile maps
// qmap / qmap_full [b][q tile]: bit j = kv tile j has some / only allowed pairs; kmap* is the transposed relation; the
// *16 maps say the same per group of 16 queries (keys).  One workgroup per (row, q tile); wave w owns the tile's 16
// queries w*16 .. w*16+15 as lane-uniform values (read back from LDS), the lanes sweep the row's keys 64 at a
// time: allowed(q, kv) <=> key[kv] == key[q] or key[kv] == key[q] without its tm bits (token_key below), so a (query,
// 64 keys) step is two compares, an OR and a ballot.  Exact: "some" bits may not miss a pair, "only" bits may not claim one.
__device__ __forceinline__ int token_key(int uid, int tm);
extern "C" __device__ int rsys_at_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");   // v_writelane_b32
__global__ __launch_bounds__(256) void attn_tilemap_kernel(AttnParams p) {
  __shared__ unsigned int any_bits, any16[4], kany[32];   // kany[j]: bit kg = keys 16 kg .. +15 of kv tile j meet a query of this tile
  __shared__ int cnt[32];
  __shared__ unsigned short kcols[32][64][4];   // [kv tile][key][query group of 16]: the key's bits against this q tile, written out as words
  __shared__ int keys[2048];                    // the row's token keys (one round of loads instead of a dependent load per kv tile)
  __shared__ int qkeys[64];                     // the tile's query keys: read back per query as a lane-uniform VECTOR register (as scalars,
                                                // 16 keys + their tm-less forms + the loop state spilled scalar registers through v_readlane)
  const int qt = blockIdx.x, b = blockIdx.y, t = threadIdx.x, l = t & 63, w = t >> 6;
  const int nt = (p.T + 63) / 64;
  const long long base = (long long)b * p.T;
  if (t == 0) any_bits = 0u;
  if (t < 4) any16[t] = 0u;
  if (t < 32) { cnt[t] = 0; kany[t] = 0u; }
  for (int i = t; i < nt * 64; i += 256) keys[i] = i < p.T ? token_key(p.uid[base + i], p.tm[base + i]) : KEY_NO_K;
  // lane i < 16 of wave w holds the key of query qt*64 + w*16 + i
  int myq = KEY_NO_Q;
  if (l < 16) { const int q = qt * 64 + w * 16 + l; if (q < p.T) myq = token_key(p.uid[base + q], p.tm[base + q]); }
  if (l < 16) qkeys[w * 16 + l] = myq;
  __syncthreads();
  int aq[16], aq0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { aq[i] = qkeys[w * 16 + i]; aq0[i] = aq[i] & ~4095; }
  unsigned int wany = 0u;
  for (int j = 0; j < nt; ++j) {
    const int ak = keys[j * 64 + l];
    bool mine = false;     // this key meets one of the wave's queries
    int n = 0;             // allowed pairs of this (q group, kv tile), wave-uniform
    unsigned long long qrow = 0ull;   // lane i < 16: the 64 keys of tile j that query w*16 + i may see
    unsigned int kcol = 0u;           // bit i: this lane's key is seen by query w*16 + i
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool hit = ak == aq[i] || ak == aq0[i];
      mine |= hit;
      n += __builtin_popcountll(__ballot(hit));
    }
    if (n) {   // (wave-uniform; a (query group, kv tile) pair without an allowed pair keeps zero bits and skips this second pass)
      int qlo = 0, qhi = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool hit = ak == aq[i] || ak == aq0[i];
        const unsigned long long hb = __ballot(hit);
        qlo = rsys_at_writelane((int)(unsigned int)hb, i, qlo);            // lane i <- query i's row of key bits
        qhi = rsys_at_writelane((int)(unsigned int)(hb >> 32), i, qhi);
        kcol |= hit ? (1u << i) : 0u;
      }
      qrow = ((unsigned long long)(unsigned int)qhi << 32) | (unsigned int)qlo;
    }
    // the pair bits the attention kernels mask with (AttnParams::qbits / kbits): every (q tile, kv tile), visited or not
    if (l < 16) p.qbits[(((long long)b * nt + qt) * nt + j) * 64 + w * 16 + l] = qrow;
    kcols[j][l][w] = (unsigned short)kcol;
    if (n) {
      wany |= 1u << j;
      const unsigned long long km = __ballot(mine);
      unsigned int kg = 0u;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) kg |= ((km >> (16 * g4)) & 0xFFFFull) ? (1u << g4) : 0u;
      if (l == 0) { atomicAdd(&cnt[j], n); atomicOr(&kany[j], kg); }
    }
  }
  if (l == 0) { any16[w] = wany; atomicOr(&any_bits, wany); }
  __syncthreads();
  for (int c = t; c < nt * 64; c += 256)   // kbits[b][kv tile c / 64][qt][key c % 64]: 512-byte runs
    p.kbits[(((long long)b * nt + (c >> 6)) * nt + qt) * 64 + (c & 63)] = *(const unsigned long long*)kcols[c >> 6][c & 63];
  if (t < nt) {
    const bool any = (any_bits >> t) & 1u, full = cnt[t] == 4096;
    if (any) atomicOr(&p.kmap[b * nt + t], 1u << qt);
    if (full) atomicOr(&p.kmap_full[b * nt + t], 1u << qt);
    if (full) atomicOr(&p.qmap_full[b * nt + qt], 1u << t);
    const unsigned int kg = kany[t];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      if ((kg >> g4) & 1u) atomicOr(&p.kmap16[(b * nt + t) * 4 + g4], 1u << qt);
  }
  if (t == 0) p.qmap[b * nt + qt] = any_bits;
  if (t < 4) p.qmap16[(b * nt + qt) * 4 + t] = any16[t];
}

// Heaviest-first launch orders (AttnParams::order_q / order_k).  blockIdx.y = 0: the q-side kernels (work of a slot = kv tiles its
// q tile visits; tiles beyond q_active: none), 1: the dK/dV kernel (q tiles its kv tile is visited by).  One workgroup per list of
// attn_work (blockIdx.x = XCD, or the single list of the fallback); rank = slots with more work + equal ones before it (stable, so
// the order is a function of the maps alone).
static int attn_heads_per_wg(const AttnParams& p);
static bool attn_kv_pairs(const AttnParams& p);   // the dK/dV launch of this shape is attn_bwd_kv32_kernel (order_k then lists kv tile PAIRS)
// kv_pairs: the dK/dV side's slots are PAIRS of kv tiles (attn_bwd_kv32_kernel: 128 keys per workgroup), work = q tiles either tile is visited by
__global__ __launch_bounds__(256) void attn_order_kernel(AttnParams p, int R, int identity, int kv_pairs) {
  extern __shared__ int ocnt[];   // [chunks of 64 slots + 1][34]: slots per (chunk, key), then their exclusive prefixes; last row: totals / bases
  const int side = blockIdx.y, nt = (p.T + 63) / 64, n_groups = p.B * p.KV;   // side 2: the forward kernel's (head, q tile PAIR) slots
  const int n_inner = side == 0 ? (p.H / p.KV / R) * nt : side == 2 ? (p.H / p.KV) * ((nt + 1) / 2) : (kv_pairs ? (nt + 1) / 2 : nt);
  int* out = side == 0 ? p.order_q : side == 2 ? p.order_q2 : p.order_k;
  if (out == nullptr) return;
  const bool lists = (n_groups & 7) == 0;
  const int ns = lists ? (n_groups >> 3) * n_inner : n_groups * n_inner, xcd = blockIdx.x;
  if (!lists && xcd > 0) return;
  out += lists ? xcd * ns : 0;
  if (identity) { for (int sl = threadIdx.x; sl < ns; sl += 256) out[sl] = sl; return; }
  const int nch = (ns + 63) >> 6, l = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto key_of = [&](int sl) -> int {
    if (sl >= ns) return 33;   // (padding of the last chunk: a key of its own)
    const int group = lists ? (sl / n_inner) * 8 + xcd : sl / n_inner, tile = (sl % n_inner) % nt, b = group / p.KV;
    const int qa = p.q_active != nullptr ? p.q_active[b] : 32;
    if (side == 0) return tile < qa ? __popc(p.qmap[b * nt + tile]) : 0;
    if (side == 2) {
      const int t0 = 2 * ((sl % n_inner) % ((nt + 1) / 2)), lim = min(nt, qa);
      return __popc((t0 < lim ? p.qmap[b * nt + t0] : 0u) | (t0 + 1 < lim ? p.qmap[b * nt + t0 + 1] : 0u));
    }
    if (kv_pairs) {
      const int t0 = 2 * (sl % n_inner);
      const unsigned int u = p.kmap[b * nt + t0] | (t0 + 1 < nt ? p.kmap[b * nt + t0 + 1] : 0u);
      return __popc(u & (qa >= 32 ? ~0u : ((1u << qa) - 1u)));
    }
    return __popc(p.kmap[b * nt + tile] & (qa >= 32 ? ~0u : ((1u << qa) - 1u)));
  };
  auto same_key = [&](int k) -> unsigned long long {   // lanes of this wave that hold the same key
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) { const unsigned long long bb = __ballot((k >> bit) & 1); m &= ((k >> bit) & 1) ? bb : ~bb; }
    return m;
  };
  for (int i = threadIdx.x; i < (nch + 1) * 34; i += 256) ocnt[i] = 0;
  __syncthreads();
  for (int c = w; c < nch; c += 4) {
    const int k = key_of(c * 64 + l);
    const unsigned long long m = same_key(k);
    if ((m & ((1ull << l) - 1ull)) == 0ull) ocnt[c * 34 + k] = __popcll(m);   // (the first lane of each key of the chunk)
  }
  __syncthreads();
  if (threadIdx.x < 34) {
    const int k = threadIdx.x;
    int run = 0;
    for (int c = 0; c < nch; ++c) { const int n = ocnt[c * 34 + k]; ocnt[c * 34 + k] = run; run += n; }
    ocnt[nch * 34 + k] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // heaviest first: the base of key k = slots with a larger key
    int acc = 0;
    for (int k = 32; k >= 0; --k) { const int n = ocnt[nch * 34 + k]; ocnt[nch * 34 + k] = acc; acc += n; }
  }
  __syncthreads();
  for (int c = w; c < nch; c += 4) {
    const int sl = c * 64 + l, k = key_of(sl);
    const unsigned long long m = same_key(k);
    if (sl < ns) out[ocnt[nch * 34 + k] + ocnt[c * 34 + k] + __popcll(m & ((1ull << l) - 1ull))] = sl;
  }
}

int launch_attn_tilemap(const AttnParams& p, hipStream_t s) {
  ARG_CHECK(p.T % 8 == 0 && (p.T + 63) / 64 <= 32, "attention: T must be a multiple of 8 and <= 2048");
  ARG_CHECK(p.qbits != nullptr && p.kbits != nullptr, "attention: the pair-bit buffers (AttnParams::qbits / kbits) are required");
  const size_t bytes = sizeof(unsigned int) * p.B * ((p.T + 63) / 64);
  if (p.maps_zero_base != nullptr) {
    HIP_CHECK(hipMemsetAsync(p.maps_zero_base, 0, p.maps_zero_bytes, s));
  } else {
    HIP_CHECK(hipMemsetAsync(p.kmap, 0, bytes, s));
    HIP_CHECK(hipMemsetAsync(p.kmap_full, 0, bytes, s));
    HIP_CHECK(hipMemsetAsync(p.qmap_full, 0, bytes, s));
    HIP_CHECK(hipMemsetAsync(p.kmap16, 0, bytes * 4, s));
  }
  hipLaunchKernelGGL(attn_tilemap_kernel, dim3((p.T + 63) / 64, p.B), dim3(256), 0, s, p);
  if (p.order_q != nullptr || p.order_k != nullptr || p.order_q2 != nullptr) {
    const int R = attn_heads_per_wg(p), nt = (p.T + 63) / 64, n_groups = p.B * p.KV;
    const int ns_max = ((n_groups & 7) == 0 ? (n_groups >> 3) : n_groups) * (p.H / p.KV) * nt;
    // (a list beyond one workgroup's LDS: keep the plain order -- the kernels read an identity permutation)
    hipLaunchKernelGGL(attn_order_kernel, dim3((n_groups & 7) == 0 ? 8 : 1, p.order_q2 != nullptr ? 3 : 2), dim3(256), (size_t)(std::min((ns_max + 63) / 64, 400) + 1) * 34 * 4, s, p, R, (ns_max + 63) / 64 > 400 ? 1 : 0,
                       attn_kv_pairs(p) ? 1 : 0);
  }
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// allowed(q, kv) = uid equal AND (tm[kv] == 0 OR tm equal).  With the token key a = uid << 12 | tm (host-checked:
// 0 <= uid < 2^19, 0 <= tm < 4096) this is  a[kv] == (a[q] & ~4095)  OR  a[kv] == a[q].  The compares run ONCE per step and map set, in
// attn_tilemap_kernel, which keeps their outcome as pair bits (AttnParams::qbits / kbits); the attention kernels mask from those.
__device__ __forceinline__ int token_key(int uid, int tm) { return (int)(((unsigned int)uid << 12) | (unsigned int)tm); }
// Masks from the precomputed pair bits: `word` = the lane's 64-bit row of the (q tile, kv tile) bit matrix, already shifted right by
// 4 g, so that the partner of the lane's score (block i, register rr) is bit 16 i + rr: a signed one-bit extract gives 0 / -1 and a
// bit-field insert picks the score or the fill -- two VALU instructions per score, the extract shared by the R heads, no key reads.
template <int R>
__device__ __forceinline__ void mask_bits_block(f32x4 (&S)[R], unsigned long long word, int i, float fill) {
  const int src = (int)(i < 2 ? (unsigned int)word : (unsigned int)(word >> 32));
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = __builtin_amdgcn_sbfe(src, 16 * (i & 1) + rr, 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float sv = S[r][rr];   // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
      S[r][rr] = __builtin_bit_cast(float, (__builtin_bit_cast(int, sv) & m) | (__builtin_bit_cast(int, fill) & ~m));
    }
  }
}
// The same mask as an ADDEND: 0 where the pair is allowed, -1e30 where not.  A score chain that starts from it ends at exactly the value
// mask_bits_block would have put there (-1e30 + q.k = -1e30 in fp32), costs no instruction per head, and the addend is the same for every
// head of the workgroup: extract + insert once per score position instead of extract + one insert per head.
__device__ __forceinline__ f32x4 mask_bias_block(unsigned long long word, int i) {
  const int src = (int)(i < 2 ? (unsigned int)word : (unsigned int)(word >> 32));
  f32x4 b;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = __builtin_amdgcn_sbfe(src, 16 * (i & 1) + rr, 1);
    b[rr] = __builtin_bit_cast(float, __builtin_bit_cast(int, -1e30f) & ~m);
  }
  return b;
}
// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each); all workgroups of one
// (row, kv head) group read the same K/V (forward, dQ) or Q/dO (dK/dV) tiles, so a group is kept on ONE XCD: its tiles
// are fetched from HBM once instead of once per XCD (rocprofv3 FETCH_SIZE: 2.9x the algorithmic bytes before).
// 1-D grid of n_groups * n_inner workgroups; falls back to the plain order when n_groups is not a multiple of 8.
// order (optional): per XCD list (or one list in the fallback) the slots sorted heaviest first (attn_order_kernel): workgroups are
// dispatched in blockIdx order, so the long ones start first and the short ones fill the tail of the launch.
__device__ __forceinline__ void attn_work(int n_groups, int n_inner, const int* __restrict__ order, int& group, int& inner) {
  const int bid = blockIdx.x;
  if ((n_groups & 7) == 0) {
    const int xcd = bid & 7;
    int slot = bid >> 3;
    if (order != nullptr) slot = order[xcd * ((n_groups >> 3) * n_inner) + slot];
    group = (slot / n_inner) * 8 + xcd;
    inner = slot % n_inner;
  } else {
    const int slot = order != nullptr ? order[bid] : bid;
    group = slot / n_inner;
    inner = slot % n_inner;
  }
}

// largest magnitude among the elements of one 16-byte chunk of a T tile (fp8 trunk: amax of a tensor while it is written)
template <typename T>
__device__ __forceinline__ float chunk_amax(const uint4& v) {
  if constexpr (is_bf16<T>::value) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, v);
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf((float)h[k]));
    return m;
  } else {
    const float4 f = __builtin_bit_cast(float4, v);
    return fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w)));
  }
}
template <typename T, int HD>
__device__ __forceinline__ void copy_out_tile(const T* Os, T* dst, long long ld, int tile_tok0, int T_len, int t, float* amax = nullptr) {
  using C = ACfg<T, HD>;
  constexpr int CPR = HD / C::E;
  float am = 0.f;
  for (int c = t; c < 64 * CPR; c += 256) {
    const int row = c / CPR, ch = c % CPR;
    if (tile_tok0 + row < T_len) {
      const uint4 v = *(const uint4*)(Os + row * C::LDD + ch * C::E);
      *(uint4*)(dst + (long long)row * ld + ch * C::E) = v;
      if (amax != nullptr) am = fmaxf(am, chunk_amax<T>(v));
    }
  }
  if (amax != nullptr) {
    am = wave_max(am);
    if ((t & 63) == 0) f8_amax_add(amax, am);
  }
}

extern "C" __device__ void rsys_at_buffer_load_lds(at_i32x4 rsrc, LDS_AS unsigned int* lds, int size, int voffset, int soffset, int offset,
                                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
// The same instruction behind an asm statement (ATTN_DMA_ASM): the compiler's wait-count pass orders every LDS read that it cannot tell
// apart from a pending LDS-DMA behind that DMA, i.e. it would make an item wait for the NEXT item's tiles as soon as it reads LDS.  Here
// it does not see the DMA; vector memory returns in issue order, so its own counted waits for later loads still cover what they must, and
// the wave's explicit s_waitcnt vmcnt(0) before the publishing barrier covers the DMA.  lds: wave-uniform LDS byte address.
// (M0 is written here and listed as clobbered, so that the compiler never takes an M0 value of its own for still valid behind the statement;
// clang's warning about a reserved register on the clobber list is silenced for this one statement: the clobber is the intent.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_asm(at_i32x4 rsrc, unsigned int lds, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
#pragma clang diagnostic pop
// (Measured per kernel on one box in round 4, intrinsic -> asm: dK/dV 193 -> 183 us, dQ 171 -> 177, forward 123 -> 127.  Round 5 found why the
// query-side kernels lost: their Q / dO fragment loads were still PENDING in the compiler's bookkeeping when the item loop began, so its
// wait-count pass kept counted vmcnt waits for them inside the loop -- counts that do not include the asm DMA, so each drained the
// next item's tiles; with those loads retired in front of the loop (an empty asm that names the registers) the asm form has no vector-memory
// wait between an item's first MFMA and its publishing wait, where the intrinsic form has a vmcnt(0) after the first third of the item.)
#ifndef ATTN_QSIDE_DMA_ASM
#define ATTN_QSIDE_DMA_ASM true
#endif
template <bool ASM>
__device__ __forceinline__ void dma16(at_i32x4 rsrc, unsigned char* lds, int voffset, int soffset) {
  if constexpr (ASM) dma16_asm(rsrc, (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)(LDS_AS unsigned char*)lds), voffset, soffset);
  else rsys_at_buffer_load_lds(rsrc, (LDS_AS unsigned int*)lds, 16, voffset, soffset, 0, 0);
}
// store_grad_tile / copy_out_tile on one such 8 KB tile (the epilogue stages through a tile buffer it no longer needs)
__device__ __forceinline__ void store_grad_tile_sw(f32x4 (&acc)[4], bool rotate, const float* rope_cos, const float* rope_sin, int pos, bf16* Os, int w, int l) {
  const int g = l >> 4, fr = l & 15, row = w * 16 + fr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float o[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
    if (rotate) {
      const int d2 = (16 * j + 4 * g) >> 1;
      const float2 cc = *(const float2*)(rope_cos + pos * 32 + d2);
      const float2 ss = *(const float2*)(rope_sin + pos * 32 + d2);
      const float a0 = o[0] * cc.x + o[1] * ss.x, a1 = -o[0] * ss.x + o[1] * cc.x;
      const float b0 = o[2] * cc.y + o[3] * ss.y, b1 = -o[2] * ss.y + o[3] * cc.y;
      o[0] = a0; o[1] = a1; o[2] = b0; o[3] = b1;
    }
    const int col = 16 * j + 4 * g;
    bf16* dst = Os + row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = (bf16)o[r];
  }
}
__device__ __forceinline__ void copy_out_tile_sw(const bf16* Os, bf16* dst, long long ld, int tile_tok0, int T_len, int t, float* amax = nullptr) {
  float am = 0.f;
#pragma unroll
  for (int c = t; c < 512; c += 256) {
    const int row = c >> 3, ch = c & 7;
    if (tile_tok0 + row < T_len) {
      const uint4 v = *(const uint4*)(Os + row * 64 + ((ch ^ (row & 7)) << 3));
      *(uint4*)(dst + (long long)row * ld + ch * 8) = v;
      if (amax != nullptr) am = fmaxf(am, chunk_amax<bf16>(v));
    }
  }
  if (amax != nullptr) {
    am = wave_max(am);
    if ((t & 63) == 0) f8_amax_add(amax, am);
  }
}
// 32 x 32 x 16 products (the 32-token-per-wave kernels: attn_fwd32_kernel, attn_bwd_kv32_kernel; LDS image and fragment maps: see the dK/dV kernel)
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ int sw32(int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 pack8f(const f32x16& v, int o) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)v[o + j];
  return r;
}
// ------------------------------------------------------------------------ forward
// R = query heads per workgroup: the R heads of one kv head's group at the SAME 64 tokens (R = 1: one head).  They share the
// staged K / V tile, every K and V^T fragment read, the tile maps and -- the mask depends on the tokens only -- the mask
// predicates; soft-max statistics and the output accumulators are per head.
// DMA (bf16, head_dim 64): the K / V tiles arrive by LDS-DMA in unpadded XOR-swizzled tiles (frag_rows_sw), as in attn_bwd_kv_dma_kernel.
template <typename T, int HD, int R, bool DMA = false>
__global__ __launch_bounds__(256, (is_bf16<T>::value && HD <= 64) ? (R == 1 ? 3 : (DMA ? 3 : 2)) : 1) void attn_fwd_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  static_assert(!DMA || (is_bf16<T>::value && HD == 64), "LDS-DMA staging: bf16, head_dim 64");
  constexpr int TILE = DMA ? 64 * 64 : C::TILE;   // elements of a staged tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = (T*)smem_raw;                       // [2][64][LDD]   (DMA: [2][64][64] swizzled)
  T* Vs = Ks + 2 * TILE;                      // [2][64][LDD]
  unsigned long long* ab = (unsigned long long*)(Vs + 2 * TILE);   // [2][64] pair bits of the 64 queries against the staged tile's keys
  const int nt = (p.T + 63) / 64;
  int grp, inner;
  attn_work(p.B * p.KV, (p.H / p.KV / R) * nt, p.order_q, grp, inner);
  const int b = grp / p.KV, kvh = grp % p.KV, h0 = kvh * (p.H / p.KV) + (inner / nt) * R, qt = inner % nt;
  if (p.q_active != nullptr && qt >= p.q_active[b]) return;   // nobody reads this query tile's output (uniform: whole workgroup)
  const int t = threadIdx.x, l = t & 63, w = t >> 6, g = l >> 4, fr = l & 15;
  const long long tok0 = (long long)b * p.T;
  const float c2 = rsqrtf((float)HD) * LOG2E;        // scores are handled in log2 units
  const int q = qt * 64 + w * 16 + fr;               // this lane's query
  const bool qv = q < p.T;
  typename M::Frag qf[R][C::NDS];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const T* qrow = (const T*)p.q + (tok0 + min(q, p.T - 1)) * p.ld + (h0 + r) * HD;
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) qf[r][s] = frag_global<T>(qrow, s * C::KS, HD, l);
  }
  float m_run[R], l_run[R];
  f32x4 oacc[R][HD / 16];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    m_run[r] = -1e30f; l_run[r] = 0.f;
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) oacc[r][j] = f32x4{0, 0, 0, 0};
  }
  if constexpr (!DMA) { zero_pad_cols<T, HD>(Ks, t); zero_pad_cols<T, HD>(Ks + C::TILE, t); }
  // (scalars through readfirstlane: loaded by vector memory -- the maps are not const -- and otherwise still pending when the item loop begins)
  const unsigned int bits = (unsigned int)__builtin_amdgcn_readfirstlane((int)p.qmap[b * nt + qt]), fullbits = (unsigned int)__builtin_amdgcn_readfirstlane((int)p.qmap_full[b * nt + qt]);
  const unsigned int wbits = __builtin_amdgcn_readfirstlane(p.qmap16[(b * nt + qt) * 4 + w]);   // kv tiles this wave's 16 queries take part in
  // K / V of this kv head, rows of this sequence ([T][HD] windows of the row-major qkv), and the rows' uid / tm (tile_load_buf)
  const at_i32x4 k_rs = at_rsrc((const T*)p.k + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  const at_i32x4 v_rs = at_rsrc((const T*)p.v + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  // the pair bits of this q tile against every kv tile: [nt][64 queries] words (AttnParams::qbits)
  const at_i32x4 qb_rs = at_rsrc(p.qbits + ((long long)b * nt + qt) * nt * 64, (long long)nt * 64 * 8);
  TileOffs<T, HD> kv_of;
  tile_offsets<T, HD>(kv_of, p.ld, t);
  const bool w0 = __builtin_amdgcn_readfirstlane(w) == 0;

  TileRegs<T, HD> rk, rv;
  unsigned long long rb = 0ull;
  int dv_[2] = {0, 0};   // DMA: wave w fills rows 16 w .. + 15 of a tile with two instructions (8 rows each); per-lane source offsets
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int row = 16 * w + 8 * k + (l >> 3); dv_[k] = (int)((row * p.ld + ((l & 7) ^ (row & 7)) * 8) * sizeof(T)); }
  }
  auto gload = [&](int kt, int buf) {   // (DMA: straight into buffer buf; else into registers, lstore(buf) follows after the arithmetic)
    const int so = (int)(kt * 64 * p.ld * sizeof(T));
    if constexpr (DMA) {
      unsigned char* kd = (unsigned char*)(Ks + buf * TILE) + (16 * w) * 128;
      unsigned char* vd = (unsigned char*)(Vs + buf * TILE) + (16 * w) * 128;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        dma16<ATTN_QSIDE_DMA_ASM>(k_rs, kd + k * 1024, dv_[k], so);
        dma16<ATTN_QSIDE_DMA_ASM>(v_rs, vd + k * 1024, dv_[k], so);
      }
    } else {
      tile_load_buf<T, HD>(rk, k_rs, kv_of, so);
      tile_load_buf<T, HD>(rv, v_rs, kv_of, so);
    }
    if (w0) rb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(qb_rs, 8 * l, kt * 512, 0));   // query l's keys of tile kt
  };
  auto lstore = [&](int buf) {
    if constexpr (!DMA) {
      tile_store<T, HD>(rk, Ks + buf * TILE, t);
      tile_store<T, HD>(rv, Vs + buf * TILE, t);
    }
    if (w0) ab[buf * 64 + l] = rb;
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA landed (the barrier follows)
  };

  int kt = next_bit(bits, 0);
  int cur = 0;
  if (kt < nt) { gload(kt, 0); lstore(0); }
  if constexpr (DMA && ATTN_QSIDE_DMA_ASM) {   // retire the Q fragments' loads in the compiler's bookkeeping before the loop (see dma16)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int s = 0; s < C::NDS; ++s) asm volatile("" : "+v"(qf[r][s]));
  }
  __syncthreads();
  while (kt < nt) {
    const int nxt = next_bit(bits, kt + 1);
    if (nxt < nt) gload(nxt, cur ^ 1);
    const T* Kc = Ks + cur * TILE;
    const T* Vc = Vs + cur * TILE;
    if ((wbits >> kt) & 1u) {   // (a wave whose 16 queries have no allowed key in this tile leaves its state untouched)
    f32x4 S[R][4];
    if (!((fullbits >> kt) & 1u)) {
      // the lane's query against the tile's 64 keys (pair bits): the mask enters the score chains as their starting value
      const unsigned long long wq = ab[cur * 64 + w * 16 + fr] >> (4 * g);
      f32x4 bias[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bias[i] = mask_bias_block(wq, i);
      first_stage_r<T, HD, R, DMA>(S, Kc, qf, l, bias);
    } else {
      first_stage_r<T, HD, R, DMA>(S, Kc, qf, l);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float tmax = -1e30f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) tmax = fmaxf(tmax, S[r][i][rr]);
      tmax = quad_rows_max(tmax);
      const float m_new = fmaxf(m_run[r], tmax * c2);
      const float mu_old = fmaxf(m_run[r], -1e20f), mu_new = fmaxf(m_new, -1e20f);
      const float alpha = fexp2(mu_old - mu_new);
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float pv = fexp2(fmaf(S[r][i][rr], c2, -mu_new));   // masked entries: exp2(-1.8e29) = 0
          S[r][i][rr] = pv;
          psum += pv;
        }
      psum = quad_rows_sum(psum);
      l_run[r] = l_run[r] * alpha + psum;
      m_run[r] = m_new;
#pragma unroll
      for (int j = 0; j < HD / 16; ++j) oacc[r][j] *= alpha;
    }
    acc_second_stage_r<T, HD, R, DMA>(oacc, S, Vc, l);
    }
    if (nxt < nt) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    kt = nxt;
  }
  // O^T (rows d, col q) -> row-major through LDS (head r in the r-th staged-tile slot), then 16-byte row stores
#pragma unroll
  for (int r = 0; r < R; ++r) {
    T* Os = Ks + r * C::TILE;   // [64][LDD]   (R <= 4: the four staged-tile slots)
    const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) {
      T* dst = Os + (w * 16 + fr) * C::LDD + 16 * j + 4 * g;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dst[rr] = from_f32<T>(oacc[r][j][rr] * inv);
    }
    if (g == 0 && qv) p.lse[((long long)b * p.H + h0 + r) * p.T + q] = (m_run[r] + log2f(l_run[r])) * (1.f / LOG2E);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r)
    copy_out_tile<T, HD>(Ks + r * C::TILE, (T*)p.o + (tok0 + qt * 64) * p.ldo + (h0 + r) * HD, p.ldo, qt * 64, p.T, t, p.f8_amax);
}


// ------------------------------------------------------------------------ forward on v_mfma_f32_32x32x16_bf16 (bf16, head_dim 64; round 6)
// The dK/dV recipe of round 5 (attn_bwd_kv32_kernel) for the forward pass: a wave owns 32 QUERIES (the lanes' l & 31), four waves = 128
// queries = two q tiles of ONE head per workgroup; every staged K / V tile feeds all of them -- the staging per FLOP of the two-head 64-query
// kernel above, half its MFMA instructions and fragment reads per FLOP, and one head's accumulators per wave (O 32 + S 32 + Q 16 registers):
// compiled for FOUR waves per SIMD.
//   S^T[kv][q] = sum_d K[kv][d] Q[q][d]:  A = K rows from LDS (ds_read_b128: row kv0 + (l & 31), d = 16 s + 8 (l >> 5) ..+7), B = Q from registers
//   O^T[d][q] += V^T[d][kv] P[kv][q]:     the S accumulator registers 8 s .. 8 s + 7 ARE the B operand of k-step s (keys 16 s + 8 (j >> 2) +
//   4 (l >> 5) + (j & 3)); A = V^T read transposed from the row-major tile (two ds_read_b64_tr_b16), exactly the dO^T operand of dV in the dK/dV kernel
// Soft-max: a query's 64 scores of a tile sit in the 32 registers of lanes l and l ^ 32: row maximum and sum are lane-local plus ONE
// v_permlane32_swap; the accumulators are rescaled only when some query's running maximum moved (wave-uniform test).
#ifndef ATTN_FWD32_WPS
#define ATTN_FWD32_WPS 4
#endif
__device__ __forceinline__ float pair_rows_max(float v) {   // over lanes l, l ^ 32
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)b[0]), __builtin_bit_cast(float, (unsigned int)b[1]));
}
__device__ __forceinline__ float pair_rows_sum(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned int)b[0]) + __builtin_bit_cast(float, (unsigned int)b[1]);
}
__global__ __launch_bounds__(256, ATTN_FWD32_WPS) void attn_fwd32_kernel(AttnParams p) {
  constexpr int HD = 64, TB = 64 * 64;   // elements of an unpadded tile
  using T = bf16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const Kb = smem_raw;                 // [2][64 kv][128 B] swizzled (sw32)
  unsigned char* const Vb = smem_raw + 2 * TB * 2;    // [2][64 kv][128 B]
  const int rep = p.H / p.KV, nt = (p.T + 63) / 64, npair = (nt + 1) / 2;
  int grp, inner;
  attn_work(p.B * p.KV, rep * npair, p.order_q2, grp, inner);
  const int b = grp / p.KV, kvh = grp % p.KV, hh = kvh * rep + inner / npair, pr = inner % npair;
  const int qa = p.q_active != nullptr ? min(p.q_active[b], nt) : nt;   // q tiles >= qa: nobody reads their output
  if (2 * pr >= qa) return;                                            // (uniform: whole workgroup)
  const int t = threadIdx.x, l = t & 63, r32 = l & 31, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qt = 2 * pr + (w >> 1);                   // this wave's q tile (may be inactive: the wave then only stages)
  const bool tile_ok = qt < qa;
  const long long tok0 = (long long)b * p.T;
  const float c2 = rsqrtf((float)HD) * LOG2E;        // scores are handled in log2 units
  const int q = qt * 64 + 32 * (w & 1) + r32;        // this lane's query
  const bool qv = tile_ok && q < p.T;
  bf16x8 qf[4];
  {
    const T* qrow = (const T*)p.q + (tok0 + min(q, p.T - 1)) * p.ld + hh * HD;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qrow + 16 * s + 8 * h);
  }
  f32x16 O[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { O[0][i] = 0.f; O[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const int t0i = b * nt + 2 * pr, t1i = b * nt + min(2 * pr + 1, nt - 1), qti = b * nt + min(qt, nt - 1);
  const unsigned int bits = (unsigned int)__builtin_amdgcn_readfirstlane((int)(p.qmap[t0i] | (2 * pr + 1 < qa ? p.qmap[t1i] : 0u)));   // kv tiles the workgroup stages
  const unsigned int fullbits = tile_ok ? (unsigned int)__builtin_amdgcn_readfirstlane((int)p.qmap_full[qti]) : 0u;
  const unsigned int wbits = tile_ok ? (unsigned int)__builtin_amdgcn_readfirstlane((int)(p.qmap16[qti * 4 + 2 * (w & 1)] | p.qmap16[qti * 4 + 2 * (w & 1) + 1])) : 0u;
  const at_i32x4 k_rs = at_rsrc((const T*)p.k + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  const at_i32x4 v_rs = at_rsrc((const T*)p.v + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  // the pair bits of this wave's q tile against every kv tile: [nt][64 queries] words (an inactive tile: an empty window, all zero)
  const at_i32x4 qb_rs = at_rsrc(p.qbits + (long long)qti * nt * 64, tile_ok ? (long long)nt * 64 * 8 : 0);
  // wave w fills rows 16 w .. 16 w + 15 of each tile with two DMA instructions (8 rows = 1 KB each): per-lane source offsets, fixed
  int dv_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = 16 * w + 8 * k + (l >> 3), ch = (l & 7) ^ sw32(row);
    dv_[k] = (int)((row * p.ld + ch * 8) * sizeof(T));
  }
  // fragment addresses inside a tile (bytes; lane constants), as in attn_bwd_kv32_kernel:
  //   row reads: row kv0 + r32, chunk 2 s + h -> slot (2 s) ^ G, G = h ^ sw32(r32)
  //   transposed reads: row kv0 + 16 s2 + 8 u + 4 h + (i >> 2), columns 32 db + 16 g16 + 4 (i & 3) -> slot L ^ (4 db) ^ u
  const int G = h ^ sw32(r32);
  const int i16 = l & 15, g16 = (l >> 4) & 1;
  const int Lc = (2 * g16 + ((i16 & 3) >> 1)) ^ ((((i16 >> 2) >> 1) & 1) << 2 | (h << 1));
  int a_row[4], a_tr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_row[s] = r32 * 128 + (((2 * s) ^ G) << 4);
#pragma unroll
  for (int v = 0; v < 4; ++v) a_tr[v] = (4 * h + (i16 >> 2)) * 128 + ((Lc ^ ((v >> 1) << 2) ^ (v & 1)) << 4) + ((i16 & 1) << 3);   // v = 2 db + u
  const unsigned int lds0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)(LDS_AS unsigned char*)smem_raw);
  unsigned long long sqb = 0ull;   // this lane's pair-bit word of the next tile on its way
  auto stage = [&](int kt, int buf) {
    const int so = (int)(kt * 64 * p.ld * sizeof(T));
    const unsigned int kd = lds0 + buf * (TB * 2) + (16 * w) * 128, vd = kd + 2 * TB * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      dma16_asm(k_rs, kd + k * 1024, dv_[k], so);
      dma16_asm(v_rs, vd + k * 1024, dv_[k], so);
    }
    sqb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(qb_rs, 8 * (32 * (w & 1) + r32), kt * 512, 0));   // this query's keys of tile kt
  };
  int kt = next_bit(bits, 0), cur = 0;
  unsigned long long wq = 0ull;
  if (kt < nt) { stage(kt, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wq = sqb; }
  // the Q fragments' loads are retired here in the compiler's bookkeeping (see dma16 / attn_bwd_kv32_kernel)
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
  asm volatile("" : "+v"(wq));
  __syncthreads();
  while (kt < nt) {
    const int nxt = next_bit(bits, kt + 1);
    if (nxt < nt) stage(nxt, cur ^ 1);   // (the other buffer: every wave left it before the barrier that ended the previous tile)
    if ((wbits >> kt) & 1u) {            // (a wave whose 32 queries have no allowed key in this tile leaves its state untouched)
      const unsigned char* Kc = Kb + cur * (TB * 2);
      const unsigned char* Vc = Vb + cur * (TB * 2);
      const bool fullt = (fullbits >> kt) & 1u;
      f32x16 S[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {   // 32 keys at a time: accumulator register r = key 32 kb + 8 (r >> 2) + 4 h + (r & 3)
        const unsigned int w32 = (unsigned int)(kb ? (wq >> 32) : wq);
        // the mask enters the score chain as its starting value (0 / -1e30); a 32 x 32 block of which every pair is allowed needs none
        if (!fullt && __builtin_amdgcn_ballot_w64(w32 != 0xFFFFFFFFu) != 0ull) {
          const int src = (int)(w32 >> (4 * h));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = __builtin_amdgcn_sbfe(src, 8 * (r >> 2) + (r & 3), 1);
            S[kb][r] = __builtin_bit_cast(float, __builtin_bit_cast(int, -1e30f) & ~m);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) S[kb] = mfma32(*(const bf16x8*)(Kc + kb * 4096 + a_row[s]), qf[s], S[kb]);   // S^T[kv][q]
      }
      float tmax = -1e30f;
#ifndef ATTN_FWD32_TIMING_NOMAX   // (timing-only builds, tools/: what the row maximum, the exponential and the row sum cost)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, S[kb][r]);
      tmax = pair_rows_max(tmax);
#else
      tmax = 0.25f;
#endif
      const float m_new = fmaxf(m_run, tmax * c2);
      const float mu_old = fmaxf(m_run, -1e20f), mu_new = fmaxf(m_new, -1e20f);
      const float alpha = fexp2(mu_old - mu_new);
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef ATTN_FWD32_TIMING_NOEXP
          const float pv = fmaf(S[kb][r], c2, -mu_new);
#else
          const float pv = fexp2(fmaf(S[kb][r], c2, -mu_new));   // masked entries: exp2(-1.8e29) = 0
#endif
          S[kb][r] = pv;
#ifndef ATTN_FWD32_TIMING_NOSUM
          psum += pv;
#endif
        }
      psum = pair_rows_sum(psum);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {   // (some query's maximum moved: after the first tiles of a user it rarely does)
#pragma unroll
        for (int i = 0; i < 16; ++i) { O[0][i] *= alpha; O[1][i] *= alpha; }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {   // k-step = 16 keys: accumulator registers 8 s2 .. 8 s2 + 7
          const bf16x8 pf = pack8f(S[kb], 8 * s2);
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int o = (32 * kb + 16 * s2) * 128;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(Vc + o + a_tr[2 * db]));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(Vc + o + 8 * 128 + a_tr[2 * db + 1]));
            O[db] = mfma32(__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), pf, O[db]);   // O^T[d][q] += V^T[d][kv] P[kv][q]
          }
        }
    }
    if (nxt < nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next tile landed (the barrier publishes it)
    __syncthreads();
    wq = sqb;
    cur ^= 1;
    kt = nxt;
  }
  // epilogue: O as bf16 rows [query][64] through the two K buffers (128 queries x 128 bytes), then row stores.
  // accumulator register r of block db: d = 32 db + 8 (r >> 2) + 4 h + (r & 3), query = r32 of this wave
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  if (h == 0 && qv) p.lse[((long long)b * p.H + hh) * p.T + q] = (m_run + log2f(l_run)) * (1.f / LOG2E);
  const int orow = 32 * w + r32;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 32 * db + 8 * g4 + 4 * h;
      bf16x4 oo; oo[0] = (bf16)(O[db][4 * g4] * inv); oo[1] = (bf16)(O[db][4 * g4 + 1] * inv); oo[2] = (bf16)(O[db][4 * g4 + 2] * inv); oo[3] = (bf16)(O[db][4 * g4 + 3] * inv);
      // (chunk slot permuted by the row so that the 8-byte stores of a half-wave spread over the banks; the copy below undoes it)
      *(bf16x4*)(Kb + orow * 128 + ((((d >> 3) ^ (orow & 7)) << 4) | ((d & 4) << 1))) = oo;
    }
  __syncthreads();
  {
    float am = 0.f;
#pragma unroll
    for (int c = t; c < 128 * 8; c += 256) {   // 16-byte chunks of the 128 rows
      const int row = c >> 3, ch = c & 7;
      const int tile = 2 * pr + (row >> 6), tok = tile * 64 + (row & 63);
      if (tile < qa && tok < p.T) {
        const uint4 v = *(const uint4*)(Kb + row * 128 + ((ch ^ (row & 7)) << 4));
        *(uint4*)((T*)p.o + (tok0 + tok) * p.ldo + hh * HD + ch * 8) = v;
        if (p.f8_amax != nullptr) am = fmaxf(am, chunk_amax<bf16>(v));
      }
    }
    if (p.f8_amax != nullptr) {
      am = wave_max(am);
      if (l == 0) f8_amax_add(p.f8_amax, am);
    }
  }
}

// query heads per workgroup of the forward / dQ kernels: the whole group of a kv head when that is 2 (every configuration of
// H / KV = 2), else 1.
static bool attn_dma_on() { return sw().attn_dma != 0; }
static int attn_heads_per_wg(const AttnParams& p) {
  return (p.H / p.KV == 2 && p.hd <= 64) ? 2 : 1;   // (head_dim 128: two heads' accumulators cost a wave per SIMD)
}

template <typename T, int HD>
static int attn_fwd_hd(const AttnParams& p, hipStream_t s) {
  using C = ACfg<T, HD>;
  const size_t sm = sizeof(T) * 4 * C::TILE + 384 * sizeof(int);
  static bool set = false;
  if (!set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, HD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, HD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    set = true;
  }
  if constexpr (is_bf16<T>::value && HD == 64) {
    if (attn_dma_on() && sw().attn_fwd32 != 0) {   // RSYS_ATTN_FWD32=1 (opt-in): 128 queries of one head per workgroup on 32 x 32 x 16 products
      static bool set32 = false;
      if (!set32) { HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 64 * 2)); set32 = true; }
      const int nt = (p.T + 63) / 64;
      hipLaunchKernelGGL(attn_fwd32_kernel, dim3(((nt + 1) / 2) * p.H * p.B), dim3(256), 4 * 64 * 64 * 2, s, p);
      HIP_CHECK(hipGetLastError());
      return RSYS_OK;
    }
    if (attn_dma_on()) {   // LDS-DMA staging (RSYS_ATTN_DMA=0: the register-staged kernels)
      const size_t sm_dma = 4 * 64 * 64 * 2 + 384 * 4;
      if (attn_heads_per_wg(p) == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, HD, 2, true>), dim3(((p.T + 63) / 64) * (p.H / 2) * p.B), dim3(256), sm_dma, s, p);
      else hipLaunchKernelGGL((attn_fwd_kernel<T, HD, 1, true>), dim3(((p.T + 63) / 64) * p.H * p.B), dim3(256), sm_dma, s, p);
      HIP_CHECK(hipGetLastError());
      return RSYS_OK;
    }
  }
  if (attn_heads_per_wg(p) == 2) hipLaunchKernelGGL((attn_fwd_kernel<T, HD, 2>), dim3(((p.T + 63) / 64) * (p.H / 2) * p.B), dim3(256), sm, s, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<T, HD, 1>), dim3(((p.T + 63) / 64) * p.H * p.B), dim3(256), sm, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

static int check_attn(const AttnParams& p, size_t esz) {
  ARG_CHECK(p.T % 8 == 0 && (p.T + 63) / 64 <= 32, "attention: T must be a multiple of 8 and <= 2048");
  ARG_CHECK(p.qbits != nullptr && p.kbits != nullptr, "attention: the pair-bit buffers (AttnParams::qbits / kbits) are required");
  ARG_CHECK(p.H % p.KV == 0, "attention: H % KV");
  ARG_CHECK((p.ld * esz) % 16 == 0 && (p.hd * esz) % 16 == 0, "attention: 16-byte row alignment");
  return RSYS_OK;
}

template <typename T>
int launch_attn_fwd(const AttnParams& p, hipStream_t s) {
  int rc = check_attn(p, sizeof(T));
  if (rc) return rc;
  switch (p.hd) {
    case 16: return attn_fwd_hd<T, 16>(p, s);
    case 32: return attn_fwd_hd<T, 32>(p, s);
    case 64: return attn_fwd_hd<T, 64>(p, s);
    case 128: return attn_fwd_hd<T, 128>(p, s);
  }
  set_error("attention: head_dim must be 16, 32, 64 or 128");
  return RSYS_ERR_ARG;
}
template int launch_attn_fwd<bf16>(const AttnParams&, hipStream_t);
template int launch_attn_fwd<float>(const AttnParams&, hipStream_t);


// gradient tile (rows d = 16j+4g+r, col = token on the lane) -> un-rotate (inverse of transformer.model.py:182-190; the
// pair (d, d+1) is two consecutive registers of the lane), transpose through LDS, store rows with 16-byte accesses
template <typename T, int HD>
__device__ __forceinline__ void store_grad_tile(f32x4 (&acc)[HD / 16], bool rotate, const float* rope_cos, const float* rope_sin,
                                                int pos, T* Os, int w, int l) {
  using C = ACfg<T, HD>;
  const int g = l >> 4, fr = l & 15;
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) {
    float o[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
    if (rotate) {
      const int d2 = (16 * j + 4 * g) >> 1;
      const float2 cc = *(const float2*)(rope_cos + pos * (HD / 2) + d2);
      const float2 ss = *(const float2*)(rope_sin + pos * (HD / 2) + d2);
      const float a0 = o[0] * cc.x + o[1] * ss.x, a1 = -o[0] * ss.x + o[1] * cc.x;
      const float b0 = o[2] * cc.y + o[3] * ss.y, b1 = -o[2] * ss.y + o[3] * cc.y;
      o[0] = a0; o[1] = a1; o[2] = b0; o[3] = b1;
    }
    T* dst = Os + (w * 16 + fr) * C::LDD + 16 * j + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = from_f32<T>(o[r]);
  }
}
// ------------------------------------------------------------------------ backward: dK, dV (one workgroup per kv tile and kv head)
template <typename T, int HD>
__global__ __launch_bounds__(256, (is_bf16<T>::value && HD <= 64) ? 3 : 1) void attn_bwd_kv_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Qs = (T*)smem_raw;                  // [2][64 q][LDD]
  T* dOs = Qs + 2 * C::TILE;             // [2][64 q][LDD]
  float* lse2 = (float*)(dOs + 2 * C::TILE);   // [2][64]
  float* dls = lse2 + 128;                      // [2][64]
  unsigned long long* kbs = (unsigned long long*)(dls + 128);   // [2][64] pair bits of the 64 keys against the staged item's queries
  const int rep = p.H / p.KV, nt = (p.T + 63) / 64;
  int grp, kvt;
  attn_work(p.B * p.KV, nt, p.order_k, grp, kvt);
  const int b = grp / p.KV, kvh = grp % p.KV;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, g = l >> 4, fr = l & 15;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;
  const int kv = kvt * 64 + w * 16 + fr;       // this lane's key/value token
  const bool kvv = kv < p.T;
  typename M::Frag kf[1][C::NDS], vf[1][C::NDS];
  {
    const T* krow = (const T*)p.k + (tok0 + min(kv, p.T - 1)) * p.ld + kvh * HD;
    const T* vrow = (const T*)p.v + (tok0 + min(kv, p.T - 1)) * p.ld + kvh * HD;
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) { kf[0][s] = frag_global<T>(krow, s * C::KS, HD, l); vf[0][s] = frag_global<T>(vrow, s * C::KS, HD, l); }
  }
  f32x4 dK[1][HD / 16], dV[1][HD / 16];
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) { dK[0][j] = f32x4{0, 0, 0, 0}; dV[0][j] = f32x4{0, 0, 0, 0}; }
  for (int i = 0; i < 2; ++i) { zero_pad_cols<T, HD>(Qs + i * C::TILE, t); zero_pad_cols<T, HD>(dOs + i * C::TILE, t); }
  const int qa = p.q_active != nullptr ? p.q_active[b] : 32;
  const unsigned int act = qa >= 32 ? ~0u : ((1u << qa) - 1u);   // query tiles whose dO can be non-zero
  const unsigned int bits = p.kmap[b * nt + kvt] & act, fullbits = p.kmap_full[b * nt + kvt];
  const unsigned int wbits = __builtin_amdgcn_readfirstlane(p.kmap16[(b * nt + kvt) * 4 + w]);   // q tiles that may see this wave's 16 keys

  // Staging runs TWO items ahead of the arithmetic: an item's global loads have a whole iteration (one item of another workgroup's
  // arithmetic would not cover their latency) before they are stored to LDS, and that store is in LDS one barrier before its use.
  // Two register sets, used alternately (the loop below is unrolled by two so that each stays in fixed registers).
  struct ItemRegs { TileRegs<T, HD> rq, rdo; int x, y; unsigned long long kb; };   // (wave 0) x, y: log-sum-exp and -delta of query l; kb: pair bits of key l
  ItemRegs R0, R1;
  R0.x = R1.x = R0.y = R1.y = 0; R0.kb = R1.kb = 0ull;
  // Q / dO of the kv head's query heads, rows of this sequence: [T][rep * HD] windows of the row-major tensors
  const at_i32x4 q_rs = at_rsrc((const T*)p.q + tok0 * p.ld + kvh * rep * HD, ((long long)(p.T - 1) * p.ld + rep * HD) * sizeof(T));
  const at_i32x4 do_rs = at_rsrc((const T*)p.dO + tok0 * p.ldo + kvh * rep * HD, ((long long)(p.T - 1) * p.ldo + rep * HD) * sizeof(T));
  TileOffs<T, HD> q_of, do_of;
  tile_offsets<T, HD>(q_of, p.ld, t);
  tile_offsets<T, HD>(do_of, p.ldo, t);
  // [rep][T] windows of lse / delta (a row past T of a head reads the next head's, finite and masked; of the last head: zero), [T] of uid / tm
  const at_i32x4 lse_rs = at_rsrc(p.lse + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  const at_i32x4 dl_rs = at_rsrc(p.delta + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  // the pair bits of this kv tile against every q tile: [nt][64 keys] words (AttnParams::kbits)
  const at_i32x4 kb_rs = at_rsrc(p.kbits + ((long long)b * nt + kvt) * nt * 64, (long long)nt * 64 * 8);
  // wave 0 (a wave-uniform branch) stages the 64 rows' scalars.  (One array per wave instead -- log-sum-exp, -delta, keys on waves 0, 1, 2 --
  // was measured 7 % SLOWER: three more uniform branches per item in every wave; profiles/r4_ab_attn_staging.log)
  const bool w0 = __builtin_amdgcn_readfirstlane(w) == 0;
  auto gload = [&](ItemRegs& r, int it) {   // it = head-in-group * 32 + q tile
    const int hh = it >> 5, qt = it & 31;
    tile_load_buf<T, HD>(r.rq, q_rs, q_of, (int)((qt * 64 * p.ld + hh * HD) * sizeof(T)));
    tile_load_buf<T, HD>(r.rdo, do_rs, do_of, (int)((qt * 64 * p.ldo + hh * HD) * sizeof(T)));
    if (w0) {   // (rows past the sequence: the next head's finite values or zero, all masked, and the key of no query)
      const int so = (hh * p.T + qt * 64) * 4;
      r.x = __builtin_bit_cast(int, __builtin_bit_cast(float, rsys_at_buffer_load_b32(lse_rs, 4 * l, so, 0)) * LOG2E);
      r.y = rsys_at_buffer_load_b32(dl_rs, 4 * l, so, 0) ^ 0x80000000;   // -delta: what the dP chains start from
      r.kb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(kb_rs, 8 * l, qt * 512, 0));   // key l against the 64 queries
    }
  };
  auto lstore = [&](const ItemRegs& r, int buf) {
    tile_store<T, HD>(r.rq, Qs + buf * C::TILE, t);
    tile_store<T, HD>(r.rdo, dOs + buf * C::TILE, t);
    if (w0) { ((int*)lse2)[buf * 64 + l] = r.x; ((int*)dls)[buf * 64 + l] = r.y; kbs[buf * 64 + l] = r.kb; }
  };
  auto next_item = [&](int from) {   // items are (head, q tile) pairs in order; returns rep*32 when exhausted
    int hh = from >> 5, qt = from & 31;
    while (hh < rep) {
      const int n = next_bit(bits, qt);
      if (n < nt) return hh * 32 + n;
      ++hh; qt = 0;
    }
    return rep * 32;
  };
  const int end = rep * 32;
  int it = next_item(0), cur = 0;
  int nxt = it < end ? next_item(it + 1) : end, nxt2 = nxt < end ? next_item(nxt + 1) : end;   // the items staged one and two ahead
  if (it < end) { gload(R0, it); lstore(R0, 0); }
  if (nxt < end) gload(R1, nxt);
  __syncthreads();
  auto step = [&](ItemRegs& rload, const ItemRegs& rstore) {   // rstore holds item nxt; item nxt2 is loaded into rload
    if (nxt2 < end) gload(rload, nxt2);
    const T* Qc = Qs + cur * C::TILE;
    const T* dOc = dOs + cur * C::TILE;
    if ((wbits >> (it & 31)) & 1u) {   // (nothing to add for a wave whose 16 keys no query of this tile may see)
    const bool fullt = (fullbits >> (it & 31)) & 1u;
    const unsigned long long wk = kbs[cur * 64 + w * 16 + fr] >> (4 * g);   // the lane's key against the item's 64 queries (mask_bits_block)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {   // 32 queries at a time: both stages, two score blocks alive (no row maximum is needed here)
      f32x4 P2[1][2], dS2[1][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * t2 + ii;
        const float4 l4 = *(const float4*)(lse2 + cur * 64 + 16 * i + 4 * g);
        const float4 d4 = *(const float4*)(dls + cur * 64 + 16 * i + 4 * g);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
        f32x4 S[1] = {f32x4{0, 0, 0, 0}}, dP[1] = {f32x4{d4.x, d4.y, d4.z, d4.w}};   // (rows = queries: the chain starts from -delta[q])
        first_stage_block_r<T, HD, 1>(S, Qc, kf, i, l);      // S[q][kv]: rows q (registers), col kv (lane)
        first_stage_block_r<T, HD, 1>(dP, dOc, vf, i, l);    // dP[q][kv] - delta[q]
        if (!fullt) mask_bits_block<1>(S, wk, i, -1e30f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = fexp2(fmaf(S[0][r], c2, -ll[r]));   // masked: exp2(-1.8e29 - lse) = 0
          P2[0][ii][r] = pv;
          dS2[0][ii][r] = pv * dP[0][r];   // (the 1/sqrt(hd) factor of dS is applied once to dK at the end)
        }
      }
      acc_second_stage_half_r<T, HD, 1>(dV, P2, dOc, t2, l);    // dV^T[d][kv] += dO^T[d][q] P[q][kv]
      acc_second_stage_half_r<T, HD, 1>(dK, dS2, Qc, t2, l);    // dK^T[d][kv] += Q^T[d][q] dS[q][kv]
    }
    }
    if (nxt < end) lstore(rstore, cur ^ 1);
    __syncthreads();
    cur ^= 1;
    it = nxt; nxt = nxt2; nxt2 = nxt2 < end ? next_item(nxt2 + 1) : end;
  };
  while (it < end) {
    step(R0, R1);
    if (it >= end) break;
    step(R1, R0);
  }
  const int pos = p.rope_pos ? p.rope_pos[tok0 + min(kv, p.T - 1)] : min(kv, p.T - 1);
  T* Os = Qs;
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) dK[0][j] *= scale;
  store_grad_tile<T, HD>(dK[0], true, p.rope_cos, p.rope_sin, pos, Os, w, l);
  __syncthreads();
  copy_out_tile<T, HD>(Os, (T*)p.dk + (tok0 + kvt * 64) * p.ldg + kvh * HD, p.ldg, kvt * 64, p.T, t, p.f8_amax ? p.f8_amax + 1 : nullptr);
  __syncthreads();
  store_grad_tile<T, HD>(dV[0], false, p.rope_cos, p.rope_sin, pos, Os, w, l);
  __syncthreads();
  copy_out_tile<T, HD>(Os, (T*)p.dv + (tok0 + kvt * 64) * p.ldg + kvh * HD, p.ldg, kvt * 64, p.T, t, p.f8_amax ? p.f8_amax + 2 : nullptr);
}


// ------------------------------------------------------------------------ dK, dV with LDS-DMA staging (bf16, head_dim 64)
// The same kernel with the Q / dO tiles brought in by `buffer_load ... lds` (no staging registers, no LDS stores) into UNPADDED tiles:
// a 64-column bf16 row is 128 bytes = 8 chunks of 16, chunk c of row r sits in slot c ^ (r & 7).  By the guide's lane groups that
// layout is free of bank conflicts for the natural ds_read_b128 fragments and for both ds_read_b64_tr_b16 reads (as the 160-byte rows
// are), it is what a DMA instruction can fill (1 KB = 8 consecutive rows, the swizzle applied on the SOURCE side: the lane that fills
// slot s of row r loads chunk s ^ (r & 7)), and it makes a tile 8 KB: 34 KB per workgroup and ~110 registers -- four workgroups per CU.
// -DATTN_KV_TRACE (tools/trace_attn_kv.sh; never in the product build): wave 0 of every workgroup stamps the wall clock (10 ns) at entry,
// after the first item is staged, after the item loop and at exit, and sums the item loop's phases -- issue of the next item's DMA and
// scalar loads, arithmetic, publishing (waits for the DMA), barrier.  rsys_attn_trace_read copies the table out.
#ifdef ATTN_KV_TRACE
__device__ unsigned long long rsys_attn_trace[16 * 8192];
#define KVT_NOW() ((unsigned long long)wall_clock64())
#else
#define KVT_NOW() 0ull
#endif
__global__ __launch_bounds__(256, 4) void attn_bwd_kv_dma_kernel(AttnParams p) {
  [[maybe_unused]] unsigned long long tr_[12] = {KVT_NOW(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int HD = 64, TB = 64 * 64;   // elements of an unpadded tile
  using T = bf16;
  using C = ACfg<T, HD>;
  // (The compiler orders an LDS read behind a pending LDS-DMA unless it can tell the two apart, so each item starts with a wait for the
  // next item's DMA it has just issued; the other three workgroups of the CU run meanwhile.  Telling them apart was tried -- one LDS
  // object per buffer and the item loop unrolled by two: no wait, but 158 registers = three workgroups per CU and 209 against 198 us, or
  // 30 spilled registers at four; profiles/r4_ab_attn_kv_dma.log.)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Qs = (T*)smem_raw;                  // [2][64 q][64] swizzled
  T* dOs = Qs + 2 * TB;                  // [2][64 q][64] swizzled
  float* lse2 = (float*)(dOs + 2 * TB);  // [2][64]
  float* dls = lse2 + 128;               // [2][64]
  unsigned long long* kbs = (unsigned long long*)(dls + 128);   // [2][64]
  const int rep = p.H / p.KV, nt = (p.T + 63) / 64;
  int grp, kvt;
  attn_work(p.B * p.KV, nt, p.order_k, grp, kvt);
  const int b = grp / p.KV, kvh = grp % p.KV;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, g = l >> 4, fr = l & 15;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;
  const int kv = kvt * 64 + w * 16 + fr;       // this lane's key/value token
  bf16x8 kf[2], vf[2];
  {
    const T* krow = (const T*)p.k + (tok0 + min(kv, p.T - 1)) * p.ld + kvh * HD;
    const T* vrow = (const T*)p.v + (tok0 + min(kv, p.T - 1)) * p.ld + kvh * HD;
#pragma unroll
    for (int s = 0; s < 2; ++s) { kf[s] = frag_global<T>(krow, s * 32, HD, l); vf[s] = frag_global<T>(vrow, s * 32, HD, l); }
  }
  f32x4 dK[1][4], dV[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { dK[0][j] = f32x4{0, 0, 0, 0}; dV[0][j] = f32x4{0, 0, 0, 0}; }
  const int qa = p.q_active != nullptr ? p.q_active[b] : 32;
  const unsigned int act = qa >= 32 ? ~0u : ((1u << qa) - 1u);
  const unsigned int bits = p.kmap[b * nt + kvt] & act, fullbits = p.kmap_full[b * nt + kvt];
  const unsigned int wbits = __builtin_amdgcn_readfirstlane(p.kmap16[(b * nt + kvt) * 4 + w]);
  const at_i32x4 q_rs = at_rsrc((const T*)p.q + tok0 * p.ld + kvh * rep * HD, ((long long)(p.T - 1) * p.ld + rep * HD) * sizeof(T));
  const at_i32x4 do_rs = at_rsrc((const T*)p.dO + tok0 * p.ldo + kvh * rep * HD, ((long long)(p.T - 1) * p.ldo + rep * HD) * sizeof(T));
  const at_i32x4 lse_rs = at_rsrc(p.lse + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  const at_i32x4 dl_rs = at_rsrc(p.delta + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  const at_i32x4 kb_rs = at_rsrc(p.kbits + ((long long)b * nt + kvt) * nt * 64, (long long)nt * 64 * 8);
  const bool w0 = __builtin_amdgcn_readfirstlane(w) == 0;
  // wave w fills rows 16 w .. 16 w + 15 of each tile with two DMA instructions (8 rows = 1 KB each): per-lane source offsets, fixed
  int qv_[2], dv_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = 16 * w + 8 * k + (l >> 3), ch = (l & 7) ^ (row & 7);
    qv_[k] = (int)((row * p.ld + ch * 8) * sizeof(T));
    dv_[k] = (int)((row * p.ldo + ch * 8) * sizeof(T));
  }
  int sx = 0, sy = 0; unsigned long long skb = 0ull;   // (wave 0) the next item's row scalars on their way to LDS
  auto stage = [&](int it, int buf) {   // it = head-in-group * 32 + q tile
    const int hh = it >> 5, qt = it & 31;
    const int qso = (int)((qt * 64 * p.ld + hh * HD) * sizeof(T)), dso = (int)((qt * 64 * p.ldo + hh * HD) * sizeof(T));
    unsigned char* qd = (unsigned char*)(Qs + buf * TB) + (16 * w) * 128;
    unsigned char* dd = (unsigned char*)(dOs + buf * TB) + (16 * w) * 128;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      dma16<true>(q_rs, qd + k * 1024, qv_[k], qso);
      dma16<true>(do_rs, dd + k * 1024, dv_[k], dso);
    }
    if (w0) {
      const int so = (hh * p.T + qt * 64) * 4;
      sx = rsys_at_buffer_load_b32(lse_rs, 4 * l, so, 0);   // (raw: used, and therefore waited for, only when published)
      sy = rsys_at_buffer_load_b32(dl_rs, 4 * l, so, 0);
      skb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(kb_rs, 8 * l, qt * 512, 0));
    }
  };
  auto publish = [&](int buf) {   // the scalars into LDS; every DMA of this wave landed
    if (w0) { lse2[buf * 64 + l] = __builtin_bit_cast(float, sx) * LOG2E; ((int*)dls)[buf * 64 + l] = sy ^ 0x80000000; kbs[buf * 64 + l] = skb; }   // log2 units; -delta
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto next_item = [&](int from) {
    int hh = from >> 5, qt = from & 31;
    while (hh < rep) {
      const int n = next_bit(bits, qt);
      if (n < nt) return hh * 32 + n;
      ++hh; qt = 0;
    }
    return rep * 32;
  };
  const int end = rep * 32;
  int it = next_item(0), cur = 0;
  if (it < end) { stage(it, 0); publish(0); }
  __syncthreads();
  tr_[1] = KVT_NOW();
  while (it < end) {
    [[maybe_unused]] const unsigned long long ta = KVT_NOW();
    const int nxt = next_item(it + 1);
    if (nxt < end) stage(nxt, cur ^ 1);   // (the other buffer: every wave left it before the barrier that ended the previous item)
    [[maybe_unused]] const unsigned long long tb = KVT_NOW();
    const T* Qc = Qs + cur * TB;
    const T* dOc = dOs + cur * TB;
    if ((wbits >> (it & 31)) & 1u) {
      const bool fullt = (fullbits >> (it & 31)) & 1u;
      const unsigned long long wk = kbs[cur * 64 + w * 16 + fr] >> (4 * g);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        f32x4 P2[2], dS2[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * t2 + ii;
          const float4 l4 = *(const float4*)(lse2 + cur * 64 + 16 * i + 4 * g);
          const float4 d4 = *(const float4*)(dls + cur * 64 + 16 * i + 4 * g);
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
          f32x4 S[1] = {f32x4{0, 0, 0, 0}}, dP = f32x4{d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            S[0] = AMma<bf16>::mma(frag_rows_sw(Qc, 16 * i, 32 * s2, l), kf[s2], S[0]);       // S[q][kv]
            dP = AMma<bf16>::mma(frag_rows_sw(dOc, 16 * i, 32 * s2, l), vf[s2], dP);          // dP[q][kv] - delta[q]
          }
          if (!fullt) mask_bits_block<1>(S, wk, i, -1e30f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = fexp2(fmaf(S[0][r], c2, -ll[r]));
            P2[ii][r] = pv;
            dS2[ii][r] = pv * dP[r];
          }
        }
        const bf16x8 pf = pack8(P2[0], P2[1]), sf = pack8(dS2[0], dS2[1]);
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) {
          dV[0][jd] = AMma<bf16>::mma(frag_tr_sw(dOc, 32 * t2, 16 * jd, l), pf, dV[0][jd]);   // dV^T[d][kv] += dO^T[d][q] P[q][kv]
          dK[0][jd] = AMma<bf16>::mma(frag_tr_sw(Qc, 32 * t2, 16 * jd, l), sf, dK[0][jd]);    // dK^T[d][kv] += Q^T[d][q] dS[q][kv]
        }
      }
    }
#ifdef ATTN_KV_TRACE
    asm volatile("" ::"v"(dK[0][0]), "v"(dV[0][0]));
    const unsigned long long tc = KVT_NOW();
#endif
    if (nxt < end) publish(cur ^ 1);
    [[maybe_unused]] const unsigned long long td = KVT_NOW();
    __syncthreads();
#ifdef ATTN_KV_TRACE
    const unsigned long long te = KVT_NOW();
    tr_[4] += tb - ta; tr_[5] += tc - tb; tr_[6] += td - tc; tr_[7] += te - td; tr_[8] += 1; tr_[9] += (wbits >> (it & 31)) & 1u;
#endif
    cur ^= 1;
    it = nxt;
  }
  tr_[2] = KVT_NOW();
  const int pos = p.rope_pos ? p.rope_pos[tok0 + min(kv, p.T - 1)] : min(kv, p.T - 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) dK[0][j] *= scale;
  store_grad_tile_sw(dK[0], true, p.rope_cos, p.rope_sin, pos, Qs, w, l);    // (dK through one Q buffer, dV through the other: one barrier fewer)
  store_grad_tile_sw(dV[0], false, p.rope_cos, p.rope_sin, pos, Qs + TB, w, l);
  __syncthreads();
  copy_out_tile_sw(Qs, (T*)p.dk + (tok0 + kvt * 64) * p.ldg + kvh * HD, p.ldg, kvt * 64, p.T, t, p.f8_amax ? p.f8_amax + 1 : nullptr);
  copy_out_tile_sw(Qs + TB, (T*)p.dv + (tok0 + kvt * 64) * p.ldg + kvh * HD, p.ldg, kvt * 64, p.T, t, p.f8_amax ? p.f8_amax + 2 : nullptr);
#ifdef ATTN_KV_TRACE
  tr_[3] = KVT_NOW();
  if (threadIdx.x == 0 && blockIdx.x < 8192) for (int i = 0; i < 12; ++i) rsys_attn_trace[blockIdx.x * 16 + i] = tr_[i];
#endif
}

// ------------------------------------------------------------------------ dK, dV on v_mfma_f32_32x32x16_bf16 (bf16, head_dim 64)
// The item loops above are bound by instruction ISSUE (profiles/r4_pmc_attention_sq.txt: the waves' active cycles add up to the SIMDs'
// issue capacity; an MFMA holds the vector issue port for 8 cycles whatever its shape).  This kernel does the same arithmetic with half
// the MFMA instructions, half the LDS fragment reads and half the staging per FLOP: a wave owns 32 keys (the lanes' l & 31) and runs
// 32 x 32 x 16 products, four waves = 128 keys = two kv tiles per workgroup, each staged Q / dO tile feeds all of them.
//   S[q][kv] = sum_d Q[q][d] K[kv][d]:  A = Q rows from LDS (ds_read_b128: row q0 + (l & 31), d = 16 s + 8 (l >> 5) ..+7), B = K from registers
//   dV^T[d][kv] += dO^T[d][q] P[q][kv]: the P accumulator registers 8 s .. 8 s + 7 ARE the B operand of k-step s (rows 16 s + 8 (j >> 2) +
//   4 (l >> 5) + (j & 3)); the matching A operand is read transposed from the row-major tile (two ds_read_b64_tr_b16: rows R .. R + 3 and R + 8 ..)
// LDS image: unpadded 128-byte rows, 16-byte chunk c of row r in slot c ^ sw32(r), sw32(r) = (r1, r2, r3) as bits (2, 1, 0): with that
// permutation the 32-row ds_read_b128 fragments (lane groups {0-3, 12-15, 20-27}, ...) and the 4-row x 4-chunk transposed reads of a
// 32-lane half both touch every bank once (checked against the guide's lane groups; the 16-row kernels' c ^ (r & 7) is 2-way here).
static bool attn_kv32_on() { return sw().attn_kv32 != 0; }   // 0 = the 16-key-per-wave kernel
static bool attn_kv_pairs(const AttnParams& p) {   // (bf16 is the caller's business: the fp32 launches never read order_k's pair form)
  return p.hd == 64 && p.is_bf16 && sw().attn_kv_dma != 0 && attn_dma_on() && attn_kv32_on();
}
#ifndef ATTN_KV32_WPS
#define ATTN_KV32_WPS 3   // waves per SIMD the kernel is compiled for: 168 registers, 5 dwords spilled OUTSIDE the item loop; 2 (189 registers) is 9 % slower (profiles/r5_ab_attn_kv32.log)
#endif
__global__ __launch_bounds__(256, ATTN_KV32_WPS) void attn_bwd_kv32_kernel(AttnParams p) {
  [[maybe_unused]] unsigned long long tr_[12] = {KVT_NOW(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (-DATTN_KV_TRACE only: tools/trace_attn_kv.sh)
  constexpr int HD = 64, TB = 64 * 64;   // elements of an unpadded tile
  using T = bf16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const Qb = smem_raw;                   // [2][64 q][128 B] swizzled
  unsigned char* const dOb = smem_raw + 2 * TB * 2;     // [2][64 q][128 B]
  float* const lse2 = (float*)(smem_raw + 4 * TB * 2);  // [2][64]
  float* const dls = lse2 + 128;                        // [2][64]
  const int rep = p.H / p.KV, nt = (p.T + 63) / 64, npair = (nt + 1) / 2;
  int grp, pr;
  attn_work(p.B * p.KV, npair, p.order_k, grp, pr);
  const int b = grp / p.KV, kvh = grp % p.KV;
  const int t = threadIdx.x, l = t & 63, r32 = l & 31, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kvt = 2 * pr + (w >> 1);                    // this wave's kv tile (may be nt: the odd tile's partner does nothing)
  const bool tile_ok = kvt < nt;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;
  const int kv = kvt * 64 + 32 * (w & 1) + r32;        // this lane's key/value token
  bf16x8 kf[4], vf[4];
  {
    const int kr = min(kv, p.T - 1);
    const T* krow = (const T*)p.k + (tok0 + kr) * p.ld + kvh * HD;
    const T* vrow = (const T*)p.v + (tok0 + kr) * p.ld + kvh * HD;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = *(const bf16x8*)(krow + 16 * s + 8 * h); vf[s] = *(const bf16x8*)(vrow + 16 * s + 8 * h); }
  }
  f32x16 dK[2], dV[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dK[0][i] = 0.f; dK[1][i] = 0.f; dV[0][i] = 0.f; dV[1][i] = 0.f; }
  const int qa = p.q_active != nullptr ? p.q_active[b] : 32;
  const unsigned int act = qa >= 32 ? ~0u : ((1u << qa) - 1u);
  const int t0i = b * nt + 2 * pr, t1i = min(2 * pr + 1, nt - 1) + b * nt;
  const unsigned int bits = (unsigned int)__builtin_amdgcn_readfirstlane((int)((p.kmap[t0i] | (2 * pr + 1 < nt ? p.kmap[t1i] : 0u)) & act));   // q tiles the workgroup stages (union of its two kv tiles)
  const int kti = b * nt + min(kvt, nt - 1);
  const unsigned int fullbits = tile_ok ? (unsigned int)__builtin_amdgcn_readfirstlane((int)p.kmap_full[kti]) : 0u;
  const unsigned int wbits = tile_ok ? (unsigned int)__builtin_amdgcn_readfirstlane((int)(p.kmap16[kti * 4 + 2 * (w & 1)] | p.kmap16[kti * 4 + 2 * (w & 1) + 1])) : 0u;
  const at_i32x4 q_rs = at_rsrc((const T*)p.q + tok0 * p.ld + kvh * rep * HD, ((long long)(p.T - 1) * p.ld + rep * HD) * sizeof(T));
  const at_i32x4 do_rs = at_rsrc((const T*)p.dO + tok0 * p.ldo + kvh * rep * HD, ((long long)(p.T - 1) * p.ldo + rep * HD) * sizeof(T));
  const at_i32x4 lse_rs = at_rsrc(p.lse + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  const at_i32x4 dl_rs = at_rsrc(p.delta + ((long long)b * p.H + kvh * rep) * p.T, (long long)rep * p.T * 4);
  // the pair bits of this wave's kv tile against every q tile: [nt][64 keys] words (an absent tile: an empty window, all zero)
  const at_i32x4 kb_rs = at_rsrc(p.kbits + (long long)kti * nt * 64, tile_ok ? (long long)nt * 64 * 8 : 0);
  // wave w fills rows 16 w .. 16 w + 15 of each tile with two DMA instructions (8 rows = 1 KB each): per-lane source offsets, fixed
  int qv_[2], dv_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = 16 * w + 8 * k + (l >> 3), ch = (l & 7) ^ sw32(row);
    qv_[k] = (int)((row * p.ld + ch * 8) * sizeof(T));
    dv_[k] = (int)((row * p.ldo + ch * 8) * sizeof(T));
  }
  // fragment addresses inside a tile (bytes; lane constants, the item / block / k-step parts are immediates):
  //   row reads: row q0 + r32, chunk 2 s + h -> slot (2 s) ^ G, G = h ^ sw32(r32)
  //   transposed reads: row q0 + 16 s2 + 8 u + 4 h + (i >> 2), columns 32 db + 16 g16 + 4 (i & 3) -> slot L ^ (4 db) ^ u
  const int G = h ^ sw32(r32);
  const int i16 = l & 15, g16 = (l >> 4) & 1;
  const int Lc = (2 * g16 + ((i16 & 3) >> 1)) ^ ((((i16 >> 2) >> 1) & 1) << 2 | (h << 1));
  int a_row[4], a_tr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_row[s] = r32 * 128 + (((2 * s) ^ G) << 4);
#pragma unroll
  for (int v = 0; v < 4; ++v) a_tr[v] = (4 * h + (i16 >> 2)) * 128 + ((Lc ^ ((v >> 1) << 2) ^ (v & 1)) << 4) + ((i16 & 1) << 3);   // v = 2 db + u
  const bool w0 = w == 0;
  const unsigned int lds0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)(LDS_AS unsigned char*)smem_raw);
  int sx = 0, sy = 0; unsigned long long skb = 0ull;   // the next item's row scalars (wave 0) and this lane's pair-bit word on their way
  auto stage = [&](int it, int buf) {   // it = head-in-group * 32 + q tile
    const int hh = it >> 5, qt = it & 31;
    const int qso = (int)((qt * 64 * p.ld + hh * HD) * sizeof(T)), dso = (int)((qt * 64 * p.ldo + hh * HD) * sizeof(T));
    const unsigned int qd = lds0 + buf * (TB * 2) + (16 * w) * 128, dd = qd + 2 * TB * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      dma16_asm(q_rs, qd + k * 1024, qv_[k], qso);
      dma16_asm(do_rs, dd + k * 1024, dv_[k], dso);
    }
    skb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(kb_rs, 8 * (32 * (w & 1) + r32), qt * 512, 0));
    if (w0) {
      const int so = (hh * p.T + qt * 64) * 4;
      sx = rsys_at_buffer_load_b32(lse_rs, 4 * l, so, 0);
      sy = rsys_at_buffer_load_b32(dl_rs, 4 * l, so, 0);
    }
  };
  auto publish = [&](int buf) {   // the scalars into LDS; every DMA of this wave landed
    // -lse / scale: what the S chains start from, so that p = exp2(c2 * S') needs no subtraction (1 / scale = 8 is exact); -delta likewise for dP
    if (w0) { lse2[buf * 64 + l] = __builtin_bit_cast(float, sx) * -8.0f; ((int*)dls)[buf * 64 + l] = sy ^ 0x80000000; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // items = (head of the group, q tile in `bits`), heads outermost: a scalar iterator, two scalar instructions per step
  const int end = rep * 32;
  auto next_item = [&](int from) {
    int hh = from >> 5, qt = from & 31;
    while (hh < rep) {
      const int n = next_bit(bits, qt);
      if (n < nt) return hh * 32 + n;
      ++hh; qt = 0;
    }
    return end;
  };
  int it = next_item(0), cur = 0;
  unsigned long long wk = 0ull;
  if (it < end) { stage(it, 0); publish(0); wk = skb; }
  // The K / V fragments' loads are retired HERE, in the compiler's own bookkeeping: left pending into the loop, its wait-count pass
  // keeps counted vmcnt waits in front of the loop's first MFMAs on every trip, and those counts do not know about the LDS-DMA issued
  // behind asm just before them -- vector memory retires in order, so each such wait would drain the DMA of the NEXT item
  // (s_waitcnt vmcnt(8) .. (1) in the ISA of the first build: the arithmetic of an item began by waiting for the next item's tiles).
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(kf[s]), "+v"(vf[s]));
  asm volatile("" : "+v"(wk));
  __syncthreads();
  tr_[1] = KVT_NOW();
  while (it < end) {
    [[maybe_unused]] const unsigned long long ta = KVT_NOW();
    const int nxt = next_item(it + 1);
    if (nxt < end) stage(nxt, cur ^ 1);   // (the other buffer: every wave left it before the barrier that ended the previous item)
    [[maybe_unused]] const unsigned long long tb = KVT_NOW();
    if ((wbits >> (it & 31)) & 1u) {
      const bool fullt = (fullbits >> (it & 31)) & 1u;
      const unsigned char* Qc = Qb + cur * (TB * 2);
      const unsigned char* dOc = dOb + cur * (TB * 2);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {   // 32 queries at a time
        f32x16 S, dP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // accumulator rows 4 k .. 4 k + 3 = queries 32 qb + 8 k + 4 h ..+3
          const float4 l4 = *(const float4*)(lse2 + cur * 64 + 32 * qb + 8 * k + 4 * h);
          const float4 d4 = *(const float4*)(dls + cur * 64 + 32 * qb + 8 * k + 4 * h);
          S[4 * k] = l4.x; S[4 * k + 1] = l4.y; S[4 * k + 2] = l4.z; S[4 * k + 3] = l4.w;       // the S chain starts from -lse[q] / scale
          dP[4 * k] = d4.x; dP[4 * k + 1] = d4.y; dP[4 * k + 2] = d4.z; dP[4 * k + 3] = d4.w;   // the dP chain starts from -delta[q]
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          S = mfma32(*(const bf16x8*)(Qc + qb * 4096 + a_row[s]), kf[s], S);       // S[q][kv]
          dP = mfma32(*(const bf16x8*)(dOc + qb * 4096 + a_row[s]), vf[s], dP);    // dP[q][kv] - delta[q]
        }
        // bit (query) of the lane's key word: query 32 qb + 8 (r >> 2) + 4 h + (r & 3).  A 32 x 32 block of which every pair is allowed
        // needs no mask (wave-uniform: one compare and a ballot): with ~5 % of the events carrying a token-mask id a 64 x 64 tile is
        // rarely full, a block of 32 keys often is
        const unsigned int w32 = (unsigned int)(qb ? (wk >> 32) : wk);
        if (!fullt && __builtin_amdgcn_ballot_w64(w32 != 0xFFFFFFFFu) != 0ull) {
          const int src = (int)(w32 >> (4 * h));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = __builtin_amdgcn_sbfe(src, 8 * (r >> 2) + (r & 3), 1);
            const float sv = S[r];
            S[r] = __builtin_bit_cast(float, (__builtin_bit_cast(int, sv) & m) | (__builtin_bit_cast(int, -1e30f) & ~m));
          }
        }
        f32x16 P, dS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fexp2(S[r] * c2);   // = exp(q.k / sqrt(hd) - lse); masked: exp2(-1.8e29) = 0
          P[r] = pv;
          dS[r] = pv * dP[r];   // (the 1/sqrt(hd) factor of dS is applied once to dK at the end)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {   // k-step = 16 queries: accumulator registers 8 s2 .. 8 s2 + 7
          const bf16x8 pf = pack8f(P, 8 * s2), sf = pack8f(dS, 8 * s2);
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int o = (32 * qb + 16 * s2) * 128;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(dOc + o + a_tr[2 * db]));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(dOc + o + 8 * 128 + a_tr[2 * db + 1]));
            dV[db] = mfma32(__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), pf, dV[db]);   // dV^T[d][kv] += dO^T[d][q] P[q][kv]
            const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(Qc + o + a_tr[2 * db]));
            const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(Qc + o + 8 * 128 + a_tr[2 * db + 1]));
            dK[db] = mfma32(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), sf, dK[db]);   // dK^T[d][kv] += Q^T[d][q] dS[q][kv]
          }
        }
      }
    }
#ifdef ATTN_KV_TRACE
    asm volatile("" ::"v"(dK[0][0]), "v"(dV[0][0]));
    const unsigned long long tc = KVT_NOW();
#endif
    if (nxt < end) publish(cur ^ 1);
    [[maybe_unused]] const unsigned long long td = KVT_NOW();
    __syncthreads();
#ifdef ATTN_KV_TRACE
    const unsigned long long te = KVT_NOW();
    tr_[4] += tb - ta; tr_[5] += tc - tb; tr_[6] += td - tc; tr_[7] += te - td; tr_[8] += 1; tr_[9] += (wbits >> (it & 31)) & 1u;
#endif
    wk = skb;
    cur ^= 1;
    it = nxt;
  }
  tr_[2] = KVT_NOW();
  // epilogue: dK (scaled, un-rotated) and dV as bf16 rows [key][64] through the two Q buffers (128 keys x 128 bytes each), then row stores.
  // accumulator register r of block db: d = 32 db + 8 (r >> 2) + 4 h + (r & 3), key = r32 of this wave
  const int pos = p.rope_pos ? p.rope_pos[tok0 + min(kv, p.T - 1)] : min(kv, p.T - 1);
  unsigned char* const Ks_ = Qb;                 // [128 keys][128 B]: wave w's keys at rows 32 w ..
  unsigned char* const Vs_ = Qb + 128 * 128;
  const int orow = 32 * w + r32;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 32 * db + 8 * g4 + 4 * h;
      float o[4] = {dK[db][4 * g4] * scale, dK[db][4 * g4 + 1] * scale, dK[db][4 * g4 + 2] * scale, dK[db][4 * g4 + 3] * scale};
      const float2 cc = *(const float2*)(p.rope_cos + pos * 32 + (d >> 1));
      const float2 ss = *(const float2*)(p.rope_sin + pos * 32 + (d >> 1));
      const float a0 = o[0] * cc.x + o[1] * ss.x, a1 = -o[0] * ss.x + o[1] * cc.x;
      const float b0 = o[2] * cc.y + o[3] * ss.y, b1 = -o[2] * ss.y + o[3] * cc.y;
      bf16x4 ko; ko[0] = (bf16)a0; ko[1] = (bf16)a1; ko[2] = (bf16)b0; ko[3] = (bf16)b1;
      bf16x4 vo; vo[0] = (bf16)dV[db][4 * g4]; vo[1] = (bf16)dV[db][4 * g4 + 1]; vo[2] = (bf16)dV[db][4 * g4 + 2]; vo[3] = (bf16)dV[db][4 * g4 + 3];
      // (chunk slot permuted by the row so that the 8-byte stores of a half-wave spread over the banks; the copy below undoes it)
      const int off = orow * 128 + ((((d >> 3) ^ (orow & 7)) << 4) | ((d & 4) << 1));
      *(bf16x4*)(Ks_ + off) = ko;
      *(bf16x4*)(Vs_ + off) = vo;
    }
  __syncthreads();
  {
    float amk = 0.f, amv = 0.f;
#pragma unroll
    for (int c = t; c < 128 * 8; c += 256) {   // 16-byte chunks of the 128 rows
      const int row = c >> 3, ch = c & 7;
      const int tile = 2 * pr + (row >> 6), key = tile * 64 + (row & 63);
      if (tile < nt && key < p.T) {
        const int off = row * 128 + ((ch ^ (row & 7)) << 4);
        const uint4 kq = *(const uint4*)(Ks_ + off), vq = *(const uint4*)(Vs_ + off);
        *(uint4*)((T*)p.dk + (tok0 + key) * p.ldg + kvh * HD + ch * 8) = kq;
        *(uint4*)((T*)p.dv + (tok0 + key) * p.ldg + kvh * HD + ch * 8) = vq;
        if (p.f8_amax != nullptr) { amk = fmaxf(amk, chunk_amax<bf16>(kq)); amv = fmaxf(amv, chunk_amax<bf16>(vq)); }
      }
    }
    if (p.f8_amax != nullptr) {
      amk = wave_max(amk); amv = wave_max(amv);
      if (l == 0) { f8_amax_add(p.f8_amax + 1, amk); f8_amax_add(p.f8_amax + 2, amv); }
    }
  }
#ifdef ATTN_KV_TRACE
  tr_[3] = KVT_NOW();
  if (threadIdx.x == 0 && blockIdx.x < 8192) for (int i = 0; i < 12; ++i) rsys_attn_trace[blockIdx.x * 16 + i] = tr_[i];
#endif
}

// ------------------------------------------------------------------------ backward: dQ (one workgroup per q tile and R heads of a kv group)
template <typename T, int HD, int R, bool DMA = false>
#ifndef ATTN_DQ_WPS
#define ATTN_DQ_WPS 3   // waves per SIMD the two-head LDS-DMA form is compiled for (4: 128 registers, 13 spilled dwords; A/B in profiles/r4_ab_attn_dq_four_waves.log)
#endif
__global__ __launch_bounds__(256, (is_bf16<T>::value && HD <= 64) ? (R == 1 ? 3 : (DMA ? ATTN_DQ_WPS : 2)) : 1) void attn_bwd_q_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  static_assert(!DMA || (is_bf16<T>::value && HD == 64), "LDS-DMA staging: bf16, head_dim 64");
  constexpr int TILE = DMA ? 64 * 64 : C::TILE;   // elements of a staged tile (DMA: unpadded, swizzled; attn_fwd_kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = (T*)smem_raw;                  // [2][64 kv][LDD]
  T* Vs = Ks + 2 * TILE;                 // [2][64 kv][LDD]
  unsigned long long* ab = (unsigned long long*)(Vs + 2 * TILE);   // [2][64] pair bits of the 64 queries against the staged tile's keys
  const int nt = (p.T + 63) / 64;
  int grp, inner;
  attn_work(p.B * p.KV, (p.H / p.KV / R) * nt, p.order_q, grp, inner);
  const int b = grp / p.KV, kvh = grp % p.KV, h0 = kvh * (p.H / p.KV) + (inner / nt) * R, qt = inner % nt;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, g = l >> 4, fr = l & 15;
  const long long tok0 = (long long)b * p.T;
  if (p.q_active != nullptr && qt >= p.q_active[b]) {   // dO of these queries is identically zero: so is dQ (uniform: whole workgroup)
    constexpr int CPRZ = HD / C::E;
    for (int c = t; c < 64 * CPRZ * R; c += 256) {
      const int r = c / (64 * CPRZ), cc = c % (64 * CPRZ), row = cc / CPRZ, ch = cc % CPRZ;
      if (qt * 64 + row < p.T) *(uint4*)((T*)p.dq + (tok0 + qt * 64 + row) * p.ldg + (h0 + r) * HD + ch * C::E) = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const float scale = rsqrtf((float)HD), c2 = scale * LOG2E;
  const int q = qt * 64 + w * 16 + fr;
  const bool qv = q < p.T;
  typename M::Frag qf[R][C::NDS], dof[R][C::NDS];
  float dl[R], lse2[R];
  f32x4 dQ[R][HD / 16];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int h = h0 + r;
    const T* qrow = (const T*)p.q + (tok0 + min(q, p.T - 1)) * p.ld + h * HD;
    const T* drow = (const T*)p.dO + (tok0 + min(q, p.T - 1)) * p.ldo + h * HD;
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) { qf[r][s] = frag_global<T>(qrow, s * C::KS, HD, l); dof[r][s] = frag_global<T>(drow, s * C::KS, HD, l); }
    const long long so = ((long long)b * p.H + h) * p.T + min(q, p.T - 1);
    // delta = rowsum(dO * O) of this lane's query: the four lanes that share a query hold disjoint quarters of d in their
    // fragments.  Written out for the dK/dV kernel, which runs after this one.
    float d_ = 0.f;
    const T* orow = (const T*)p.o + (tok0 + min(q, p.T - 1)) * p.ldo + h * HD;
#pragma unroll
    for (int s = 0; s < C::NDS; ++s) d_ += frag_dot(dof[r][s], frag_global<T>(orow, s * C::KS, HD, l));
    d_ += __shfl_xor(d_, 16, 64);
    d_ += __shfl_xor(d_, 32, 64);
    if (!qv) d_ = 0.f;
    if (g == 0 && qv) p.delta[so] = d_;
    dl[r] = d_;
    lse2[r] = qv ? p.lse[so] * LOG2E : 0.f;
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) dQ[r][j] = f32x4{0, 0, 0, 0};
  }
  if constexpr (!DMA) for (int i = 0; i < 2; ++i) { zero_pad_cols<T, HD>(Ks + i * C::TILE, t); zero_pad_cols<T, HD>(Vs + i * C::TILE, t); }
  // (scalars through readfirstlane: loaded by vector memory -- the maps are not const -- and otherwise still pending when the item loop begins)
  const unsigned int bits = (unsigned int)__builtin_amdgcn_readfirstlane((int)p.qmap[b * nt + qt]), fullbits = (unsigned int)__builtin_amdgcn_readfirstlane((int)p.qmap_full[b * nt + qt]);
  const unsigned int wbits = __builtin_amdgcn_readfirstlane(p.qmap16[(b * nt + qt) * 4 + w]);
  // K / V of this kv head, rows of this sequence ([T][HD] windows of the row-major qkv), and the rows' uid / tm (tile_load_buf)
  const at_i32x4 k_rs = at_rsrc((const T*)p.k + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  const at_i32x4 v_rs = at_rsrc((const T*)p.v + tok0 * p.ld + kvh * HD, ((long long)(p.T - 1) * p.ld + HD) * sizeof(T));
  // the pair bits of this q tile against every kv tile: [nt][64 queries] words (AttnParams::qbits)
  const at_i32x4 qb_rs = at_rsrc(p.qbits + ((long long)b * nt + qt) * nt * 64, (long long)nt * 64 * 8);
  TileOffs<T, HD> kv_of;
  tile_offsets<T, HD>(kv_of, p.ld, t);
  const bool w0 = __builtin_amdgcn_readfirstlane(w) == 0;
  TileRegs<T, HD> rk, rv;
  unsigned long long rb = 0ull;
  int dv_[2] = {0, 0};   // DMA: per-lane source offsets of the wave's two instructions per tile (attn_fwd_kernel)
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int row = 16 * w + 8 * k + (l >> 3); dv_[k] = (int)((row * p.ld + ((l & 7) ^ (row & 7)) * 8) * sizeof(T)); }
  }
  auto gload = [&](int kt, int buf) {
    const int so = (int)(kt * 64 * p.ld * sizeof(T));
    if constexpr (DMA) {
      unsigned char* kd = (unsigned char*)(Ks + buf * TILE) + (16 * w) * 128;
      unsigned char* vd = (unsigned char*)(Vs + buf * TILE) + (16 * w) * 128;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        dma16<ATTN_QSIDE_DMA_ASM>(k_rs, kd + k * 1024, dv_[k], so);
        dma16<ATTN_QSIDE_DMA_ASM>(v_rs, vd + k * 1024, dv_[k], so);
      }
    } else {
      tile_load_buf<T, HD>(rk, k_rs, kv_of, so);
      tile_load_buf<T, HD>(rv, v_rs, kv_of, so);
    }
    if (w0) rb = __builtin_bit_cast(unsigned long long, rsys_at_buffer_load_b64(qb_rs, 8 * l, kt * 512, 0));   // query l's keys of tile kt
  };
  auto lstore = [&](int buf) {
    if constexpr (!DMA) {
      tile_store<T, HD>(rk, Ks + buf * TILE, t);
      tile_store<T, HD>(rv, Vs + buf * TILE, t);
    }
    if (w0) ab[buf * 64 + l] = rb;
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  int kt = next_bit(bits, 0), cur = 0;
  if (kt < nt) { gload(kt, 0); lstore(0); }
  if constexpr (DMA && ATTN_QSIDE_DMA_ASM) {   // retire the Q / dO fragments' and row scalars' loads before the loop (see dma16)
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int s = 0; s < C::NDS; ++s) asm volatile("" : "+v"(qf[r][s]), "+v"(dof[r][s]));
      asm volatile("" : "+v"(dl[r]), "+v"(lse2[r]));
    }
  }
  __syncthreads();
  while (kt < nt) {
    const int nxt = next_bit(bits, kt + 1);
    if (nxt < nt) gload(nxt, cur ^ 1);
    const T* Kc = Ks + cur * TILE;
    const T* Vc = Vs + cur * TILE;
    if ((wbits >> kt) & 1u) {
    const bool partial = !((fullbits >> kt) & 1u);
    const unsigned long long wq = ab[cur * 64 + w * 16 + fr] >> (4 * g);   // the lane's query against the tile's 64 keys (mask_bits_block)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {   // 32 keys at a time: both stages, two score blocks per head alive
      f32x4 dS[R][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * t2 + ii;
        f32x4 S[R], dP[R];
        const f32x4 s0 = partial ? mask_bias_block(wq, i) : f32x4{0, 0, 0, 0};   // the mask as the chains' starting value, one for every head
#pragma unroll
        for (int r = 0; r < R; ++r) { S[r] = s0; dP[r] = f32x4{-dl[r], -dl[r], -dl[r], -dl[r]}; }
        first_stage_block_r<T, HD, R, DMA>(S, Kc, qf, i, l);      // S^T[kv][q] (+ mask)
        first_stage_block_r<T, HD, R, DMA>(dP, Vc, dof, i, l);    // dP^T[kv][q] - delta[q]
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float pv = fexp2(fmaf(S[r][rr], c2, -lse2[r]));
            dS[r][ii][rr] = pv * dP[r][rr];   // (the 1/sqrt(hd) factor of dS is applied once to dQ at the end)
          }
      }
      acc_second_stage_half_r<T, HD, R, DMA>(dQ, dS, Kc, t2, l);    // dQ^T[d][q] += K^T[d][kv] dS^T[kv][q]
    }
    }
    if (nxt < nt) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    kt = nxt;
  }
  const int pos = p.rope_pos ? p.rope_pos[tok0 + min(q, p.T - 1)] : min(q, p.T - 1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) dQ[r][j] *= scale;
    store_grad_tile<T, HD>(dQ[r], true, p.rope_cos, p.rope_sin, pos, Ks + r * C::TILE, w, l);   // (head r in the r-th staged-tile slot)
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r)
    copy_out_tile<T, HD>(Ks + r * C::TILE, (T*)p.dq + (tok0 + qt * 64) * p.ldg + (h0 + r) * HD, p.ldg, qt * 64, p.T, t, p.f8_amax);
}

template <typename T, int HD>
static int attn_bwd_hd(const AttnParams& p, hipStream_t s) {
  using C = ACfg<T, HD>;
  const size_t sm_kv = sizeof(T) * 4 * C::TILE + 512 * 4;
  const size_t sm_q = sizeof(T) * 4 * C::TILE + 384 * 4;
  static bool set = false;
  if (!set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_kv));
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_q_kernel<T, HD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_q));
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_q_kernel<T, HD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_q));
    set = true;
  }
  // the dQ kernel also produces delta = rowsum(dO * O), which the dK/dV kernel reads
  bool q_done = false;
  if constexpr (is_bf16<T>::value && HD == 64) {
    if (attn_dma_on()) {
      const size_t sm_dma = 4 * 64 * 64 * 2 + 384 * 4;
      if (attn_heads_per_wg(p) == 2) hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD, 2, true>), dim3(((p.T + 63) / 64) * (p.H / 2) * p.B), dim3(256), sm_dma, s, p);
      else hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD, 1, true>), dim3(((p.T + 63) / 64) * p.H * p.B), dim3(256), sm_dma, s, p);
      q_done = true;
    }
  }
  if (!q_done) {
    if (attn_heads_per_wg(p) == 2) hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD, 2>), dim3(((p.T + 63) / 64) * (p.H / 2) * p.B), dim3(256), sm_q, s, p);
    else hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD, 1>), dim3(((p.T + 63) / 64) * p.H * p.B), dim3(256), sm_q, s, p);
  }
  HIP_CHECK(hipGetLastError());
  // (two adjacent kv tiles per workgroup -- Q / dO staging and every fragment read shared by 32 keys per wave -- measured 6 % slower:
  // 254 registers, two waves per SIMD; profiles/r4_ab_attn_dkv_two_key_tiles.log)
  if constexpr (is_bf16<T>::value && HD == 64) {
    const bool dma = sw().attn_kv_dma != 0;   // A/B switch of this kernel alone
    if (attn_kv_pairs(p)) {   // 32 keys per wave on 32 x 32 x 16 products, two kv tiles per workgroup (order_k lists the pairs)
      const int npair = ((p.T + 63) / 64 + 1) / 2;
      hipLaunchKernelGGL(attn_bwd_kv32_kernel, dim3(npair * p.KV * p.B), dim3(256), 4 * 64 * 64 * 2 + 256 * 4, s, p);
      HIP_CHECK(hipGetLastError());
      return RSYS_OK;
    }
    if (dma && attn_dma_on()) {
      hipLaunchKernelGGL(attn_bwd_kv_dma_kernel, dim3(((p.T + 63) / 64) * p.KV * p.B), dim3(256), 4 * 64 * 64 * 2 + 512 * 4, s, p);
      HIP_CHECK(hipGetLastError());
      return RSYS_OK;
    }
  }
  hipLaunchKernelGGL((attn_bwd_kv_kernel<T, HD>), dim3(((p.T + 63) / 64) * p.KV * p.B), dim3(256), sm_kv, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename T>
int launch_attn_bwd(const AttnParams& p, hipStream_t s) {
  int rc = check_attn(p, sizeof(T));
  if (rc) return rc;
  ARG_CHECK(p.H / p.KV <= 8, "attention: at most 8 query heads per kv head");
  switch (p.hd) {
    case 16: return attn_bwd_hd<T, 16>(p, s);
    case 32: return attn_bwd_hd<T, 32>(p, s);
    case 64: return attn_bwd_hd<T, 64>(p, s);
    case 128: return attn_bwd_hd<T, 128>(p, s);
  }
  set_error("attention: head_dim must be 16, 32, 64 or 128");
  return RSYS_ERR_ARG;
}
template int launch_attn_bwd<bf16>(const AttnParams&, hipStream_t);
template int launch_attn_bwd<float>(const AttnParams&, hipStream_t);

}  // namespace rsys

#ifdef ATTN_KV_TRACE
extern "C" __attribute__((visibility("default"))) int rsys_attn_trace_read(void* dst, unsigned long long n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rsys::rsys_attn_trace), n);
}
#endif
