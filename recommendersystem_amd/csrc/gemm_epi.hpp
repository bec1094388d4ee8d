// Fused GEMM epilogues, LDS-staged form: used by the 128x128 kernel (gemm.hip), which passes its accumulators through
// LDS and hands every lane W consecutive columns of one output row; epi_item applies the epilogue to those W values and
// writes them with 16-byte accesses.  (The 256-wide kernels write straight from registers: gemm_epi_reg.hpp.)
// Also declares the launchers of the LDS-DMA kernels (gemm8p.hip, gemm8c.hip) for the dispatcher in gemm.hip.
#pragma once
#include <utility>

#include "gemm.hpp"

namespace rsys {

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: accumulator tiles must be indexed by constants or hipcc moves them to scratch
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// nv (<= W) values to dst; full groups leave as 16-byte stores (8-byte bf16 stores run at about half the rate)
template <typename CT, int W>
__device__ __forceinline__ void store_vec(CT* dst, const float* v, int nv) {
  if (nv == W) {
    if constexpr (is_bf16<CT>::value) {
      if constexpr (W == 8) { bf16x8 pk; for (int k = 0; k < 8; ++k) pk[k] = (bf16)v[k]; *(bf16x8*)dst = pk; }
      else { bf16x4 pk; for (int k = 0; k < 4; ++k) pk[k] = (bf16)v[k]; *(bf16x4*)dst = pk; }
    } else {
#pragma unroll
      for (int k = 0; k < W; k += 4) *(float4*)(dst + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
  } else {
    for (int k = 0; k < nv; ++k) dst[k] = from_f32<CT>(v[k]);
  }
}

// One item of the primary output tile: columns [col, col + nv) of row `row`, accumulator values in v.
template <typename CT, int W>
__device__ __forceinline__ void epi_item(const GemmParams& p, long long row, int col, float (&v)[W], int nv, bool outf32) {
  switch (p.epi) {
    case EPI_STORE:
#pragma unroll
      for (int k = 0; k < W; ++k) v[k] *= p.alpha;
      break;
    case EPI_ACCUM:
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < nv) v[k] += ((const float*)p.C)[row * p.ldc + col + k];
      break;
    case EPI_BIAS:
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < nv) v[k] += p.bias[col + k];
      break;
    case EPI_RESIDUAL:
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < nv) v[k] += p.resid[row * p.ldr + col + k];
      break;
    case EPI_TABLE: {
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < nv) v[k] += p.E[row * p.ldc + col + k] + p.bias[col + k];
      float* d32 = (float*)p.C + row * p.ldc + col;
      if (nv == W) {
#pragma unroll
        for (int k = 0; k < W; k += 4) *(float4*)(d32 + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
      } else for (int k = 0; k < nv; ++k) d32[k] = v[k];
      store_vec<CT, W>((CT*)p.C2 + row * p.ldc2 + col, v, nv);
      return;
    }
    case EPI_GELU: {
      float ge[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (k < nv) v[k] += p.bias[col + k];
        ge[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752f));
      }
      store_vec<CT, W>((CT*)p.C + row * p.ldc + col, v, nv);
      store_vec<CT, W>((CT*)p.C2 + row * p.ldc2 + col, ge, nv);
      return;
    }
    case EPI_QKV_ROPE:
      // rotate interleaved pairs (transformer.model.py:182-190): W consecutive columns = W/2 pairs of one head
#pragma unroll
      for (int k = 0; k < W; ++k) v[k] *= p.alpha;
      if (col < p.n_q + p.n_k) {
        const int pos = p.rope_pos ? p.rope_pos[row] : (int)(row % p.T);
        const int cc = col < p.n_q ? col : col - p.n_q;
        const int d2 = (cc & (p.hd - 1)) >> 1;
        const float* cs = p.rope_cos + pos * (p.hd >> 1) + d2;
        const float* sn = p.rope_sin + pos * (p.hd >> 1) + d2;
#pragma unroll
        for (int k = 0; k < W; k += 2) {
          const float c = cs[k >> 1], s2 = sn[k >> 1];
          const float a0 = v[k] * c - v[k + 1] * s2, a1 = v[k] * s2 + v[k + 1] * c;
          v[k] = a0; v[k + 1] = a1;
        }
      }
      break;
    case EPI_SWIGLU_BWD: {
      // column col = i index of dg; a,b live at (i>>4)*32 + (i&15) (+16) of the interleaved [a|b] rows;
      // N % 16 == 0 and W | 16, so the W columns of an item are one aligned group of a and one of b
      const long long base = row * p.ldc + (long long)(col >> 4) * 32 + (col & 15);
      float av[W], bv[W], da[W], db[W];
      if constexpr (is_bf16<CT>::value && W == 8) {
        const bf16x8 a8 = *(const bf16x8*)((const CT*)p.C2 + base), b8 = *(const bf16x8*)((const CT*)p.C2 + base + 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) { av[k] = (float)a8[k]; bv[k] = (float)b8[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) { av[k] = to_f32(((const CT*)p.C2)[base + k]); bv[k] = to_f32(((const CT*)p.C2)[base + 16 + k]); }
      }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float sg = 1.f / (1.f + __expf(-av[k]));
        da[k] = v[k] * bv[k] * sg * (1.f + av[k] * (1.f - sg));
        db[k] = v[k] * av[k] * sg;
      }
      store_vec<CT, W>((CT*)p.C + base, da, W);
      store_vec<CT, W>((CT*)p.C + base + 16, db, W);
      return;
    }
    default: break;   // SWIGLU: plain store of the primary tile
  }
  if (outf32) {
    float* dst = (float*)p.C + row * p.ldc + col;
    if (nv == W) {
#pragma unroll
      for (int k = 0; k < W; k += 4) *(float4*)(dst + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    } else for (int k = 0; k < nv; ++k) dst[k] = v[k];
  } else {
    CT* dstc = (CT*)p.C + row * p.ldc + col;
    if (p.accum) {
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < nv) v[k] += to_f32(dstc[k]);
    }
    store_vec<CT, W>(dstc, v, nv);
  }
}

// row-major bf16 operands, one 256x256x64 LDS-DMA pipeline (gemm8p.hip); returns RSYS_OK or an error.
// gemm8p_eligible: the problem satisfies that kernel's layout / size preconditions.
bool gemm8p_eligible(const GemmParams& p);
int launch_gemm8p(const GemmParams& p, hipStream_t s);
// the same pipeline with one operand stream across the workgroup's output tiles and the epilogue overlapped with the next
// tile's first K tile (gemm8c.hip): the epilogue classes of the training step; launch_gemm8p forwards eligible problems
bool gemm8c_eligible(const GemmParams& p);
int launch_gemm8c(const GemmParams& p, hipStream_t s);
// gemm4p.hip: the long-K member of the family (plain bf16 store, whole tiles); launch_gemm8c forwards what gemm4p_takes()
bool gemm4p_eligible(const GemmParams& p);
bool gemm4p_takes(const GemmParams& p);
int launch_gemm4p(const GemmParams& p, hipStream_t s);
// gemm4k.hip: the K-major split-K member of the four-wave loops (weight gradients); grids and work lists are gemm8p's
bool gemm4k_eligible(const GemmParams& p);
int launch_gemm4k(const GemmParams& p, int grid, hipStream_t s);
int launch_gemm4k_group(const GemmParams* d_probs, const int* d_off, const unsigned int* d_work, int grid, hipStream_t s);
bool gemm8p_forwards_to_8c(const GemmParams& p);   // what launch_gemm8p will do with p (timing tags)
// the same pipeline for K-major bf16 operands with split-K fp32 atomics (weight gradients); picks its own K split
bool gemm8p_tn_eligible(const GemmParams& p);
int launch_gemm8p_tn(const GemmParams& p, hipStream_t s);
bool gemm8p_tn_store_eligible(const GemmParams& p);   // K-major operands, one K split, fp32 C stored / accumulated (device-side K limit allowed)
int launch_gemm8p_tn_store(const GemmParams& p, hipStream_t s);
// the row-major pipeline with the same split-K mapping and atomic epilogue (long K, few output tiles, K-contiguous operands)
bool gemm8p_nt_splitk_eligible(const GemmParams& p);
int launch_gemm8p_nt_splitk(const GemmParams& p, hipStream_t s);
int gemm8p_splits(const GemmParams& p, bool k_major);            // K splits the two split-K forms choose (gemm8p.hip)
int gemm_slab_begin(const GemmParams& p, hipStream_t s);         // deterministic split-K: clear the slab / sum it into C (gemm.hip)
int gemm_slab_end(const GemmParams& p, hipStream_t s);

}  // namespace rsys
