// Register-resident epilogue of the LDS-DMA GEMM kernels (gemm8p.hip, gemm8c.hip): every wave owns a 128x64 block
// of the output as 8x4 MFMA 16x16 accumulator blocks of the TRANSPOSED product (B fragment as the first MFMA
// operand), so lane (fq, fr) = (lane >> 4, lane & 15) holds FOUR CONSECUTIVE COLUMNS of one row:
//   acc[i][j][r] = C[wm0 + 16 i + fr][wn0 + 16 j + 4 fq + r].
// Every fused epilogue is lane-local in this layout (RoPE pairs, the [16 a | 16 b] SwiGLU groups = blocks j, j+1)
// and the results leave straight from registers: f32 outputs as 16-byte stores; bf16 outputs after one
// v_permlane16_swap per dword between two blocks, which gives every lane 8 consecutive columns (16 bytes,
// 64-byte row segments per wave instruction).  No LDS round trip, no barrier: the stores are in flight when the
// workgroup retires.
#pragma once
#include "gemm.hpp"
#include "gemm_epi.hpp"

namespace rsys {

// wm0 / wn0: first row / column of the wave's block; full: the whole workgroup tile lies inside the matrix.
// MODE 1 / 0: the caller knows at compile time that the tile is full / an edge tile (the persistent kernel runs its
// full tiles and its edge tiles in two separate loops); -1: decided by `full` at run time.
// ECF >= 0: the epilogue class is known at compile time (gemm8c.hip: one kernel per class; EPI_QKV_ROPE there means the
// implicit positions row % T), else it is p.epi.
// NRB: row blocks of 16 the wave owns (8; 4 in the 128-row tiles of gemm8c's HALF form, which leave acc[4..7] unused).
template <int MODE = -1, int ECF = -1, int NRB = 8>
__device__ __forceinline__ void epilogue_regs(const GemmParams& p, const f32x4 (&acc)[8][4], int wm0, int wn0, bool full,
                                              int fq, int fr) {
  auto pk2 = [](float a, float b) __attribute__((always_inline)) -> unsigned int {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v; v[0] = (bf16)a; v[1] = (bf16)b;
    return __builtin_bit_cast(unsigned int, v);
  };
  const bool cf32 = p.c_f32 != 0;

  // x: 4 values at columns cx + 4 fq + r, y: 4 values at columns cy + 4 fq + r of row `rowp` (T-typed, element
  // pointer of the row).  After the swaps lane rows (fq) 0/2 hold columns cx + 8 (fq>>1) + [0,8), rows 1/3 the same
  // of cy.  Every lane takes part in the swaps; only the store is guarded.
  // Everything below is straight-line code per epilogue class (no branch around a load): hipcc then keeps exact vmcnt
  // counts, and because operands of row block i+1 are requested BEFORE the stores of row block i are issued, no wait
  // ever has to drain a store (memory operations retire in issue order).  Addresses are uniform base pointers plus
  // 32-bit byte offsets (launcher: every operand spans < 4 GB), so a load / store costs about one vector add.
  // FULL tiles (all 256x256 outputs inside the matrix) carry no clamps or store masks; edge tiles clamp the load
  // addresses and mask the stores.  The launcher guarantees N % 8 == 0: a lane's group of 4 (f32) or 8 (bf16)
  // columns is inside or outside as a whole.
  auto run = [&](auto EC, auto FULLC) __attribute__((always_inline))  {
    constexpr int ecv = decltype(EC)::value;
    constexpr int ec = ecv == 100 ? (int)EPI_QKV_ROPE : ecv;
    constexpr bool rope_explicit = ecv == 100;   // per-row positions given (inference)
    constexpr bool FULL = decltype(FULLC)::value;
    const bool outf32 = cf32 || ec == EPI_ACCUM || ec == EPI_RESIDUAL;
    const int lrow = wm0 + fr;                                         // row of row block 0
    const int c4 = wn0 + 4 * fq;                                        // first column of the lane's 4 in block 0 (block j: + 16 j)
    const int c8 = wn0 + ((fq & 1) << 4) + ((fq >> 1) << 3);            // first column of the lane's 8 after a pair swap of blocks (0,1); (2,3): + 32
    auto rowoff = [&](int i, long long ld, int esz) __attribute__((always_inline)) -> unsigned int  {   // byte offset of the lane's row in row block i
      int r = lrow + 16 * i;
      if constexpr (!FULL) r = min(r, p.M - 1);
      return (unsigned int)r * (unsigned int)(ld * esz);
    };
    auto colclamp = [&](int col, int ncols) __attribute__((always_inline)) -> int  { if constexpr (FULL) return col; else return col < ncols ? col : 0; };
    auto ldf4 = [&](const void* base, unsigned int off) __attribute__((always_inline)) -> float4  { return *(const float4*)((const char*)base + off); };
    auto ldf2 = [&](const void* base, unsigned int off) __attribute__((always_inline)) -> float2  { return *(const float2*)((const char*)base + off); };
    // x: 4 values of block jx at columns .. + 4 fq + r, y: the same of the partner block.  After the swaps lane rows
    // (fq) 0/2 hold 8 consecutive columns of x's block, rows 1/3 of y's.  Every lane takes part in the swaps.
    auto store_pair = [&](void* base, unsigned int off, bool ok, const float (&x)[4], const float (&y)[4]) __attribute__((always_inline))  {
      unsigned int x0 = pk2(x[0], x[1]), x1 = pk2(x[2], x[3]), y0 = pk2(y[0], y[1]), y1 = pk2(y[2], y[3]);
      auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      if (FULL || ok) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
        *(u32x4*)((char*)base + off) = v;   // (nontemporal stores measured 3 % slower in the step)
      }
    };
    auto store_f32 = [&](void* base, unsigned int off, bool ok, const float (&x)[4]) __attribute__((always_inline))  {
      if (FULL || ok) {
        const f32x4 v = {x[0], x[1], x[2], x[3]};
        *(f32x4*)((char*)base + off) = v;
      }
    };
    auto unpack4 = [](float lo, float hi, float (&o)[4]) __attribute__((always_inline)) {
      const bf16x4 q = __builtin_bit_cast(bf16x4, make_float2(lo, hi));
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (float)q[r];
    };

    // per-column vectors, the same for all rows
    float4 bias4[4];
    if constexpr (ec == EPI_BIAS || ec == EPI_GELU || ec == EPI_TABLE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bias4[j] = ldf4(p.bias, (unsigned int)colclamp(c4 + 16 * j, p.N) * 4u);
      if constexpr (!FULL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(bias4[j].x), "v"(bias4[j].y), "v"(bias4[j].z), "v"(bias4[j].w));
      }
    }
    // RoPE: byte offset of the lane's (cos, sin) pair inside one position's row for every block; blocks beyond the
    // q and k columns read position 0 (cos 1, sin 0: the rotation is the identity there)
    unsigned int rope_d[4], rope_m[4]; unsigned int rope_p0 = 0;
    if constexpr (ec == EPI_QKV_ROPE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = c4 + 16 * j;
        const int cc = col < p.n_q ? col : col - p.n_q;
        rope_d[j] = (unsigned int)(((cc & (p.hd - 1)) >> 1) * 4);
        rope_m[j] = col < p.n_q + p.n_k ? 0xFFFFFFFFu : 0u;
      }
      if constexpr (!rope_explicit) rope_p0 = (unsigned int)lrow % (unsigned int)p.T;
    }

    // per-row-block operands, requested one row block ahead (16 registers per stage)
    struct Pre { float4 f[4]; };
    auto request = [&](auto I, Pre& pre) __attribute__((always_inline))  {
      constexpr int i = decltype(I)::value;
      if constexpr (ec == EPI_ACCUM || ec == EPI_RESIDUAL || ec == EPI_TABLE) {
        const void* base = ec == EPI_ACCUM ? (const void*)p.C : ec == EPI_RESIDUAL ? (const void*)p.resid : (const void*)p.E;
        const unsigned int ro = rowoff(i, ec == EPI_RESIDUAL ? p.ldr : p.ldc, 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = ldf4(base, ro + (unsigned int)colclamp(c4 + 16 * j, p.N) * 4u);
      } else if constexpr (ec == EPI_QKV_ROPE) {
        // f[j] = {cos0, cos1, sin0, sin1} of the lane's two pairs in block j
        unsigned int pos;
        if constexpr (rope_explicit) pos = (unsigned int)p.rope_pos[min(lrow + 16 * i, p.M - 1)];
        else {
          const unsigned int q = rope_p0 + 16u * i;   // row % T without a division per row block
          pos = p.T >= 128 ? (q >= (unsigned int)p.T ? q - (unsigned int)p.T : q) : q % (unsigned int)p.T;
        }
        const unsigned int po = pos * (unsigned int)((p.hd >> 1) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned int off = (po & rope_m[j]) + rope_d[j];
          // interleaved table {cos p, sin p, cos p+1, sin p+1}: one 16-byte load per block (launcher: rope_cs is set).  No branch
          // around a load here -- a uniform `if (table A) else (table B)` made hipcc drain the loads at every row block:
          // qkv_fwd 0.98 -> 1.42 ms per step
          const float4 v = ldf4(p.rope_cs, off * 2u);
          pre.f[j] = make_float4(v.x, v.z, v.y, v.w);
        }
      } else if constexpr (ec == EPI_SWIGLU_BWD) {
        // saved a, b of dg column c live at (c>>4)*32 + (c&15) (+16) of the [a|b] rows: f[j] = {a (4 bf16), b (4 bf16)}.
        // Loaded the way store_pair writes -- 16 bytes per lane, 64-byte row segments -- and brought back to "4 columns of
        // a, 4 of b" with the same two swaps (the transformation is an involution); half the load instructions of 8-byte loads.
        const unsigned int ro = rowoff(i, p.ldc2, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int oc = colclamp(((wn0 >> 4) + j) * 32 + ((fq & 1) << 4) + ((fq >> 1) << 3), 2 * p.N);
          pre.f[j] = *(const float4*)((const char*)p.C2 + ro + (unsigned int)oc * 2u);   // raw; un-swapped where it is consumed
        }
      }
    };
    float amx0 = 0.f, amx1 = 0.f;   // fp8 trunk: running amax of the tensors this epilogue produces (rows / columns beyond the matrix repeat the last ones)
    const bool want_amax = (ec == EPI_SWIGLU || ec == EPI_SWIGLU_BWD) && p.f8_amax_out != nullptr;
    auto finish = [&](auto I, const Pre& pre) __attribute__((always_inline))  {
      constexpr int i = decltype(I)::value;
      const bool rowok = FULL || lrow + 16 * i < p.M;
      if constexpr (!FULL && (ec == EPI_ACCUM || ec == EPI_RESIDUAL || ec == EPI_TABLE || ec == EPI_QKV_ROPE || ec == EPI_SWIGLU_BWD)) {
        // Edge tiles: the only consumers of the requested operands are masked stores, and hipcc sinks the wait for a
        // load into the masked block with them; where the mask is empty the load then stays "pending" for the compiler's
        // wait-count pass, and in a persistent kernel it would guard the first instructions of the next tile's K loop
        // on every trip.  An unconditional (empty) use retires the loads here.
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(pre.f[j].x), "v"(pre.f[j].y), "v"(pre.f[j].z), "v"(pre.f[j].w));
      }
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r];
      if constexpr (ec == EPI_STORE) {
        if (p.alpha != 1.f) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] *= p.alpha;
        }
      } else if constexpr (ec == EPI_BIAS || ec == EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j][0] += bias4[j].x; v[j][1] += bias4[j].y; v[j][2] += bias4[j].z; v[j][3] += bias4[j].w; }
      } else if constexpr (ec == EPI_ACCUM || ec == EPI_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j][0] += pre.f[j].x; v[j][1] += pre.f[j].y; v[j][2] += pre.f[j].z; v[j][3] += pre.f[j].w; }
      } else if constexpr (ec == EPI_TABLE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j][0] += pre.f[j].x + bias4[j].x; v[j][1] += pre.f[j].y + bias4[j].y;
          v[j][2] += pre.f[j].z + bias4[j].z; v[j][3] += pre.f[j].w + bias4[j].w;
        }
      } else if constexpr (ec == EPI_QKV_ROPE) {
        // rotate interleaved pairs (transformer.model.py:182-190): 4 consecutive columns = 2 pairs of one head
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c0 = pre.f[j].x, c1 = pre.f[j].y, s0 = pre.f[j].z, s1 = pre.f[j].w;
          const float a0 = v[j][0] * c0 - v[j][1] * s0, a1 = v[j][0] * s0 + v[j][1] * c0;
          const float a2 = v[j][2] * c1 - v[j][3] * s1, a3 = v[j][2] * s1 + v[j][3] * c1;
          v[j][0] = a0; v[j][1] = a1; v[j][2] = a2; v[j][3] = a3;
        }
      }

      if constexpr (ec == EPI_SWIGLU_BWD) {
        // acc = dg; (da, db) of one block pair up: 32 consecutive columns of the [a|b] layout
        const unsigned int ro = rowoff(i, p.ldc, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float av[4], bv[4], da[4], db[4];
          auto r0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned int, pre.f[j].x), __builtin_bit_cast(unsigned int, pre.f[j].z), false, false);
          auto r1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned int, pre.f[j].y), __builtin_bit_cast(unsigned int, pre.f[j].w), false, false);
          unpack4(__builtin_bit_cast(float, (unsigned int)r0[0]), __builtin_bit_cast(float, (unsigned int)r1[0]), av);
          unpack4(__builtin_bit_cast(float, (unsigned int)r0[1]), __builtin_bit_cast(float, (unsigned int)r1[1]), bv);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-av[r]));
            da[r] = v[j][r] * bv[r] * sg * (1.f + av[r] * (1.f - sg));
            db[r] = v[j][r] * av[r] * sg;
          }
          const int oc = ((wn0 >> 4) + j) * 32 + ((fq & 1) << 4) + ((fq >> 1) << 3);
          if (want_amax && (FULL || (rowok && wn0 + 16 * j + 4 * fq < p.N))) {   // (this lane's own 4 columns of dg, before the pair swap)
            amx0 = fmaxf(amx0, fmaxf(fmaxf(fabsf(da[0]), fabsf(da[1])), fmaxf(fabsf(da[2]), fabsf(da[3]))));
            amx1 = fmaxf(amx1, fmaxf(fmaxf(fabsf(db[0]), fabsf(db[1])), fmaxf(fabsf(db[2]), fabsf(db[3]))));
          }
          store_pair(p.C, ro + (unsigned int)oc * 2u, rowok && oc < 2 * p.N, da, db);
        }
      } else if constexpr (ec == EPI_GELU) {
        const unsigned int ro = rowoff(i, p.ldc, 2), ro2 = rowoff(i, p.ldc2, 2);
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          float gx[4], gy[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gx[r] = 0.5f * v[jp][r] * (1.f + erff(v[jp][r] * 0.70710678118654752f));
            gy[r] = 0.5f * v[jp + 1][r] * (1.f + erff(v[jp + 1][r] * 0.70710678118654752f));
          }
          const int oc = c8 + jp * 16;
          store_pair(p.C, ro + (unsigned int)oc * 2u, rowok && oc < p.N, v[jp], v[jp + 1]);
          store_pair(p.C2, ro2 + (unsigned int)oc * 2u, rowok && oc < p.N, gx, gy);
        }
      } else if constexpr (ec == EPI_SWIGLU) {
        // blocks (0,1) and (2,3) are [16 a | 16 b] groups: C gets [a|b] as is, C2 the products g = silu(a) * b
        float g0[4], g1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          g0[r] = v[0][r] * __builtin_amdgcn_rcpf(1.f + __expf(-v[0][r])) * v[1][r];
          g1[r] = v[2][r] * __builtin_amdgcn_rcpf(1.f + __expf(-v[2][r])) * v[3][r];
        }
        if (want_amax && (FULL || rowok)) {   // (g0 / g1: this lane's 4 columns of blocks (0,1) / (2,3), before the pair swap)
          if (FULL || wn0 + 4 * fq < p.N) amx0 = fmaxf(amx0, fmaxf(fmaxf(fabsf(g0[0]), fabsf(g0[1])), fmaxf(fabsf(g0[2]), fabsf(g0[3]))));
          if (FULL || wn0 + 32 + 4 * fq < p.N) amx0 = fmaxf(amx0, fmaxf(fmaxf(fabsf(g1[0]), fabsf(g1[1])), fmaxf(fabsf(g1[2]), fabsf(g1[3]))));
        }
        const unsigned int ro = rowoff(i, p.ldc, 2), ro2 = rowoff(i, p.ldc2, 2);
        store_pair(p.C, ro + (unsigned int)c8 * 2u, rowok && c8 < p.N, v[0], v[1]);
        store_pair(p.C, ro + (unsigned int)(c8 + 32) * 2u, rowok && c8 + 32 < p.N, v[2], v[3]);
        const int gc = (wn0 >> 1) + ((fq & 1) << 4) + ((fq >> 1) << 3);
        store_pair(p.C2, ro2 + (unsigned int)gc * 2u, rowok && gc * 2 < p.N, g0, g1);
      } else if constexpr (ec == EPI_TABLE) {
        const unsigned int ro = rowoff(i, p.ldc, 4), ro2 = rowoff(i, p.ldc2, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) store_f32(p.C, ro + (unsigned int)(c4 + 16 * j) * 4u, rowok && c4 + 16 * j < p.N, v[j]);
        store_pair(p.C2, ro2 + (unsigned int)c8 * 2u, rowok && c8 < p.N, v[0], v[1]);
        store_pair(p.C2, ro2 + (unsigned int)(c8 + 32) * 2u, rowok && c8 + 32 < p.N, v[2], v[3]);
      } else {
        if (outf32) {
          const unsigned int ro = rowoff(i, p.ldc, 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) store_f32(p.C, ro + (unsigned int)(c4 + 16 * j) * 4u, rowok && c4 + 16 * j < p.N, v[j]);
        } else {
          const unsigned int ro = rowoff(i, p.ldc, 2);
          store_pair(p.C, ro + (unsigned int)c8 * 2u, rowok && c8 < p.N, v[0], v[1]);
          store_pair(p.C, ro + (unsigned int)(c8 + 32) * 2u, rowok && c8 + 32 < p.N, v[2], v[3]);
        }
      }
    };
    // software pipeline over the 8 (NRB) row blocks: the operands of block i+1 are in flight while block i is finished.  (Two
    // blocks of lookahead measured the same -- 31 K vs 33 K cycles per tile for the residual epilogue,
    // tools/micro/gemm8p_trace.hip: with every CU in its epilogue at once these loads are bandwidth-, not latency-bound
    // -- and cost 16 more registers.)  The memory clobbers keep each request ahead of the stores that follow it in
    // program order: the persistent kernel counts on "the stores of the last row block follow the last load".
    Pre pa, pb;
    request(std::integral_constant<int, 0>{}, pa);
    static_for<NRB / 2>([&](auto H) __attribute__((always_inline))  {
      constexpr int i = decltype(H)::value * 2;
      request(std::integral_constant<int, i + 1>{}, pb);
      asm volatile("" ::: "memory");
      finish(std::integral_constant<int, i>{}, pa);
      if constexpr (i + 2 < NRB) { request(std::integral_constant<int, i + 2>{}, pa); asm volatile("" ::: "memory"); }
      finish(std::integral_constant<int, i + 1>{}, pb);
    });
    if constexpr (ec == EPI_SWIGLU || ec == EPI_SWIGLU_BWD) {
      if (want_amax) {   // (the stored values are bf16: the amax of the rounded values is the rounded amax)
        amx0 = wave_max(amx0);
        if constexpr (ec == EPI_SWIGLU_BWD) amx1 = wave_max(amx1);
        if ((threadIdx.x & 63) == 0) {
          f8_amax_add(p.f8_amax_out, bf16_rounded(amx0));   // (no read-and-compare first: it would drain the tile's stores)
          if constexpr (ec == EPI_SWIGLU_BWD) f8_amax_add(p.f8_amax_out + 1, bf16_rounded(amx1));
        }
      }
    }
  };
  auto run2 = [&](auto EC) __attribute__((always_inline))  {
    if constexpr (MODE == 1) run(EC, std::true_type{});
    else if constexpr (MODE == 0) run(EC, std::false_type{});
    else { if (full) run(EC, std::true_type{}); else run(EC, std::false_type{}); }
  };
  if constexpr (ECF >= 0) { run2(std::integral_constant<int, ECF>{}); return; }
  switch (p.epi) {
    case EPI_STORE: run2(std::integral_constant<int, EPI_STORE>{}); break;
    case EPI_ACCUM: run2(std::integral_constant<int, EPI_ACCUM>{}); break;
    case EPI_BIAS: run2(std::integral_constant<int, EPI_BIAS>{}); break;
    case EPI_RESIDUAL: run2(std::integral_constant<int, EPI_RESIDUAL>{}); break;
    case EPI_QKV_ROPE:
      if (p.rope_pos) run2(std::integral_constant<int, 100>{}); else run2(std::integral_constant<int, EPI_QKV_ROPE>{});
      break;
    case EPI_SWIGLU: run2(std::integral_constant<int, EPI_SWIGLU>{}); break;
    case EPI_TABLE: run2(std::integral_constant<int, EPI_TABLE>{}); break;
    case EPI_GELU: run2(std::integral_constant<int, EPI_GELU>{}); break;
    case EPI_SWIGLU_BWD: run2(std::integral_constant<int, EPI_SWIGLU_BWD>{}); break;
    default: break;
  }
}

}  // namespace rsys
