// MFMA GEMM, gfx950.  128x128 output tile per 256-thread workgroup (4 waves as
// 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles), 128 bytes of K per LDS row,
// register-staged double-buffered LDS, XOR-swizzled so that both the
// ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads are
// bank-conflict free, XCD-aware tile order (cdna_hip_programming.md T1/T2/T10).
#include <utility>

#include "gemm.hpp"

namespace rsys {

constexpr int BM = 128, BN = 128;

template <typename CT> struct MmaT;
template <> struct MmaT<bf16> {
  static constexpr int EPC = 8;     // elements per 16-byte chunk
  static constexpr int BK = 64;     // K per tile (128 B per row)
  static constexpr int KSTEPS = 2;  // MFMA k-steps per tile (32 each)
  using Frag = bf16x8;
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<float> {
  static constexpr int EPC = 4;
  static constexpr int BK = 32;
  static constexpr int KSTEPS = 8;  // 4 each
  using Frag = float;
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// ---- staging registers: one 16-byte LDS chunk worth of source data
template <typename CT, bool F32SRC> struct Staged { uint4 v; };
template <> struct Staged<bf16, true> { float4 a, b; };

// Guarded 16-byte chunk load without a branch: an invalid chunk reads element 0 of the operand
// (always mapped) and is then zeroed by a select, so the loads of a tile issue back to back.
template <typename CT, bool F32SRC>
__device__ __forceinline__ Staged<CT, F32SRC> load_chunk(const void* base, long long off, bool valid) {
  Staged<CT, F32SRC> r;
  off = valid ? off : 0;
  if constexpr (is_bf16<CT>::value && F32SRC) {
    const float4* p = (const float4*)((const float*)base + off);
    float4 a = p[0], b = p[1];
    const float4 z = make_float4(0, 0, 0, 0);
    r.a = valid ? a : z; r.b = valid ? b : z;
  } else {
    uint4 v = *(const uint4*)((const CT*)base + off);
    r.v = valid ? v : make_uint4(0, 0, 0, 0);
  }
  return r;
}

template <typename CT, bool F32SRC>
__device__ __forceinline__ uint4 chunk_bits(const Staged<CT, F32SRC>& r) {
  if constexpr (is_bf16<CT>::value && F32SRC) {
    bf16x8 o;
    o[0] = (bf16)r.a.x; o[1] = (bf16)r.a.y; o[2] = (bf16)r.a.z; o[3] = (bf16)r.a.w;
    o[4] = (bf16)r.b.x; o[5] = (bf16)r.b.y; o[6] = (bf16)r.b.z; o[7] = (bf16)r.b.w;
    return *(uint4*)&o;
  } else {
    return r.v;
  }
}

// swizzle of the 32-byte block index of a K-major bf16 tile row (see header comment)
__device__ __forceinline__ int kmf(int k) { return (((k >> 3) & 1) << 2) | (k & 3); }

template <typename CT, bool KM>
__device__ __forceinline__ int lds_chunk_offset(int c) {
  if constexpr (!KM) {
    int row = c >> 3, ch = c & 7;
    return row * 128 + ((ch ^ (row & 7)) << 4);
  } else if constexpr (is_bf16<CT>::value) {
    int krow = c >> 4, ch = c & 15;
    return krow * 256 + ((((ch >> 1) ^ kmf(krow))) << 5) + ((ch & 1) << 4);
  } else {
    int krow = c >> 5, ch = c & 31;
    return krow * 512 + (ch << 4);
  }
}

template <typename CT, bool KM>
__device__ __forceinline__ typename MmaT<CT>::Frag load_frag(const unsigned char* tile, int r0, int s, int l) {
  const int g = l >> 4, i = l & 15;
  if constexpr (is_bf16<CT>::value) {
    if constexpr (!KM) {
      int row = r0 + i;
      int chunk = 4 * s + g;
      return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
      int q = i >> 2, pp = i & 3;
      int krow = 32 * s + 8 * g + q;
      int bm = r0 >> 4;
      int f = ((g & 1) << 2) | q;
      const unsigned char* a = tile + krow * 256 + ((bm ^ f) << 5) + pp * 8;
      bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a));
      bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a + 4 * 256));
      return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  } else {
    if constexpr (!KM) {
      int row = r0 + i;
      return *(const float*)(tile + row * 128 + ((s ^ (row & 7)) << 4) + g * 4);
    } else {
      int krow = 4 * s + g;
      return *(const float*)(tile + krow * 512 + (r0 + i) * 4);
    }
  }
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: accumulator tiles must be indexed by constants or hipcc moves them to scratch
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <typename CT>
__device__ __forceinline__ void store_c(const GemmParams& p, void* C, long long ld, bool f32, long long row, int col, float v) {
  if (f32) ((float*)C)[row * ld + col] = v;
  else ((CT*)C)[row * ld + col] = from_f32<CT>(v);
}

template <typename CT, bool AF32, bool BF32, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  using MT = MmaT<CT>;
  constexpr int EPC = MT::EPC, BK = MT::BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD remap: blocks that share an XCD (bid % 8) take consecutive tiles
    int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = blockIdx.y * per;
  const int kt1 = min(ktiles, kt0 + per);
  if (kt0 >= kt1) return;

  const int t = threadIdx.x, l = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  // LDS: [buf 0: A 16K | B 16K][buf 1: A 16K | B 16K]

  Staged<CT, AF32> ra[4];
  Staged<CT, BF32> rb[4];

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = t + 256 * j;
      if constexpr (!AKM) {
        int row = c >> 3, ch = c & 7;
        long long gr = min(m0 + row, p.M - 1);
        int kk = k0 + ch * EPC;
        ra[j] = load_chunk<CT, AF32>(p.A, gr * p.lda + kk, kk < p.K);
      } else {
        constexpr int CPR = 128 / EPC;
        int krow = c / CPR, ch = c % CPR;
        long long kk = k0 + krow;
        int mm = m0 + ch * EPC;
        ra[j] = load_chunk<CT, AF32>(p.A, kk * p.lda + mm, kk < p.K && mm < p.M);
      }
      if constexpr (!BKM) {
        int row = c >> 3, ch = c & 7;
        long long gr = min(n0 + row, p.N - 1);
        int kk = k0 + ch * EPC;
        rb[j] = load_chunk<CT, BF32>(p.B, gr * p.ldb + kk, kk < p.K);
      } else {
        constexpr int CPR = 128 / EPC;
        int krow = c / CPR, ch = c % CPR;
        long long kk = k0 + krow;
        int nn = n0 + ch * EPC;
        rb[j] = load_chunk<CT, BF32>(p.B, kk * p.ldb + nn, kk < p.K && nn < p.N);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = t + 256 * j;
      *(uint4*)(smem + buf * 32768 + lds_chunk_offset<CT, AKM>(c)) = chunk_bits<CT, AF32>(ra[j]);
      *(uint4*)(smem + buf * 32768 + 16384 + lds_chunk_offset<CT, BKM>(c)) = chunk_bits<CT, BF32>(rb[j]);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gload(kt0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < MT::KSTEPS; ++s) {
      typename MT::Frag a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = load_frag<CT, AKM>(smem + cur * 32768, wr * 64 + i * 16, s, l);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag<CT, BKM>(smem + cur * 32768 + 16384, wc * 64 + j * 16, s, l);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MT::mma(a[i], b[j], acc[i][j]);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ------------------------------------------------------------------ epilogue
  // acc[i][j]: rows m0 + wr*64 + i*16 + 4*(l>>4) + r (r = 0..3), column n0 + wc*64 + j*16 + (l&15)
  const int fq = l >> 4, fr = l & 15;
  const bool cf32 = p.c_f32 != 0;
  const long long rbase = m0 + wr * 64 + 4 * fq;
  const int cbase = n0 + wc * 64 + fr;
  auto tiles = [&](auto&& fn) {
    static_for<4>([&](auto i) { static_for<4>([&](auto j) { fn(i, j, rbase + i * 16, cbase + j * 16); }); });
  };
  switch (p.epi) {
    case EPI_STORE:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) store_c<CT>(p, p.C, p.ldc, cf32, row0 + r, col, v[r] * p.alpha);
        }
      });
      break;
    case EPI_ACCUM:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) ((float*)p.C)[(row0 + r) * p.ldc + col] += v[r];
        }
      });
      break;
    case EPI_ATOMIC:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) atomicAdd(&((float*)p.C)[(row0 + r) * p.ldc + col], v[r]);
        }
      });
      break;
    case EPI_BIAS:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
          const float bv = p.bias[col];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) store_c<CT>(p, p.C, p.ldc, cf32, row0 + r, col, v[r] + bv);
        }
      });
      break;
    case EPI_RESIDUAL:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) ((float*)p.C)[(row0 + r) * p.ldc + col] = p.resid[(row0 + r) * p.ldr + col] + v[r];
        }
      });
      break;
    case EPI_TABLE:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
          const float bv = p.bias[col];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) {
              const long long row = row0 + r;
              const float f = v[r] + p.E[row * p.ldc + col] + bv;
              ((float*)p.C)[row * p.ldc + col] = f;
              ((CT*)p.C2)[row * p.ldc2 + col] = from_f32<CT>(f);
            }
        }
      });
      break;
    case EPI_GELU:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        if (col < p.N) {
          const float bv = p.bias[col];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) {
              const long long row = row0 + r;
              const float z = v[r] + bv;
              ((CT*)p.C)[row * p.ldc + col] = from_f32<CT>(z);
              ((CT*)p.C2)[row * p.ldc2 + col] = from_f32<CT>(0.5f * z * (1.f + erff(z * 0.70710678118654752f)));
            }
        }
      });
      break;
    case EPI_SWIGLU:
      // columns interleaved in 16-wide blocks [a | b]: tiles (i, 2jj) and (i, 2jj+1) of one lane pair up
      static_for<4>([&](auto i) {
        static_for<2>([&](auto jj) {
          const f32x4 va = acc[i][2 * jj], vb = acc[i][2 * jj + 1];
          const long long row0 = rbase + i * 16;
          const int col_a = cbase + jj * 32, col_b = col_a + 16;
          const int gcol = ((n0 + wc * 64) >> 1) + jj * 16 + fr;
          if (col_b < p.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const long long row = row0 + r;
              if (row < p.M) {
                const float a = va[r], b = vb[r];
                ((CT*)p.C)[row * p.ldc + col_a] = from_f32<CT>(a);
                ((CT*)p.C)[row * p.ldc + col_b] = from_f32<CT>(b);
                const float sg = 1.f / (1.f + __expf(-a));
                ((CT*)p.C2)[row * p.ldc2 + gcol] = from_f32<CT>(a * sg * b);
              }
            }
          }
        });
      });
      break;
    case EPI_QKV_ROPE:
    case EPI_STORE_HEADS_T:
      tiles([&](auto i, auto j, long long row0, int col) {
        const f32x4 v = acc[i][j];
        // region / head bookkeeping (16 columns of an MFMA tile never straddle a head: hd % 16 == 0)
        const bool cok = col < p.N;
        const int ccol = cok ? col : 0;
        int region = 0, cc = ccol;
        void* XT = p.C2;
        int heads = p.N / p.hd;
        if (p.epi == EPI_QKV_ROPE) {
          if (ccol < p.n_q) { region = 0; cc = ccol; XT = p.qT; heads = p.n_q / p.hd; }
          else if (ccol < p.n_q + p.n_k) { region = 1; cc = ccol - p.n_q; XT = p.kT; heads = p.n_k / p.hd; }
          else { region = 2; cc = ccol - p.n_q - p.n_k; XT = p.vT; heads = (p.N - p.n_q - p.n_k) / p.hd; }
        }
        const int head = cc / p.hd, d = cc % p.hd;
        const bool rot = (p.epi == EPI_QKV_ROPE) && region < 2;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = v[r];
          const float partner = __shfl_xor(x, 1, 64);
          if (rot) {
            const long long row = min(row0 + r, (long long)p.M - 1);
            const int pos = p.rope_pos ? p.rope_pos[row] : (int)(row % p.T);
            const float c = p.rope_cos[pos * (p.hd >> 1) + (d >> 1)];
            const float sn = p.rope_sin[pos * (p.hd >> 1) + (d >> 1)];
            x = (l & 1) ? (partner * sn + x * c) : (x * c - partner * sn);
          }
          o[r] = x;
        }
        if (cok) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) ((CT*)p.C)[(row0 + r) * p.ldc + col] = from_f32<CT>(o[r]);
          if (XT != nullptr && row0 < p.M) {
            const long long b = row0 / p.T;
            const int t0 = (int)(row0 % p.T);
            CT* dst = (CT*)XT + ((b * heads + head) * p.hd + d) * (long long)p.T + t0;
            if (row0 + 3 < p.M) {
              if constexpr (is_bf16<CT>::value) {
                bf16x4 pk; pk[0] = (bf16)o[0]; pk[1] = (bf16)o[1]; pk[2] = (bf16)o[2]; pk[3] = (bf16)o[3];
                *(bf16x4*)dst = pk;
              } else {
                *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (row0 + r < p.M) dst[r] = from_f32<CT>(o[r]);
            }
          }
        }
      });
      break;
    default: break;
  }
}

template <typename CT, bool AF32, bool BF32, bool AKM, bool BKM>
static int launch_one(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  dim3 grid(tiles, p.splitk, 1);
  hipLaunchKernelGGL((gemm_kernel<CT, AF32, BF32, AKM, BKM>), grid, dim3(256), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename CT>
int launch_gemm(const GemmParams& p0, bool a_f32, bool b_f32, bool a_km, bool b_km, hipStream_t s) {
  GemmParams p = p0;
  constexpr int EPC = MmaT<CT>::EPC;
  if (p.splitk < 1) p.splitk = 1;
  ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
  ARG_CHECK(p.splitk == 1 || p.epi == EPI_ATOMIC, "gemm: split-K needs the atomic epilogue");
  const int ea = (is_bf16<CT>::value && a_f32) ? 4 : EPC;
  const int eb = (is_bf16<CT>::value && b_f32) ? 4 : EPC;
  ARG_CHECK(p.lda % ea == 0 && p.ldb % eb == 0, "gemm: leading dimensions must keep 16-byte row alignment");
  ARG_CHECK(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "gemm: operand base must be 16-byte aligned");
  if (!a_km) ARG_CHECK(p.lda >= ((p.K + EPC - 1) / EPC) * EPC, "gemm: row-major A rows must be padded to a whole chunk");
  else ARG_CHECK(p.lda >= ((p.M + EPC - 1) / EPC) * EPC, "gemm: K-major A rows must be padded to a whole chunk");
  if (!b_km) ARG_CHECK(p.ldb >= ((p.K + EPC - 1) / EPC) * EPC, "gemm: row-major B rows must be padded to a whole chunk");
  else ARG_CHECK(p.ldb >= ((p.N + EPC - 1) / EPC) * EPC, "gemm: K-major B rows must be padded to a whole chunk");
  if (p.epi == EPI_QKV_ROPE || p.epi == EPI_STORE_HEADS_T)
    ARG_CHECK(p.hd % 16 == 0 && p.T % 4 == 0, "gemm: head epilogues need hd % 16 == 0 and T % 4 == 0");
  if (p.epi == EPI_SWIGLU) ARG_CHECK(p.N % 32 == 0, "gemm: swiglu epilogue needs N % 32 == 0");
  if constexpr (!is_bf16<CT>::value) { a_f32 = false; b_f32 = false; }
  if (!a_km && !b_km) {
    if (!a_f32 && !b_f32) return launch_one<CT, false, false, false, false>(p, s);
    if (a_f32 && !b_f32) return launch_one<CT, true, false, false, false>(p, s);
  } else if (!a_km && b_km) {
    if (!a_f32 && !b_f32) return launch_one<CT, false, false, false, true>(p, s);
    if (a_f32 && !b_f32) return launch_one<CT, true, false, false, true>(p, s);
  } else if (a_km && b_km) {
    if (!a_f32 && !b_f32) return launch_one<CT, false, false, true, true>(p, s);
    if (a_f32 && !b_f32) return launch_one<CT, true, false, true, true>(p, s);
  }
  set_error("gemm: operand layout/type combination not instantiated");
  return RSYS_ERR_ARG;
}

template int launch_gemm<bf16>(const GemmParams&, bool, bool, bool, bool, hipStream_t);
template int launch_gemm<float>(const GemmParams&, bool, bool, bool, bool, hipStream_t);

}  // namespace rsys
