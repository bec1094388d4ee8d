// MFMA GEMM, gfx950.  128x128 output tile per 256-thread workgroup (4 waves as
// 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles), 128 bytes of K per LDS row,
// register-staged double-buffered LDS, XOR-swizzled so that both the
// ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads are
// bank-conflict free, XCD-aware tile order (cdna_hip_programming.md T1/T2/T10).
#include <cstdlib>
#include <utility>

#include "gemm.hpp"
#include "gemm_epi.hpp"

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

// ---- staging registers: one 16-byte LDS chunk worth of source data (+ its validity mask)
template <typename CT, bool F32SRC> struct Staged { uint4 v; unsigned int msk; };
template <> struct Staged<bf16, true> { uint4 a, b; unsigned int msk; };

// 16-byte chunk load from (uniform base + 32-bit byte offset).  A guarded (invalid) chunk is redirected to
// element 0 of the operand (always mapped) by the caller and zeroed with an AND mask when it is written to
// LDS (chunk_bits), i.e. after the MFMA phase the load overlaps with; the loaded bits are always consumed, so
// hipcc cannot sink the load into a conditional block (which would turn its counted vmcnt waits into vmcnt(0)).
template <typename CT, bool F32SRC>
__device__ __forceinline__ Staged<CT, F32SRC> load_chunk_at(const char* base, unsigned int voff, bool valid) {
  Staged<CT, F32SRC> r;
  r.msk = valid ? 0xFFFFFFFFu : 0u;
  if constexpr (is_bf16<CT>::value && F32SRC) {
    const uint4* p = (const uint4*)(base + voff);
    r.a = p[0]; r.b = p[1];
  } else {
    r.v = *(const uint4*)(base + voff);
  }
  return r;
}

template <typename CT, bool F32SRC, bool FULL>
__device__ __forceinline__ uint4 chunk_bits(const Staged<CT, F32SRC>& r) {
  const unsigned int m = r.msk;
  if constexpr (is_bf16<CT>::value && F32SRC) {
    uint4 a = r.a, b = r.b;
    if constexpr (!FULL) { a.x &= m; a.y &= m; a.z &= m; a.w &= m; b.x &= m; b.y &= m; b.z &= m; b.w &= m; }
    const float4 fa = *(float4*)&a, fb = *(float4*)&b;
    bf16x8 o;
    o[0] = (bf16)fa.x; o[1] = (bf16)fa.y; o[2] = (bf16)fa.z; o[3] = (bf16)fa.w;
    o[4] = (bf16)fb.x; o[5] = (bf16)fb.y; o[6] = (bf16)fb.z; o[7] = (bf16)fb.w;
    return *(uint4*)&o;
  } else {
    uint4 v = r.v;
    if constexpr (!FULL) { v.x &= m; v.y &= m; v.z &= m; v.w &= m; }
    return v;
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

template <typename CT>
__device__ __forceinline__ void store_c(const GemmParams& p, void* C, long long ld, bool f32, long long row, int col, float v) {
  if (f32) ((float*)C)[row * ld + col] = v;
  else ((CT*)C)[row * ld + col] = from_f32<CT>(v);
}

template <typename CT, bool AF32, bool BF32, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  using MT = MmaT<CT>;
  constexpr int EPC = MT::EPC, BK = MT::BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];   // [buf 0: A 16K | B 16K][buf 1: A 16K | B 16K]

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  constexpr int ESA = (is_bf16<CT>::value && AF32) ? 4 : (int)sizeof(CT);   // bytes per source element
  constexpr int ESB = (is_bf16<CT>::value && BF32) ? 4 : (int)sizeof(CT);

  // ---- state of the current work item (an output tile and a K range)
  int cm0 = 0, cn0 = 0, kt0 = 0, kt1 = 0, cur_split = 0;
  // per-thread staging geometry: 4 A chunks + 4 B chunks of 16 bytes per tile.  Global address of a chunk =
  // uniform 64-bit tile base (SGPRs) + 32-bit per-thread byte offset (VGPR): the steady-state loads need no
  // vector address arithmetic.
  unsigned int aoff[4], boff[4];       // byte offset of the chunk relative to the block base
  int arow[4], brow[4];                // K-major: k row inside the tile; row-major: chunk's first k inside the tile
  bool aval[4], bval[4];               // K-major: the chunk's m/n range is inside the matrix
  int ldsa[4], ldsb[4];
  const char* Ablk = nullptr; const char* Bblk = nullptr;
  bool full = true;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ldsa[j] = lds_chunk_offset<CT, AKM>(t + 256 * j);
    ldsb[j] = 16384 + lds_chunk_offset<CT, BKM>(t + 256 * j);
  }

  // Work item -> (output tile, K split), XCD-aware (blocks are dealt round-robin over the 8 XCDs, each with a
  // private L2): without split-K, blocks of one XCD take consecutive tiles (they share the A rows); with split-K
  // (weight gradients: K = tokens) all tiles of one K split run on ONE XCD, so the K-major operand rows of that
  // split are fetched from the fabric once instead of once per XCD.  Returns false for an item with no work.
  auto setup = [&](int tile, int split) -> bool {
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    cm0 = m0; cn0 = n0;
    if (p.m_dev != nullptr && m0 >= *p.m_dev) return false;
    const int ktiles = ((p.k_dev != nullptr ? min(p.K, *p.k_dev) : p.K) + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    kt0 = split * per;
    kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = t + 256 * j;
      if constexpr (!AKM) {
        const int row = c >> 3, ch = c & 7;
        const int gr = min(m0 + row, p.M - 1) - m0;
        aoff[j] = (unsigned int)(((long long)gr * p.lda + ch * EPC) * ESA);
        arow[j] = ch * EPC; aval[j] = true;
      } else {
        constexpr int CPR = 128 / EPC;
        const int krow = c / CPR, ch = c % CPR;
        aoff[j] = (unsigned int)(((long long)krow * p.lda + ch * EPC) * ESA);
        arow[j] = krow; aval[j] = (m0 + ch * EPC) < p.M;
      }
      if constexpr (!BKM) {
        const int row = c >> 3, ch = c & 7;
        const int gr = min(n0 + row, p.N - 1) - n0;
        boff[j] = (unsigned int)(((long long)gr * p.ldb + ch * EPC) * ESB);
        brow[j] = ch * EPC; bval[j] = true;
      } else {
        constexpr int CPR = 128 / EPC;
        const int krow = c / CPR, ch = c % CPR;
        boff[j] = (unsigned int)(((long long)krow * p.ldb + ch * EPC) * ESB);
        brow[j] = krow; bval[j] = (n0 + ch * EPC) < p.N;
      }
    }
    Ablk = (const char*)p.A + (AKM ? (long long)m0 : (long long)m0 * p.lda) * ESA;
    Bblk = (const char*)p.B + (BKM ? (long long)n0 : (long long)n0 * p.ldb) * ESB;
    full = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (p.K % BK == 0);   // no chunk of any tile needs a guard
    return true;
  };

  f32x4 acc[4][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  auto compute = [&](int cur) {
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
  };

  // Software pipeline, two register sets: the loads of tile k+2 are issued before the MFMAs of tile k while
  // tile k+1 is in flight.  The steady-state loop has no branch around a memory operation, so hipcc keeps exact
  // vmcnt counts (the LDS store of tile k+1 waits for ITS loads only); the last one..three phases are peeled so
  // that no block waits for loads it does not consume.
  Staged<CT, AF32> ra0[4], ra1[4];
  Staged<CT, BF32> rb0[4], rb1[4];
  auto gload = [&](Staged<CT, AF32>(&ra)[4], Staged<CT, BF32>(&rb)[4], int kt, auto FULLC) {
    constexpr bool FULL = decltype(FULLC)::value;
    const int k0 = kt * BK;
    const char* At = Ablk + (AKM ? (long long)k0 * p.lda : (long long)k0) * ESA;   // uniform
    const char* Bt = Bblk + (BKM ? (long long)k0 * p.ldb : (long long)k0) * ESB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (FULL) {
        ra[j] = load_chunk_at<CT, AF32>(At, aoff[j], true);
        rb[j] = load_chunk_at<CT, BF32>(Bt, boff[j], true);
      } else {
        const bool va = aval[j] && (k0 + arow[j] < p.K);
        const bool vb = bval[j] && (k0 + brow[j] < p.K);
        ra[j] = load_chunk_at<CT, AF32>(va ? At : (const char*)p.A, va ? aoff[j] : 0u, va);
        rb[j] = load_chunk_at<CT, BF32>(vb ? Bt : (const char*)p.B, vb ? boff[j] : 0u, vb);
      }
    }
  };
  auto lstore = [&](const Staged<CT, AF32>(&ra)[4], const Staged<CT, BF32>(&rb)[4], int buf, auto FULLC) {
    constexpr bool FULL = decltype(FULLC)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(uint4*)(smem + buf * 32768 + ldsa[j]) = chunk_bits<CT, AF32, FULL>(ra[j]);
      *(uint4*)(smem + buf * 32768 + ldsb[j]) = chunk_bits<CT, BF32, FULL>(rb[j]);
    }
  };
  // first two tiles of the current item into the register sets (issued as early as possible: in the persistent
  // kernel BEFORE the epilogue of the previous tile, which hides their HBM latency)
  auto prologue_loads = [&](auto FC) {
    gload(ra0, rb0, kt0, FC);
    if (kt1 - kt0 > 1) gload(ra1, rb1, kt0 + 1, FC);
  };
  // Phase p computes tile p from LDS buffer p&1, after issuing the loads of tile p+2 and before storing tile p+1.
  auto mainloop = [&](auto FC) {
    const int n = kt1 - kt0;
    lstore(ra0, rb0, 0, FC);
    __syncthreads();
    int pz = 0;
    for (; pz + 3 < n; pz += 2) {
      gload(ra0, rb0, kt0 + pz + 2, FC);     // even phase: buffer 0 = tile pz, set 1 = tile pz+1 (in flight)
      compute(0);
      lstore(ra1, rb1, 1, FC);
      __syncthreads();
      gload(ra1, rb1, kt0 + pz + 3, FC);     // odd phase: buffer 1 = tile pz+1, set 0 = tile pz+2 (in flight)
      compute(1);
      lstore(ra0, rb0, 0, FC);
      __syncthreads();
    }
    const int rem = n - pz;              // 1, 2 or 3 phases left
    if (rem == 3) {
      gload(ra0, rb0, kt0 + pz + 2, FC);
      compute(0);
      lstore(ra1, rb1, 1, FC);
      __syncthreads();
      compute(1);
      lstore(ra0, rb0, 0, FC);
      __syncthreads();
      compute(0);
    } else if (rem == 2) {
      compute(0);
      lstore(ra1, rb1, 1, FC);
      __syncthreads();
      compute(1);
    } else {
      compute(0);
    }
    __syncthreads();
  };

  auto epilogue = [&](const int m0, const int n0) {
  // ------------------------------------------------------------------ epilogue
  // The accumulators are staged through LDS (the operand tiles are dead by now) so that every global
  // access of the epilogue is a full 128..512-byte row segment with 8..16 bytes per lane, instead of the
  // MFMA layout's 2..4-byte column-strided stores.  acc[i][j][r]: row wr*64 + i*16 + 4*(l>>4) + r,
  // column wc*64 + j*16 + (l&15) of the 128x128 tile; the tile leaves in two 64-row halves.
  const int fq = l >> 4, fr = l & 15;
  constexpr int CS_LD = 132;   // f32 row-major staging tile [64][132]
  constexpr int CT_LD = 68;    // f32 staging tile of the SwiGLU product [64][68]
  float* Cs = (float*)smem;
  const bool cf32 = p.c_f32 != 0;   // (the main loop ends with a barrier: the operand tiles are dead)
  if (p.epi == 99) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = acc[3][3][3]; return; }   // timing experiment: no epilogue

  if (p.epi == EPI_ATOMIC) {
    // split-K weight gradients: 64 lanes add 256 contiguous bytes per wave-instruction (full-rate shape)
    for (int half = 0; half < 2; ++half) {
      if (wr == half) {
        static_for<4>([&](auto i) { static_for<4>([&](auto j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[(i * 16 + 4 * fq + r) * CS_LD + wc * 64 + j * 16 + fr] = acc[i][j][r];
        }); });
      }
      __syncthreads();
      const int col = n0 + (t & 127);
      for (int it = 0; it < 32; ++it) {
        const int row_l = it * 2 + (t >> 7);
        const long long row = m0 + half * 64 + row_l;
        if (row < p.M && col < p.N) {
          const float v = p.alpha * Cs[row_l * CS_LD + (t & 127)];
          if (p.slab != nullptr) p.slab[((long long)cur_split * p.M + row) * p.N + col] = v;   // deterministic mode: summed in split order afterwards
          else atomicAdd(&((float*)p.C)[row * p.ldc + col], v);
        }
      }
      __syncthreads();
    }
    return;
  }

  // copy-out of a staged [64][*] f32 tile: every lane handles W consecutive columns of one row, so that the
  // global accesses are 16 bytes per lane (bf16: W = 8, f32: W = 4; 8-byte stores run at ~half the rate)
  const bool outf32 = cf32 || p.epi == EPI_ACCUM || p.epi == EPI_RESIDUAL;
  auto copy_primary = [&](int half, auto WC) {
    constexpr int W = decltype(WC)::value;
    constexpr int CPR = 128 / W, ITEMS = 64 * CPR / 256;
#pragma unroll 2
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = t + 256 * it;
      const int row_l = idx / CPR, cw = (idx % CPR) * W;
      const long long row = m0 + half * 64 + row_l;
      const int col = n0 + cw;
      if (row >= p.M || col >= p.N) continue;
      const int nv = min(W, p.N - col);
      float v[W];
#pragma unroll
      for (int k = 0; k < W; k += 4) {
        const float4 q = *(const float4*)&Cs[row_l * CS_LD + cw + k];
        v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
      }
      epi_item<CT, W>(p, row, col, v, nv, outf32);
    }
  };
  const bool wide = is_bf16<CT>::value && (!outf32 || p.epi == EPI_TABLE);   // bf16 outputs (incl. the bf16 copy of EPI_TABLE): 8 columns per lane

  for (int half = 0; half < 2; ++half) {
    // ---- primary tile, row-major
    if (wr == half) {
      static_for<4>([&](auto i) { static_for<4>([&](auto j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(i * 16 + 4 * fq + r) * CS_LD + wc * 64 + j * 16 + fr] = acc[i][j][r];
      }); });
    }
    __syncthreads();
    if (wide) copy_primary(half, std::integral_constant<int, 8>{});
    else copy_primary(half, std::integral_constant<int, 4>{});
    __syncthreads();

    // ---- SwiGLU product g = silu(a)*b: columns are interleaved in 16-wide blocks [a | b], lane-local pairs
    if (p.epi == EPI_SWIGLU) {
      if (wr == half) {
        static_for<4>([&](auto i) { static_for<2>([&](auto jj) {
          const f32x4 va = acc[i][2 * jj], vb = acc[i][2 * jj + 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = va[r], b = vb[r];
            Cs[(i * 16 + 4 * fq + r) * CT_LD + wc * 32 + jj * 16 + fr] = a / (1.f + __expf(-a)) * b;
          }
        }); });
      }
      __syncthreads();
      constexpr int GW = is_bf16<CT>::value ? 8 : 4, GCPR = 64 / GW;
      for (int it = 0; it < 64 * GCPR / 256; ++it) {
        const int idx = t + 256 * it;
        const int row_l = idx / GCPR, cw = (idx % GCPR) * GW;
        const long long row = m0 + half * 64 + row_l;
        const int gcol = (n0 >> 1) + cw;
        if (row >= p.M || gcol * 2 >= p.N) continue;
        float v[GW];
#pragma unroll
        for (int k = 0; k < GW; k += 4) {
          const float4 q = *(const float4*)&Cs[row_l * CT_LD + cw + k];
          v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
        store_vec<CT, GW>((CT*)p.C2 + row * p.ldc2 + gcol, v, GW);
      }
      __syncthreads();
    }
  }
  };   // epilogue

  {
    int tile, split;
    if (p.splitk == 1) {
      const int bid = blockIdx.x;
      const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
      split = 0;
    } else {
      // launcher guarantees splitk % 8 == 0: XCD x owns splits x, x+8, ...; inside an XCD tiles vary fastest
      const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
      tile = local % ntiles;
      split = xcd + 8 * (local / ntiles);
    }
    cur_split = split;
    if (!setup(tile, split)) return;
    if (full) { prologue_loads(std::true_type{}); mainloop(std::true_type{}); }
    else { prologue_loads(std::false_type{}); mainloop(std::false_type{}); }
    epilogue(cm0, cn0);
  }
}

// C[row][col] += sum over splits (in index order) of slab[split][row][col]: the deterministic end of a split-K product
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N, float* C, long long ldc) {
  const long long n4 = (long long)M * N / 4, stride = (long long)M * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 acc = *(const float4*)(slab + 4 * i);
    for (int sp = 1; sp < splits; ++sp) {
      const float4 v = *(const float4*)(slab + sp * stride + 4 * i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const long long e = 4 * i, row = e / N; const int col = (int)(e % N);
    float4* c = (float4*)(C + row * ldc + col);
    float4 o = *c; o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w; *c = o;
  }
}
int gemm_slab_begin(const GemmParams& p, hipStream_t s) {
  if (p.slab == nullptr) return RSYS_OK;
  const long long need = (long long)p.splitk * p.M * p.N;
  ARG_CHECK(need <= p.slab_floats, "gemm: split-K slab too small");
  ARG_CHECK(p.N % 4 == 0 && p.ldc % 4 == 0, "gemm: deterministic split-K needs N % 4 == 0");
  HIP_CHECK(hipMemsetAsync(p.slab, 0, (size_t)need * 4, s));   // (splits without work, rows beyond a device-side limit)
  return RSYS_OK;
}
int gemm_slab_end(const GemmParams& p, hipStream_t s) {
  if (p.slab == nullptr) return RSYS_OK;
  const long long n4 = (long long)p.M * p.N / 4;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 4096)), dim3(256), 0, s, p.slab, p.splitk, p.M, p.N,
                     (float*)p.C, p.ldc);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename CT, bool AF32, bool BF32, bool AKM, bool BKM>
static int launch_one(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const bool slab = p.slab != nullptr && p.epi == EPI_ATOMIC && p.splitk > 1;
  GemmParams q = p;
  if (!slab) q.slab = nullptr;
  { const int rc_ = gemm_slab_begin(q, s); if (rc_ != RSYS_OK) return rc_; }
  hipLaunchKernelGGL((gemm_kernel<CT, AF32, BF32, AKM, BKM>), dim3(tiles * p.splitk), dim3(256), 0, s, q);
  HIP_CHECK(hipGetLastError());
  return gemm_slab_end(q, s);
}

// which kernel a row-major bf16 problem goes to: 0 = 128x128 register-staged (this file), 2 = 256x256 LDS-DMA (gemm8c.hip for the
// epilogue classes it has kernels for, else gemm8p.hip).  RSYS_GEMM_KERNEL=1 / 2 forces one of them where it is eligible (tests,
// A/B timing).  (The 256x128 two-per-CU sibling of rounds 1-3 left the library: slower than gemm8c on every shape of the step,
// profiles/r4_gemm4w_two_per_cu_stagger.log; its source is kept under tools/micro/ for measurements.)
static int pick_rowmajor_kernel(const GemmParams& p) {
  const int hint = sw().gemm_kernel;
  if (hint == 1) return 0;
  const bool e8 = gemm8p_eligible(p);
  if (hint == 2) return e8 ? 2 : 0;
  const int Me = (p.m_dev != nullptr && p.m_expect > 0) ? std::min(p.M, p.m_expect) : p.M;   // rows the launch is expected to compute
  const long long t256 = (long long)((Me + 255) / 256) * ((p.N + 255) / 256);
  if (e8 && t256 >= 128) return 2;   // at least half the CUs get a 256x256 tile
  return 0;
}

// K-major bf16 operands with the atomic epilogue (weight gradients): the LDS-DMA pipeline unless RSYS_GEMM_KERNEL_TN=1
static bool use_8p_tn(const GemmParams& p) {
  const int e = sw().gemm_kernel_tn;
  if (e == 1) return false;
  if (!gemm8p_tn_eligible(p)) return false;
  const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  // measured (tools/dbg/dw_small_outputs.py, round 4, after the kernel's LDS-DMA stopped being drained every phase): against the
  // 128x128 kernel it is 25-40 % faster from 16 output tiles of 256 x 256 on (1024 x 1024: 926 against 744 TFLOP/s; 2048 x 1024: 964
  // against 688), level at 8-12 tiles and slower at 4 (too few workgroups per K split).  The step's per-layer gradients at cfg-3
  // (4..22 tiles) go through the grouped launch instead (model.hip); this rule serves the shapes that do not (cfg-4, the production shape).
  return e == 2 || t256 >= 16;
}

// K-major bf16 operands, fp32 output stored or accumulated (no split-K): the tied head's table gradient (10^5 x D outputs, K = the live
// selected rows).  On the LDS-DMA pipeline when the output fills the chip; RSYS_GEMM_KERNEL_TN=1 keeps the 128x128 kernel (A/B).
static bool use_8p_tn_store(const GemmParams& p) {
  const int e = sw().gemm_kernel_tn;
  if (e == 1) return false;
  if (!gemm8p_tn_store_eligible(p)) return false;
  return e == 2 || (long long)((p.M + 255) / 256) * ((p.N + 255) / 256) >= 128;
}

// row-major bf16 operands with the atomic epilogue: the LDS-DMA split-K form when the output is large enough for it
static bool use_8p_nt_splitk(const GemmParams& p) {
  if (p.epi != EPI_ATOMIC || !gemm8p_nt_splitk_eligible(p)) return false;
  if (sw().gemm_kernel_nt_splitk == 2 || (p.flags & 128)) return true;   // 2: force; flags bit 7: the caller wants the 256x256 split-K form for a few tiles (table_forward's tail)
  // (RSYS_GEMM_KERNEL_NT_SPLITK=2: tools/ab_dw_rowmajor.py: the trunk's weight-gradient shapes on K-contiguous copies)
  // from 32 output tiles on; from 8 when K is long enough that the K splits fill the chip by themselves (cfg-2's metadata-projection
  // gradient: 256 x 6148 outputs = 25 tiles, K = 100 K: 0.59 -> 0.33 ms against the 128x128 kernel, profiles/r6b_*)
  const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  return t256 >= 32 || (t256 >= 8 && (long long)p.K * t256 >= 32LL * 16384);
}

// row-major A x K-major B with the atomic epilogue (the tied head's dEw = dlogits . F: a few hundred live rows, K = the vocabulary): the LDS-DMA
// mixed-layout form when K is long; RSYS_GEMM_KERNEL_MIX: 1 = that rule, 0 = never (the 128x128 kernel), 2 = wherever eligible
static bool use_8p_mix(const GemmParams& p) {
  const int mode = sw().gemm_kernel_mix;
  if (mode == 0 || sw().gemm_kernel == 1) return false;
  GemmParams q = p; q.K = p.K & ~63;
  if (!gemm8p_mix_eligible(q)) return false;
  // measured in the step (profiles/r6_ab_head_dx_mixed_layout.log): cfg-3's dEw (N = 512, ~10^5 K, a few hundred to a thousand live rows) 0.301 -> 0.220 ms
  // per step; cfg-2's (N = 256: one tile column, three row tiles -- 240 workgroups that each add a 256 KB partial tile atomically) 0.135 -> 0.156: not taken
  return mode == 2 || (p.K >= 8192 && p.N >= 512);
}

const char* gemm_kernel_name(const GemmParams& p0, bool bf16_mode, bool a_f32, bool b_f32, bool a_km, bool b_km) {
  GemmParams p = p0;
  if (p.splitk < 1) p.splitk = 1;
  if (bf16_mode && !a_km && !b_km && !a_f32 && !b_f32) {
    const int k = pick_rowmajor_kernel(p);
    if (k == 2) return gemm8p_forwards_to_8c(p) ? (gemm4p_takes(p) ? "4p" : "8c") : "8p";
  }
  if (bf16_mode && a_km && b_km && !a_f32 && !b_f32 && use_8p_tn(p)) { GemmParams q = p; q.splitk = 8; return gemm4k_eligible(q) ? "4k" : "8t"; }
  if (bf16_mode && a_km && b_km && !a_f32 && !b_f32 && use_8p_tn_store(p)) return "8ts";
  if (bf16_mode && !a_km && !b_km && !a_f32 && !b_f32 && use_8p_nt_splitk(p)) return "8s";
  if (bf16_mode && !a_km && b_km && !a_f32 && !b_f32 && use_8p_mix(p)) return "8m";
  return a_km ? "tn" : (b_km ? "nn" : "nt");
}

template <typename CT>
int launch_gemm(const GemmParams& p0, bool a_f32, bool b_f32, bool a_km, bool b_km, hipStream_t s) {
  GemmParams p = p0;
  constexpr int EPC = MmaT<CT>::EPC;
  if (p.splitk < 1) p.splitk = 1;
  if (p.splitk > 1) p.splitk = (p.splitk + 7) / 8 * 8;   // one K split never straddles XCDs (see the kernel's work mapping)
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
  if (p.epi == EPI_QKV_ROPE)
    ARG_CHECK(p.hd % 16 == 0 && p.N % 4 == 0 && p.n_q % 4 == 0 && p.n_k % 4 == 0, "gemm: rope epilogue needs hd % 16 == 0");
  if (p.epi == EPI_SWIGLU) ARG_CHECK(p.N % 32 == 0 && p.ldc2 % 4 == 0, "gemm: swiglu epilogue needs N % 32 == 0");
  if (p.epi == EPI_SWIGLU_BWD) ARG_CHECK(p.N % 16 == 0 && p.ldc2 == p.ldc && ((uintptr_t)p.C2 % 16) == 0, "gemm: swiglu-bwd epilogue needs N % 16 == 0");
  ARG_CHECK(p.ldc % 4 == 0 && ((uintptr_t)p.C % 16) == 0, "gemm: C rows must keep 16-byte alignment (ldc % 4 == 0)");
  if (is_bf16<CT>::value && !(p.c_f32 || p.epi == EPI_ACCUM || p.epi == EPI_RESIDUAL || p.epi == EPI_ATOMIC))
    ARG_CHECK(p.ldc % 8 == 0 && (p.C2 == nullptr || (p.ldc2 % 8 == 0 && ((uintptr_t)p.C2 % 16) == 0)), "gemm: bf16 outputs need ldc % 8 == 0");
  if (p.epi == EPI_QKV_ROPE) ARG_CHECK((p.hd & (p.hd - 1)) == 0, "gemm: head_dim must be a power of two");
  if constexpr (!is_bf16<CT>::value) { a_f32 = false; b_f32 = false; }
  if constexpr (is_bf16<CT>::value) {
    if (!a_km && !b_km && !a_f32 && !b_f32 && use_8p_nt_splitk(p)) return launch_gemm8p_nt_splitk(p, s);
    if (!a_km && !b_km && !a_f32 && !b_f32) {
      const int k = pick_rowmajor_kernel(p);
      if (k == 2) return launch_gemm8p(p, s);
    }
    if (!a_km && b_km && !a_f32 && !b_f32 && use_8p_mix(p)) {
      GemmParams q = p; q.K = p.K & ~63;
      { const int rc_ = launch_gemm8p_mix(q, s); if (rc_ != RSYS_OK) return rc_; }
      if ((p.K & 63) == 0) return RSYS_OK;
      GemmParams t = p;   // the last K % 64 columns of A / rows of B, added by the 128x128 kernel (guarded loads)
      t.A = (const char*)p.A + (size_t)q.K * 2; t.B = (const char*)p.B + (size_t)q.K * p.ldb * 2; t.K = p.K & 63; t.splitk = 1;
      return launch_one<CT, false, false, false, true>(t, s);
    }
    if (a_km && b_km && !a_f32 && !b_f32 && use_8p_tn(p)) return launch_gemm8p_tn(p, s);
    if (a_km && b_km && !a_f32 && !b_f32 && use_8p_tn_store(p)) return launch_gemm8p_tn_store(p, s);
  }
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

template <typename CT>
long long gemm_slab_need(const GemmParams& p0, bool a_f32, bool b_f32, bool a_km, bool b_km) {
  GemmParams p = p0;
  if (p.epi != EPI_ATOMIC) return 0;
  if (p.splitk < 1) p.splitk = 1;
  if (p.splitk > 1) p.splitk = (p.splitk + 7) / 8 * 8;
  if constexpr (is_bf16<CT>::value) {
    if (!a_km && !b_km && !a_f32 && !b_f32 && use_8p_nt_splitk(p)) return (long long)gemm8p_splits(p, false) * p.M * p.N;
    if (a_km && b_km && !a_f32 && !b_f32 && use_8p_tn(p)) return (long long)gemm8p_splits(p, true) * p.M * p.N;
  }
  return p.splitk > 1 ? (long long)p.splitk * p.M * p.N : 0;
}
template long long gemm_slab_need<bf16>(const GemmParams&, bool, bool, bool, bool);
template long long gemm_slab_need<float>(const GemmParams&, bool, bool, bool, bool);

template int launch_gemm<bf16>(const GemmParams&, bool, bool, bool, bool, hipStream_t);
template int launch_gemm<float>(const GemmParams&, bool, bool, bool, bool, hipStream_t);

}  // namespace rsys
