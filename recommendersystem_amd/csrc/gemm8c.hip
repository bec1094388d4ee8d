// bf16 MFMA GEMM for row-major operands, gfx950: C[M,N] = sum_k A[m][k] * B[n][k] -- the persistent 256x256 LDS-DMA pipeline of
// gemm8p.hip (same tile, same LDS image, same four phases per K tile: read that header first) with the OUTPUT STREAM OVERLAPPED:
//
//  * one operand stream per workgroup.  The K tiles of all the output tiles a workgroup walks form one sequence g = 0, 1, 2, ...;
//    phase P1 / P2 of K tile g request the second halves of K tile g + 1 and P3 / P4 the first halves of g + 2 WHATEVER output tile
//    those belong to, so a new output tile finds its first K tile landed and its second one in flight (gemm8p stops requesting two
//    K tiles before the end of a tile, bursts 128 KB per CU after it and waits for them: ~6 K of ~35 K cycles per tile at K = 512).
//    The per-lane source offsets are the same for every tile (the tile origin is part of the wave-uniform buffer descriptor, whose
//    num_records also bounds rows >= M / columns >= N of edge tiles: they read zeros and are never stored), the requests are
//    unconditional (behind the last tile the descriptor is empty), so every vmcnt is a compile-time constant;
//  * staged epilogue (classes without operand loads: plain store, SwiGLU).  The first K tile of the NEXT output tile starts its
//    accumulator chains from the constant 0, so the registers of quadrant q of the finished tile are free for its MFMAs as soon as
//    that quadrant has been converted and its stores issued: phase Pq of that K tile carries "epilogue piece q" in its load section,
//    beside the partner wave's MFMA section (the two wave rows of a SIMD run one barrier apart).  The stores drain behind the
//    following MFMAs; the counted waits allow for them (vector memory retires in issue order);
//  * classes whose epilogue loads operands (residual, accumulate, table, SwiGLU backward, RoPE) keep the register epilogue of
//    gemm_epi_reg.hpp between two tiles; the second halves of the next tile's K tile 1 are requested in front of it (run-ahead), so
//    the stores of its last row blocks have until the second K tile to drain.
// Full tiles and edge tiles run in two passes (edge epilogues mask their stores; see gemm8p.hip).
#include <algorithm>

#include "gemm.hpp"
#include "gemm_epi.hpp"
#include "gemm_epi_reg.hpp"

namespace rsys {

namespace {

constexpr int C8_BM = 256, C8_BN = 256, C8_BK = 64;
#ifndef C8_DEBUG
#define C8_DEBUG 0   // timing-only builds of tools/micro/gemm8c_dev.hip: 1 = no epilogue at all (accumulators kept alive), 2 = staged pieces without their stores
#endif

typedef __attribute__((ext_vector_type(4))) int c8_i32x4;
extern "C" __device__ void rsys_c8_raw_buffer_load_lds(c8_i32x4 rsrc, LDS_AS unsigned int* lds, int size, int voffset, int soffset,
                                                       int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

#define C8_BARRIER()                         \
  do {                                       \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");          \
  } while (0)

// operand window of one K tile of one output tile: wave-uniform (SGPRs)
struct C8Cur {
  const char* a; const char* b;   // A + m0 lda 2 + kt 128, B + n0 ldb 2 + kt 128
  unsigned int ra, rb;            // num_records: the last valid row's K tile ends at ra / rb; 0 = no tile (reads nothing)
  int kt;
};
// position in the workgroup's run of output tiles: index and (row, column) of the tile, kept incrementally (no division per tile)
struct C8Tile { int t, tm, tn; };

// vector-memory operations a staged epilogue issues in phase ph (0..3) of the next tile's first K tile
template <int EC, bool CF32>
constexpr int c8_piece_ops(int ph) {
  if (EC == EPI_STORE) return CF32 ? 8 : 4;
  if (EC == EPI_SWIGLU) return (ph == 1 || ph == 3) ? 8 : 4;
  return 0;
}
#ifndef C8_NO_STAGED   // (tools/micro/gemm8c_dev.hip builds a second binary without the staged epilogue: what the stream alone buys)
template <int EC, bool CF32> constexpr bool c8_staged() { return EC == EPI_STORE || EC == EPI_SWIGLU; }
#else
template <int EC, bool CF32> constexpr bool c8_staged() { return false; }
#endif
// lower bound on the stores a register epilogue issues after its last load (see gemm8p.hip `pend`)
template <int EC, bool CF32>
constexpr int c8_pend() {
  if (EC == EPI_ACCUM || EC == EPI_RESIDUAL || EC == EPI_SWIGLU_BWD) return 8;
  if (EC == EPI_TABLE) return 12;
  if (EC == EPI_QKV_ROPE) return CF32 ? 8 : 4;
  return 0;
}

template <int N> __device__ __forceinline__ void c8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EC, bool CF32>
__global__ __launch_bounds__(512) void gemm8c_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];   // [buf][A h0 | A h1 | B h0 | B h1] x 16 KB
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int fq = l >> 4, fr = l & 15;
  const int bid = blockIdx.x, nblk = gridDim.x;

  // ---- the workgroup's run of output tiles (XCD-aware, as gemm8p.hip)
  const int tiles_n = (p.N + C8_BN - 1) / C8_BN;
  const int rows = __builtin_amdgcn_readfirstlane(p.m_dev != nullptr ? min(*p.m_dev, p.M) : p.M);   // (a vector load: make it scalar again)
  const int ntiles = ((rows + C8_BM - 1) / C8_BM) * tiles_n;
  int tile_first, tile_end, tile_step;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    tile_first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_end = (xcd < r ? (xcd + 1) * (q + 1) : r * (q + 1) + (xcd + 1 - r) * q);
    tile_step = (nblk + 7 - xcd) >> 3;
  }
  if (tile_first >= tile_end) return;
  const int nt = p.K / C8_BK;   // launcher: K % 64 == 0, nt >= 2
  const int step_m = tile_step / tiles_n, step_n = tile_step % tiles_n;

  // ---- per-lane DMA source offsets, the same for every tile: instruction j of wave w fills the 1 KB piece (w*2+j) of a
  // half-tile = local rows (w*2+j)*8 + [0,8); lane l -> local row + (l>>3), stored slot l&7 (swizzle on the source side)
  unsigned int aoff[2][2], boff[2][2];   // [j][h]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lr = (w * 2 + j) * 8 + (l >> 3);
    const int c = (l & 7) ^ ((lr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      aoff[j][h] = (unsigned int)((lr >> 6) * 128 + h * 64 + (lr & 63)) * (unsigned int)(p.lda * 2) + (unsigned int)(c * 16);
      boff[j][h] = (unsigned int)((lr >> 5) * 64 + h * 32 + (lr & 31)) * (unsigned int)(p.ldb * 2) + (unsigned int)(c * 16);
    }
  }
  unsigned char* const dma_base = smem + w * 2048;   // + buf*65536 + X*32768 + h*16384 + j*1024
  auto rsrc_of = [&](const char* base, unsigned int rec) __attribute__((always_inline)) -> c8_i32x4 {
    const unsigned long long a = (unsigned long long)base;
    c8_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)((a >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)rec);
    r[3] = 0x00020000;
    return r;
  };
  auto stage_a = [&](const C8Cur& c, auto H, int bo) __attribute__((always_inline)) {
    constexpr int h = decltype(H)::value;
    const c8_i32x4 rs = rsrc_of(c.a, c.ra);
    rsys_c8_raw_buffer_load_lds(rs, (LDS_AS unsigned int*)(dma_base + bo + h * 16384), 16, (int)aoff[0][h], 0, 0, 0);
    rsys_c8_raw_buffer_load_lds(rs, (LDS_AS unsigned int*)(dma_base + bo + h * 16384 + 1024), 16, (int)aoff[1][h], 0, 0, 0);
  };
  auto stage_b = [&](const C8Cur& c, auto H, int bo) __attribute__((always_inline)) {
    constexpr int h = decltype(H)::value;
    const c8_i32x4 rs = rsrc_of(c.b, c.rb);
    rsys_c8_raw_buffer_load_lds(rs, (LDS_AS unsigned int*)(dma_base + bo + 32768 + h * 16384), 16, (int)boff[0][h], 0, 0, 0);
    rsys_c8_raw_buffer_load_lds(rs, (LDS_AS unsigned int*)(dma_base + bo + 32768 + h * 16384 + 1024), 16, (int)boff[1][h], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment read offsets: lane (fq, fr) takes chunk kk*4+fq of local row base+fr
  const int sw0 = ((fq ^ (fr >> 1)) << 4), sw1 = (((4 + fq) ^ (fr >> 1)) << 4);
  const int a_rd0 = (wr * 64 + fr) * 128 + sw0, a_rd1 = (wr * 64 + fr) * 128 + sw1;
  const int b_rd0 = 32768 + (wc * 32 + fr) * 128 + sw0, b_rd1 = 32768 + (wc * 32 + fr) * 128 + sw1;

  f32x4 acc[8][4];
  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](int bo, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd0);
      af[i][1] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd1);
    }
  };
  auto read_b = [&](bf16x8(&bf)[2][2], int bo, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf[j][0] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd0);
      bf[j][1] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd1);
    }
  };
  // 16 MFMAs of quadrant (ih, jh); C0: the chains start from the constant 0 (first K tile of an output tile)
#ifndef C8_SHADOW
#define C8_SHADOW 1   // bf16 plain-store class: convert quadrant q + 1 of the finished tile between the MFMAs of phase q (0: in the load section)
#endif
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  u32x4 pkq[4];   // bf16 plain-store class: one converted quadrant (a 16-byte store per row block), made a phase before it is stored
  auto pk2 = [](float a, float b) __attribute__((always_inline)) -> unsigned int {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v; v[0] = (bf16)a; v[1] = (bf16)b;
    return __builtin_bit_cast(unsigned int, v);
  };
  // row block ii of quadrant (ih, jh) of the FINISHED tile -> pkq[ii]: the lane's 8 consecutive bf16 columns after the pair swap
  auto convert_rb = [&](auto IH, auto JH, auto II) __attribute__((always_inline)) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value, ii = decltype(II)::value, i = ih * 4 + ii;
    const f32x4 x = acc[i][jh * 2], y = acc[i][jh * 2 + 1];
    auto r0 = __builtin_amdgcn_permlane16_swap(pk2(x[0], x[1]), pk2(y[0], y[1]), false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(pk2(x[2], x[3]), pk2(y[2], y[3]), false, false);
    pkq[ii] = u32x4{r0[0], r1[0], r0[1], r1[1]};
  };
  auto mma_q = [&](auto IH, auto JH, auto C0, const bf16x8(&bf)[2][2], auto CIH, auto CJH) __attribute__((always_inline)) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value;
    constexpr bool c0 = decltype(C0)::value;
    constexpr int cih = decltype(CIH)::value;   // >= 0: convert row block i of quadrant (cih, cjh) of the finished tile behind the MFMAs of row block i
    __builtin_amdgcn_s_setprio(1);
    static_for<4>([&](auto i) {
      static_for<2>([&](auto j) {
        if constexpr (c0)
          acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else
          acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][0], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
        acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[i][1], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
      });
      if constexpr (cih >= 0) convert_rb(CIH, CJH, i);
    });
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- staged epilogue pieces (full tiles only): quadrant (ih, jh) of the finished tile whose wave block starts at (wm0, wn0).
  // Layout as gemm_epi_reg.hpp: acc[i][j][r] = C[wm0 + 16 i + fr][wn0 + 16 j + 4 fq + r].
  unsigned int hold[8];   // SwiGLU: packed g of the quadrant that comes first of a block pair (2 dwords per row block)
  auto st_pair_pk = [&](void* base, unsigned int off, unsigned int x0, unsigned int x1, unsigned int y0, unsigned int y1) __attribute__((always_inline)) {
    auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    if constexpr (C8_DEBUG == 2) asm volatile("" ::"v"(v), "v"(off));
    else *(u32x4*)((char*)base + off) = v;
  };
  auto keep_acc = [&]() __attribute__((always_inline)) {
    static_for<8>([&](auto i) { static_for<4>([&](auto j) { const f32x4 v = acc[i][j]; asm volatile("" ::"v"(v)); }); });
  };
  auto piece = [&](auto IH, auto JH, int wm0, int wn0) __attribute__((always_inline)) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value;
    if constexpr (C8_DEBUG == 1) { static_for<4>([&](auto ii) { static_for<2>([&](auto jj) { const f32x4 v = acc[ih * 4 + ii][jh * 2 + jj]; asm volatile("" ::"v"(v)); }); }); return; }
    const int lrow = wm0 + fr;
    const int c8 = wn0 + ((fq & 1) << 4) + ((fq >> 1) << 3) + jh * 32;   // first of the lane's 8 columns after the pair swap
    if constexpr (EC == EPI_STORE) {
      if constexpr (CF32) {
        const int c4 = wn0 + 4 * fq + jh * 32;
        static_for<4>([&](auto ii) {
          constexpr int i = ih * 4 + decltype(ii)::value;
          const unsigned int ro = (unsigned int)(lrow + 16 * i) * (unsigned int)(p.ldc * 4);
          static_for<2>([&](auto jj) {
            constexpr int j = jh * 2 + decltype(jj)::value;
            const f32x4 v = acc[i][j];   // (launcher: alpha == 1)
            *(f32x4*)((char*)p.C + ro + (unsigned int)(c4 + 16 * decltype(jj)::value) * 4u) = v;
          });
        });
      } else if constexpr (C8_SHADOW) {
        // (launcher: alpha == 1) the quadrant was converted between the MFMAs of the phase before (the first one right here)
        if constexpr (ih == 0 && jh == 0) static_for<4>([&](auto ii) { convert_rb(IH, JH, ii); });
        static_for<4>([&](auto ii) {
          constexpr int i = ih * 4 + decltype(ii)::value;
          const unsigned int ro = (unsigned int)(lrow + 16 * i) * (unsigned int)(p.ldc * 2);
          if constexpr (C8_DEBUG == 2) asm volatile("" ::"v"(pkq[decltype(ii)::value]), "v"(ro));
          else *(u32x4*)((char*)p.C + ro + (unsigned int)c8 * 2u) = pkq[decltype(ii)::value];
        });
      } else {
        static_for<4>([&](auto ii) {
          constexpr int i = ih * 4 + decltype(ii)::value;
          const unsigned int ro = (unsigned int)(lrow + 16 * i) * (unsigned int)(p.ldc * 2);
          const f32x4 x = acc[i][jh * 2], y = acc[i][jh * 2 + 1];
          st_pair_pk(p.C, ro + (unsigned int)c8 * 2u, pk2(x[0], x[1]), pk2(x[2], x[3]), pk2(y[0], y[1]), pk2(y[2], y[3]));
        });
      }
    } else if constexpr (EC == EPI_SWIGLU) {
      // blocks (2 jh, 2 jh + 1) are a [16 a | 16 b] group: C gets [a|b] as is; g = silu(a) * b of block pairs (0,1) and (2,3)
      // share a 16-byte store of C2, so the quadrant that comes first in phase order (jh 0 for ih 0, jh 1 for ih 1) holds its g
      constexpr bool first = (ih == 0) ? (jh == 0) : (jh == 1);
      const int gc = (wn0 >> 1) + ((fq & 1) << 4) + ((fq >> 1) << 3);
      static_for<4>([&](auto ii) {
        constexpr int k = decltype(ii)::value, i = ih * 4 + k;
        const unsigned int ro = (unsigned int)(lrow + 16 * i) * (unsigned int)(p.ldc * 2);
        const f32x4 a = acc[i][jh * 2], b = acc[i][jh * 2 + 1];
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = a[r] * __builtin_amdgcn_rcpf(1.f + __expf(-a[r])) * b[r];
        st_pair_pk(p.C, ro + (unsigned int)c8 * 2u, pk2(a[0], a[1]), pk2(a[2], a[3]), pk2(b[0], b[1]), pk2(b[2], b[3]));
        const unsigned int g0 = pk2(g[0], g[1]), g1 = pk2(g[2], g[3]);
        if constexpr (first) { hold[2 * k] = g0; hold[2 * k + 1] = g1; }
        else {
          const unsigned int ro2 = (unsigned int)(lrow + 16 * i) * (unsigned int)(p.ldc2 * 2);
          if constexpr (jh == 1) st_pair_pk(p.C2, ro2 + (unsigned int)gc * 2u, hold[2 * k], hold[2 * k + 1], g0, g1);   // x = blocks (0,1), y = (2,3)
          else st_pair_pk(p.C2, ro2 + (unsigned int)gc * 2u, g0, g1, hold[2 * k], hold[2 * k + 1]);
        }
      });
    }
  };

  // ---- one K tile.  n = window of K tile g + 1, s = of g + 2 (advanced by the caller).  Waits: P2 leaves the four half-tiles
  // requested after A1(g) in flight (+ X2 operations of epilogue pieces / a register epilogue issued after it), P4 the three
  // requested after B1(g+1) (+ X4).  PIECES: epilogue piece q of the previous tile (wave block at pm0, pn0) in phase q.
  // DMA12 = false: B1 / A1 of g + 1 were requested ahead of a register epilogue.
  C8Cur cn, cs;
  int bo = 0;
  auto body = [&](auto C0, auto PIECES, auto DMA12, auto X2, auto X4, int pm0, int pn0) __attribute__((always_inline)) {
    constexpr bool pieces = decltype(PIECES)::value, dma12 = decltype(DMA12)::value;
    constexpr int x2 = decltype(X2)::value, x4 = decltype(X4)::value;
    constexpr bool shadow = pieces && C8_SHADOW && EC == EPI_STORE && !CF32;
    using CV1 = std::integral_constant<int, shadow ? 0 : -1>;   // row half of the quadrant converted in P1 (-1: none)
    using CV2 = std::integral_constant<int, shadow ? 1 : -1>;   // ... in P2 and P3
    const int bn = bo ^ 65536;
    // P1
    read_b(bf0, bo, 0);
    read_a(bo, 0);
    if constexpr (dma12) stage_b(cn, I1{}, bn);
    if constexpr (pieces) { piece(I0{}, I0{}, pm0, pn0); __builtin_amdgcn_sched_barrier(0); }
    C8_BARRIER();
    mma_q(I0{}, I0{}, C0, bf0, CV1{}, I1{});   // (converts quadrant (0, 1) when staged)
    C8_BARRIER();
    // P2
    read_b(bf1, bo, 1);
    if constexpr (dma12) stage_a(cn, I1{}, bn);
    if constexpr (pieces) { piece(I0{}, I1{}, pm0, pn0); __builtin_amdgcn_sched_barrier(0); }
    c8_wait_vm<8 + x2>();
    C8_BARRIER();
    mma_q(I0{}, I1{}, C0, bf1, CV2{}, I1{});   // (1, 1)
    C8_BARRIER();
    // P3
    read_a(bo, 1);
    stage_b(cs, I0{}, bo);
    if constexpr (pieces) { piece(I1{}, I1{}, pm0, pn0); __builtin_amdgcn_sched_barrier(0); }
    C8_BARRIER();
    mma_q(I1{}, I1{}, C0, bf1, CV2{}, I0{});   // (1, 0)
    C8_BARRIER();
    // P4
    stage_a(cs, I0{}, bo);
    if constexpr (pieces) { piece(I1{}, I0{}, pm0, pn0); __builtin_amdgcn_sched_barrier(0); }
    c8_wait_vm<6 + x4>();
    C8_BARRIER();
    mma_q(I1{}, I0{}, C0, bf0, std::integral_constant<int, -1>{}, I0{});
    C8_BARRIER();
    bo = bn;
  };
  using T_ = std::true_type; using F_ = std::false_type;
  auto run_pass = [&](auto FULLC) __attribute__((always_inline)) {
    constexpr bool WANT = decltype(FULLC)::value;
    constexpr bool STAGED = WANT && c8_staged<EC, CF32>();
    constexpr int PEND = WANT ? c8_pend<EC, CF32>() : 0;
    auto tile_full = [&](const C8Tile& x) __attribute__((always_inline)) -> bool {
      return x.tm * C8_BM + C8_BM <= p.M && x.tn * C8_BN + C8_BN <= p.N;
    };
    auto tile_inc = [&](C8Tile& x) __attribute__((always_inline)) {
      x.t += tile_step; x.tm += step_m; x.tn += step_n;
      if (x.tn >= tiles_n) { x.tn -= tiles_n; ++x.tm; }
    };
    auto next_tile = [&](C8Tile x) __attribute__((always_inline)) -> C8Tile {   // the next tile of this pass (t >= tile_end: none)
      if (x.t < tile_end) { tile_inc(x); while (x.t < tile_end && tile_full(x) != WANT) tile_inc(x); }
      return x;
    };
    auto set_tile = [&](C8Cur& c, const C8Tile& x) __attribute__((always_inline)) {
      c.kt = 0;
      if (x.t < tile_end) {
        const int m0 = x.tm * C8_BM, n0 = x.tn * C8_BN;
        c.a = (const char*)p.A + (long long)m0 * p.lda * 2;
        c.b = (const char*)p.B + (long long)n0 * p.ldb * 2;
        // (launcher: 256 rows of either operand span < 4 GB)
        c.ra = (unsigned int)min(p.M - 1 - m0, 255) * (unsigned int)(p.lda * 2) + 128u;
        c.rb = (unsigned int)min(p.N - 1 - n0, 255) * (unsigned int)(p.ldb * 2) + 128u;
      } else { c.a = (const char*)p.A; c.b = (const char*)p.B; c.ra = 0; c.rb = 0; }
    };
    C8Tile tile{tile_first, tile_first / tiles_n, tile_first % tiles_n};
    while (tile.t < tile_end && tile_full(tile) != WANT) tile_inc(tile);
    if (tile.t >= tile_end) return;
    // The request stream runs two K tiles ahead of the MFMAs.  cs = window of K tile g + 2, cn = of g + 1; nx = the first K tile
    // of the output tile after the one cs is in, computed once per output tile outside the K loop (at the tile transitions),
    // so that advancing a window inside the K loop is a few scalar moves.
    C8Tile ts = tile;            // the output tile cs is in
    C8Cur nx; bool nx_stale = false;
    auto advance = [&](C8Cur& c) __attribute__((always_inline)) {
      if (c.kt + 1 < nt) { ++c.kt; c.a += C8_BK * 2; c.b += C8_BK * 2; }
      else { c = nx; nx_stale = true; }
    };
    auto refresh_nx = [&]() __attribute__((always_inline)) {
      if (nx_stale) { ts = next_tile(ts); nx_stale = false; }
      set_tile(nx, next_tile(ts));
    };
    // K tile 0 (all four half-tiles) and the first halves of K tile 1: 12 DMA instructions per wave
    C8Cur c0;
    set_tile(c0, tile);
    set_tile(nx, next_tile(ts));
    bo = 0;
    stage_b(c0, I0{}, 0); stage_a(c0, I0{}, 0); stage_b(c0, I1{}, 0); stage_a(c0, I1{}, 0);
    cn = c0; advance(cn);
    stage_b(cn, I0{}, 65536); stage_a(cn, I0{}, 65536);
    cs = cn; advance(cs);
    if (nx_stale) refresh_nx();   // (nt == 2: the stream is already in the second output tile)
    c8_wait_vm<6>();   // B0, A0, B1 of K tile 0 have landed
    C8_BARRIER();
    if (wr == 1) C8_BARRIER();   // the second wave row runs one barrier behind the first
    auto step = [&]() __attribute__((always_inline)) { cn = cs; advance(cs); };
    using Z = std::integral_constant<int, 0>;
    body(T_{}, F_{}, T_{}, Z{}, Z{}, 0, 0); step();
#pragma unroll 1
    for (int kt = 1; kt < nt; ++kt) { body(F_{}, F_{}, T_{}, Z{}, Z{}, 0, 0); step(); }
    C8Tile tnext = next_tile(tile);
#pragma unroll 1
    while (tnext.t < tile_end) {
      const int em0 = tile.tm * C8_BM, en0 = tile.tn * C8_BN;
      refresh_nx();
      if constexpr (STAGED) {
        constexpr int s1 = c8_piece_ops<EC, CF32>(0), s2 = c8_piece_ops<EC, CF32>(1), s3 = c8_piece_ops<EC, CF32>(2), s4 = c8_piece_ops<EC, CF32>(3);
        body(T_{}, T_{}, T_{}, std::integral_constant<int, s1 + s2>{}, std::integral_constant<int, s1 + s2 + s3 + s4>{}, em0 + wr * 128, en0 + wc * 64); step();
        body(F_{}, F_{}, T_{}, std::integral_constant<int, s2 + s3 + s4>{}, Z{}, 0, 0); step();
      } else {
        // run-ahead: second halves of the next K tile but one (their LDS slots were last read a phase ago), then the register epilogue
        stage_b(cn, I1{}, bo ^ 65536); stage_a(cn, I1{}, bo ^ 65536);
        if constexpr (C8_DEBUG == 1) keep_acc(); else
        epilogue_regs<WANT ? 1 : 0, EC>(p, acc, em0 + wr * 128, en0 + wc * 64, WANT, fq, fr);
        if constexpr (!WANT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): edge epilogues leave masked loads "pending" for hipcc
        body(T_{}, F_{}, F_{}, std::integral_constant<int, PEND>{}, std::integral_constant<int, PEND>{}, 0, 0); step();
        body(F_{}, F_{}, T_{}, std::integral_constant<int, PEND>{}, Z{}, 0, 0); step();
      }
#pragma unroll 1
      for (int kt = 2; kt < nt; ++kt) { body(F_{}, F_{}, T_{}, Z{}, Z{}, 0, 0); step(); }
      tile = tnext;
      tnext = next_tile(tile);
    }
    {   // last tile of the pass: nothing left to overlap with
      const int em0 = tile.tm * C8_BM, en0 = tile.tn * C8_BN;
      if (wr == 0) C8_BARRIER();   // rejoin (equal barrier counts)
      if constexpr (C8_DEBUG == 1) keep_acc(); else
      epilogue_regs<WANT ? 1 : 0, EC>(p, acc, em0 + wr * 128, en0 + wc * 64, WANT, fq, fr);
      if constexpr (!WANT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
  };
  {   // timing experiment (tools/micro/gemm8c_dev.hip): workgroup j of an XCD starts (j % groups) * naps sleeps late, so that the
      // workgroups' epilogues (the HBM-bound part of a tile) do not all fall into the same microseconds
    const int groups = (p.flags >> 8) & 0xFF, naps = (p.flags >> 16) & 0xFF;
    if (groups > 1) { const int d = ((bid >> 3) % groups) * naps; for (int k = 0; k < d; ++k) __builtin_amdgcn_s_sleep(32); }
  }
  run_pass(std::true_type{});
  run_pass(std::false_type{});
}

}  // namespace

// classes with a kernel of their own; everything else (bias, GELU, explicit RoPE positions, fp8 amax producers, accumulate-into-bf16)
// stays on gemm8p.hip
bool gemm8c_eligible(const GemmParams& p) {
  if (!gemm8p_eligible(p)) return false;
  if (p.K < 2 * C8_BK || p.f8 != 0 || p.f8_amax_out != nullptr) return false;
  if ((unsigned long long)p.lda * 2 * 256 >= (1ull << 32) || (unsigned long long)p.ldb * 2 * 256 >= (1ull << 32)) return false;
  switch (p.epi) {
    case EPI_STORE: return p.alpha == 1.f;
    case EPI_SWIGLU: case EPI_RESIDUAL: case EPI_ACCUM: case EPI_SWIGLU_BWD: case EPI_TABLE: return true;
    case EPI_QKV_ROPE: return p.rope_pos == nullptr && !p.c_f32;
    default: return false;
  }
}

int launch_gemm8c(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  int cus = 256;
  { static int n = 0; if (n == 0) { int dev = 0, v = 0; n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; } cus = n; }
  const int tiles = ((p.M + C8_BM - 1) / C8_BM) * ((p.N + C8_BN - 1) / C8_BN);
  const dim3 grid((p.flags & 2) && p.m_dev == nullptr ? tiles : std::min(tiles, cus)), blk(512);
  switch (p.epi) {
    case EPI_STORE:
      if (p.c_f32) hipLaunchKernelGGL((gemm8c_kernel<EPI_STORE, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_STORE, false>), grid, blk, 0, s, p);
      break;
    case EPI_SWIGLU: hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU, false>), grid, blk, 0, s, p); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL((gemm8c_kernel<EPI_RESIDUAL, true>), grid, blk, 0, s, p); break;
    case EPI_ACCUM: hipLaunchKernelGGL((gemm8c_kernel<EPI_ACCUM, true>), grid, blk, 0, s, p); break;
    case EPI_SWIGLU_BWD: hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU_BWD, false>), grid, blk, 0, s, p); break;
    case EPI_TABLE: hipLaunchKernelGGL((gemm8c_kernel<EPI_TABLE, true>), grid, blk, 0, s, p); break;
    case EPI_QKV_ROPE: hipLaunchKernelGGL((gemm8c_kernel<EPI_QKV_ROPE, false>), grid, blk, 0, s, p); break;
    default: set_error("gemm8c: epilogue class without a kernel"); return RSYS_ERR_ARG;
  }
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
