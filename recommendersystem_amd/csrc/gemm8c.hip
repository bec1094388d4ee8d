// bf16 MFMA GEMM for row-major operands, gfx950: C[M,N] = sum_k A[m][k] * B[n][k] -- the persistent 256x256 LDS-DMA pipeline of
// gemm8p.hip (same tile, same LDS image, same four phases per K tile: read that header first) with ONE OPERAND STREAM PER
// WORKGROUP.  The K tiles of all the output tiles a workgroup walks form one sequence g = 0, 1, 2, ...; phases P1 / P2 of K tile g
// request the second halves of K tile g + 1 and P3 / P4 the first halves of g + 2 WHATEVER output tile those belong to, so a new
// output tile finds its first K tile landed and its second one in flight.  (gemm8p stops requesting two K tiles before the end
// of a tile, bursts 128 KB per CU after it and waits for them with the MFMA pipe idle: without any epilogue this kernel runs the
// step's K = 512 shapes at 1.26 - 1.49 PFLOP/s, the rate of its K loop at K = 8192; profiles/r4_gemm8c_no_epilogue_timing_only.log.)
//  * The per-lane source offsets are the same for every tile: the tile origin is part of the wave-uniform buffer descriptor, whose
//    num_records also bounds rows >= M / columns >= N of edge tiles (they read zeros and are never stored).  The requests are
//    unconditional -- behind the last tile the descriptor is empty -- so every vmcnt is a compile-time constant, and a window
//    advances by a few scalar selects per K tile (the next output tile's window is computed once per tile, outside the K loop).
//  * Between two output tiles the register epilogue of gemm_epi_reg.hpp runs as in gemm8p, one kernel per epilogue class
//    (straight-line code, exact vmcnt counts).  The second halves of the next tile's K tile 1 are requested in front of it
//    (run-ahead), so the stores of its last row blocks have until the second K tile to drain; the accumulator chains of a tile's
//    first K tile start from the constant 0 (no zero fill).
//  * What was tried on top and is NOT here (tools/micro/gemm8c_dev.hip at commit "gemm8c: persistent 256x256 GEMM with one
//    operand stream ...", logs profiles/r4_gemm8c_*.log): the epilogue cut into four quadrant pieces carried by the load sections
//    of the next tile's first K tile (the C = 0 start frees a quadrant's registers as soon as it is stored).  Bit-identical, and
//    no faster for outputs that stay in the Infinity Cache, 14 - 27 % SLOWER for outputs that go to HBM (N = 2816, 4096): vector
//    memory retires in issue order per wave, so a wave that waits for its LDS-DMA also waits for every older store, and stores to
//    HBM need longer than the two K tiles of prefetch the LDS allows.  Starting the workgroups of an XCD staggered in time changed
//    nothing either: the output stream is bound per CU (bytes in flight / store latency), not by the CUs storing together.
//  * HALF form (round 6, template parameter): 128 x 256 output tiles for outputs with fewer 256 x 256 tiles than CUs (cfg-2's N = 256
//    products: 128 tiles, half the chip idle while the other half streams its output).  Same LDS image, same barriers and the same
//    request stream minus the second row half of A: the A image holds 128 rows, wave row wr owns rows [64 wr, 64 wr + 64) as 4 x 4
//    accumulator blocks, phases P1 / P2 carry the tile's 2 x 16 MFMAs per wave and P3 / P4 only issue requests; one counted wait
//    per K tile (P4: B1 of the next K tile has landed).  Bit-identical to the full form (same K order per accumulator).
#include <algorithm>

#include "gemm.hpp"
#include "gemm_epi.hpp"
#include "gemm_epi_reg.hpp"

namespace rsys {

namespace {

constexpr int C8_BM = 256, C8_BN = 256, C8_BK = 64;
#ifndef C8_DEBUG
#define C8_DEBUG 0   // 1: timing-only build of tools/micro/gemm8c_dev.hip without any epilogue (accumulators kept alive): what the K loop alone takes
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

// lower bound on the stores a register epilogue issues after its last load (see gemm8p.hip `pend`)
template <int EC, bool CF32, bool HALF = false>
constexpr int c8_pend() {
  if (EC == EPI_STORE) return (CF32 ? 32 : 16) / (HALF ? 2 : 1);   // (no loads: all stores)
  if (EC == EPI_SWIGLU) return HALF ? 12 : 24;
  if (EC == EPI_ACCUM || EC == EPI_RESIDUAL || EC == EPI_SWIGLU_BWD) return 8;   // (the stores of the last two row blocks)
  if (EC == EPI_TABLE) return 12;
  if (EC == EPI_QKV_ROPE) return CF32 ? 8 : 4;
  return 0;
}

template <int N> __device__ __forceinline__ void c8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EC, bool CF32, bool HALF = false>
__global__ __launch_bounds__(512) void gemm8c_kernel(GemmParams p) {
  constexpr int BMT = HALF ? 128 : C8_BM;   // rows of an output tile
  constexpr int WROWS = BMT / 2;            // rows of a wave row's block
  __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];   // [buf][A h0 | A h1 | B h0 | B h1] x 16 KB
  const int t = threadIdx.x, l0 = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int bid = blockIdx.x, nblk = gridDim.x;

  // ---- the workgroup's run of output tiles (XCD-aware, as gemm8p.hip)
  const int tiles_n = (p.N + C8_BN - 1) / C8_BN;
  const int rows = __builtin_amdgcn_readfirstlane(p.m_dev != nullptr ? min(*p.m_dev, p.M) : p.M);   // (a vector load: make it scalar again)
  const int ntiles = ((rows + BMT - 1) / BMT) * tiles_n;
  int tile_first, tile_end, tile_step;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    tile_first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_end = (xcd < r ? (xcd + 1) * (q + 1) : r * (q + 1) + (xcd + 1 - r) * q);
    tile_step = (nblk + 7 - xcd) >> 3;
  }
  if (tile_first >= tile_end) return;
  // flags bits 3-6 = R > 0 (the launcher sets 4): the tile sequence runs band by band of R tile rows, down the band's columns (4 x 1 pieces) instead of row by row, so
  // the 32 workgroups an XCD runs side by side cover 4 row blocks x 8 column blocks and share 12 operand blocks per K step through the
  // XCD's L2 where a row-major run of a wide output shares 1 + 32 (launcher: outputs more than eight tiles wide)
  const int band_rows = (p.flags >> 3) & 15;
  const bool patch = band_rows != 0;
  const bool rev = (p.flags & 256) != 0 && !patch;   // flags bit 8: the run walks the tile ROWS from the last to the first (row-major order otherwise unchanged)
  const int tiles_m_all = ntiles / tiles_n;
  auto tile_pos = [&](C8Tile& x) __attribute__((always_inline)) {
    if (x.t >= ntiles) return;   // (past the end: never dereferenced)
    const int per_band = band_rows * tiles_n, band = x.t / per_band, rem = x.t - band * per_band;
    const int rows_here = min(band_rows, tiles_m_all - band * band_rows);
    const int c = rem / rows_here;
    x.tm = band * band_rows + (rem - c * rows_here); x.tn = c;
  };
  const int nt = p.K / C8_BK;   // launcher: K % 64 == 0, nt >= 2
  const int step_m = tile_step / tiles_n, step_n = tile_step % tiles_n;

  // ---- per-lane DMA source offsets, the same for every tile: instruction j of wave w fills the 1 KB piece (w*2+j) of a
  // half-tile = local rows (w*2+j)*8 + [0,8); lane l -> local row + (l>>3), stored slot l&7 (swizzle on the source side)
  // and the fragment read offsets: lane (fq, fr) takes chunk kk*4+fq of local row base+fr.  All of it is derived from a fresh,
  // opaque copy of the lane id after every epilogue, so that none of these 14 registers is held through one.
  unsigned int aoff[2][2], boff[2][2];   // [j][h]
  int fq, fr, a_rd0, a_rd1, b_rd0, b_rd1;
  auto lane_setup = [&]() __attribute__((always_inline)) {
    int l = l0; asm volatile("" : "+v"(l));
    fq = l >> 4; fr = l & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = (w * 2 + j) * 8 + (l >> 3);
      const int c = (l & 7) ^ ((lr >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // (HALF: the A image's first half holds the tile's 128 rows in order, the second half is never requested)
        aoff[j][h] = (unsigned int)(HALF ? lr : (lr >> 6) * 128 + h * 64 + (lr & 63)) * (unsigned int)(p.lda * 2) + (unsigned int)(c * 16);
        boff[j][h] = (unsigned int)((lr >> 5) * 64 + h * 32 + (lr & 31)) * (unsigned int)(p.ldb * 2) + (unsigned int)(c * 16);
      }
    }
    const int sw0 = ((fq ^ (fr >> 1)) << 4), sw1 = (((4 + fq) ^ (fr >> 1)) << 4);
    a_rd0 = (wr * 64 + fr) * 128 + sw0; a_rd1 = (wr * 64 + fr) * 128 + sw1;
    b_rd0 = 32768 + (wc * 32 + fr) * 128 + sw0; b_rd1 = 32768 + (wc * 32 + fr) * 128 + sw1;
  };
  lane_setup();
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
  auto mma_q = [&](auto IH, auto JH, auto C0, const bf16x8(&bf)[2][2]) __attribute__((always_inline)) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value;
    constexpr bool c0 = decltype(C0)::value;
    __builtin_amdgcn_s_setprio(1);
    static_for<4>([&](auto i) {
      static_for<2>([&](auto j) {
        if constexpr (c0)
          acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else
          acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][0], af[i][0], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
        acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][1], af[i][1], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
      });
    });
    __builtin_amdgcn_s_setprio(0);
  };

  auto keep_acc = [&]() __attribute__((always_inline)) {   // (timing-only build: the MFMAs must not become dead code)
    static_for<8>([&](auto i) { static_for<4>([&](auto j) { const f32x4 v = acc[i][j]; asm volatile("" ::"v"(v)); }); });
  };

  // ---- one K tile.  cn = window of K tile g + 1, cs = of g + 2 (advanced by the caller).  Waits: P2 leaves the four half-tiles
  // requested after A1(g) in flight, P4 the three requested after B1(g+1); X2 / X4 more where the stores of a register epilogue
  // were issued after the guarded half-tile (they retire behind it).  C0: first K tile of an output tile.  DMA12 = false: B1 / A1
  // of g + 1 were requested ahead of a register epilogue.
  C8Cur cn, cs;
  int bo = 0;
  auto body = [&](auto C0, auto DMA12, auto X2, auto X4) __attribute__((always_inline)) {
    constexpr bool dma12 = decltype(DMA12)::value;
    constexpr int x2 = decltype(X2)::value, x4 = decltype(X4)::value;
    const int bn = bo ^ 65536;
    // P1
    read_b(bf0, bo, 0);
    read_a(bo, 0);
    if constexpr (dma12) stage_b(cn, I1{}, bn);
    C8_BARRIER();
    mma_q(I0{}, I0{}, C0, bf0);
    C8_BARRIER();
    // P2
    read_b(bf1, bo, 1);
    if constexpr (!HALF) {
      if constexpr (dma12) stage_a(cn, I1{}, bn);
      c8_wait_vm<8 + x2>();
    }
    C8_BARRIER();
    mma_q(I0{}, I1{}, C0, bf1);
    C8_BARRIER();
    // P3
    if constexpr (!HALF) read_a(bo, 1);
    stage_b(cs, I0{}, bo);
    C8_BARRIER();
    if constexpr (!HALF) mma_q(I1{}, I1{}, C0, bf1);
    C8_BARRIER();
    // P4 (HALF: the two half-tiles requested after B1 of the next K tile stay in flight)
    stage_a(cs, I0{}, bo);
    c8_wait_vm<(HALF ? 4 : 6) + x4>();
    C8_BARRIER();
    if constexpr (!HALF) mma_q(I1{}, I0{}, C0, bf0);
    C8_BARRIER();
    bo = bn;
  };
  using T_ = std::true_type; using F_ = std::false_type;
  auto run_pass = [&](auto FULLC) __attribute__((always_inline)) {
    constexpr bool WANT = decltype(FULLC)::value;
    constexpr int PEND = WANT ? c8_pend<EC, CF32, HALF>() : 0;
    auto tile_full = [&](const C8Tile& x) __attribute__((always_inline)) -> bool {
      return x.tm * BMT + BMT <= p.M && x.tn * C8_BN + C8_BN <= p.N;
    };
    auto tile_inc = [&](C8Tile& x) __attribute__((always_inline)) {
      x.t += tile_step;
      if (patch) { tile_pos(x); return; }
      if (rev) { if (x.t < ntiles) { const int r = x.t / tiles_n; x.tm = tiles_m_all - 1 - r; x.tn = x.t - r * tiles_n; } return; }
      x.tm += step_m; x.tn += step_n;
      if (x.tn >= tiles_n) { x.tn -= tiles_n; ++x.tm; }
    };
    auto next_tile = [&](C8Tile x) __attribute__((always_inline)) -> C8Tile {   // the next tile of this pass (t >= tile_end: none)
      if (x.t < tile_end) { tile_inc(x); while (x.t < tile_end && tile_full(x) != WANT) tile_inc(x); }
      return x;
    };
    auto set_tile = [&](C8Cur& c, const C8Tile& x) __attribute__((always_inline)) {
      c.kt = 0;
      if (x.t < tile_end) {
        const int m0 = x.tm * BMT, n0 = x.tn * C8_BN;
        c.a = (const char*)p.A + (long long)m0 * p.lda * 2;
        c.b = (const char*)p.B + (long long)n0 * p.ldb * 2;
        // (launcher: 256 rows of either operand span < 4 GB)
        c.ra = (unsigned int)min(p.M - 1 - m0, BMT - 1) * (unsigned int)(p.lda * 2) + 128u;
        c.rb = (unsigned int)min(p.N - 1 - n0, 255) * (unsigned int)(p.ldb * 2) + 128u;
      } else { c.a = (const char*)p.A; c.b = (const char*)p.B; c.ra = 0; c.rb = 0; }
    };
    C8Tile tile{tile_first, tile_first / tiles_n, tile_first % tiles_n};
    if (patch) tile_pos(tile);
    if (rev) tile.tm = tiles_m_all - 1 - tile.tm;
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
    // K tile 0 (all four half-tiles; HALF: three) and the first halves of K tile 1: 12 (10) DMA instructions per wave
    C8Cur c0;
    set_tile(c0, tile);
    set_tile(nx, next_tile(ts));
    bo = 0;
    stage_b(c0, I0{}, 0); stage_a(c0, I0{}, 0); stage_b(c0, I1{}, 0);
    if constexpr (!HALF) stage_a(c0, I1{}, 0);
    cn = c0; advance(cn);
    stage_b(cn, I0{}, 65536); stage_a(cn, I0{}, 65536);
    cs = cn; advance(cs);
    if (nx_stale) refresh_nx();   // (nt == 2: the stream is already in the second output tile)
    c8_wait_vm<(HALF ? 4 : 6)>();   // B0, A0, B1 of K tile 0 have landed
    C8_BARRIER();
    if (wr == 1) C8_BARRIER();   // the second wave row runs one barrier behind the first
    auto step = [&]() __attribute__((always_inline)) { cn = cs; advance(cs); };
    using Z = std::integral_constant<int, 0>;
    using PD = std::integral_constant<int, PEND>;
    body(T_{}, T_{}, Z{}, Z{}); step();
#pragma unroll 1
    for (int kt = 1; kt < nt; ++kt) { body(F_{}, T_{}, Z{}, Z{}); step(); }
    C8Tile tnext = next_tile(tile);
#pragma unroll 1
    while (tnext.t < tile_end) {
      const int em0 = tile.tm * BMT, en0 = tile.tn * C8_BN;
      refresh_nx();
      // run-ahead: second halves of the next K tile but one (their LDS slots were last read a phase ago), then the register epilogue
      stage_b(cn, I1{}, bo ^ 65536);
      if constexpr (!HALF) stage_a(cn, I1{}, bo ^ 65536);
      if constexpr (C8_DEBUG == 1) keep_acc(); else
      epilogue_regs<WANT ? 1 : 0, EC, HALF ? 4 : 8>(p, acc, em0 + wr * WROWS, en0 + wc * 64, WANT, fq, fr);
      if constexpr (!WANT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): edge epilogues leave masked loads "pending" for hipcc
      lane_setup();
      body(T_{}, F_{}, PD{}, PD{}); step();
      body(F_{}, T_{}, PD{}, Z{}); step();
#pragma unroll 1
      for (int kt = 2; kt < nt; ++kt) { body(F_{}, T_{}, Z{}, Z{}); step(); }
      tile = tnext;
      tnext = next_tile(tile);
    }
    {   // last tile of the pass: nothing left to overlap with
      const int em0 = tile.tm * BMT, en0 = tile.tn * C8_BN;
      if (wr == 0) C8_BARRIER();   // rejoin (equal barrier counts)
      if constexpr (C8_DEBUG == 1) keep_acc(); else
      epilogue_regs<WANT ? 1 : 0, EC, HALF ? 4 : 8>(p, acc, em0 + wr * WROWS, en0 + wc * 64, WANT, fq, fr);
      if constexpr (!WANT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
  };
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

// The HALF form (128 x 256 tiles) has kernels for the classes that meet narrow outputs in a training step (plain bf16 store, fp32
// residual, QKV + RoPE, SwiGLU forward / backward).  RSYS_GEMM8C_HALF: 1 (default) = when the output has fewer 256 x 256 tiles than the chip
// has CUs, 0 = never, 2 = wherever a kernel exists (tests: both forms on every shape).
static bool c8_half_class(const GemmParams& p) {
  switch (p.epi) {
    case EPI_STORE: return !p.c_f32;
    case EPI_RESIDUAL: case EPI_QKV_ROPE: case EPI_SWIGLU: case EPI_SWIGLU_BWD: return true;
    default: return false;
  }
}
bool gemm8c_uses_half(const GemmParams& p, int cus) {
  const int mode = sw().gemm8c_half;
  if (mode == 0 || !c8_half_class(p)) return false;
  if (mode == 2) return true;
  const long long t256 = (long long)((p.M + C8_BM - 1) / C8_BM) * ((p.N + C8_BN - 1) / C8_BN);
  // ... and the SwiGLU backward at K <= 256 whatever its tile count: four K tiles of MFMAs against an epilogue that reads and writes 256 KB per
  // tile -- half tiles interleave the two finer across the chip (cfg-2's w2_dx 0.475 -> 0.418 ms per step; at K = 512 every class LOSES
  // 8 - 25 % in the HALF form, profiles/r6k_*)
  return t256 < cus || (p.epi == EPI_SWIGLU_BWD && p.K <= 256);
}

int launch_gemm8c(const GemmParams& p0, hipStream_t s) {
  if (gemm4p_takes(p0)) return launch_gemm4p(p0, s);
  GemmParams p = p0;
  { const int mode = sw().gemm_reverse; if (mode == 2) p.flags |= 256; else if (mode == 0) p.flags &= ~256; }   // (bit 8: tile rows walked from the last to the first)
  int cus = 256;
  { static int n = 0; if (n == 0) { int dev = 0, v = 0; n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; } cus = n; }
  const bool half = gemm8c_uses_half(p, cus);
  const int bmt = half ? 128 : C8_BM;
  const int tiles = ((p.M + bmt - 1) / bmt) * ((p.N + C8_BN - 1) / C8_BN);
  const dim3 grid((p.flags & 2) && p.m_dev == nullptr ? tiles : std::min(tiles, cus)), blk(512);
  {
    // RSYS_GEMM_PATCH=0: row-major tile order everywhere (A/B), 2: band order everywhere (the GEMM tests compare the orders)
    const int mode = sw().gemm_patch, rows = 4;
    const int tiles_n = (p.N + C8_BN - 1) / C8_BN;
    // wide AND tall outputs only: 4096 x 120000 (16 tile rows: every XCD already holds all of A) measured 6 % slower in band order
    if (mode == 2 || (mode == 1 && tiles_n > 8 && tiles >= 32 * tiles_n)) p.flags |= rows << 3;
  }
  switch (p.epi) {
    case EPI_STORE:
      if (p.c_f32) hipLaunchKernelGGL((gemm8c_kernel<EPI_STORE, true>), grid, blk, 0, s, p);
      else if (half) hipLaunchKernelGGL((gemm8c_kernel<EPI_STORE, false, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_STORE, false>), grid, blk, 0, s, p);
      break;
    case EPI_SWIGLU:
      if (half) hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU, false, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU, false>), grid, blk, 0, s, p);
      break;
    case EPI_RESIDUAL:
      if (half) hipLaunchKernelGGL((gemm8c_kernel<EPI_RESIDUAL, true, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_RESIDUAL, true>), grid, blk, 0, s, p);
      break;
    case EPI_ACCUM: hipLaunchKernelGGL((gemm8c_kernel<EPI_ACCUM, true>), grid, blk, 0, s, p); break;
    case EPI_SWIGLU_BWD:
      if (half) hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU_BWD, false, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_SWIGLU_BWD, false>), grid, blk, 0, s, p);
      break;
    case EPI_TABLE: hipLaunchKernelGGL((gemm8c_kernel<EPI_TABLE, true>), grid, blk, 0, s, p); break;
    case EPI_QKV_ROPE:
      if (half) hipLaunchKernelGGL((gemm8c_kernel<EPI_QKV_ROPE, false, true>), grid, blk, 0, s, p);
      else hipLaunchKernelGGL((gemm8c_kernel<EPI_QKV_ROPE, false>), grid, blk, 0, s, p);
      break;
    default: set_error("gemm8c: epilogue class without a kernel"); return RSYS_ERR_ARG;
  }
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
