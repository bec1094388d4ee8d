// bf16 MFMA GEMM for row-major operands, gfx950: C[M,N] = sum_k A[m][k] * B[n][k].
//
// 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave = 8x4 MFMA 16x16x32 tiles),
// K in tiles of 64.  Operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers), two 64 KB
// LDS buffers, one workgroup per CU.  A K tile is consumed in four phases, one 64x32 quadrant of the wave's
// output each; every phase issues its LDS reads and one 16 KB half-tile of DMA, then the 16 MFMAs between two raw
// s_barriers.  The two wave rows run one barrier apart, so on every SIMD one wave is in its MFMA section while the
// other issues loads (cdna_hip_programming.md 5, "256^2 8-phase": two K tiles = 8 phases per loop trip).
//
// Half-tiles: "A h" = rows {wr*128 + h*64 + [0,64)} for both wave rows (exactly what phase reads "A h" touch),
// "B h" = columns {wc*64 + h*32 + [0,32)} for the four wave columns.  Schedule of tile t (buffer t & 1):
//   P1: read B0, A0     DMA B1(t+1)                 MFMA A0 x B0
//   P2: read B1         DMA A1(t+1)   vmcnt(8)      MFMA A0 x B1
//   P3: read A1         DMA B0(t+2)                 MFMA A1 x B1
//   P4:                 DMA A0(t+2)   vmcnt(6)      MFMA A1 x B0      (B0 stays in registers)
// WAR: a half-tile is overwritten two or more phases after its last read (the waves of the other row may still have
// that read in flight one phase later).  RAW: the counted vmcnt sits before the first barrier of a phase and the
// data is read from the next phase on: P2's wait retires A1(t) (read in P3), P4's retires B0, A0, B1 of t+1.
// LDS image of a half-tile: 128 rows of 128 B, 16-byte chunk c of local row r at r*128 + ((c ^ ((r>>1)&7)) << 4):
// the 16 lanes of a ds_read_b128 group (16 rows, one chunk) cover all 64 banks.  LDS-DMA writes lane-linear, so
// the permutation is applied to the per-lane SOURCE address.
#include <algorithm>
#include <vector>

#include "gemm.hpp"
#include "gemm_epi.hpp"
#include "gemm_epi_reg.hpp"

namespace rsys {

#define GLB_AS __attribute__((address_space(1)))

namespace {

constexpr int T8_BM = 256, T8_BN = 256, T8_BK = 64;

typedef __attribute__((ext_vector_type(4))) int i32x4;
// buffer_load_dwordx4 ... lds: LDS[m0-base + lane*16] <- 16 bytes at (descriptor base + voffset).  A wave-uniform
// descriptor (SGPRs) plus a loop-invariant 32-bit per-lane offset: the K loop advances the descriptor base with
// scalar adds and spends no vector registers or VALU work on addresses.
extern "C" __device__ void rsys_raw_buffer_load_lds(i32x4 rsrc, LDS_AS unsigned int* lds, int size, int voffset, int soffset,
                                                    int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ i32x4 make_rsrc(const char* base) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)((a >> 32) & 0xFFFFu));   // stride 0
  r[2] = -1;                                                                       // num_records: 4 GB window
  r[3] = 0x00020000;                                                               // raw buffer, 32-bit data format
  return r;
}
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned int voff, unsigned char* lds) {
  rsys_raw_buffer_load_lds(rsrc, (LDS_AS unsigned int*)lds, 16, (int)voff, 0, 0, 0);
}

// The same instruction behind asm volatile, for the K-major forms: their fragments are read with ds_read_b64_tr_b16 (an intrinsic),
// and hipcc's wait-count pass cannot tell such a read apart from a pending LDS-DMA it knows of, so it put s_waitcnt vmcnt(0) in front
// of the reads of P1, P2 and P3 -- every K tile waited for the half-tile requested one phase earlier (found in the ISA in round 4;
// the row-major form's plain 16-byte LDS loads are told apart and get no such wait).  Here the compiler does not see the DMA at
// all: the counted waits of the K loop are the only ones, and the K loop ends with nothing in flight.  lds: wave-uniform LDS byte address.
// The statement writes M0 and lists it as clobbered: the compiler tracks and merges its own M0 initialisations (LDS-DMA intrinsic, movrel,
// readlane / writelane through M0) and must not take an earlier value of its own for still valid behind this statement.  (clang warns
// that M0 is a reserved register; the clobber is what the warning's note asks to be aware of, and it is the intent.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_asm(i32x4 rsrc, unsigned int voff, unsigned int lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop

#define T8_BARRIER()                         \
  do {                                       \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");          \
  } while (0)

// KM = false: row-major operands (A [M][K], B [N][K]).  KM = true: K-major operands (A [K][M], B [K][N]: the weight
// gradients dW = dY^T X with K = tokens), split-K with fp32 atomics; the LDS image of a half-tile is then [64 k][256 B]
// and the fragments are read with ds_read_b64_tr_b16 (two per fragment).
// Row-major form (KM = false) is PERSISTENT: the grid has at most one workgroup per CU and a workgroup walks a strided
// run of output tiles.  After the last K tile of a tile it first issues the LDS-DMA of the next tile's first two K tiles
// (both LDS buffers are free by then) and only then runs the epilogue, so the next tile's HBM latency hides behind
// the epilogue and the epilogue's stores drain behind the next tile's first MFMAs; the counted vmcnt waits of the
// first K tiles allow for those stores (vector memory operations retire in issue order).  A device-side row count
// (*m_dev, the head GEMMs) just shortens the tile run: no idle workgroups.
// SK (row-major operands only): split-K with the work mapping and the LDS-staged fp32-atomic epilogue of the K-major
// form instead of the persistent tile run -- for products with few output tiles and a very long K whose operands exist
// as row-major (K-contiguous) copies: the metadata-projection gradient on transposed copies of dF and Meta.
// GROUP (K-major form only): the workgroup's (tile, K split) comes from a work list (gemm8p_group_kernel below) instead of the
// block index: many small products in one launch.
// F8 (row-major operands only): 1-byte operands, A e4m3 (1) or e5m2 (2), B e4m3.  A K tile is still 128 bytes per row, i.e. 128
// elements: the LDS image, the DMA pattern and the two 16-byte fragment reads per lane are byte for byte those of the bf16 form;
// the two reads together are the lane's 32 k values of ONE v_mfma_f32_16x16x128_f8f6f4 (both operands use the same k order, so
// which 32 of the 128 a lane group holds does not matter).  The accumulators are multiplied by the descale of the operands'
// tensor-wise scales before the epilogue (GemmParams::f8_*).
// MIX (K-major form only): A is ROW-major ([M][K], as in the KM = false form: same requests, same LDS image, same fragment reads), B stays
// K-major -- the tied head's dEw = dlogits . F with F [V][D]: the gradient rows stream K-contiguous, the table is read as it lies.  Both
// fragment sources hand a lane the same eight k of a 32-wide step, so the two halves combine without a transpose anywhere.
template <bool KM, bool SK, bool GROUP, int F8 = 0, bool MIX = false>
__device__ __forceinline__ void gemm8p_body(const GemmParams& p, const int bid, const int nblk, const int g_tile, const int g_split) {
  static_assert(!(KM && SK), "the K-major form is always split-K");
  static_assert(!GROUP || KM || SK, "grouped launches exist for the split-K forms");
  static_assert(!F8 || !KM, "fp8 operands are row-major");
  static_assert(!MIX || (KM && GROUP), "the mixed-layout form is a K-major split-K form whose (tile, split) comes from its kernel");
  constexpr bool KMA = KM && !MIX, KMB = KM;   // layout of each operand
  constexpr bool PERSIST = !KM && !SK;
  constexpr int KE = F8 ? 128 : T8_BK;     // elements per K tile
  constexpr int ES = F8 ? 1 : 2;           // bytes per element
  __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];   // [buf][A h0 | A h1 | B h0 | B h1] x 16 KB
  const int t = threadIdx.x, l0 = t & 63;
  int l = l0;   // refreshed per tile through an opaque move: nothing derived from it is carried across an epilogue
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  int fq = l >> 4, fr = l & 15;

  // ---- output tiles, XCD-aware: workgroups are dealt round-robin over the 8 XCDs; each XCD owns a contiguous run of
  // tiles (tn fastest) and its workgroups walk that run side by side, so neighbouring tiles share A rows / B rows
  // through the XCD's private L2
  const int tiles_n = (p.N + T8_BN - 1) / T8_BN;
  // K-major store / accumulate form: a device-side K limit (rows of the operands that are live: the head GEMMs over the selected positions)
  [[maybe_unused]] const int keff = (KM && p.k_dev != nullptr) ? __builtin_amdgcn_readfirstlane(max(min(*p.k_dev, p.K), 0)) : p.K;
  int tile, tile_first = 0, tile_end = 0, tile_step = 1, kt0 = 0, nt, split_id = 0;
  if constexpr (PERSIST) {
    const int rows = p.m_dev != nullptr ? min(*p.m_dev, p.M) : p.M;
    const int ntiles = ((rows + T8_BM - 1) / T8_BM) * tiles_n;
    const int xcd = bid & 7, G = nblk;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_end = (xcd < r ? (xcd + 1) * (q + 1) : r * (q + 1) + (xcd + 1 - r) * q);
    tile_step = (G + 7 - xcd) >> 3;                    // workgroups on this XCD
    if (tile >= tile_end) return;                      // uniform: whole workgroup leaves
    tile_first = tile;
    nt = p.K / KE;                                     // launcher: K % 64 (128) == 0, nt >= 2
  } else {
    // split-K (launcher: splitk % 8 == 0): XCD x owns the K splits x, x+8, ...; inside an XCD the tiles of one split
    // vary fastest, so the K-major operand rows of a split are fetched from HBM by one L2 only
    int split;
    if constexpr (GROUP) { tile = g_tile; split = g_split; }
    else {
      const int tiles_m = (p.M + T8_BM - 1) / T8_BM, ntiles = tiles_m * tiles_n;
      const int xcd = bid & 7, local = bid >> 3;
      tile = local % ntiles;
      split = xcd + 8 * (local / ntiles);
      if constexpr (KM) {
        if (p.epi != EPI_ATOMIC) {   // store / accumulate form (one K split, launch_gemm8p_tn_store): XCD x owns a contiguous run of tiles
          const int q = ntiles >> 3, r = ntiles & 7;
          const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
          if (local >= cnt) return;
          tile = first + local;
          split = 0;
        }
      }
      // The ~32 workgroups an XCD runs side by side walk their K range in step and share operand columns through its L2: take the
      // tiles in bands of 8 tile columns (tn fastest inside a band), so that 32 neighbours are 4 x 8 tiles = 12 operand panels rather
      // than 1.5 rows of a wide product (dW2 at the production shape, 8 x 22 tiles: 24 panels; PMC: 9.3 GB fetched for 2.0 GB of operands)
      if (tiles_n > 8) {
        const int full = tiles_n & ~7, band = tiles_m * 8;
        int tm, tn;
        if (tile < tiles_m * full) { const int g = tile / band, r = tile - g * band; tm = r >> 3; tn = g * 8 + (r & 7); }
        else { const int r = tile - tiles_m * full, wd = tiles_n - full; tm = r / wd; tn = full + r - tm * wd; }
        tile = tm * tiles_n + tn;
      }
    }
    split_id = split;
    const int ktiles = (keff + KE - 1) / KE, per = (ktiles + p.splitk - 1) / p.splitk;
    kt0 = split * per;
    nt = min(ktiles, kt0 + per) - kt0;
    if (KM && p.epi != EPI_ATOMIC) nt = max(nt, 1);   // (no live K row: the window below is empty, the tile is stored as zeros)
    if (nt <= 0) return;
  }
  int m0 = (tile / tiles_n) * T8_BM, n0 = (tile % tiles_n) * T8_BN;
  // ---- DMA source offsets (bytes from the operand base, k tile 0).  Instruction j of wave w fills the 1 KB piece
  // (w*2+j) of a half-tile = local rows (w*2+j)*8 + [0,8); lane l -> local row + (l>>3), stored slot l&7.
  unsigned int aoff[2][2], boff[2][2];   // [j][h]
  auto tile_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (!KMA || !KMB) {
      const int lr = (w * 2 + j) * 8 + (l >> 3);
      const int c = (l & 7) ^ ((lr >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int grow = min(m0 + (lr >> 6) * 128 + h * 64 + (lr & 63), p.M - 1);   // clamped rows are never stored
        const int gcol = min(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31), p.N - 1);
        if constexpr (!KMA) aoff[j][h] = (unsigned int)(((long long)grow * p.lda + c * (16 / ES)) * ES);
        if constexpr (!KMB) boff[j][h] = (unsigned int)(((long long)gcol * p.ldb + c * (16 / ES)) * ES);
      }
    }
    if constexpr (KMA || KMB) {
      // K-major: piece (w*2+j) = k rows (w*2+j)*4 + [0,4) of the half-tile, 256 B = eight 16-column blocks per row;
      // lane l -> row + (l>>4), 16-byte chunk l&15.  Block mb of row k is stored at block mb ^ f(k),
      // f(k) = ((k>>3)&1)<<2 | (k&3): the eight rows one half-wave of a transposed read touches land on distinct banks.
      const int krow = (w * 2 + j) * 4 + (l >> 4), ch = l & 15;
      const int f = (((krow >> 3) & 1) << 2) | (krow & 3);
      const int mb = (ch >> 1) ^ f, half = ch & 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ma = min(m0 + (mb >> 2) * 128 + h * 64 + (mb & 3) * 16 + half * 8, p.M - 8);   // (clamped columns are never added)
        const int nb = min(n0 + (mb >> 1) * 64 + h * 32 + (mb & 1) * 16 + half * 8, p.N - 8);
        if constexpr (KMA) aoff[j][h] = (unsigned int)(((long long)krow * p.lda + ma) * 2);
        if constexpr (KMB) boff[j][h] = (unsigned int)(((long long)krow * p.ldb + nb) * 2);
      }
    }
  }
  };
  tile_offsets();
  // Running operand windows (wave-uniform, SGPRs): a_cur / b_cur = base of K tile kt, a_rem / b_rem = bytes from there
  // to the end of the operand (K-major only: rows past K read as zeros, so the K tail of the last split needs no guard).
  // The K loop advances them with two scalar adds per operand instead of rebuilding 64-bit products per DMA.
  const long long stepA = KMA ? (long long)T8_BK * p.lda * 2 : (long long)T8_BK * 2;
  const long long stepB = KMB ? (long long)T8_BK * p.ldb * 2 : (long long)T8_BK * 2;
  const char* a_cur = (const char*)p.A + (long long)kt0 * stepA;
  const char* b_cur = (const char*)p.B + (long long)kt0 * stepB;
  long long a_rem = KMA ? (long long)keff * p.lda * 2 - (long long)kt0 * stepA : 0;
  long long b_rem = KMB ? (long long)keff * p.ldb * 2 - (long long)kt0 * stepB : 0;
  auto window = [&](auto KMX, const char* cur, long long rem, long long step, int d) __attribute__((always_inline)) -> i32x4 {
    i32x4 r = make_rsrc(cur + d * step);
    if constexpr (decltype(KMX)::value) {
      const long long left = rem - d * step;
      r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned int)(left <= 0 ? 0 : (left > 0xFFFFFFFFll ? 0xFFFFFFFFll : left)));
    }
    return r;
  };
  unsigned char* const dma_base = smem + w * 2048;   // + buf*65536 + X*32768 + h*16384 + j*1024
  // stage_x(bo, h, d): half-tile h of K tile (current + d) into the buffer at byte offset bo
  [[maybe_unused]] const unsigned int dma_lds = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)(LDS_AS unsigned char*)dma_base);
  auto stage_a = [&](int bo, auto H, int d) {
    constexpr int h = decltype(H)::value;
    const i32x4 rs = window(std::integral_constant<bool, KMA>{}, a_cur, a_rem, stepA, d);
    if constexpr (KM) {
      const unsigned int at = (unsigned int)__builtin_amdgcn_readfirstlane((int)(dma_lds + bo + h * 16384));
      dma16_asm(rs, aoff[0][h], at);
      dma16_asm(rs, aoff[1][h], at + 1024);
    } else {
      dma16(rs, aoff[0][h], dma_base + bo + h * 16384);
      dma16(rs, aoff[1][h], dma_base + bo + h * 16384 + 1024);
    }
  };
  auto stage_b = [&](int bo, auto H, int d) {
    constexpr int h = decltype(H)::value;
    const i32x4 rs = window(std::integral_constant<bool, KMB>{}, b_cur, b_rem, stepB, d);
    if constexpr (KM) {
      const unsigned int at = (unsigned int)__builtin_amdgcn_readfirstlane((int)(dma_lds + bo + 32768 + h * 16384));
      dma16_asm(rs, boff[0][h], at);
      dma16_asm(rs, boff[1][h], at + 1024);
    } else {
      dma16(rs, boff[0][h], dma_base + bo + 32768 + h * 16384);
      dma16(rs, boff[1][h], dma_base + bo + 32768 + h * 16384 + 1024);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment read offsets: lane (fq, fr) takes chunk kk*4+fq of local row base+fr
  int a_rd0, a_rd1, b_rd0, b_rd1;
  auto lane_offsets = [&]() __attribute__((always_inline)) {
    const int sw0 = ((fq ^ (fr >> 1)) << 4), sw1 = (((4 + fq) ^ (fr >> 1)) << 4);
    a_rd0 = (wr * 64 + fr) * 128 + sw0; a_rd1 = (wr * 64 + fr) * 128 + sw1;
    b_rd0 = 32768 + (wc * 32 + fr) * 128 + sw0; b_rd1 = 32768 + (wc * 32 + fr) * 128 + sw1;
  };
  lane_offsets();
  // K-major: lane (fq, fr) supplies the address of row k = 32 kk + 8 fq + (fr>>2) (+4 for the second read), columns
  // 4 (fr&3) .. +3 of block mb; the 16-lane group receives the block's 4 x 16 piece transposed (column fr on lane fr)
  int a_rdk[4], b_rdk[2];
  {
    const int q = fr >> 2, pp = fr & 3, fK = ((fq & 1) << 2) | q, rowb = (8 * fq + q) * 256 + pp * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_rdk[i] = rowb + (((wr * 4 + i) ^ fK) << 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_rdk[j] = 32768 + rowb + (((wc * 2 + j) ^ fK) << 5);
  }
  auto tr_frag = [&](int off) __attribute__((always_inline)) -> bf16x8 {
    const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(smem + off));
    const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(smem + off + 1024));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  f32x4 acc[8][4];

  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](int bo, int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!KMA) {
        af[i][0] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd0);
        af[i][1] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd1);
      } else {
        af[i][0] = tr_frag(bo + h * 16384 + a_rdk[i]);
        af[i][1] = tr_frag(bo + h * 16384 + 8192 + a_rdk[i]);
      }
    }
  };
  auto read_b = [&](bf16x8(&bf)[2][2], int bo, int h) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (!KMB) {
        bf[j][0] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd0);
        bf[j][1] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd1);
      } else {
        bf[j][0] = tr_frag(bo + h * 16384 + b_rdk[j]);
        bf[j][1] = tr_frag(bo + h * 16384 + 8192 + b_rdk[j]);
      }
    }
  };
#ifndef T8_VARIANT
#define T8_VARIANT 0   // tools/micro/gemm8p_trace.hip experiments: bit 0 = all LDS reads retired before the first MFMA of a phase, bit 1 = no s_setprio, bit 2 = two half-tiles less in flight at the waits
#endif
  auto mma_q = [&](auto IH, auto JH, const bf16x8(&bf)[2][2]) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value;
    if constexpr (T8_VARIANT & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(T8_VARIANT & 2)) __builtin_amdgcn_s_setprio(1);
    static_for<4>([&](auto i) { static_for<2>([&](auto j) {
      if constexpr (F8 != 0) {
        typedef __attribute__((ext_vector_type(8))) int i32x8;
        typedef __attribute__((ext_vector_type(4))) int i32x4v;
        const i32x4v b0 = __builtin_bit_cast(i32x4v, bf[j][0]), b1 = __builtin_bit_cast(i32x4v, bf[j][1]);
        const i32x4v a0 = __builtin_bit_cast(i32x4v, af[i][0]), a1 = __builtin_bit_cast(i32x4v, af[i][1]);
        const i32x8 bb = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7), aa = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        // first operand (cbsz) = the B matrix (weights, e4m3), second (blgp) = the A matrix (e4m3 or e5m2); scales unused (0)
        acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bb, aa, acc[ih * 4 + i][jh * 2 + j], 0, F8 == 2 ? 1 : 0, 0, 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], af[i][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
      }
    }); });
    if constexpr (!(T8_VARIANT & 2)) __builtin_amdgcn_s_setprio(0);
  };

  // One K tile per trip; the buffer index is a run-time offset and the ends of the pipeline are handled by
  // wave-uniform branches around the DMA issue (one loop body: the register allocation of the accumulators is the
  // same for every tile).  K tiles 0 and 1 are complete in the prologue, so tile 0 issues nothing in P1 / P2.
  // Waits: P2 leaves the four half-tiles issued after A1(t) in flight, P4 the three issued after B1(t+1); at the end of
  // K fewer are outstanding.  `pend` = stores of the previous tile's epilogue that were issued after this tile's
  // prologue: they retire after the prologue's DMA and before everything issued later, so the waits that guard
  // prologue data (P2, P4 of K tile 0, P2 of K tile 1) allow for them.
  auto wait_vm = [&](auto BASE, int extra) __attribute__((always_inline)) {
    constexpr int b = decltype(BASE)::value;
    switch (extra) {
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 4) : "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 8) : "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 12) : "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 16) : "memory"); break;
      case 24: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 24) : "memory"); break;
      case 32: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b + 32) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b) : "memory"); break;
    }
  };
  using W0 = std::integral_constant<int, 0>;
  using W2 = std::integral_constant<int, 2>;
  using W6 = std::integral_constant<int, 6>;
  using W8 = std::integral_constant<int, 8>;
  using W10 = std::integral_constant<int, 10>;
  int pend = 0;
  typedef const __attribute__((address_space(4))) float* cfloat_p;   // scalar loads: no vector memory counter involved
  int kseg_next = -1, kseg_idx = 0;
  auto tile_body = [&](int kt) {
    const int bo = (kt & 1) << 16, bn = bo ^ 65536;
    if constexpr (F8 != 0) {
      if (kt == kseg_next) {   // a K segment with another scale begins: bring the sums so far into its units
        const float r = ((cfloat_p)p.f8_desc)[16 + kseg_idx];
        ++kseg_idx;
        // (no run-time index into p: in the grouped kernel p is a local copy, and an indexed member sent the whole struct to scratch
        //  memory with a scratch load -- and the s_waitcnt vmcnt(0) in front of it -- at the head of every K tile)
        const int kb = kseg_idx == 1 ? p.f8_kb[1] : (kseg_idx == 2 ? p.f8_kb[2] : 0);
        kseg_next = kb > 0 ? kb : -1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] *= r;
      }
    }
    // P1
    read_b(bf0, bo, 0);
    read_a(bo, 0);
    if (kt > 0 && kt + 1 < nt) stage_b(bn, I1{}, 1);
    T8_BARRIER();
    mma_q(I0{}, I0{}, bf0);
    T8_BARRIER();
    // P2
    read_b(bf1, bo, 1);
    if (kt > 0 && kt + 1 < nt) stage_a(bn, I1{}, 1);
    if (kt < 2 && pend) { if (kt + 1 < nt) wait_vm(W8{}, pend); else wait_vm(W0{}, pend); }
    else if (kt + 1 < nt) { if constexpr (T8_VARIANT & 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T8_BARRIER();
    mma_q(I0{}, I1{}, bf1);
    T8_BARRIER();
    // P3
    read_a(bo, 1);
    if (kt + 2 < nt) stage_b(bo, I0{}, 2);
    T8_BARRIER();
    mma_q(I1{}, I1{}, bf1);
    T8_BARRIER();
    // P4
    if (kt + 2 < nt) stage_a(bo, I0{}, 2);
    if (kt == 0 && pend) { if (kt + 2 < nt) wait_vm(W6{}, pend); else wait_vm(W2{}, pend); }
    else if (kt + 2 < nt) { if constexpr (T8_VARIANT & 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    T8_BARRIER();
    mma_q(I1{}, I0{}, bf0);
    T8_BARRIER();
    a_cur += stepA; b_cur += stepB; a_rem -= stepA; b_rem -= stepB;
  };
  // (K-major forms, round 4: requesting every half-tile of K tile t+2 as soon as its slot is free for both wave rows -- after the
  // first barrier of P2 / P3 / P4 -- with each wait one phase before the read it guards: 5.5 phases of lead instead of 3-5.  Equal at
  // the production shape's products, 10 % slower in the grouped launch of cfg-3; profiles/r4_kmajor_dma_waits.log.  Not kept.)
  // K tile 0 into buffer 0 and (nt > 1) K tile 1 into buffer 1: 16 DMA instructions per wave
  auto prologue = [&]() __attribute__((always_inline)) {
    stage_b(0, I0{}, 0); stage_a(0, I0{}, 0); stage_b(0, I1{}, 0); stage_a(0, I1{}, 0);
    if (nt > 1) { stage_b(65536, I0{}, 1); stage_a(65536, I0{}, 1); stage_b(65536, I1{}, 1); stage_a(65536, I1{}, 1); }
  };
  // Full tiles and edge tiles of the workgroup's run are processed in two passes, each with its own copy of the loops:
  // in edge tiles the consumers of some epilogue loads are masked stores, which leaves loads "pending" for the
  // compiler's wait-count pass on some paths; inside one loop nest that would put a vmcnt(0) at the head of every
  // K-loop trip.  The edge pass ends each tile with a real s_waitcnt vmcnt(0) instead (and so lets its stores drain).
  auto is_full = [&](int tl) __attribute__((always_inline)) -> bool {
    return (tl / tiles_n) * T8_BM + T8_BM <= p.M && (tl % tiles_n) * T8_BN + T8_BN <= p.N;
  };
#ifdef RSYS_8P_TRACE   // instrumented build of tools/micro/gemm8p_trace.hip: where a workgroup's cycles go, per tile section
  unsigned long long tr_sum[5] = {0, 0, 0, 0, 0}, tr_at = 0, tr_tiles = 0;
#define TR_START() (tr_at = __builtin_amdgcn_s_memtime())
#define TR_MARK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tr_sum[i] += n_ - tr_at; tr_at = n_; } while (0)
#else
#define TR_START() ((void)0)
#define TR_MARK(i) ((void)0)
#endif
  auto run_tiles = [&](auto FULLC) __attribute__((always_inline)) {
  constexpr bool WANT = decltype(FULLC)::value;
  if constexpr (PERSIST) {
    tile = tile_first;
    while (tile < tile_end && is_full(tile) != WANT) tile += tile_step;
    if (tile >= tile_end) return;
    m0 = (tile / tiles_n) * T8_BM; n0 = (tile % tiles_n) * T8_BN;
    a_cur = (const char*)p.A; b_cur = (const char*)p.B;
    tile_offsets();
  }
  pend = 0;
  if constexpr (PERSIST) {
    if (p.flags & 4) {   // timing experiment: every other workgroup of an XCD starts p.T microsecond-ish naps late
      if ((bid >> 3) & 1) for (int k = 0; k < p.T; ++k) __builtin_amdgcn_s_sleep(32);
    }
  }
  prologue();
  TR_START();

  for (;;) {   // tiles of this pass (one trip for the K-major form)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  kseg_next = (F8 != 0 && p.f8_kb[0] > 0) ? p.f8_kb[0] : -1; kseg_idx = 0;   // (-1: no boundary ahead)
  // B0, A0, B1 of K tile 0 have landed
  if (nt > 1) wait_vm(W10{}, pend); else wait_vm(W2{}, pend);
  T8_BARRIER();
  if (wr == 1) T8_BARRIER();   // the second wave row runs one barrier behind the first
  TR_MARK(0);

#pragma unroll 1
  for (int kt = 0; kt < nt; ++kt) tile_body(kt);
  if (wr == 0) T8_BARRIER();   // rejoin (equal barrier counts): every wave is done with both LDS buffers
  TR_MARK(1);

  // ---- next tile: its first two K tiles are requested before this tile's results are written
  const int em0 = m0, en0 = n0;
  bool more = false;
  if constexpr (PERSIST) {
    tile += tile_step;
    while (tile < tile_end && is_full(tile) != WANT) tile += tile_step;
    more = tile < tile_end;
    if (more) {
      m0 = (tile / tiles_n) * T8_BM; n0 = (tile % tiles_n) * T8_BN;
      a_cur = (const char*)p.A; b_cur = (const char*)p.B;
      tile_offsets();
      prologue();
    }
  }
  TR_MARK(2);
  // ------------------------------------------------------------------ epilogue (gemm_epi_reg.hpp)
  if (p.epi == 99) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = acc[7][3][3] + acc[3][1][2]; if (!more) return; l = l0; asm volatile("" : "+v"(l)); fq = l >> 4; fr = l & 15; lane_offsets(); tile_offsets(); continue; }   // timing experiment: no epilogue
  if constexpr (F8 != 0 && PERSIST) {
    const cfloat_p d = (cfloat_p)p.f8_desc;
    const int seg = p.f8_seg_cols > 0 ? (en0 + wc * 64) / p.f8_seg_cols : 0;   // (segments are multiples of 64 columns: uniform per wave)
    const float ce = d[p.f8_alt ? 0 : seg], co = d[p.f8_alt ? 1 : seg];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i][0] *= ce; acc[i][1] *= co; acc[i][2] *= ce; acc[i][3] *= co; }
  }
  if constexpr (PERSIST) {
    epilogue_regs<WANT ? 1 : 0>(p, acc, em0 + wr * 128, en0 + wc * 64, WANT, fq, fr);
    // A lower bound on the vector memory instructions of the epilogue that follow its last load: all stores where the
    // epilogue has no loads, else the stores of its last two row blocks (the last operand request precedes them).  A
    // smaller number only makes the next tile wait for more of the stores than it must.
    pend = 0;
    if constexpr (WANT) {
      if (more && !(p.flags & 1)) {
        switch (p.epi) {
          case EPI_STORE: pend = p.c_f32 ? 32 : 16; break;
          case EPI_SWIGLU: pend = 24; break;
          case EPI_ACCUM: case EPI_RESIDUAL: case EPI_SWIGLU_BWD: pend = 8; break;
          case EPI_TABLE: pend = 12; break;
          case EPI_QKV_ROPE: pend = p.c_f32 ? 8 : 4; break;
          default: break;
        }
      }
    } else __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  } else {
    // split-K partial sums: fp32 atomics.  An atomic wave instruction runs at full rate only when its 64 lanes add 256
    // contiguous bytes, so each wave passes its 128x64 block through a private LDS patch, 32 rows at a time
    // ([32][68] f32; the operand tiles are dead: every wave is past the last barrier and no DMA is in flight).
    constexpr int CS_LD = 68;
    float* const Cs = (float*)(smem + w * (32 * CS_LD * 4));
    float* const C = (float*)p.C;
    const int wm0 = em0 + wr * 128, wn0 = en0 + wc * 64;
    const int col = wn0 + l;
    [[maybe_unused]] const cfloat_p f8d = (cfloat_p)p.f8_desc;
    auto stage_acc = [&](auto Q) __attribute__((always_inline)) {
      constexpr int q = decltype(Q)::value;
      static_for<2>([&](auto ii) { static_for<4>([&](auto j) {
        *(f32x4*)&Cs[(ii * 16 + fr) * CS_LD + j * 16 + 4 * fq] = acc[q * 2 + ii][j];   // lane: 4 consecutive columns of one row
      }); });
    };
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      switch (q) {
        case 0: stage_acc(std::integral_constant<int, 0>{}); break;
        case 1: stage_acc(std::integral_constant<int, 1>{}); break;
        case 2: stage_acc(std::integral_constant<int, 2>{}); break;
        default: stage_acc(std::integral_constant<int, 3>{}); break;
      }
      if (col < p.N) {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          const int row = wm0 + q * 32 + r;
          if (row < p.M) {
            float v = p.alpha * Cs[r * CS_LD + l];
            int orow = row;
            if constexpr (F8 != 0) {   // descale of the row's gradient tensor; de-interleaved rows back to W13's row blocks
              v *= f8d[p.f8_rseg > 0 ? row / p.f8_rseg : 0];
              if (p.f8_rowmode == 1) { const int half = p.M >> 1, isb = row >= half ? 1 : 0, i = row - isb * half; orow = ((i >> 4) << 5) + (isb << 4) + (i & 15); }
            }
            if (p.slab != nullptr) p.slab[((long long)split_id * p.M + row) * p.N + col] = v;   // deterministic mode (gemm.hpp)
            else if (KM && p.epi == EPI_STORE) C[(long long)orow * p.ldc + col] = v;        // (one K split: 256 contiguous bytes per wave instruction)
            else if (KM && p.epi == EPI_ACCUM) C[(long long)orow * p.ldc + col] += v;
            else atomicAdd(C + (long long)orow * p.ldc + col, v);
          }
        }
      }
    }
  }
  TR_MARK(3);
#ifdef RSYS_8P_TRACE
  ++tr_tiles;
#endif
  if (!more) break;
  // (the DMA offsets of the next tile were consumed by its prologue above; recompute them and the LDS read offsets from
  // a fresh copy of the lane id instead of holding 12 registers through the epilogue)
  l = l0; asm volatile("" : "+v"(l));
  fq = l >> 4; fr = l & 15;
  lane_offsets();
  tile_offsets();
  TR_MARK(4);
  }
  };
  if constexpr (!PERSIST) run_tiles(std::true_type{});
  else { run_tiles(std::true_type{}); run_tiles(std::false_type{}); }
#ifdef RSYS_8P_TRACE
  if (p.trace != nullptr && t == 0) {
    for (int i = 0; i < 5; ++i) p.trace[bid * 8 + i] = tr_sum[i];
    p.trace[bid * 8 + 5] = tr_tiles;
  }
#endif
}

template <bool KM, bool SK = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmParams p) {
  gemm8p_body<KM, SK, false>(p, blockIdx.x, gridDim.x, 0, 0);
}

template <int F8>
__global__ __launch_bounds__(512) void gemm8p_f8_kernel(GemmParams p) {
  gemm8p_body<false, false, false, F8>(p, blockIdx.x, gridDim.x, 0, 0);
}
// fp8 split-K (A = dY8^T e5m2, B = X8^T e4m3, both K-contiguous): one product per launch / many products per launch
__global__ __launch_bounds__(512) void gemm8p_f8sk_kernel(GemmParams p) {
  gemm8p_body<false, true, false, 2>(p, blockIdx.x, gridDim.x, 0, 0);
}
__global__ __launch_bounds__(512) void gemm8p_group_f8_kernel(const GemmParams* __restrict__ probs, const int* __restrict__ xcd_off,
                                                              const unsigned int* __restrict__ work) {
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int o = xcd_off[xcd];
  if (local >= xcd_off[xcd + 1] - o) return;
  const unsigned int wk = work[o + local];     // product << 24 | split << 16 | tile
  const GemmParams p = probs[wk >> 24];
  gemm8p_body<false, true, true, 2>(p, blockIdx.x, gridDim.x, (int)(wk & 0xFFFFu), (int)((wk >> 16) & 0xFFu));
}

// Grouped K-major split-K launch: the weight gradients dW = dY^T X of MANY layers in one grid.  Each product alone is too small
// for the chip (4 - 22 output tiles of 256^2 at cfg-3), all of them together are not.  The host deals the products to the 8
// XCDs (work lists, gemm8p_group_plan): a workgroup of XCD x takes entry (blockIdx.x >> 3) of list x = (product, K split, tile),
// the tiles of one (product, split) adjacent, so that the K-major operand rows of a product are fetched from HBM by one L2 and
// shared by the product's tiles.  (Workgroups are dealt round-robin over the XCDs; which XCD gets which list is speed only.)
__global__ __launch_bounds__(512) void gemm8p_group_kernel(const GemmParams* __restrict__ probs, const int* __restrict__ xcd_off,
                                                           const unsigned int* __restrict__ work) {
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int o = xcd_off[xcd];
  if (local >= xcd_off[xcd + 1] - o) return;
  const unsigned int wk = work[o + local];     // product << 24 | split << 16 | tile
  const GemmParams p = probs[wk >> 24];
  gemm8p_body<true, false, true>(p, blockIdx.x, gridDim.x, (int)(wk & 0xFFFFu), (int)((wk >> 16) & 0xFFu));
}

// Mixed layouts (gemm8p_body's MIX): C[M][N] += A[M][K] (row-major) . B[K][N] (K-major), split-K with fp32 atomics, over the rows the device says are
// live (*m_dev: the tied head's dEw = dlogits . F over the selected positions).  The grid is fixed (the launcher does not know the row count); the
// kernel deals (tile, K split) pairs itself: XCD x owns the splits x, x + 8, ..., as many per XCD as its workgroups can take at once, the tiles of one
// split side by side (they share the split's K-major rows through that XCD's L2); more tiles than workgroups per XCD: a workgroup takes several pairs.
__global__ __launch_bounds__(512) void gemm8p_mix_kernel(GemmParams p) {
  const int rows = p.m_dev != nullptr ? __builtin_amdgcn_readfirstlane(max(min(*p.m_dev, p.M), 0)) : p.M;
  const int ntiles = ((rows + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  if (ntiles == 0) return;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int S = max(1, G / ntiles);
  GemmParams q = p;
  q.M = rows; q.m_dev = nullptr; q.splitk = 8 * S;
  for (int i = local; i < ntiles * S; i += G) {
    if (i != local) __syncthreads();   // (the previous pair's epilogue staged through the LDS the next prologue fills)
    gemm8p_body<true, false, true, 0, true>(q, blockIdx.x, gridDim.x, i % ntiles, xcd + 8 * (i / ntiles));
  }
}

}  // namespace

// ------------------------------------------------------------------ grouped launch plan
struct GemmGroupPlan {
  void* dev = nullptr;          // [probs | xcd_off | work]
  const GemmParams* d_probs = nullptr; const int* d_off = nullptr; const unsigned int* d_work = nullptr;
  int grid = 0, n = 0, splitk = 1;
  bool f8 = false;
  bool asm4k = false;           // every product can run on gemm4k.hip (bf16, atomic, no slab): launch_gemm8p_group takes its kernel for the same work lists
  double flops = 0.0;
  // ordered (deterministic) form: every product's K splits store their partial tiles into its own slab [split][M][N] (all of them in
  // one allocation, cleared per launch) and one batched kernel adds the splits in index order into C afterwards
  float* slab_all = nullptr; size_t slab_bytes = 0; int max_mn4 = 0;
};

// grid.y = product: C[row][col] += sum over its splits (in index order) of slab[split][row][col]
__global__ void slab_reduce_group_kernel(const GemmParams* __restrict__ probs) {
  const GemmParams& p = probs[blockIdx.y];
  const long long n4 = (long long)p.M * p.N / 4, stride = (long long)p.M * p.N;
  float* C = (float*)p.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 acc = *(const float4*)(p.slab + 4 * i);
    for (int sp = 1; sp < p.splitk; ++sp) {
      const float4 v = *(const float4*)(p.slab + sp * stride + 4 * i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const long long e = 4 * i, row = e / p.N; const int col = (int)(e % p.N);
    float4* c = (float4*)(C + row * p.ldc + col);
    float4 o = *c; o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w; *c = o;
  }
}

bool gemm8p_group_eligible(const GemmParams& p) { return (p.f8 ? gemm8p_f8_splitk_eligible(p) : gemm8p_tn_eligible(p)) && p.slab == nullptr; }

// Deals `n` (<= 128) K-major products (EPI_ATOMIC, fp32 C, zero or accumulating) to the XCDs and uploads the plan.  The plan
// holds device pointers of the operands: it stays valid while those buffers do.  Synchronous (one small H2D copy).
int gemm8p_group_plan_create(const GemmParams* probs, int n, GemmGroupPlan** out, bool ordered) {
  ARG_CHECK(n >= 1 && n <= 128, "grouped GEMM: 1..128 products");
  int cus = 256;
  { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
  const int per_xcd = std::max(1, cus / 8);
  std::vector<int> tiles(n), ktiles(n);
  int ktmax = 0;
  const bool f8 = probs[0].f8 != 0;
  for (int i = 0; i < n; ++i) {
    ARG_CHECK((probs[i].f8 != 0) == f8, "grouped GEMM: fp8 and bf16 products in one plan");
    ARG_CHECK(gemm8p_group_eligible(probs[i]), "grouped GEMM: product not eligible for the split-K LDS-DMA kernel");
    tiles[i] = ((probs[i].M + T8_BM - 1) / T8_BM) * ((probs[i].N + T8_BN - 1) / T8_BN);
    ARG_CHECK(tiles[i] < 65536, "grouped GEMM: too many tiles");
    ktiles[i] = f8 ? probs[i].K / 128 : (probs[i].K + T8_BK - 1) / T8_BK;
    ktmax = std::max(ktmax, ktiles[i]);
  }
  // products to XCDs: largest first onto the least loaded list (work = tiles x K tiles)
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (long long)tiles[a] * ktiles[a] > (long long)tiles[b] * ktiles[b]; });
  std::vector<std::vector<int>> lists(8);
  long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i : order) {
    int best = 0;
    for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
    lists[best].push_back(i); load[best] += (long long)tiles[i] * ktiles[i];
  }
  // one K-split count for the whole group: the one whose slowest XCD finishes first (a workgroup costs its K tiles plus
  // ~24 K-tile times of prologue and 256 KB of atomics; a list runs in rounds of one workgroup per CU)
  const int force = sw().debug_8g_splitk;
  int best_s = 1; double best_t = 1e300;
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && (ktmax + s - 1) / s < 8) break;
    double worst = 0.0;
    for (int x = 0; x < 8; ++x) {
      // items in list order, greedy onto per_xcd slots
      std::vector<double> slot(per_xcd, 0.0);
      for (int i : lists[x]) {
        const int per = (ktiles[i] + s - 1) / s;
        for (int sp = 0; sp < s; ++sp) {
          const int nt = std::min(ktiles[i], (sp + 1) * per) - sp * per;
          if (nt <= 0) continue;
          for (int t = 0; t < tiles[i]; ++t) { auto it = std::min_element(slot.begin(), slot.end()); *it += nt + 24.0; }
        }
      }
      worst = std::max(worst, *std::max_element(slot.begin(), slot.end()));
    }
    if (worst < best_t) { best_t = worst; best_s = s; }
  }
  if (force > 0) best_s = std::min(force, 255);
  std::vector<GemmParams> hp(probs, probs + n);
  std::vector<int> off(9, 0);
  std::vector<unsigned int> work;
  int longest = 0;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) { hp[i].splitk = best_s; if (hp[i].alpha == 0.f) hp[i].alpha = 1.f; flops += 2.0 * hp[i].M * hp[i].N * (double)hp[i].K; }
  float* slab_all = nullptr; size_t slab_bytes = 0; int max_mn4 = 0;
  if (ordered) {
    ARG_CHECK(!f8, "grouped GEMM: the ordered form exists for the bf16 products only");
    long long total = 0;
    for (int i = 0; i < n; ++i) {
      ARG_CHECK(hp[i].N % 4 == 0 && hp[i].ldc % 4 == 0, "grouped GEMM: the ordered split-K sum needs N % 4 == 0");
      total += (long long)best_s * hp[i].M * hp[i].N;
      max_mn4 = std::max(max_mn4, (int)((long long)hp[i].M * hp[i].N / 4));
    }
    slab_bytes = (size_t)total * 4;
    if (hipMalloc((void**)&slab_all, slab_bytes) != hipSuccess) { set_error("grouped GEMM: hipMalloc of the split-K slabs failed"); return RSYS_ERR_HIP; }
    long long at = 0;
    for (int i = 0; i < n; ++i) { hp[i].slab = slab_all + at; hp[i].slab_floats = (long long)best_s * hp[i].M * hp[i].N; at += hp[i].slab_floats; }
  }
  for (int x = 0; x < 8; ++x) {
    off[x] = (int)work.size();
    for (int i : lists[x]) {
      const int per = (ktiles[i] + best_s - 1) / best_s;
      for (int sp = 0; sp < best_s; ++sp) {
        if (sp * per >= ktiles[i]) continue;
        for (int t = 0; t < tiles[i]; ++t) work.push_back(((unsigned int)i << 24) | ((unsigned int)sp << 16) | (unsigned int)t);
      }
    }
    longest = std::max(longest, (int)work.size() - off[x]);
  }
  off[8] = (int)work.size();
  GemmGroupPlan* pl = new GemmGroupPlan();
  const size_t b_probs = (sizeof(GemmParams) * n + 255) / 256 * 256, b_off = 256, b_work = (work.size() * 4 + 255) / 256 * 256;
  pl->slab_all = slab_all; pl->slab_bytes = slab_bytes; pl->max_mn4 = max_mn4;
  if (hipMalloc(&pl->dev, b_probs + b_off + b_work) != hipSuccess) { if (slab_all) hipFree(slab_all); delete pl; set_error("grouped GEMM: hipMalloc failed"); return RSYS_ERR_HIP; }
  pl->d_probs = (const GemmParams*)pl->dev; pl->d_off = (const int*)((char*)pl->dev + b_probs); pl->d_work = (const unsigned int*)((char*)pl->dev + b_probs + b_off);
  bool ok = hipMemcpy((void*)pl->d_probs, hp.data(), sizeof(GemmParams) * n, hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMemcpy((void*)pl->d_off, off.data(), 9 * 4, hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMemcpy((void*)pl->d_work, work.data(), work.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { hipFree(pl->dev); if (slab_all) hipFree(slab_all); delete pl; set_error("grouped GEMM: plan upload failed"); return RSYS_ERR_HIP; }
  pl->grid = 8 * longest; pl->n = n; pl->splitk = best_s; pl->flops = flops; pl->f8 = f8;
  pl->asm4k = !f8;
  for (int i = 0; i < n && pl->asm4k; ++i) pl->asm4k = gemm4k_eligible(hp[i]);
  *out = pl;
  return RSYS_OK;
}
void gemm8p_group_plan_destroy(GemmGroupPlan* pl) { if (pl) { if (pl->dev) hipFree(pl->dev); if (pl->slab_all) hipFree(pl->slab_all); delete pl; } }
double gemm8p_group_flops(const GemmGroupPlan* pl) { return pl->flops; }
bool gemm8p_group_on_4k(const GemmGroupPlan* pl) { return pl->asm4k && !pl->f8 && sw().gemm4k != 0; }   // (the launch goes to gemm4k.hip: its tag says so)
int gemm8p_group_splitk(const GemmGroupPlan* pl) { return pl->splitk; }
int launch_gemm8p_group(const GemmGroupPlan* pl, hipStream_t s) {
  if (pl->grid <= 0) return RSYS_OK;
  if (pl->slab_all) HIP_CHECK(hipMemsetAsync(pl->slab_all, 0, pl->slab_bytes, s));   // (K splits without work leave their part untouched)
  if (pl->f8) hipLaunchKernelGGL(gemm8p_group_f8_kernel, dim3(pl->grid), dim3(512), 0, s, pl->d_probs, pl->d_off, pl->d_work);
  else if (pl->asm4k && sw().gemm4k != 0) return launch_gemm4k_group(pl->d_probs, pl->d_off, pl->d_work, pl->grid, s);
  else hipLaunchKernelGGL(gemm8p_group_kernel, dim3(pl->grid), dim3(512), 0, s, pl->d_probs, pl->d_off, pl->d_work);
  HIP_CHECK(hipGetLastError());
  if (pl->slab_all) {
    hipLaunchKernelGGL(slab_reduce_group_kernel, dim3((unsigned)std::min(std::max((pl->max_mn4 + 255) / 256, 1), 512), pl->n), dim3(256), 0, s, pl->d_probs);
    HIP_CHECK(hipGetLastError());
  }
  return RSYS_OK;
}

bool gemm8p_eligible(const GemmParams& p) {
  if (p.splitk > 1 || p.epi == EPI_ATOMIC || p.k_dev != nullptr || p.accum) return false;
  if (p.K % T8_BK != 0 || p.K < 2 * T8_BK) return false;
  if (p.lda % 8 != 0 || p.ldb % 8 != 0) return false;
  if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldb * 2 >= (1ull << 32)) return false;
  if (p.N % 8 != 0) return false;   // whole 8-column groups per lane in the epilogue
  const unsigned long long lim = 1ull << 32;   // 32-bit byte offsets in the epilogue
  const bool cf = p.c_f32 || p.epi == EPI_ACCUM || p.epi == EPI_RESIDUAL || p.epi == EPI_TABLE;
  if ((unsigned long long)p.M * p.ldc * (cf ? 4 : 2) >= lim) return false;
  if (p.C2 != nullptr && (unsigned long long)p.M * p.ldc2 * 2 >= lim) return false;
  if (p.epi == EPI_RESIDUAL && (unsigned long long)p.M * p.ldr * 4 >= lim) return false;
  if (p.epi == EPI_QKV_ROPE && (p.alpha != 1.f || p.rope_cs == nullptr)) return false;
  if (p.epi == EPI_SWIGLU && (p.N % 32 != 0 || p.ldc2 % 8 != 0)) return false;
  return true;
}

// K-major operands + split-K atomics (weight gradients)
bool gemm8p_tn_eligible(const GemmParams& p) {
  if (p.epi != EPI_ATOMIC || !p.c_f32 || p.k_dev != nullptr || p.m_dev != nullptr) return false;
  if (p.M % 8 != 0 || p.N % 8 != 0 || p.M < 8 || p.N < 8 || p.lda % 8 != 0 || p.ldb % 8 != 0) return false;
  if ((unsigned long long)64 * p.lda * 2 + (unsigned long long)p.M * 2 >= (1ull << 31)) return false;
  if ((unsigned long long)64 * p.ldb * 2 + (unsigned long long)p.N * 2 >= (1ull << 31)) return false;
  return true;
}

// K splits of the two split-K forms (a multiple of 8: one split never straddles XCDs): fill whole rounds of the 256 CUs while
// keeping the K range of a workgroup long against its fixed cost (first tiles from HBM + 256 KB of atomics ~ 16 K tiles)
int gemm8p_splits(const GemmParams& p, bool k_major) {
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  const int ktiles = (p.K + T8_BK - 1) / T8_BK;
  int best = 8; double best_score = -1.0;
  for (int sk = 8; sk <= 256; sk += 8) {
    const int per = (ktiles + sk - 1) / sk;
    if (per < 2 && sk > 8) break;
    const long long wgs = (long long)tiles * sk;
    const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    const double score = eff * per / (per + 16.0);   // (tools/scan_splitk_8t.py: 24 splits 694 TFLOP/s, 40 splits 660 on the dWp shape)
    if (score > best_score) { best_score = score; best = sk; }
  }
  const int force = sw().debug_8t_splitk;   // scans (tools/)
  return (k_major && force > 0) ? (force + 7) / 8 * 8 : best;
}

// K-major operands, ONE K split, fp32 output stored or accumulated with plain memory operations (the tied head's table gradient
// dF[v][:] (+)= sum_rows dlogits[row][v] Ew[row][:]: few hundred live rows -- a device-side K limit -- against 10^5 output rows)
bool gemm8p_tn_store_eligible(const GemmParams& p) {
  if ((p.epi != EPI_STORE && p.epi != EPI_ACCUM) || !p.c_f32 || p.m_dev != nullptr || p.splitk > 1 || p.alpha != 1.f || p.f8 != 0) return false;
  if (p.M % 8 != 0 || p.N % 8 != 0 || p.M < 8 || p.N < 8 || p.lda % 8 != 0 || p.ldb % 8 != 0) return false;
  if ((unsigned long long)64 * p.lda * 2 + (unsigned long long)p.M * 2 >= (1ull << 31)) return false;
  if ((unsigned long long)64 * p.ldb * 2 + (unsigned long long)p.N * 2 >= (1ull << 31)) return false;
  if ((unsigned long long)p.K * p.lda * 2 >= (1ull << 40) || (unsigned long long)p.K * p.ldb * 2 >= (1ull << 40)) return false;
  return true;
}
int launch_gemm8p_tn_store(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.splitk = 1; p.slab = nullptr;
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  hipLaunchKernelGGL(gemm8p_kernel<true>, dim3(8 * ((tiles + 7) / 8)), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

int launch_gemm8p_tn(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  p.splitk = gemm8p_splits(p, true);
  if (gemm4k_eligible(p)) return launch_gemm4k(p, tiles * p.splitk, s);   // (no slab: the four-wave register-named loop, same grid and work mapping)
  { const int rc_ = gemm_slab_begin(p, s); if (rc_ != RSYS_OK) return rc_; }
  hipLaunchKernelGGL(gemm8p_kernel<true>, dim3(tiles * p.splitk), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return gemm_slab_end(p, s);
}

// row-major operands + split-K atomics (see the SK template parameter)
bool gemm8p_nt_splitk_eligible(const GemmParams& p) {
  if (p.epi != EPI_ATOMIC || !p.c_f32 || p.k_dev != nullptr || p.m_dev != nullptr) return false;
  if (p.K % T8_BK != 0 || p.K < 16 * T8_BK || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.N % 8 != 0) return false;
  if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldb * 2 >= (1ull << 32)) return false;
  return true;
}

int launch_gemm8p_nt_splitk(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  p.splitk = gemm8p_splits(p, false);
  { const int rc_ = gemm_slab_begin(p, s); if (rc_ != RSYS_OK) return rc_; }
  hipLaunchKernelGGL((gemm8p_kernel<false, true>), dim3(tiles * p.splitk), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return gemm_slab_end(p, s);
}

// row-major A, K-major B, split-K atomics (gemm8p_mix_kernel); the caller passes K as a multiple of 64 (a row-major operand has no zero-filled
// K tail: gemm.hip adds the last K % 64 columns with the 128x128 kernel)
bool gemm8p_mix_eligible(const GemmParams& p) {
  if (p.epi != EPI_ATOMIC || !p.c_f32 || p.k_dev != nullptr || p.slab != nullptr || p.f8 != 0) return false;
  if (p.K < 16 * T8_BK || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.N % 8 != 0 || p.N < 8 || p.M < 1) return false;
  if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32)) return false;                                  // row-major A: 32-bit row offsets
  if ((unsigned long long)64 * p.ldb * 2 + (unsigned long long)p.N * 2 >= (1ull << 31)) return false;      // K-major B: a K tile's window
  if ((unsigned long long)p.K * p.ldb * 2 >= (1ull << 40)) return false;
  return true;
}
static int cu_count();
int launch_gemm8p_mix(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  ARG_CHECK(p.K % T8_BK == 0, "gemm8p_mix: K must be a multiple of 64 (the caller splits the tail off)");
  hipLaunchKernelGGL(gemm8p_mix_kernel, dim3(8 * std::max(1, cu_count() / 8)), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// at most one workgroup per CU (128 KB of LDS each): the kernel walks the remaining tiles itself
static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

// fp8 operands: the same eligibility in bytes (K tiles of 128 elements)
bool gemm8p_f8_eligible(const GemmParams& p) {
  if (p.f8 != 1 && p.f8 != 2) return false;
  if (p.f8_desc == nullptr) return false;
  if (p.splitk > 1 || p.epi == EPI_ATOMIC || p.k_dev != nullptr || p.accum) return false;
  if (p.K % 128 != 0 || p.K < 256) return false;
  if (p.lda % 16 != 0 || p.ldb % 16 != 0) return false;
  if ((unsigned long long)p.M * p.lda >= (1ull << 32) || (unsigned long long)p.N * p.ldb >= (1ull << 32)) return false;
  if (p.N % 8 != 0) return false;
  if (p.f8_seg_cols > 0 && p.f8_seg_cols % 64 != 0) return false;
  if (p.f8_alt && p.N % 32 != 0) return false;
  for (int j = 0; j < 3; ++j) if (p.f8_kb[j] < 0 || p.f8_kb[j] >= p.K / 128 || (j > 0 && p.f8_kb[j] > 0 && p.f8_kb[j] <= p.f8_kb[j - 1])) return false;
  if (p.f8_seg_cols > 0 && (p.N + p.f8_seg_cols - 1) / p.f8_seg_cols > 16) return false;
  const unsigned long long lim = 1ull << 32;
  const bool cf = p.c_f32 || p.epi == EPI_ACCUM || p.epi == EPI_RESIDUAL || p.epi == EPI_TABLE;
  if ((unsigned long long)p.M * p.ldc * (cf ? 4 : 2) >= lim) return false;
  if (p.C2 != nullptr && (unsigned long long)p.M * p.ldc2 * 2 >= lim) return false;
  if (p.epi == EPI_RESIDUAL && (unsigned long long)p.M * p.ldr * 4 >= lim) return false;
  if (p.epi == EPI_QKV_ROPE && (p.alpha != 1.f || p.rope_cs == nullptr)) return false;
  if (p.epi == EPI_SWIGLU && (p.N % 32 != 0 || p.ldc2 % 8 != 0)) return false;
  return true;
}

int launch_gemm8p_f8(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  ARG_CHECK(gemm8p_f8_eligible(p), "fp8 GEMM: shape / operand layout not supported by the 256x256 fp8 pipeline (K % 128, K >= 256, strides % 16)");
  if (p.alpha == 0.f) p.alpha = 1.f;
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  const dim3 grid((p.flags & 2) && p.m_dev == nullptr ? tiles : std::min(tiles, cu_count()));
  if (p.f8 == 1) hipLaunchKernelGGL(gemm8p_f8_kernel<1>, grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL(gemm8p_f8_kernel<2>, grid, dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

bool gemm8p_f8_splitk_eligible(const GemmParams& p) {
  if (p.f8 != 2 || p.f8_desc == nullptr) return false;
  if (p.epi != EPI_ATOMIC || !p.c_f32 || p.k_dev != nullptr || p.m_dev != nullptr) return false;
  if (p.K % 128 != 0 || p.K < 256 || p.lda % 16 != 0 || p.ldb % 16 != 0 || p.N % 8 != 0) return false;
  if ((unsigned long long)p.M * p.lda >= (1ull << 32) || (unsigned long long)p.N * p.ldb >= (1ull << 32)) return false;
  if (p.f8_rseg > 0 && (p.M + p.f8_rseg - 1) / p.f8_rseg > 16) return false;
  if (p.f8_rowmode == 1 && p.M % 32 != 0) return false;
  if (p.f8_rowmode != 0 && p.slab != nullptr) return false;   // (the ordered slab sum has no row map)
  return true;
}

int launch_gemm8p_f8_splitk(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  ARG_CHECK(gemm8p_f8_splitk_eligible(p), "fp8 split-K GEMM: shape / operand layout not supported");
  if (p.alpha == 0.f) p.alpha = 1.f;
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  {   // K splits as for the bf16 row-major split-K form, in K tiles of 128
    const int ktiles = p.K / 128;
    int best = 8; double best_score = -1.0;
    for (int sk = 8; sk <= 256; sk += 8) {
      const int per = (ktiles + sk - 1) / sk;
      if (per < 2 && sk > 8) break;
      const long long wgs = (long long)tiles * sk;
      const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
      const double score = eff * per / (per + 16.0);
      if (score > best_score) { best_score = score; best = sk; }
    }
    p.splitk = best;
  }
  { const int rc_ = gemm_slab_begin(p, s); if (rc_ != RSYS_OK) return rc_; }
  hipLaunchKernelGGL(gemm8p_f8sk_kernel, dim3(tiles * p.splitk), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return gemm_slab_end(p, s);
}

// the one-stream form of this pipeline (gemm8c.hip) takes the epilogue classes it has kernels for; RSYS_GEMM8C=0: A/B switch
bool gemm8p_forwards_to_8c(const GemmParams& p) {
  const int dbg = sw().debug_8p;
  const int use_8c = sw().gemm8c;
  return use_8c && dbg == 0 && p.epi != 99 && gemm8c_eligible(p);
}

int launch_gemm8p(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  const int dbg = sw().debug_8p;
  p.flags |= dbg;
  if (gemm8p_forwards_to_8c(p0)) return launch_gemm8c(p0, s);
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  hipLaunchKernelGGL(gemm8p_kernel<false>, dim3((p.flags & 2) && p.m_dev == nullptr ? tiles : std::min(tiles, cu_count())), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
