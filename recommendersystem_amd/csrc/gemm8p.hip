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
#include "gemm.hpp"
#include "gemm_epi.hpp"

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

#define T8_BARRIER()                         \
  do {                                       \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");          \
  } while (0)

__global__ __launch_bounds__(512) void gemm8p_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];   // [buf][A h0 | A h1 | B h0 | B h1] x 16 KB
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int fq = l >> 4, fr = l & 15;

  // ---- output tile, XCD-aware: workgroups are dealt round-robin over the 8 XCDs; each XCD takes a contiguous run
  // of tiles (tn fastest), which share A rows / B rows through its private L2
  const int tiles_n = (p.N + T8_BN - 1) / T8_BN, tiles_m = (p.M + T8_BM - 1) / T8_BM;
  const int ntiles = tiles_m * tiles_n;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * T8_BM, n0 = tn * T8_BN;
  if (p.m_dev != nullptr && m0 >= *p.m_dev) return;   // uniform: whole workgroup leaves
  const int nt = p.K / T8_BK;                          // launcher: K % 64 == 0, nt >= 2
  if (p.dbg > 0) {
    // EXPERIMENT: de-synchronise the CUs (odd CUs start half a tile late) so that one half's output bursts overlap the
    // other half's MFMA phases
    const unsigned int cu = __builtin_amdgcn_s_getreg(6660);   // HW_ID.cu_id
    if (blockIdx.x < 256 && (cu & 1)) {
      for (int k = 0; k < p.dbg; ++k) __builtin_amdgcn_s_sleep(127);
    }
  }

  // ---- DMA source offsets (bytes from the operand base, k tile 0).  Instruction j of wave w fills the 1 KB piece
  // (w*2+j) of a half-tile = local rows (w*2+j)*8 + [0,8); lane l -> local row + (l>>3), stored slot l&7.
  unsigned int aoff[2][2], boff[2][2];   // [j][h]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lr = (w * 2 + j) * 8 + (l >> 3);
    const int c = (l & 7) ^ ((lr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int grow = min(m0 + (lr >> 6) * 128 + h * 64 + (lr & 63), p.M - 1);   // clamped rows are never stored
      const int gcol = min(n0 + (lr >> 5) * 64 + h * 32 + (lr & 31), p.N - 1);
      aoff[j][h] = (unsigned int)(((long long)grow * p.lda + c * 8) * 2);
      boff[j][h] = (unsigned int)(((long long)gcol * p.ldb + c * 8) * 2);
    }
  }
  const char* Ab = (const char*)p.A;
  const char* Bb = (const char*)p.B;
  unsigned char* const dma_base = smem + w * 2048;   // + buf*65536 + X*32768 + h*16384 + j*1024
  auto stage_a = [&](int bo, auto H, int kt) {
    constexpr int h = decltype(H)::value;
    const i32x4 rs = make_rsrc(Ab + (long long)kt * (T8_BK * 2));
    dma16(rs, aoff[0][h], dma_base + bo + h * 16384);
    dma16(rs, aoff[1][h], dma_base + bo + h * 16384 + 1024);
  };
  auto stage_b = [&](int bo, auto H, int kt) {
    constexpr int h = decltype(H)::value;
    const i32x4 rs = make_rsrc(Bb + (long long)kt * (T8_BK * 2));
    dma16(rs, boff[0][h], dma_base + bo + 32768 + h * 16384);
    dma16(rs, boff[1][h], dma_base + bo + 32768 + h * 16384 + 1024);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment read offsets: lane (fq, fr) takes chunk kk*4+fq of local row base+fr
  const int sw0 = ((fq ^ (fr >> 1)) << 4), sw1 = (((4 + fq) ^ (fr >> 1)) << 4);
  const int a_rd0 = (wr * 64 + fr) * 128 + sw0, a_rd1 = (wr * 64 + fr) * 128 + sw1;
  const int b_rd0 = 32768 + (wc * 32 + fr) * 128 + sw0, b_rd1 = 32768 + (wc * 32 + fr) * 128 + sw1;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](int bo, int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd0);
      af[i][1] = *(const bf16x8*)(smem + bo + h * 16384 + i * 2048 + a_rd1);
    }
  };
  auto read_b = [&](bf16x8(&bf)[2][2], int bo, int h) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf[j][0] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd0);
      bf[j][1] = *(const bf16x8*)(smem + bo + h * 16384 + j * 2048 + b_rd1);
    }
  };
  auto mma_q = [&](auto IH, auto JH, const bf16x8(&bf)[2][2]) {
    constexpr int ih = decltype(IH)::value, jh = decltype(JH)::value;
    __builtin_amdgcn_s_setprio(1);
    static_for<4>([&](auto i) { static_for<2>([&](auto j) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], af[i][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
    }); });
    __builtin_amdgcn_s_setprio(0);
  };

  // One K tile per trip; the buffer index is a run-time offset and the tail of the pipeline is handled by
  // wave-uniform branches around the DMA issue (one loop body: the register allocation of the accumulators is the
  // same for every tile).  Waits: P2 leaves the four half-tiles issued after A1(t) in flight, P4 the three issued
  // after B1(t+1); at the end of K fewer are outstanding.
  auto tile_body = [&](int kt) {
    const int bo = (kt & 1) << 16, bn = bo ^ 65536;
    // P1
    read_b(bf0, bo, 0);
    read_a(bo, 0);
    if (kt + 1 < nt) stage_b(bn, I1{}, kt + 1);
    T8_BARRIER();
    mma_q(I0{}, I0{}, bf0);
    T8_BARRIER();
    // P2
    read_b(bf1, bo, 1);
    if (kt + 1 < nt) { stage_a(bn, I1{}, kt + 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T8_BARRIER();
    mma_q(I0{}, I1{}, bf1);
    T8_BARRIER();
    // P3
    read_a(bo, 1);
    if (kt + 2 < nt) stage_b(bo, I0{}, kt + 2);
    T8_BARRIER();
    mma_q(I1{}, I1{}, bf1);
    T8_BARRIER();
    // P4
    if (kt + 2 < nt) { stage_a(bo, I0{}, kt + 2); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    T8_BARRIER();
    mma_q(I1{}, I0{}, bf0);
    T8_BARRIER();
  };

  // ---- prologue: tile 0 complete, B0/A0 of tile 1
  stage_b(0, I0{}, 0); stage_a(0, I0{}, 0); stage_b(0, I1{}, 0); stage_a(0, I1{}, 0);
  stage_b(65536, I0{}, 1); stage_a(65536, I0{}, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  T8_BARRIER();
  if (wr == 1) T8_BARRIER();   // the second wave row runs one barrier behind the first

#pragma unroll 1
  for (int kt = 0; kt < nt; ++kt) tile_body(kt);
  if (wr == 0) T8_BARRIER();   // rejoin (equal barrier counts)

  // ------------------------------------------------------------------ epilogue
  // The MFMAs computed the TRANSPOSED product (B fragment as the first operand), so lane (fq, fr) holds, for the
  // 16x16 block (i, j) of its wave's 128x64 output, FOUR CONSECUTIVE COLUMNS of one row:
  //   acc[i][j][r] = C[wm0 + 16 i + fr][wn0 + 16 j + 4 fq + r].
  // Every fused epilogue is lane-local in this layout (RoPE pairs, the [16 a | 16 b] SwiGLU groups = blocks j, j+1)
  // and the results leave straight from registers: f32 outputs as 16-byte stores; bf16 outputs after one
  // v_permlane16_swap per dword between two blocks, which gives every lane 8 consecutive columns (16 bytes,
  // 64-byte row segments per wave instruction).  No LDS round trip, no barrier: the stores are in flight when the
  // workgroup retires and drain under the next workgroup's prologue.
  if (p.epi == 99) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = acc[7][3][3] + acc[3][1][2]; return; }   // timing experiment: no epilogue
  const bool cf32 = p.c_f32 != 0;
  const int wm0 = m0 + wr * 128, wn0 = n0 + wc * 64;

  auto pk2 = [](float a, float b) -> unsigned int {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v; v[0] = (bf16)a; v[1] = (bf16)b;
    return __builtin_bit_cast(unsigned int, v);
  };
  // x: 4 values at columns cx + 4 fq + r, y: 4 values at columns cy + 4 fq + r of row `rowp` (T-typed, element
  // pointer of the row).  After the swaps lane rows (fq) 0/2 hold columns cx + 8 (fq>>1) + [0,8), rows 1/3 the same
  // of cy.  Every lane takes part in the swaps; only the store is guarded.
  // Everything below is straight-line code per epilogue class (no branch around a load): hipcc then keeps exact vmcnt
  // counts, and because operands of row block i+1 are requested BEFORE the stores of row block i are issued, no wait
  // ever has to drain a store (memory operations retire in issue order).  Out-of-range rows / columns are handled by
  // clamping the load addresses and masking the stores; the launcher guarantees N % 8 == 0, so a lane's group of
  // 4 (f32) or 8 (bf16) columns is inside or outside as a whole.
  auto store_pair = [&](bf16* rowp, bool rowok, int cx, int cy, int ncols, const float (&x)[4], const float (&y)[4]) {
    // x: 4 values at columns cx + 4 fq + r, y: at cy + 4 fq + r.  After the swaps lane rows (fq) 0/2 hold columns
    // cx + 8 (fq>>1) + [0,8), rows 1/3 the same of cy.  Every lane takes part in the swaps; only the store is masked.
    unsigned int x0 = pk2(x[0], x[1]), x1 = pk2(x[2], x[3]), y0 = pk2(y[0], y[1]), y1 = pk2(y[2], y[3]);
    auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    const int col = ((fq & 1) ? cy : cx) + 8 * (fq >> 1);
    if (rowok && col < ncols) *(uint4*)(rowp + col) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  };
  auto store_f32 = [&](float* rowp, bool rowok, int col, int ncols, const float (&x)[4]) {
    if (rowok && col < ncols) *(float4*)(rowp + col) = make_float4(x[0], x[1], x[2], x[3]);
  };
  auto ldf4 = [&](const float* rowp, int col, int ncols) -> float4 { return *(const float4*)(rowp + (col < ncols ? col : 0)); };
  auto ldt4 = [&](const bf16* rowp, int col, int ncols) -> bf16x4 { return *(const bf16x4*)(rowp + (col < ncols ? col : 0)); };

  auto run = [&](auto EC) {
    constexpr int ec = decltype(EC)::value;
    const bool outf32 = cf32 || ec == EPI_ACCUM || ec == EPI_RESIDUAL;
    const int cb = wn0 + 4 * fq;   // this lane's first column in block 0; block j adds 16 j
    // per-column vectors, the same for all rows
    float4 bias4[4];
    if constexpr (ec == EPI_BIAS || ec == EPI_GELU || ec == EPI_TABLE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bias4[j] = ldf4(p.bias, cb + 16 * j, p.N);
    }
    const unsigned int rope_p0 = ec == EPI_QKV_ROPE ? (unsigned int)(wm0 + fr) % (unsigned int)p.T : 0u;
    // per-row-block operands, requested one row block ahead (16 registers per stage)
    struct Pre { float4 f[4]; };
    auto request = [&](auto I, Pre& pre) {
      constexpr int i = decltype(I)::value;
      const long long rl = min((long long)(wm0 + i * 16 + fr), (long long)p.M - 1);
      if constexpr (ec == EPI_ACCUM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = ldf4((const float*)p.C + rl * p.ldc, cb + 16 * j, p.N);
      } else if constexpr (ec == EPI_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = ldf4(p.resid + rl * p.ldr, cb + 16 * j, p.N);
      } else if constexpr (ec == EPI_TABLE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pre.f[j] = ldf4(p.E + rl * p.ldc, cb + 16 * j, p.N);
      } else if constexpr (ec == EPI_QKV_ROPE) {
        // (cos, sin) of the lane's two pairs in every block: f[j] = {cos0, cos1, sin0, sin1}
        // position = row % T, without a 64-bit division per row block: one 32-bit modulo per lane (rope_p0), then +16 i
        int pos;
        if (p.rope_pos) pos = p.rope_pos[rl];
        else {
          const unsigned int q = rope_p0 + 16u * i;
          pos = (int)(p.T >= 128 ? (q >= (unsigned int)p.T ? q - (unsigned int)p.T : q) : q % (unsigned int)p.T);
          if (wm0 + i * 16 + fr >= p.M) pos = 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = cb + 16 * j;
          const int cc = col < p.n_q ? col : col - p.n_q;
          const int d2 = (cc & (p.hd - 1)) >> 1;
          const float2 c2 = *(const float2*)(p.rope_cos + pos * (p.hd >> 1) + d2);
          const float2 s2 = *(const float2*)(p.rope_sin + pos * (p.hd >> 1) + d2);
          pre.f[j] = make_float4(c2.x, c2.y, s2.x, s2.y);
        }
      } else if constexpr (ec == EPI_SWIGLU_BWD) {
        // saved a, b of dg column c live at (c>>4)*32 + (c&15) (+16) of the [a|b] rows: f[j] = {a (4 bf16), b (4 bf16)}
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ob = ((wn0 + 16 * j) >> 4) * 32 + 4 * fq;
          const bf16x4 a4 = ldt4((const bf16*)p.C2 + rl * p.ldc2, ob, 2 * p.N);
          const bf16x4 b4 = ldt4((const bf16*)p.C2 + rl * p.ldc2, ob + 16, 2 * p.N);
          pre.f[j] = make_float4(__builtin_bit_cast(float2, a4).x, __builtin_bit_cast(float2, a4).y,
                                 __builtin_bit_cast(float2, b4).x, __builtin_bit_cast(float2, b4).y);
        }
      } else if constexpr (ec == EPI_STORE) {
        if (!outf32 && p.accum) {   // C (T) += result: LoRA updates; f[j] holds the old 4 bf16 values in .x, .y
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x4 o4 = ldt4((const bf16*)p.C + rl * p.ldc, cb + 16 * j, p.N);
            pre.f[j] = make_float4(__builtin_bit_cast(float2, o4).x, __builtin_bit_cast(float2, o4).y, 0.f, 0.f);
          }
        }
      }
    };
    auto unpack4 = [](float lo, float hi, float (&o)[4]) {
      const bf16x4 q = __builtin_bit_cast(bf16x4, make_float2(lo, hi));
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (float)q[r];
    };
    auto finish = [&](auto I, const Pre& pre) {
      constexpr int i = decltype(I)::value;
      const long long row = wm0 + i * 16 + fr;
      const bool rowok = row < p.M;
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r];
      if constexpr (ec == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] *= p.alpha;
        if (!outf32 && p.accum) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float o[4]; unpack4(pre.f[j].x, pre.f[j].y, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] += o[r];
          }
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
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] *= p.alpha;
          if (cb + 16 * j < p.n_q + p.n_k) {
            const float c0 = pre.f[j].x, c1 = pre.f[j].y, s0 = pre.f[j].z, s1 = pre.f[j].w;
            const float a0 = v[j][0] * c0 - v[j][1] * s0, a1 = v[j][0] * s0 + v[j][1] * c0;
            const float a2 = v[j][2] * c1 - v[j][3] * s1, a3 = v[j][2] * s1 + v[j][3] * c1;
            v[j][0] = a0; v[j][1] = a1; v[j][2] = a2; v[j][3] = a3;
          }
        }
        if (!outf32 && p.accum) {   // (LoRA B update of the rotated q / v columns; small problems only)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const long long rl = min(row, (long long)p.M - 1);
            const bf16x4 o4 = ldt4((const bf16*)p.C + rl * p.ldc, cb + 16 * j, p.N);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] += (float)o4[r];
          }
        }
      }

      if constexpr (ec == EPI_SWIGLU_BWD) {
        // acc = dg; (da, db) of one block pair up: 32 consecutive columns of the [a|b] layout
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float av[4], bv[4], da[4], db[4];
          unpack4(pre.f[j].x, pre.f[j].y, av); unpack4(pre.f[j].z, pre.f[j].w, bv);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = 1.f / (1.f + __expf(-av[r]));
            da[r] = v[j][r] * bv[r] * sg * (1.f + av[r] * (1.f - sg));
            db[r] = v[j][r] * av[r] * sg;
          }
          const int ob = ((wn0 + 16 * j) >> 4) * 32;
          store_pair((bf16*)p.C + row * p.ldc, rowok, ob, ob + 16, 2 * p.N, da, db);
        }
      } else if constexpr (ec == EPI_GELU) {
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          float gx[4], gy[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gx[r] = 0.5f * v[jp][r] * (1.f + erff(v[jp][r] * 0.70710678118654752f));
            gy[r] = 0.5f * v[jp + 1][r] * (1.f + erff(v[jp + 1][r] * 0.70710678118654752f));
          }
          store_pair((bf16*)p.C + row * p.ldc, rowok, wn0 + jp * 16, wn0 + jp * 16 + 16, p.N, v[jp], v[jp + 1]);
          store_pair((bf16*)p.C2 + row * p.ldc2, rowok, wn0 + jp * 16, wn0 + jp * 16 + 16, p.N, gx, gy);
        }
      } else if constexpr (ec == EPI_SWIGLU) {
        // blocks (0,1) and (2,3) are [16 a | 16 b] groups: C gets [a|b] as is, C2 the products g = silu(a) * b
        float g0[4], g1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          g0[r] = v[0][r] / (1.f + __expf(-v[0][r])) * v[1][r];
          g1[r] = v[2][r] / (1.f + __expf(-v[2][r])) * v[3][r];
        }
        store_pair((bf16*)p.C + row * p.ldc, rowok, wn0, wn0 + 16, p.N, v[0], v[1]);
        store_pair((bf16*)p.C + row * p.ldc, rowok, wn0 + 32, wn0 + 48, p.N, v[2], v[3]);
        store_pair((bf16*)p.C2 + row * p.ldc2, rowok, (wn0 >> 1), (wn0 >> 1) + 16, p.N >> 1, g0, g1);
      } else if constexpr (ec == EPI_TABLE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store_f32((float*)p.C + row * p.ldc, rowok, cb + 16 * j, p.N, v[j]);
        store_pair((bf16*)p.C2 + row * p.ldc2, rowok, wn0, wn0 + 16, p.N, v[0], v[1]);
        store_pair((bf16*)p.C2 + row * p.ldc2, rowok, wn0 + 32, wn0 + 48, p.N, v[2], v[3]);
      } else {
        if (outf32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) store_f32((float*)p.C + row * p.ldc, rowok, cb + 16 * j, p.N, v[j]);
        } else {
          store_pair((bf16*)p.C + row * p.ldc, rowok, wn0, wn0 + 16, p.N, v[0], v[1]);
          store_pair((bf16*)p.C + row * p.ldc, rowok, wn0 + 32, wn0 + 48, p.N, v[2], v[3]);
        }
      }
    };
    Pre pa, pb;
    request(std::integral_constant<int, 0>{}, pa);
    static_for<4>([&](auto H) {
      constexpr int i = decltype(H)::value * 2;
      request(std::integral_constant<int, i + 1>{}, pb);
      finish(std::integral_constant<int, i>{}, pa);
      if constexpr (i + 2 < 8) request(std::integral_constant<int, i + 2>{}, pa);
      finish(std::integral_constant<int, i + 1>{}, pb);
    });
  };
  switch (p.epi) {
    case EPI_STORE: run(std::integral_constant<int, EPI_STORE>{}); break;
    case EPI_ACCUM: run(std::integral_constant<int, EPI_ACCUM>{}); break;
    case EPI_BIAS: run(std::integral_constant<int, EPI_BIAS>{}); break;
    case EPI_RESIDUAL: run(std::integral_constant<int, EPI_RESIDUAL>{}); break;
    case EPI_QKV_ROPE: run(std::integral_constant<int, EPI_QKV_ROPE>{}); break;
    case EPI_SWIGLU: run(std::integral_constant<int, EPI_SWIGLU>{}); break;
    case EPI_TABLE: run(std::integral_constant<int, EPI_TABLE>{}); break;
    case EPI_GELU: run(std::integral_constant<int, EPI_GELU>{}); break;
    case EPI_SWIGLU_BWD: run(std::integral_constant<int, EPI_SWIGLU_BWD>{}); break;
    default: break;
  }
}

}  // namespace

bool gemm8p_eligible(const GemmParams& p) {
  if (p.splitk > 1 || p.epi == EPI_ATOMIC || p.k_dev != nullptr) return false;
  if (p.K % T8_BK != 0 || p.K < 2 * T8_BK) return false;
  if (p.lda % 8 != 0 || p.ldb % 8 != 0) return false;
  if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldb * 2 >= (1ull << 32)) return false;
  if (p.N % 8 != 0) return false;   // whole 8-column groups per lane in the epilogue
  if (p.epi == EPI_SWIGLU && (p.N % 32 != 0 || p.ldc2 % 8 != 0)) return false;
  return true;
}

int launch_gemm8p(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + T8_BM - 1) / T8_BM) * ((p.N + T8_BN - 1) / T8_BN);
  hipLaunchKernelGGL(gemm8p_kernel, dim3(tiles), dim3(512), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
