// (Measurement only since round 4: not part of librsys_hip.so.  tools/micro/gemm8c_dev.hip at commit 278633b timed it against gemm8c.)
// bf16 MFMA GEMM for row-major operands, gfx950: C[M,N] = sum_k A[m][k] * B[n][k] -- the short-K sibling of
// gemm8p.hip.  The training step's GEMMs have K = 512..2816 and write as many bytes as they read: with one
// 256x256 workgroup per CU the prologue (first tiles from HBM) and the epilogue (the output burst) of every tile
// are exposed, about 10 us against 12 us of MFMA work at K = 512.  Here TWO independent workgroups share a CU:
// 256x128 output tile per 256-thread workgroup (4 waves as 2(M) x 2(N), 128x64 per wave = 8x4 MFMA 16x16x32 blocks,
// the same register layout and epilogue as gemm8p), K in tiles of 32, three 24 KB LDS stages filled by LDS-DMA
// (72 KB per workgroup, 256 VGPRs per wave: 2 workgroups = 2 waves per SIMD).  While one workgroup waits for its
// first tiles or drains its outputs, the other one owns the MFMA pipes.
//
// K tile t (stage t % 3):  s_waitcnt vmcnt(6)  -- own DMA of tile t landed, tile t+1 may be in flight
//                          s_barrier           -- everybody's tile t is visible; everybody finished tile t-1
//                          DMA tile t+2 -> stage (t+2) % 3 = (t-1) % 3
//                          12 x ds_read_b128 (4 B + 8 A fragments), 32 MFMA
// LDS image of a stage: rows of 64 B (32 k), 16-byte chunk c of row r at r*64 + ((c ^ ((-(r>>2)) & 3)) << 4): with
// the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) every group covers all 64 banks.  LDS-DMA writes
// lane-linear, so the permutation is applied to the per-lane SOURCE address.
#include "../../recommendersystem_amd/csrc/gemm.hpp"
#include "../../recommendersystem_amd/csrc/gemm_epi.hpp"
#include "../../recommendersystem_amd/csrc/gemm_epi_reg.hpp"

namespace rsys {

namespace {

constexpr int T4_BM = 256, T4_BN = 128, T4_BK = 32, T4_STAGE = 24576;

typedef __attribute__((ext_vector_type(4))) int i32x4;
extern "C" __device__ void rsys4_raw_buffer_load_lds(i32x4 rsrc, LDS_AS unsigned int* lds, int size, int voffset, int soffset,
                                                     int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ i32x4 make_rsrc4(const char* base) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)((a >> 32) & 0xFFFFu));   // stride 0
  r[2] = -1;                                                                       // num_records: 4 GB window
  r[3] = 0x00020000;                                                               // raw buffer, 32-bit data format
  return r;
}
__device__ __forceinline__ void dma16_4(i32x4 rsrc, unsigned int voff, unsigned char* lds) {
  rsys4_raw_buffer_load_lds(rsrc, (LDS_AS unsigned int*)lds, 16, (int)voff, 0, 0, 0);
}

#define T4_BARRIER()                         \
  do {                                       \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");          \
  } while (0)

__global__ __launch_bounds__(256, 2) void gemm4w_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * T4_STAGE];   // [stage][A 256 x 64 B | B 128 x 64 B]
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int fq = l >> 4, fr = l & 15;

  // ---- output tile, XCD-aware (see gemm8p.hip)
  const int tiles_n = (p.N + T4_BN - 1) / T4_BN, tiles_m = (p.M + T4_BM - 1) / T4_BM;
  const int ntiles = tiles_m * tiles_n;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
  }
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * T4_BM, n0 = tn * T4_BN;
  if (p.m_dev != nullptr && m0 >= *p.m_dev) return;   // uniform: whole workgroup leaves
  const int nt = p.K / T4_BK;                          // launcher: K % 32 == 0
#ifdef RSYS_4W_STAGGER   // tools/micro/gemm8c_dev.hip: the second workgroup of every CU starts (flags >> 16) naps late
  if (blockIdx.x < 512 && ((blockIdx.x >> 3) & 32)) for (int k = 0; k < ((p.flags >> 16) & 0xFF); ++k) __builtin_amdgcn_s_sleep(32);
#endif

  // ---- DMA source offsets.  A stage is 24 pieces of 1 KB (16 rows x 64 B): pieces 0..15 = A rows, 16..23 = B rows.
  // Wave w issues pieces w, w+4, ... (6 per K tile: 4 of A, 2 of B); lane l -> row 16 * piece + (l >> 2), slot l & 3.
  unsigned int aoff[4], boff[2];
  {
    const int rr = l >> 2;
    const int c = (l & 3) ^ ((-(rr >> 2)) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int grow = min(m0 + (w + 4 * j) * 16 + rr, p.M - 1);   // clamped rows are never stored
      aoff[j] = (unsigned int)(((long long)grow * p.lda + c * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gcol = min(n0 + (w + 4 * j) * 16 + rr, p.N - 1);
      boff[j] = (unsigned int)(((long long)gcol * p.ldb + c * 8) * 2);
    }
  }
  const char* Ab = (const char*)p.A;
  const char* Bb = (const char*)p.B;
  auto stage_tile = [&](int kt, int so) __attribute__((always_inline)) {   // so: byte offset of the stage
    const i32x4 ra = make_rsrc4(Ab + (long long)kt * (T4_BK * 2));
    const i32x4 rb = make_rsrc4(Bb + (long long)kt * (T4_BK * 2));
    unsigned char* const d = smem + so + w * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16_4(ra, aoff[j], d + j * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16_4(rb, boff[j], d + 16384 + j * 4096);
  };

  // ---- fragment read offsets: lane (fq, fr) takes chunk fq of row base + fr
  const int sw = ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  const int a_rd = (wr * 128 + fr) * 64 + sw;            // block i: + i * 1024
  const int b_rd = 16384 + (wc * 64 + fr) * 64 + sw;     // block j: + j * 1024

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tiles 0 and 1
  stage_tile(0, 0);
  if (nt > 1) stage_tile(1, T4_STAGE);

  int so = 0;   // stage offset of tile kt
#pragma unroll 1
  for (int kt = 0; kt < nt; ++kt) {
    if (kt + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T4_BARRIER();
    const int so2 = so >= T4_STAGE ? so - T4_STAGE : so + 2 * T4_STAGE;   // stage of tile kt + 2 = stage of tile kt - 1
    if (kt + 2 < nt) stage_tile(kt + 2, so2);
    bf16x8 bfr[4], afr[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(smem + so + b_rd + j * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) afr[i] = *(const bf16x8*)(smem + so + a_rd + i * 1024);
    __builtin_amdgcn_s_setprio(1);
    static_for<8>([&](auto i) { static_for<4>([&](auto j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], afr[i], acc[i][j], 0, 0, 0);   // transposed product
    }); });
    __builtin_amdgcn_s_setprio(0);
    so = so + T4_STAGE >= 3 * T4_STAGE ? 0 : so + T4_STAGE;
  }

  // ------------------------------------------------------------------ epilogue (gemm_epi_reg.hpp)
  if (p.epi == 99) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = acc[7][3][3] + acc[3][1][2]; return; }   // timing experiment: no epilogue
  epilogue_regs(p, acc, m0 + wr * 128, n0 + wc * 64, m0 + T4_BM <= p.M && n0 + T4_BN <= p.N, fq, fr);
}

}  // namespace

bool gemm4w_eligible(const GemmParams& p);
int launch_gemm4w(const GemmParams& p, hipStream_t s);
bool gemm4w_eligible(const GemmParams& p) {
  if (p.splitk > 1 || p.epi == EPI_ATOMIC || p.k_dev != nullptr || p.accum) return false;
  if (p.K % T4_BK != 0) return false;
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

int launch_gemm4w(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + T4_BM - 1) / T4_BM) * ((p.N + T4_BN - 1) / T4_BN);
  hipLaunchKernelGGL(gemm4w_kernel, dim3(tiles), dim3(256), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
