// MEASUREMENT ONLY (tools/micro/gemm4h_dev.hip; not part of the library): bit-identical to gemm8c.hip and 11-33 % SLOWER on the step's
// shapes (profiles/r4_gemm4h_two_workgroups_per_cu.log) -- two independent workgroups per CU do not overlap one's output stream with the
// other's K loop either: identical workgroups started together stay in phase, and the 256 x 128 tile needs 1.5x the LDS fill per FLOP.
//
// bf16 MFMA GEMM for row-major operands whose OUTPUT STREAM is most of the launch (gfx950): C[M,N] = sum_k A[m][k] * B[n][k].
//
// gemm8c.hip runs one 8-wave workgroup per CU: its K loop (1.3-1.5 PFLOP/s) and its register epilogue (the output bytes at the chip's
// HBM copy rate) take turns, and nothing inside one wave can overlap them (vector memory retires in issue order: DESIGN 4c).  Here a
// workgroup is HALF of that -- four waves, a 256 x 128 tile, the same 128 x 64 block and the same register epilogue per wave -- so
// two INDEPENDENT workgroups share a CU (2 x 72 KB of LDS, 2 waves per SIMD): while one streams its tile out, the other runs its K
// loop.  The price is the tile's arithmetic intensity (85 instead of 128 FLOP per byte of LDS fill), so the dispatcher picks this
// kernel only where the epilogue's bytes outweigh the K loop (launch_gemm4h / gemm4h_eligible).
//
// K in stages of 32 (one MFMA k-step): A 256 rows x 64 B + B 128 rows x 64 B = 24 KB per stage, three stages.  A stage is filled by
// LDS-DMA, wave w bringing rows 64 w .. + 63 of A (4 instructions of 16 rows) and rows 32 w .. + 31 of B (2); a 64-byte row keeps
// 16-byte chunk c in slot c ^ 2 ((row >> 2) & 1), which makes the ds_read_b128 fragments conflict-free (the swizzle is applied on the
// source side).  Per stage and wave: wait for the stage (counted vmcnt), barrier, request stage + 2, 12 fragment reads, 32 MFMAs.
#include <algorithm>

#include "../../recommendersystem_amd/csrc/gemm.hpp"
#include "../../recommendersystem_amd/csrc/gemm_epi.hpp"
#include "../../recommendersystem_amd/csrc/gemm_epi_reg.hpp"

namespace rsys {

namespace {

typedef __attribute__((ext_vector_type(4))) int h4_i32x4;
constexpr int H4_BM = 256, H4_BN = 128, H4_BK = 32, H4_ST = 3;
constexpr int H4_A_BYTES = H4_BM * 64, H4_B_BYTES = H4_BN * 64, H4_STAGE = H4_A_BYTES + H4_B_BYTES;   // 16 KB + 8 KB

// The DMA behind an asm statement: the compiler's wait-count pass would order every LDS read it cannot tell apart from a pending LDS-DMA
// behind it (attention.hip, dma16_asm); the kernel keeps its own counted waits.  lds: wave-uniform LDS byte address.
__device__ __forceinline__ void h4_dma(h4_i32x4 rsrc, unsigned int lds, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
template <int N> __device__ __forceinline__ void h4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EC>
__global__ __launch_bounds__(256, 2) void gemm4h_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 1, wc = w & 1, fq = l >> 4, fr = l & 15;
  const int tiles_m = (p.M + H4_BM - 1) / H4_BM, tiles_n = (p.N + H4_BN - 1) / H4_BN;
  // tile of this workgroup: the tiles of one row block of A stay on one XCD (its A rows and the whole of B come from that L2)
  int tm, tn;
  {
    const int bid = blockIdx.x;
    if ((tiles_m & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; tm = (slot / tiles_n) * 8 + xcd; tn = slot % tiles_n; }
    else { tm = bid / tiles_n; tn = bid % tiles_n; }
  }
  // (probe: the second workgroup of every CU starts late by flags >> 16 naps of ~4 us, so that the two slots of a CU run out of phase)
  if (blockIdx.x >= 256 && blockIdx.x < 512) for (int i = 0; i < (p.flags >> 16); ++i) __builtin_amdgcn_s_sleep(127);
  const int nk = p.K / H4_BK;
  // descriptors over the whole operands: rows past M / N read as zeros (their outputs are masked by the edge epilogue)
  auto rsrc_of = [&](const void* base, long long bytes) __attribute__((always_inline)) -> h4_i32x4 {
    const unsigned long long a = (unsigned long long)base;
    h4_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned int)((a >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned int)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const h4_i32x4 a_rs = rsrc_of(p.A, (long long)p.M * p.lda * 2), b_rs = rsrc_of(p.B, (long long)p.N * p.ldb * 2);
  // per-lane source offsets inside a 16-row DMA piece: lane -> row l >> 2, slot l & 3 holds chunk (l & 3) ^ 2 ((l >> 4) & 1)
  const int sw_l = (l & 3) ^ (((l >> 4) & 1) << 1);
  const int a_vo = (int)(((l >> 2) * p.lda + sw_l * 8) * 2), b_vo = (int)(((l >> 2) * p.ldb + sw_l * 8) * 2);
  const unsigned int lds0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)(LDS_AS unsigned char*)smem);
  const int a_so0 = (int)(((long long)(tm * H4_BM + w * 64) * p.lda) * 2), b_so0 = (int)(((long long)(tn * H4_BN + w * 32) * p.ldb) * 2);
  const int a_so16 = (int)(16 * p.lda * 2), b_so16 = (int)(16 * p.ldb * 2);
  auto stage = [&](int kt, int slot) __attribute__((always_inline)) {
    const unsigned int base = lds0 + slot * H4_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) h4_dma(a_rs, base + (w * 64 + i * 16) * 64, a_vo, a_so0 + i * a_so16 + kt * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) h4_dma(b_rs, base + H4_A_BYTES + (w * 32 + j * 16) * 64, b_vo, b_so0 + j * b_so16 + kt * 64);
  };
  // fragment read offsets: lane (fq, fr) takes chunk fq of local row base + fr
  const int sw_r = (fq ^ (((fr >> 2) & 1) << 1)) << 4;
  const int a_rd = (wr * 128 + fr) * 64 + sw_r, b_rd = H4_A_BYTES + (wc * 64 + fr) * 64 + sw_r;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  if (nk > 1) stage(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) h4_wait_vm<6>(); else h4_wait_vm<0>();   // stage kt landed (the six requests of stage kt + 1 may be in flight)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // ... for every wave; and every wave has left stage kt - 1
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) stage(kt + 2, slot == 0 ? 2 : slot - 1);
    const unsigned char* sb = smem + slot * H4_STAGE;
    bf16x8 af[8], bf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = *(const bf16x8*)(sb + b_rd + j * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(sb + a_rd + i * 1024);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    slot = slot == 2 ? 0 : slot + 1;
  }
  const bool full = (tm + 1) * H4_BM <= p.M && (tn + 1) * H4_BN <= p.N;
  if (full) epilogue_regs<1, EC>(p, acc, tm * H4_BM + wr * 128, tn * H4_BN + wc * 64, true, fq, fr);
  else epilogue_regs<0, EC>(p, acc, tm * H4_BM + wr * 128, tn * H4_BN + wc * 64, false, fq, fr);
}

}  // namespace

bool gemm4h_eligible(const GemmParams& p) {
  if (!gemm8c_eligible(p)) return false;      // the same operand layout and epilogue classes
  if (p.m_dev != nullptr || p.k_dev != nullptr) return false;
  if (p.K % H4_BK != 0 || p.K < 2 * H4_BK) return false;
  if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 31) || (unsigned long long)p.N * p.ldb * 2 >= (1ull << 31)) return false;
  return true;
}

int launch_gemm4h(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + H4_BM - 1) / H4_BM) * ((p.N + H4_BN - 1) / H4_BN);
  const size_t sm = (size_t)H4_ST * H4_STAGE;
#define H4_LAUNCH(EC)                                                                                                              \
  do {                                                                                                                             \
    static bool set = false;                                                                                                       \
    if (!set) { HIP_CHECK(hipFuncSetAttribute((const void*)gemm4h_kernel<EC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); set = true; } \
    hipLaunchKernelGGL(gemm4h_kernel<EC>, dim3(tiles), dim3(256), sm, s, p);                                                       \
  } while (0)
  switch (p.epi) {
    case EPI_STORE: H4_LAUNCH(EPI_STORE); break;
    case EPI_SWIGLU: H4_LAUNCH(EPI_SWIGLU); break;
    case EPI_RESIDUAL: H4_LAUNCH(EPI_RESIDUAL); break;
    case EPI_SWIGLU_BWD: H4_LAUNCH(EPI_SWIGLU_BWD); break;
    default: set_error("gemm4h: epilogue class not built"); return RSYS_ERR_ARG;
  }
#undef H4_LAUNCH
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

}  // namespace rsys
