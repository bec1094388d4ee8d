// Probe for ADVICE r5 (medium): the epilogue of attn_bwd_q_kernel<bf16,64,2> at 02e2619 (the build that returned a wrong dQ column on lanes
// 48-63 about once in three launches) and the -fno-slp-vectorize build that has never failed differ in packing -- and BOTH overwrite a data
// register of a `ds_write2_b64` with the very next VALU instruction (profiles/r6_packed_f32_epilogue_isa_diff.md: `ds_write2_b64 v6, v[8:9],
// v[2:3]` / `v_pk_mul_f32 v[2:3], ...` in the failing build, `ds_write2_b64 v6, v[8:9], v[10:11]` / `v_mul_f32 v8, ...` in the shipped one).
// LLVM's hazard recogniser inserts a wait state only behind stores of MORE than 64 bits per data operand (ds_write_b128); a ds_write2_b64
// carries two 64-bit operands and gets none.  If the LDS pipe read its second operand late -- the last 16-lane pass last -- a VALU write that
// follows at once could reach the register first: wrong LDS contents on lanes 48-63 only, which is what the failure looked like.
// Every lane stores two known 64-bit values with ds_write2_b64 (or one with ds_write_b64), overwrites a data register in the next
// instruction (packed or scalar VALU, with or without a wait state in between), reads the LDS back and compares with what it meant to
// store; mismatches are counted per lane quarter.  `mfma`: 16 MFMAs per iteration around the sequence (the failing kernel's neighbourhood).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/ds_store_war_probe.hip -o tools/micro/bin/ds_store_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE, bool MFMA>
__global__ __launch_bounds__(256, 3) void probe(int iters, unsigned long long* bad, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 64];
  const int t = threadIdx.x;
  const unsigned int addr = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds + t * 64;
  unsigned int local_bad = 0;
  float live_all = 0.f;
  float2 x = make_float2(1.0f + t * 0.001f, 2.0f - t * 0.002f), y = make_float2(0.5f, -0.25f);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa, fb;
  for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * (t + k)); fb[k] = (__bf16)(0.02f * (k - t)); }
  for (int i = 0; i < iters; ++i) {
    // what is meant to be stored: depends on lane and iteration, never equal to what the overwriting instruction produces
    float2 d0 = make_float2(__uint_as_float(0x3f800000u | ((unsigned)(i * 2654435761u + t) & 0x7fffffu)), __uint_as_float(0x40000000u | ((unsigned)(i * 40503u + 7 * t) & 0x7fffffu)));
    float2 d1 = make_float2(__uint_as_float(0x40400000u | ((unsigned)(i * 69069u + 3 * t) & 0x7fffffu)), __uint_as_float(0x40800000u | ((unsigned)(i * 1103515245u + 11 * t) & 0x7fffffu)));
    const float2 w0 = d0, w1 = d1;
    if constexpr (MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    }
    float4 rr; float live = 0.f;
    // The stored values sit in FIXED registers v[100:103] (clobbered: the compiler keeps out of them), so that the overwriting instruction can
    // name a whole pair or one dword of it; the movs in front are separated from the store by wait states of their own.
#define PROBE_SEQ(OVERWRITE)                                                                                                    \
    asm volatile("v_mov_b32 v100, %[a]\n\tv_mov_b32 v101, %[b]\n\tv_mov_b32 v102, %[c]\n\tv_mov_b32 v103, %[d]\n\ts_nop 7\n\t" OVERWRITE \
                 "\n\ts_nop 7\n\tv_add_f32 %[live], v100, v102\n\t"                                                              \
                 : [live] "=v"(live)                                                                                            \
                 : [a] "v"(d0.x), [b] "v"(d0.y), [c] "v"(d1.x), [d] "v"(d1.y), [addr] "v"(addr), [x] "v"(x), [y] "v"(y), [xs] "v"(x.x), [ys] "v"(y.x) \
                 : "memory", "v100", "v101", "v102", "v103")
    if constexpr (MODE == 0)        // the failing build's pattern: a packed multiply overwrites data1 in the next instruction
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\tv_pk_mul_f32 v[102:103], %[x], %[y]");
    else if constexpr (MODE == 1)   // the shipped build's pattern: a scalar multiply overwrites data0's low dword in the next instruction
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\tv_mul_f32 v100, %[xs], %[ys]");
    else if constexpr (MODE == 2)   // a packed multiply overwrites data0
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\tv_pk_mul_f32 v[100:101], %[x], %[y]");
    else if constexpr (MODE == 3)   // MODE 0 with one wait state in between
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\ts_nop 0\n\tv_pk_mul_f32 v[102:103], %[x], %[y]");
    else if constexpr (MODE == 4)   // two single 64-bit stores, the second one's data overwritten at once
      PROBE_SEQ("ds_write_b64 %[addr], v[100:101]\n\tds_write_b64 %[addr], v[102:103] offset:32\n\tv_pk_mul_f32 v[102:103], %[x], %[y]");
    else if constexpr (MODE == 5)   // a packed FMA overwrites data1 in place (three sources, crossed low half: the failing block's form)
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\tv_pk_fma_f32 v[102:103], %[x], %[y], v[102:103] op_sel:[0,1,0]");
    else                            // MODE 6: a scalar multiply overwrites data1's high dword
      PROBE_SEQ("ds_write2_b64 %[addr], v[100:101], v[102:103] offset1:4\n\tv_mul_f32 v103, %[xs], %[ys]");
#undef PROBE_SEQ
    if constexpr (MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read2_b64 %0, %1 offset1:4\n\ts_waitcnt lgkmcnt(0)\n\t" : "=&v"(rr) : "v"(addr) : "memory");
    const bool ok = __float_as_uint(rr.x) == __float_as_uint(w0.x) && __float_as_uint(rr.y) == __float_as_uint(w0.y) &&
                    __float_as_uint(rr.z) == __float_as_uint(w1.x) && __float_as_uint(rr.w) == __float_as_uint(w1.y);
    local_bad += ok ? 0u : 1u;
    live_all += live;
  }
  if (local_bad) atomicAdd(&bad[(t & 63) >> 4], (unsigned long long)local_bad);
  if (acc[0] == 12345.678f || live_all == 98765.4321f) sink[0] = acc[1] + x.x + live_all;
}

template <int MODE>
static void run(const char* what, int blocks, int iters, int launches, bool mfma, unsigned long long* bad, float* sink) {
  hipMemset(bad, 0, 32);
  for (int l = 0; l < launches; ++l) {
    if (mfma) hipLaunchKernelGGL((probe<MODE, true>), dim3(blocks), dim3(256), 0, 0, iters, bad, sink);
    else hipLaunchKernelGGL((probe<MODE, false>), dim3(blocks), dim3(256), 0, 0, iters, bad, sink);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", what); exit(3); }
  unsigned long long hb[4];
  hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
  printf("%-92s %s  stores %.2e  mismatches by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n", what, mfma ? "mfma " : "plain",
         (double)blocks * 256 * iters * launches, hb[0], hb[1], hb[2], hb[3]);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 2048, launches = argc > 3 ? atoi(argv[3]) : 2;
  unsigned long long* bad; float* sink;
  if (hipMalloc(&bad, 32) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 2; }
  for (int m = 0; m < 2; ++m) {
    run<0>("ds_write2_b64 d0, d1 ; v_pk_mul_f32 d1 (failing build's pattern)", blocks, iters, launches, m, bad, sink);
    run<5>("ds_write2_b64 d0, d1 ; v_pk_fma_f32 d1 op_sel:[0,1,0] in place", blocks, iters, launches, m, bad, sink);
    run<2>("ds_write2_b64 d0, d1 ; v_pk_mul_f32 d0", blocks, iters, launches, m, bad, sink);
    run<1>("ds_write2_b64 d0, d1 ; v_mul_f32 d0.lo (shipped build's pattern)", blocks, iters, launches, m, bad, sink);
    run<4>("ds_write_b64 d0 ; ds_write_b64 d1 ; v_pk_mul_f32 d1", blocks, iters, launches, m, bad, sink);
    run<6>("ds_write2_b64 d0, d1 ; v_mul_f32 d1.hi", blocks, iters, launches, m, bad, sink);
    run<3>("ds_write2_b64 d0, d1 ; s_nop 0 ; v_pk_mul_f32 d1", blocks, iters, launches, m, bad, sink);
  }
  return 0;
}
