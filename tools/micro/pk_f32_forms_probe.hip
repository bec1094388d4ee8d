// Which packed-f32 instruction forms miscompute on MI355X while the SIMD's matrix core is busy?  (round 6)
//
// tools/micro/pk_f32_crossed_probe.hip reproduces the round-4 fault of attn_bwd_q_kernel<bf16,64,2> IN ISOLATION once MFMAs run around the
// packed sequence at three waves per SIMD (profiles/r6_pk_probe_mfma.log: 0 mismatches in 4.3e9 sequences alone, 129 444 with MFMAs, every
// one on lanes 48-63).  This probe takes the sequence apart: one packed instruction per form, operands fresh from global_load_dwordx2,
// compared half by half with the same arithmetic done by scalar v_mul_f32 / v_fma_f32 (one rounding each, as the packed halves);
// mismatches are counted per form, per half (lo / hi) and per lane quarter.  op_sel:[a,b,c] picks the dword (0 = low, 1 = high) of each
// source that the LOW half of the result uses, op_sel_hi:[d,e,f] the dwords the HIGH half uses (defaults [0,0,0] / [1,1,1]).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pk_f32_forms_probe.hip -o tools/micro/bin/pk_f32_forms_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(4))) float probe_f32x4;

struct Form { const char* text; int fma; int sel[3]; int hi[3]; };
__host__ __device__ constexpr Form form_of(int f) {
  switch (f) {
    case 0: return {"v_pk_mul_f32 (plain)", 0, {0, 0, 0}, {1, 1, 1}};
    case 1: return {"v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,1]   (low half reads both high dwords; the T product of the failing block)", 0, {1, 1, 0}, {0, 1, 1}};
    case 2: return {"v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0]   (src1.lo broadcast, src0 swapped)", 0, {1, 0, 0}, {0, 0, 1}};
    case 3: return {"v_pk_mul_f32 op_sel_hi:[0,1]                (src0.lo broadcast: the scale multiply)", 0, {0, 0, 0}, {0, 1, 1}};
    case 4: return {"v_pk_fma_f32 (plain)", 1, {0, 0, 0}, {1, 1, 1}};
    case 5: return {"v_pk_fma_f32 op_sel:[0,1,0]                 (src1.hi broadcast: low half reads a high dword; the failing form)", 1, {0, 1, 0}, {1, 1, 1}};
    case 6: return {"v_pk_fma_f32 op_sel_hi:[1,0,1]              (src1.lo broadcast: high half reads a low dword)", 1, {0, 0, 0}, {1, 0, 1}};
    case 7: return {"v_pk_fma_f32 op_sel:[1,0,0]                 (low half reads src0.hi)", 1, {1, 0, 0}, {1, 1, 1}};
    case 8: return {"v_pk_fma_f32 op_sel:[0,0,1]                 (low half reads src2.hi)", 1, {0, 0, 1}, {1, 1, 1}};
    case 9: return {"v_pk_fma_f32 op_sel:[1,1,1] op_sel_hi:[0,0,0] (both halves crossed)", 1, {1, 1, 1}, {0, 0, 0}};
    // the remaining forms of the SHIPPED library (llvm-objdump of librsys_hip.so: 640 / 463 / 50 / 4 instructions of forms 10 / 3 / 11 / 4)
    case 10: return {"v_pk_mul_f32 V, V, S op_sel_hi:[1,0]        (src1 = SGPR pair, its low dword broadcast)", 0, {0, 0, 0}, {1, 0, 1}};
    default: return {"v_pk_add_f32 (plain)", 2, {0, 0, 0}, {1, 1, 1}};
  }
}
constexpr int NFORMS = 12;

template <int F>
__device__ __forceinline__ float2 packed(float2 a, float2 b, float2 c) {
  float2 d;
  if constexpr (F == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
  else if constexpr (F == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
  else if constexpr (F == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=&v"(d) : "v"(a), "v"(b));
  else if constexpr (F == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
  else if constexpr (F == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else if constexpr (F == 10) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "s"(b));
  else asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float pick(float2 v, int i) { return i ? v.y : v.x; }
template <int F>
__device__ __forceinline__ float2 scalar_ref(float2 a, float2 b, float2 c) {
  constexpr Form f = form_of(F);
  float lo, hi;
  if constexpr (f.fma == 1) {
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(pick(a, f.sel[0])), "v"(pick(b, f.sel[1])), "v"(pick(c, f.sel[2])));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(pick(a, f.hi[0])), "v"(pick(b, f.hi[1])), "v"(pick(c, f.hi[2])));
  } else if constexpr (f.fma == 2) {
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(pick(a, f.sel[0])), "v"(pick(b, f.sel[1])));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(pick(a, f.hi[0])), "v"(pick(b, f.hi[1])));
  } else {
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(pick(a, f.sel[0])), "v"(pick(b, f.sel[1])));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(pick(a, f.hi[0])), "v"(pick(b, f.hi[1])));
  }
  return make_float2(lo, hi);
}

// bad: [form][half][lane quarter]
template <int F, int MFMA>   // 0: no MFMA in the kernel; 1: MFMAs of the SAME wave around the instruction; 2: MFMAs only in the OTHER waves of the SIMD
__global__ __launch_bounds__(256, 3) void probe(const float2* __restrict__ src, int n_src, int iters, unsigned long long* bad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int bad_lo = 0, bad_hi = 0;
  probe_f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  probe_bf16x8 fa, fb;
  for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * ((t + k) & 31)); fb[k] = (__bf16)(0.02f * ((k - t) & 15)); }
  float2 a = src[t % n_src];
  if constexpr (MFMA == 2) {
    // waves 1-3 of a workgroup (one per SIMD, as wave 0) only run MFMAs: with three workgroups per CU every SIMD holds packed-instruction
    // waves and MFMA waves of different workgroups side by side
    if ((threadIdx.x >> 6) != 0) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
      }
      if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) bad[NFORMS * 8] = 1;
      return;
    }
  }
  for (int i = 0; i < iters; ++i) {
    if constexpr (MFMA == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
    }
    float2 b = src[(t * 7 + i * 131) % n_src];
    const float2 c = src[(t * 13 + i * 17 + 5) % n_src];
    if constexpr (F == 10) {   // (an SGPR pair: the value of lane 0, the same for the whole wave)
      b.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.x)));
      b.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.y)));
    }
    const float2 d = packed<F>(a, b, c), r = scalar_ref<F>(a, b, c);
    bad_lo += __float_as_uint(d.x) != __float_as_uint(r.x);
    bad_hi += __float_as_uint(d.y) != __float_as_uint(r.y);
    if constexpr (MFMA == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[k & 3], 0, 0, 0);
    }
    a = make_float2(b.x * 0.5f + c.y, r.y * 0.25f + c.x);   // keep the operands moving (bounded)
    if (!(fabsf(a.x) < 4.f)) a.x = 0.37f;
    if (!(fabsf(a.y) < 4.f)) a.y = -0.81f;
  }
  const int q = (threadIdx.x & 63) >> 4;
  if (bad_lo) atomicAdd(&bad[(F * 2 + 0) * 4 + q], (unsigned long long)bad_lo);
  if (bad_hi) atomicAdd(&bad[(F * 2 + 1) * 4 + q], (unsigned long long)bad_hi);
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) bad[NFORMS * 8] = 1;   // (the MFMA chains stay live)
}

template <int F>
static void launch_form(int mfma, int blocks, const float2* d, int n_src, int iters, unsigned long long* bad) {
  if (mfma == 1) hipLaunchKernelGGL((probe<F, 1>), dim3(blocks), dim3(256), 0, 0, d, n_src, iters, bad);
  else if (mfma == 2) hipLaunchKernelGGL((probe<F, 2>), dim3(blocks), dim3(256), 0, 0, d, n_src, iters, bad);
  else hipLaunchKernelGGL((probe<F, 0>), dim3(blocks), dim3(256), 0, 0, d, n_src, iters, bad);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 2048, launches = argc > 3 ? atoi(argv[3]) : 2;
  const int n_src = 1 << 20;
  std::vector<float2> h(n_src);
  unsigned int s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v.x = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; s = s * 1664525u + 1013904223u; v.y = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; }
  float2* d; unsigned long long* bad;
  if (hipMalloc(&d, n_src * sizeof(float2)) != hipSuccess || hipMalloc(&bad, (NFORMS * 8 + 8) * 8) != hipSuccess) { printf("alloc failed\n"); return 2; }
  (void)hipMemcpy(d, h.data(), n_src * sizeof(float2), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    (void)hipMemset(bad, 0, (NFORMS * 8 + 8) * 8);
    for (int l = 0; l < launches; ++l) {
      launch_form<0>(mode, blocks, d, n_src, iters, bad); launch_form<1>(mode, blocks, d, n_src, iters, bad);
      launch_form<2>(mode, blocks, d, n_src, iters, bad); launch_form<3>(mode, blocks, d, n_src, iters, bad);
      launch_form<4>(mode, blocks, d, n_src, iters, bad); launch_form<5>(mode, blocks, d, n_src, iters, bad);
      launch_form<6>(mode, blocks, d, n_src, iters, bad); launch_form<7>(mode, blocks, d, n_src, iters, bad);
      launch_form<8>(mode, blocks, d, n_src, iters, bad); launch_form<9>(mode, blocks, d, n_src, iters, bad);
      launch_form<10>(mode, blocks, d, n_src, iters, bad); launch_form<11>(mode, blocks, d, n_src, iters, bad);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
    std::vector<unsigned long long> hb(NFORMS * 8);
    (void)hipMemcpy(hb.data(), bad, NFORMS * 8 * 8, hipMemcpyDeviceToHost);
    printf("== %s: %.2e instructions per form; mismatches [low half: lanes 0-15 16-31 32-47 48-63 | high half: ...]\n",
           mode == 1 ? "MFMAs of the same wave around the instruction, three waves per SIMD" : mode == 2 ? "MFMAs ONLY in other waves (waves 1-3 of every workgroup; the packed instructions run in wave 0)" : "no MFMA in the kernel",
           (double)blocks * (mode == 2 ? 64 : 256) * iters * launches);
    for (int f = 0; f < NFORMS; ++f)
      printf("  %-118s lo %llu %llu %llu %llu | hi %llu %llu %llu %llu\n", form_of(f).text, hb[f * 8], hb[f * 8 + 1], hb[f * 8 + 2], hb[f * 8 + 3],
             hb[f * 8 + 4], hb[f * 8 + 5], hb[f * 8 + 6], hb[f * 8 + 7]);
  }
  return 0;
}
