// Probe for VERDICT r4 item 3 / ADVICE r4 (medium): does the packed-f32 form that failed in attn_bwd_q_kernel<bf16,64,2> at 02e2619
// miscompute IN ISOLATION?  The failing value was the LOW half of
//     v_pk_fma_f32 vD, vP, vC, vT op_sel:[0,1,0]          (lo = P.lo * C.hi + T.lo,  hi = P.hi * C.hi + T.hi)
// with T = v_pk_mul_f32 vP, vS op_sel:[1,1] op_sel_hi:[0,1] (lo = P.hi * S.hi,        hi = P.lo * S.hi)
// and C, S pairs fresh from global_load_dwordx2 behind a counted s_waitcnt (profiles/r5_packed_f32_isa_analysis.md).
// Every lane runs that exact sequence (inline asm, same op_sel bits, operands loaded by global_load_dwordx2 right in front of it) on
// pseudo-random data `iters` times and compares both halves with the same arithmetic done by scalar v_mul_f32 / v_fma_f32; mismatches are
// counted per lane quarter.  One bounded run (default 2048 workgroups x 256 threads x 4096 iterations = 2.1e9 packed pairs); it is a
// measurement of the instruction form, not a retry loop of the failing kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pk_f32_crossed_probe.hip -o tools/micro/bin/pk_f32_crossed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// MFMA (round 6): the failing kernel runs three waves per SIMD whose neighbours are inside MFMA-heavy item loops while a wave is in its
// epilogue; with MFMA = true every iteration also issues 2 x 8 MFMAs (16x16x32 bf16, independent chains) around the packed sequence, so
// that on every SIMD the packed instructions of one wave meet the matrix-core operand reads of the others.
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(4))) float probe_f32x4;
template <bool MFMA>
__global__ __launch_bounds__(256, 3) void probe(const float2* __restrict__ src, int n_src, int iters, unsigned long long* bad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int local_bad = 0;
  float2 P = src[t % n_src];
  probe_f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  probe_bf16x8 fa, fb;
  for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * ((t + k) & 31)); fb[k] = (__bf16)(0.02f * ((k - t) & 15)); }
  for (int i = 0; i < iters; ++i) {
    if constexpr (MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[k & 3], 0, 0, 0);
    }
    const float2* pc = src + ((t * 7 + i * 131) % n_src);
    const float2* ps = src + ((t * 13 + i * 17 + 5) % n_src);
    float2 C, S, T, D, Dn;
    float r_lo, r_hi, rn_hi, t_lo, t_hi;
    // the loads, the counted wait and the packed sequence in ONE statement (the compiler neither reorders nor pads inside it)
    asm volatile(
        "global_load_dwordx2 %0, %7, off\n\t"
        "global_load_dwordx2 %1, %8, off\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_pk_mul_f32 %2, %6, %1 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %3, %6, %0, %2 op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %4, %6, %0, %2 op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        : "=&v"(C), "=&v"(S), "=&v"(T), "=&v"(D), "=&v"(Dn), "=v"(t_lo)
        : "v"(P), "v"(pc), "v"(ps)
        : "memory");
    // reference: the same products and sums, one scalar instruction each (fma contraction is what the packed op does too)
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t_lo) : "v"(P.y), "v"(S.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t_hi) : "v"(P.x), "v"(S.y));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r_lo) : "v"(P.x), "v"(C.y), "v"(t_lo));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r_hi) : "v"(P.y), "v"(C.y), "v"(t_hi));
    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(rn_hi) : "v"(P.y), "v"(C.y), "v"(t_hi));
    const bool ok = __float_as_uint(D.x) == __float_as_uint(r_lo) && __float_as_uint(D.y) == __float_as_uint(r_hi) &&
                    __float_as_uint(Dn.y) == __float_as_uint(rn_hi) && __float_as_uint(T.x) == __float_as_uint(t_lo) &&
                    __float_as_uint(T.y) == __float_as_uint(t_hi);
    local_bad += ok ? 0u : 1u;
    if constexpr (MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[k & 3], 0, 0, 0);
    }
    P = make_float2(C.x * 0.5f + S.x, D.y * 0.25f + S.y);   // keep the operands moving (bounded: |.| stays O(1))
    if (!(fabsf(P.x) < 4.f)) P.x = 0.37f;
    if (!(fabsf(P.y) < 4.f)) P.y = -0.81f;
  }
  if (local_bad) atomicAdd(&bad[(threadIdx.x & 63) >> 4], (unsigned long long)local_bad);
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) bad[4] = 1;   // (the MFMA chains stay live)
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 4096, launches = argc > 3 ? atoi(argv[3]) : 4;
  const int n_src = 1 << 20;
  std::vector<float2> h(n_src);
  unsigned int s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v.x = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; s = s * 1664525u + 1013904223u; v.y = ((int)(s >> 8) % 20001 - 10000) * 1e-4f; }
  float2* d; unsigned long long* bad;
  if (hipMalloc(&d, n_src * sizeof(float2)) != hipSuccess || hipMalloc(&bad, 8 * 8) != hipSuccess) { printf("alloc failed\n"); return 2; }
  hipMemcpy(d, h.data(), n_src * sizeof(float2), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(bad, 0, 64);
    for (int l = 0; l < launches; ++l) {
      if (mode) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, 0, d, n_src, iters, bad);
      else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, 0, d, n_src, iters, bad);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 3; }
    unsigned long long hb[4];
    hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
    const double total = (double)blocks * 256 * iters * launches;
    printf("%s: packed sequences executed %.3e (per lane quarter %.3e); mismatches by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           mode ? "with MFMAs around the sequence, 3 waves per SIMD" : "alone", total, total / 4, hb[0], hb[1], hb[2], hb[3]);
  }
  return 0;
}
