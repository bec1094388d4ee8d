// Where do the cycles of the persistent 256x256 GEMM go?  Builds gemm8p.hip with RSYS_8P_TRACE (s_memtime marks around
// the sections of a tile) and prints, per shape, the average cycles per tile a workgroup spends
//   wait   : from "previous tile done" to "first K tile landed" (initial counted wait + barrier)
//   main   : the K loop
//   issue  : requesting the next tile's first two K tiles
//   epi    : the register epilogue (operand loads, conversions, stores issued)
//   setup  : recomputing lane / tile offsets
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/micro/gemm8p_trace.hip -o /tmp/gemm8p_trace && /tmp/gemm8p_trace
#define RSYS_8P_TRACE 1
#include "../../recommendersystem_amd/csrc/gemm8p.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

namespace rsys { void set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); } }

static void run(int M, int N, int K, int epi, int c_f32, int stagger = 0) {
  using namespace rsys;
  void *A, *B, *C, *C2, *R; unsigned long long* tr;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C2, (size_t)M * N * 2); hipMalloc(&R, (size_t)M * N * 4);
  hipMalloc(&tr, 8 * 8 * 1024);
  {   // uniform random bf16 in [-1, 1): operand data sets the power draw and with it the clock (constant data reads 15-20 % high)
    std::vector<unsigned short> h((size_t)std::max(M, N) * K);
    unsigned int x = 0x1234567u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f; unsigned int u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  }
  hipMemset(R, 0, (size_t)M * N * 4);
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.epi = epi; p.c_f32 = c_f32;
  p.alpha = 1.f; p.splitk = 1; p.trace = tr;
  if (stagger) { p.flags |= 4; p.T = stagger; }
  if (epi == EPI_RESIDUAL) { p.resid = (const float*)R; p.ldr = N; p.c_f32 = 1; }
  if (epi == EPI_SWIGLU) { p.C2 = C2; p.ldc2 = N / 2; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_gemm8p(p, nullptr);
  hipMemset(tr, 0, 8 * 8 * 1024);
  hipEventRecord(e0, nullptr);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch_gemm8p(p, nullptr);
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(8 * 1024);
  hipMemcpy(h.data(), tr, 8 * 8 * 1024, hipMemcpyDeviceToHost);
  double s[5] = {0, 0, 0, 0, 0}, tiles = 0; int wgs = 0;
  for (int b = 0; b < 1024; ++b) if (h[b * 8 + 5]) { ++wgs; tiles += (double)h[b * 8 + 5]; for (int i = 0; i < 5; ++i) s[i] += (double)h[b * 8 + i]; }
  const double us = ms * 1000.0 / reps, tot = s[0] + s[1] + s[2] + s[3] + s[4];
  const double per_wg = tot / wgs;   // cycles a workgroup was busy in the last launch
  printf("stagger %2d M=%6d N=%5d K=%5d epi=%d f32=%d : %8.1f us  %7.1f TF/s | %d wgs, %.1f tiles/wg | cycles/tile wait %6.0f main %6.0f issue %5.0f epi %6.0f setup %4.0f | ticks/us %.1f\n",
         stagger, M, N, K, epi, c_f32, us, 2.0 * M * N * K / us * 1e-6, wgs, tiles / wgs, s[0] / tiles, s[1] / tiles, s[2] / tiles, s[3] / tiles, s[4] / tiles,
         per_wg / us);
  hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(R); hipFree(tr);
}

int main() {
  using namespace rsys;
  const int NT = 65536;
  const int stagger = getenv("TRACE_STAGGER") ? atoi(getenv("TRACE_STAGGER")) : 0;   // 1 us-ish naps for every other workgroup
  run(NT, 1024, 512, EPI_STORE, 0, stagger);
  run(NT, 2816, 512, EPI_SWIGLU, 0, stagger);
  run(NT, 2816, 512, EPI_STORE, 0, stagger);
  run(NT, 512, 1408, EPI_RESIDUAL, 1, stagger);
  run(NT, 1408, 512, EPI_STORE, 0, stagger);
  run(NT, 512, 512, EPI_RESIDUAL, 1, stagger);
  run(NT, 512, 2816, EPI_STORE, 0, stagger);
  run(4096, 4096, 4096, EPI_STORE, 0, stagger);
  run(8192, 8192, 8192, EPI_STORE, 0, stagger);
  return 0;
}
