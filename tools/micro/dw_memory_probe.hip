// What bounds the K-major weight-gradient kernel (gemm8p_kernel<true>: dW = dY^T X, K = 65 536 tokens)?  The same launch with
//   hbm      : the real operands (436 MB for dW13, streamed once from HBM),
//   resident : row stride 8 elements -- every K row overlaps the previous one, 1 MB in all: the DMA path runs, memory latency is an L2 hit,
//   none     : row stride 0 -- the descriptors are empty, every DMA returns zeros without touching memory: the kernel's own ceiling.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Iinclude tools/micro/dw_memory_probe.hip -o tools/micro/bin/dw_memory_probe
#include "../../recommendersystem_amd/csrc/gemm8p.hip"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rsys {
void set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }
int gemm_slab_begin(const GemmParams&, hipStream_t) { return 0; }
int gemm_slab_end(const GemmParams&, hipStream_t) { return 0; }
bool gemm8c_eligible(const GemmParams&) { return false; }   // (gemm8c.hip is not part of this build)
int launch_gemm8c(const GemmParams&, hipStream_t) { return RSYS_ERR_ARG; }
}

static void run(const char* name, int M, int N, int K, long long lda, long long ldb, const void* A, const void* B, float* C) {
  using namespace rsys;
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = N; p.epi = EPI_ATOMIC; p.c_f32 = 1; p.alpha = 1.f;
  if (!gemm8p_tn_eligible(p)) { printf("%s: not eligible\n", name); return; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_gemm8p_tn(p, nullptr);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch_gemm8p_tn(p, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, tf = 2.0 * M * N * (double)K / (us * 1e-6) / 1e12;
  printf("  %-9s M=%5d N=%5d K=%6d splits=%3d : %8.1f us  %7.1f TFLOP/s  fill %5.2f TB/s\n", name, M, N, K, gemm8p_splits(p, true), us, tf,
         ((M + 255) / 256) * ((N + 255) / 256) * 512.0 * K * 2 / (us * 1e-6) / 1e12);
}

int main() {
  const int K = 65536;
  const int shapes[4][2] = {{2816, 512}, {512, 1408}, {1024, 512}, {512, 512}};
  void *A, *B; float* C;
  hipMalloc(&A, (size_t)K * 2816 * 2); hipMalloc(&B, (size_t)K * 1408 * 2); hipMalloc(&C, (size_t)2816 * 1408 * 4);
  {
    std::vector<unsigned short> h((size_t)K * 2816);
    unsigned int x = 0x1234567u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f; unsigned int u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)K * 1408 * 2, hipMemcpyHostToDevice);
  }
  for (auto& s : shapes) {
    const int M = s[0], N = s[1];
    printf("dW %d x %d\n", M, N);
    run("hbm", M, N, K, M, N, A, B, C);
    run("resident", M, N, K, 8, 8, A, B, C);
    run("none", M, N, K, 0, 0, A, B, C);
  }
  return 0;
}
