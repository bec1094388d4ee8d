// Harness of recommendersystem_amd/csrc/gemm4p.hip (the generated-assembly K loop of tools/micro/gemm4a.hip in a persistent
// form): launches it beside launch_gemm8c on the same operands, compares the outputs bit for bit over three launches and times both with HIP
// events, alternating.  Timing-only / measurement variants of the loop (gen_gemm4p_asm.py --no-mfma, --a-empty, --b-empty, --pf into a scratch
// directory) are built with -DGEMM4P_ASM_INC='"<dir>/gemm4p_asm.inc"': tools/micro/build_gemm4p_variants.sh.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Irecommendersystem_amd/csrc tools/micro/gemm4p.hip -o tools/micro/bin/gemm4p
#include "../../recommendersystem_amd/csrc/gemm8p.hip"
#include "../../recommendersystem_amd/csrc/gemm8c.hip"
#include "../../recommendersystem_amd/csrc/gemm4p.hip"
#include "../../recommendersystem_amd/csrc/switches.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

namespace rsys {
void set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }
int gemm_slab_begin(const GemmParams&, hipStream_t) { return 0; }
int gemm_slab_end(const GemmParams&, hipStream_t) { return 0; }

}  // namespace rsys

static unsigned int g_seed = 0x1234567u;
static void fill_bf16(void* d, size_t n) {
  std::vector<unsigned short> h(n);
  for (auto& v : h) {   // standard-normal operands (what tools/bench_vendor_gemm.py and torch.randn feed)
    g_seed = g_seed * 1664525u + 1013904223u; const float u1 = (float)(g_seed >> 8) * (1.0f / 16777216.0f) + 1e-7f;
    g_seed = g_seed * 1664525u + 1013904223u; const float u2 = (float)(g_seed >> 8) * (1.0f / 16777216.0f);
    const float f = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    unsigned int u; memcpy(&u, &f, 4); v = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1)) >> 16);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

static void run(int M, int N, int K, int reps) {
  using namespace rsys;
  void *A, *B, *C0, *C1;
  const int pad = getenv("G4_PAD") ? atoi(getenv("G4_PAD")) : 0;   // extra elements per operand row (leading dimension K + pad): does the row stride matter?
  const int ld = K + pad;
  hipMalloc(&A, (size_t)M * ld * 2); hipMalloc(&B, (size_t)N * ld * 2); hipMalloc(&C0, (size_t)M * N * 2); hipMalloc(&C1, (size_t)M * N * 2);
  fill_bf16(A, (size_t)M * ld); fill_bf16(B, (size_t)N * ld);
  hipMemset(C0, 0xEE, (size_t)M * N * 2); hipMemset(C1, 0xEE, (size_t)M * N * 2);
  GemmParams p{};
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = ld; p.ldb = ld; p.ldc = N; p.epi = EPI_STORE; p.c_f32 = 0; p.alpha = 1.f; p.splitk = 1;
  const int tiles = (M / 256) * (N / 256), tiles_n = N / 256;
  if (M % 256 || N % 256 || K % 128 || K < 256) { printf("M=%d N=%d K=%d: not a shape of the asm kernel\n", M, N, K); return; }
  auto launch4 = [&]() {
    GemmParams q = p; q.C = C1;
    if (tiles_n > 8 && tiles >= 32 * tiles_n) q.flags |= 4 << 3;   // launch_gemm8c's band rule
    hipLaunchKernelGGL(gemm4p_kernel, dim3(std::min(tiles, 256)), dim3(256), 0, nullptr, q);   // (launch_gemm4p without its switch)
  };
  auto launch8 = [&]() { GemmParams q = p; q.C = C0; launch_gemm8c(q, nullptr); };
  launch8(); launch4();
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(2); }
  std::vector<unsigned short> h0((size_t)M * N), h1((size_t)M * N);
  size_t bad = 0, first = 0;
  for (int rep = 0; rep < 3; ++rep) {   // (the persistent hand-over is timing dependent: compare more than one launch)
    if (rep) { hipMemset(C1, 0xEE, (size_t)M * N * 2); launch4(); hipDeviceSynchronize(); }
    hipMemcpy(h0.data(), C0, h0.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), C1, h1.size() * 2, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { if (!bad) first = i; ++bad; }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2] = {1e30f, 1e30f};
  for (int round = 0; round < 6; ++round)
    for (int wi = 0; wi < 2; ++wi) {
      const int which = (round & 1) ? 1 - wi : wi;
      for (int i = 0; i < 2; ++i) { if (which) launch4(); else launch8(); }
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < reps; ++i) { if (which) launch4(); else launch8(); }
      hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
      float tms = 0; hipEventElapsedTime(&tms, e0, e1);
      if (round >= 2 && tms < ms[which]) ms[which] = tms;
    }
  const double fl = 2.0 * M * N * (double)K;
  printf("M=%6d N=%6d K=%5d ld=%5d : gemm8c %8.1f us %7.1f TF/s | gemm4p (asm, persistent) %8.1f us %7.1f TF/s | x%.3f | %s (%zu mismatching elements in 3 launches, first %zu)\n", M, N, K, ld,
         ms[0] * 1000 / reps, fl / (ms[0] / reps) * 1e-9, ms[1] * 1000 / reps, fl / (ms[1] / reps) * 1e-9, ms[0] / ms[1], bad ? "MISMATCH" : "bit-identical", bad, first);
  fflush(stdout);
  hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  setenv("RSYS_GEMM8C", "1", 1);
  setenv("RSYS_GEMM4P", "0", 1);   // (launch_gemm8c would forward the long-K shapes to gemm4p: the left column is gemm8c itself)
  if (argc > 4) { run(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), reps); return 0; }
  run(2048, 2048, 1024, 2);          // 64 tiles: one per workgroup
  run(4096, 4096, 1024, 2);          // 256 tiles
  run(8192, 4096, 512, 2);           // 512 tiles, two per workgroup, the shortest K the loop takes... (nt = 8)
  run(65536, 512, 2816, reps);       // w13_dx at cfg-3
  run(65536, 512, 1408, reps);       // w2_fwd's K with a plain store
  run(65536, 512, 1024, reps);
  run(65536, 1024, 512, reps);
  run(8192, 8192, 8192, reps);
  run(131072, 2048, 11264, 2);       // w13_dx at the production shape
  run(131072, 2048, 2048, reps);     // o_dx at the production shape
  return 0;
}
