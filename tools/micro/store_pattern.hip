// Micro-benchmark: HBM write rate of wave64 16-byte-per-lane stores as a function of the row-segment shape one
// instruction covers (the GEMM epilogue question: 16 rows x 64 B vs 8 x 128 B vs 4 x 256 B vs 1 KB contiguous).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// Output matrix [M][N] bf16 (2 B), N = 1024 columns (2 KB rows).  A workgroup of 512 threads (8 waves) writes a 256 x 256
// tile; wave w owns rows (w>>2)*128.., cols (w&3)*64..; SEG = bytes of one row one instruction covers (64, 128, 256).
template <int SEG>
__global__ __launch_bounds__(512) void store_tile(unsigned char* C, int M, int N) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int tiles_n = N / 256;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const long long ld = (long long)N * 2;
  if constexpr (SEG == 1024) {   // contiguous: the tile is just 128 KB somewhere
    unsigned char* base = C + (long long)blockIdx.x * 131072 + w * 16384;
#pragma unroll
    for (int i = 0; i < 16; ++i) *(uint4*)(base + i * 1024 + l * 16) = make_uint4(i, l, w, 7);
  } else {
    constexpr int LPR = SEG / 16;        // lanes per row segment
    constexpr int RPI = 64 / LPR;        // rows per instruction
    // wave block: 128 rows x 128 B; instruction k covers rows [k*RPI ..) x SEG bytes at column offset
    unsigned char* base = C + ((long long)tm * 256 + (w >> 2) * 128) * ld + (tn * 256 + (w & 3) * 64) * 2;
    constexpr int CPW = 128 / SEG;       // column pieces per wave row (SEG <= 128), or fraction
    if constexpr (SEG <= 128) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int piece = k % CPW, rblk = k / CPW;
        const int row = rblk * RPI + l / LPR;
        *(uint4*)(base + row * ld + piece * SEG + (l % LPR) * 16) = make_uint4(k, l, w, 7);
      }
    } else {   // SEG == 256: pretend the wave owns 64 rows x 256 B
      unsigned char* b2 = C + ((long long)tm * 256 + (w >> 1) * 64) * ld + (tn * 256 + (w & 1) * 128) * 2;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = k * 4 + l / 16;
        *(uint4*)(b2 + row * ld + (l % 16) * 16) = make_uint4(k, l, w, 7);
      }
    }
  }
}

template <int SEG>
static void run(unsigned char* C, int M, int N, const char* name) {
  const int tiles = (M / 256) * (N / 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  store_tile<SEG><<<tiles, 512>>>(C, M, N);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) store_tile<SEG><<<tiles, 512>>>(C, M, N);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms * 100.f, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
}

int main() {
  const int M = 262144, N = 1024;   // 512 MiB of bf16: not cache resident
  unsigned char* C; hipMalloc(&C, (size_t)M * N * 2);
  run<1024>(C, M, N, "1 KB contiguous / instr");
  run<256>(C, M, N, "4 rows x 256 B / instr");
  run<128>(C, M, N, "8 rows x 128 B / instr");
  run<64>(C, M, N, "16 rows x 64 B / instr");
  run<1024>(C, M, N, "1 KB contiguous / instr");
  return 0;
}
