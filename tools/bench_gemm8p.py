"""A/B of the two row-major bf16 GEMM kernels (RSYS_GEMM_KERNEL=1: 128x128 register-staged, 2: 256x256 LDS-DMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg

NT = 65536
SHAPES = [(NT, 1024, 512, False), (NT, 512, 512, True), (NT, 2816, 512, False), (NT, 512, 1408, True), (NT, 1408, 512, False),
          (NT, 512, 2816, False), (NT, 512, 1024, False), (4096, 120000, 512, False), (200001, 512, 6208, True),
          (4096, 4096, 4096, False), (8192, 8192, 8192, False)]
for (M, N, K, cf32) in SHAPES:
    for k in ("2", "3", "2", "3"):
        os.environ["RSYS_GEMM_KERNEL"] = k
        print("kernel", k, end="  ")
        bg.run(M, N, K, False, False, c_f32=cf32, reps=10)
