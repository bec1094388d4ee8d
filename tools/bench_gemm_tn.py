"""A/B of the K-major (weight-gradient) GEMM kernels: RSYS_GEMM_KERNEL_TN=1 (128x128 register-staged) vs 2 (LDS-DMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
NT = 65536
for (M, N, K) in [(8192, 8192, 8192), (2816, 512, NT), (512, 1408, NT), (1024, 512, NT), (512, 512, NT), (512, 6208, 200000)]:
    for k in ("1", "2", "1", "2"):
        os.environ["RSYS_GEMM_KERNEL_TN"] = k
        print("tn kernel", k, end="  ")
        bg.run(M, N, K, True, True, c_f32=True, splitk=8, reps=6)
