"""K-major weight-gradient products with FEW output tiles (4 .. 22 of 256 x 256) on the 128 x 128 register-staged kernel (RSYS_GEMM_KERNEL_TN=1)
against the 256 x 256 LDS-DMA kernel (=2): where does the second one win since its DMA is no longer drained every phase?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_gemm as bg
shapes = [(512, 512, 65536, 32), (1024, 512, 65536, 32), (512, 1408, 65536, 16), (2816, 512, 65536, 8), (1024, 1024, 131072, 16), (2048, 1024, 131072, 8)]
for mode in ("1", "2"):
    os.environ["RSYS_GEMM_KERNEL_TN"] = mode
    print("# RSYS_GEMM_KERNEL_TN=" + mode)
    for (M, N, K, sk) in shapes:
        bg.run(M, N, K, True, True, c_f32=True, splitk=sk, reps=6)
