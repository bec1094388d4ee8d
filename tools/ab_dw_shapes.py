"""The trunk's four weight-gradient shapes on the 128x128 K-major kernel, with the split counts model.hip picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["RSYS_GEMM_KERNEL_TN"] = "1"
import bench_gemm as bg
NT = 65536
for (M, N, sk) in [(2816, 512, 8), (512, 1408, 16), (1024, 512, 16), (512, 512, 32)]:
    bg.run(M, N, NT, True, True, c_f32=True, splitk=sk, reps=8)
