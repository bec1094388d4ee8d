"""Split-K scan of the 128x128 K-major kernel on the trunk's weight-gradient shapes (what model.hip's pick_splitk should pick)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["RSYS_GEMM_KERNEL_TN"] = "1"
import bench_gemm as bg
NT = 65536
for (M, N) in [(2816, 512), (512, 1408), (1024, 512), (512, 512)]:
    for sk in (8, 16, 24, 32, 40, 48, 56, 64, 96, 128):
        bg.run(M, N, NT, True, True, c_f32=True, splitk=sk, reps=6)
