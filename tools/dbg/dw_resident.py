"""Is the K-major weight-gradient kernel bound by how many operand bytes a CU keeps in flight against HBM latency?  The same 22-tile
product (dW13: 2816 x 512) on operands streamed from HBM (K = 65 536 tokens: 436 MB) and on operands that stay cache-resident across
repetitions (K = 4 096: 27 MB), with as many split-K parts as fill the chip in both cases."""
import os, sys
os.environ["RSYS_GEMM_KERNEL_TN"] = "2"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_gemm as bg
for rep in range(2):
    bg.run(2816, 512, 65536, True, True, c_f32=True, splitk=12, reps=20)
    bg.run(2816, 512, 8192, True, True, c_f32=True, splitk=12, reps=100)
    bg.run(2816, 512, 4096, True, True, c_f32=True, splitk=12, reps=200)
