import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RSYS_GEMM_KERNEL"] = "2"
import bench_gemm as bg
bg.run(8192, 8192, 8192, False, False, reps=3)
bg.run(65536, 512, 2816, False, False, reps=3)
