"""One K-major GEMM shape on the LDS-DMA kernel, for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RSYS_GEMM_KERNEL_TN", "2")
import bench_gemm as bg
bg.run(2048, 2048, 65536, True, True, c_f32=True, splitk=8, reps=3)
os.environ["RSYS_GEMM_KERNEL"] = "2"
bg.run(65536, 2048, 2048, False, False, c_f32=False, reps=3)
