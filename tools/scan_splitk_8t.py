"""Split-K scan of the K-major LDS-DMA kernel on the metadata-projection gradient shape (one process per setting)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); import bench_gemm as bg; bg.run(512, 6208, 200001, True, True, c_f32=True, splitk=8, reps=6)" % HERE
for sk in (0, 8, 16, 24, 32, 40, 48, 64, 0):
    print("RSYS_DEBUG_8T_SPLITK =", sk, end="  ", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSYS_GEMM_KERNEL_TN="2", RSYS_DEBUG_8T_SPLITK=str(sk)), check=True)
