"""Timing experiment: 256x256 kernel with and without its epilogue (RSYS_DEBUG_EPI=99)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
NT = 65536
os.environ["RSYS_GEMM_KERNEL"] = "2"
for (M, N, K, cf32) in [(NT, 1024, 512, False), (NT, 512, 512, True), (NT, 1024, 128, False), (NT, 1024, 1024, False), (NT, 1024, 2048, False)]:
    for e in (None, "99", None, "99"):
        if e: os.environ["RSYS_DEBUG_EPI"] = e
        else: os.environ.pop("RSYS_DEBUG_EPI", None)
        print("epi", e, end="  ")
        bg.run(M, N, K, False, False, c_f32=cf32, reps=10)
