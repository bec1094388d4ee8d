"""Timing experiment: odd CUs start late (RSYS_DEBUG_STAGGER = number of 127*64-clock sleeps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
NT = 65536
os.environ["RSYS_GEMM_KERNEL"] = "2"
for (M, N, K, cf32) in [(NT, 1024, 512, False), (NT, 512, 512, True), (NT, 2816, 512, False)]:
    for e in (None, "1", "2", "3", "4", "6", None):
        if e: os.environ["RSYS_DEBUG_STAGGER"] = e
        else: os.environ.pop("RSYS_DEBUG_STAGGER", None)
        print("stagger", e, end="  ")
        bg.run(M, N, K, False, False, c_f32=cf32, reps=10)
