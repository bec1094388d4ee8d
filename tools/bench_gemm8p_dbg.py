"""Timing experiment on the persistent 256x256 kernel: RSYS_DEBUG_8P bit 0 = no allowance for pending stores in the
first waits of a tile, bit 1 = one workgroup per tile (non-persistent grid)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = '''
import os, sys
sys.path.insert(0, %r)
import bench_gemm as bg
NT = 65536
os.environ["RSYS_GEMM_KERNEL"] = "2"
for (M, N, K, cf32) in [(NT, 1024, 512, False), (NT, 512, 512, True), (NT, 2816, 512, False), (NT, 512, 1408, True), (NT, 1408, 512, False), (4096, 120000, 512, False)]:
    bg.run(M, N, K, False, False, c_f32=cf32, reps=10)
''' % HERE
for dbg in ("0", "1", "2", "3", "0"):
    print("RSYS_DEBUG_8P =", dbg, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSYS_DEBUG_8P=dbg), check=True)
