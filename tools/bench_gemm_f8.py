"""fp8 against bf16 on the persistent 256x256 pipeline at the trunk's shapes (plain store epilogue, bf16 output; wall clock per call incl. one sync)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
from recommendersystem_amd import _lib
lib = bg.lib
os.environ["RSYS_GEMM_KERNEL"] = "2"

def run_f8(M, N, K, a_fmt=0, reps=10):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 120, M * K, dtype=np.uint8); b = rng.integers(0, 120, N * K, dtype=np.uint8)   # positive finite codes of both formats
    A = bg.dev(M * K); B = bg.dev(N * K); Cc = bg.dev(M * N * 2); D = bg.dev(256)
    lib.rsys_dev_h2d(A, a.ctypes.data, a.nbytes); lib.rsys_dev_h2d(B, b.ctypes.data, b.nbytes)
    d = np.full(32, 1e-3, np.float32); lib.rsys_dev_h2d(D, d.ctypes.data, d.nbytes)
    args = (A, B, Cc, M, N, K, K, K, N, a_fmt, 0, D, 0, 0, 0, 0, 0)
    assert lib.rsys_op_gemm_f8(*args) == 0, _lib.last_error()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.rsys_op_gemm_f8(*args)
    dt = (time.perf_counter() - t0) / reps
    print(f"fp8  M={M:6d} N={N:5d} K={K:6d}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s")
    for p in (A, B, Cc, D):
        lib.rsys_dev_free(p)

NT = 65536
for (N, K) in [(1536, 512), (512, 512), (2816, 512), (512, 1408), (1408, 512), (512, 2816), (512, 1536), (8192, 8192)]:
    M = 8192 if N == 8192 else NT
    print("bf16 ", end=""); bg.run(M, N, K, False, False, reps=10)
    run_f8(M, N, K)
    run_f8(M, N, K, a_fmt=1)
