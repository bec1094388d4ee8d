"""GEMM timing with operands rotated over several buffers (defeats the 256 MB infinity cache: the in-step situation)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm as bg
lib = bg.lib

def run(M, N, K, c_f32=False, nbuf=6, reps=12):
    As = [bg.dev(M * K * 2) for _ in range(nbuf)]
    B = bg.dev(N * K * 2)
    Cs = [bg.dev(M * N * (4 if c_f32 else 2)) for _ in range(nbuf)]
    for a in As: bg.fill(a, M * K, 1)
    bg.fill(B, N * K, 2)
    def call(i):
        return lib.rsys_op_gemm(1, As[i % nbuf], B, Cs[i % nbuf], M, N, K, K, K, N, 0, 0, 0, int(c_f32), 1)
    assert call(0) == 0
    t0 = time.perf_counter()
    for i in range(reps): call(i + 1)
    dt = (time.perf_counter() - t0) / reps
    by = M * K * 2 + N * K * 2 + M * N * (4 if c_f32 else 2)
    print(f"kernel {os.environ.get('RSYS_GEMM_KERNEL')} M={M} N={N} K={K} cf32={int(c_f32)} nbuf={nbuf}: {dt*1e6:8.1f} us {2.0*M*N*K/dt/1e12:7.1f} TF/s {by/dt/1e12:5.2f} TB/s")
    for p in As + Cs + [B]: lib.rsys_dev_free(p)

NT = 65536
if __name__ == "__main__":
  for (M, N, K, cf) in [(NT, 1024, 512, False), (NT, 2816, 512, False), (NT, 512, 512, True)]:
    for nbuf in (1, 6):
      for k in ("1", "2", "3"):
        os.environ["RSYS_GEMM_KERNEL"] = k
        run(M, N, K, cf, nbuf)
