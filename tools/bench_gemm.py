"""Micro-benchmark of the GEMM kernel family through rsys_op_gemm (plain store epilogue)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystem_amd import _lib
lib = _lib.lib()

def dev(nbytes):
    p = C.c_void_p(); assert lib.rsys_dev_alloc(C.byref(p), nbytes) == 0; return p

def fill(p, n_u16, seed):
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(n_u16).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    lib.rsys_dev_h2d(p, a.ctypes.data, a.nbytes)

def run(M, N, K, a_km, b_km, a_f32=False, c_f32=False, splitk=1, reps=10):
    lda = M if a_km else K; ldb = N if b_km else K
    ea = 4 if a_f32 else 2
    A = dev(M * K * ea); B = dev(N * K * 2); Cc = dev(M * N * (4 if (c_f32 or splitk > 1) else 2))
    if a_f32:
        a = np.random.default_rng(1).standard_normal(M * K).astype(np.float32); lib.rsys_dev_h2d(A, a.ctypes.data, a.nbytes)
    else:
        fill(A, M * K, 1)
    fill(B, N * K, 2)
    args = (1, A, B, Cc, M, N, K, lda, ldb, N, int(a_km), int(b_km), int(a_f32), int(c_f32 or splitk > 1), splitk)
    assert lib.rsys_op_gemm(*args) == 0, _lib.last_error()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.rsys_op_gemm(*args)
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * M * N * K
    by = M * K * ea + N * K * 2 + M * N * (4 if c_f32 else 2)
    print(f"M={M:6d} N={N:5d} K={K:6d} akm={int(a_km)} bkm={int(b_km)} af32={int(a_f32)} cf32={int(c_f32)} sk={splitk:2d}: "
          f"{dt*1e6:8.1f} us  {fl/dt/1e12:7.1f} TF/s  {by/dt/1e12:5.2f} TB/s(min bytes)")
    for p in (A, B, Cc):
        lib.rsys_dev_free(p)

def table():
  NT = 65536
  print("# forward shapes (NT)")
  run(NT, 1024, 512, False, False)
  run(NT, 512, 512, False, False, c_f32=True)
  run(NT, 2816, 512, False, False)
  run(NT, 512, 1408, False, False, c_f32=True)
  run(4096, 120000, 512, False, False)
  run(200001, 512, 6208, False, False, c_f32=True)
  print("# dx shapes (NN)")
  run(NT, 1408, 512, False, True, a_f32=True)
  run(NT, 512, 2816, False, True)
  run(NT, 512, 512, False, True, a_f32=True)
  run(NT, 512, 1024, False, True)
  print("# dw shapes (TN, split-K)")
  run(512, 1408, NT, True, True, a_f32=True, splitk=22)
  run(2816, 512, NT, True, True, splitk=11)
  run(512, 512, NT, True, True, a_f32=True, splitk=64)
  run(1024, 512, NT, True, True, splitk=32)
  run(512, 6208, 200001, True, True, a_f32=True, splitk=5)
  print("# big square")
  run(8192, 8192, 8192, False, False)
  run(8192, 8192, 8192, False, True)
  run(8192, 8192, 8192, True, True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        M, N, K, akm, bkm = [int(x) for x in sys.argv[1:6]]
        run(M, N, K, bool(akm), bool(bkm), reps=5)
    else:
        table()
