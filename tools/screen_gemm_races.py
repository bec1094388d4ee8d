"""Race screen for the LDS-DMA GEMM kernels (their correctness rests on counted vmcnt waits and barrier placement:
an early LDS read passes whenever the DMA happens to land first).  Many shapes x repeats on asymmetric integer data,
exact comparison with a float64 reference, all in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_ops as T

rng = np.random.default_rng(2024)
bad = 0; n = 0
cases = []
for _ in range(36):
    M = int(rng.choice([256, 512, 768, 1024, 2048, 4096, 8192, 1000, 3000, 16384, 20000]))
    N = int(rng.choice([256, 512, 1024, 1408, 2816, 264, 200, 776]))
    K = int(rng.choice([128, 192, 256, 512, 576, 1024, 1408, 2816]))
    cases.append((M, N, K))
for (M, N, K) in cases:
    for kern, env in (("8c", {"RSYS_GEMM_KERNEL": "2", "RSYS_GEMM8C": "1"}), ("8p", {"RSYS_GEMM_KERNEL": "2", "RSYS_GEMM8C": "0"})):
        os.environ.update(env)
        for rep in range(2):
            out, ref = T.run_gemm(1, M, N, K, False, False, c_f32=bool(rep), integer=True, seed=M + N + K + rep)
            exp = ref.astype(np.float32) if rep else T._bf16_round(ref.astype(np.float32))
            n += 1
            if not np.array_equal(out, exp):
                bad += 1; print("MISMATCH", kern, M, N, K, rep, int((out != exp).sum()))
os.environ["RSYS_GEMM_KERNEL_TN"] = "2"
for _ in range(24):
    M = int(rng.choice([256, 512, 1024, 520, 2816])); N = int(rng.choice([256, 512, 1408, 264, 6208]))
    K = int(rng.choice([4096, 8192, 65536, 5000, 20001]))
    for rep in range(2):
        out, ref = T.run_gemm(1, M, N, K, True, True, c_f32=True, splitk=2, integer=True, seed=M + N + K + rep)
        n += 1
        if not np.array_equal(out, ref.astype(np.float32)):
            bad += 1; print("MISMATCH 8t", M, N, K, rep, int((out != ref.astype(np.float32)).sum()))
# the K-major kernel's DMA is issued behind asm volatile since round 4 (the compiler no longer adds its own vmcnt(0) before the transposed
# LDS reads): the counted waits of the K loop carry it alone.  Long K ranges per split, wide products (banded tile order), odd K tiles.
for (M, N, K) in [(2048, 5632, 32768), (4096, 2048, 65536), (11264, 2048, 16384), (512, 6208, 131072), (2048, 2048, 131072),
                  (264, 2568, 12352), (1024, 4104, 9000)]:
    for rep in range(3):
        out, ref = T.run_gemm(1, M, N, K, True, True, c_f32=True, splitk=2, integer=True, seed=M + N + K + rep)
        n += 1
        if not np.array_equal(out, ref.astype(np.float32)):
            bad += 1; print("MISMATCH 8t long", M, N, K, rep, int((out != ref.astype(np.float32)).sum()))
# round 6: gemm4p (the hand-over between two tiles of a workgroup rests on vmcnt's 6-bit range and on in-order retirement behind 64 stores; whole
# tiles only), gemm8c's HALF form and its reverse walk of the tile rows, the mixed-layout split-K kernel of the head's dEw
for k in ("RSYS_GEMM_KERNEL_TN",):
    os.environ.pop(k, None)
os.environ.update({"RSYS_GEMM_KERNEL": "2", "RSYS_GEMM8C": "1", "RSYS_GEMM4P": "2"})
for _ in range(28):
    M = 256 * int(rng.choice([1, 2, 5, 33, 64, 65, 130, 257])); N = 256 * int(rng.choice([1, 2, 3, 4, 9])); K = 128 * int(rng.choice([2, 3, 4, 5, 11, 22]))
    for rep in range(3):
        out, ref = T.run_gemm(1, M, N, K, False, False, c_f32=False, integer=True, seed=M + N + K + rep)
        n += 1
        if not np.array_equal(out, T._bf16_round(ref.astype(np.float32))):
            bad += 1; print("MISMATCH 4p", M, N, K, rep, int((out != T._bf16_round(ref.astype(np.float32))).sum()))
os.environ.update({"RSYS_GEMM4P": "0", "RSYS_GEMM8C_HALF": "2", "RSYS_GEMM_REVERSE": "2"})
for _ in range(16):
    M = int(rng.choice([256, 1000, 3000, 16384, 20000, 33000])); N = int(rng.choice([256, 512, 264, 776, 1408])); K = int(rng.choice([128, 192, 256, 576, 1408]))
    for rep in range(2):
        out, ref = T.run_gemm(1, M, N, K, False, False, c_f32=False, integer=True, seed=M + N + K + rep)
        n += 1
        if not np.array_equal(out, T._bf16_round(ref.astype(np.float32))):
            bad += 1; print("MISMATCH 8c half + reverse", M, N, K, rep)
for k in ("RSYS_GEMM4P", "RSYS_GEMM8C_HALF", "RSYS_GEMM_REVERSE", "RSYS_GEMM_KERNEL"):
    os.environ.pop(k, None)
os.environ["RSYS_GEMM_KERNEL_MIX"] = "2"
from recommendersystem_amd import _lib
lib = _lib.lib()
for _ in range(16):
    M = int(rng.choice([512, 2048, 4096, 8704])); N = int(rng.choice([256, 512, 264, 1024])); K = int(rng.choice([1024, 4096 + 40, 8192, 20001]))
    rows = int(rng.choice([1, 300, 1300, 2600, 9000]))
    A = rng.integers(-2, 3, (M, K)).astype(np.float32); B = rng.integers(-2, 3, (K, N)).astype(np.float32)
    lda = (K + 7) // 8 * 8; ldb = (N + 7) // 8 * 8
    Ap = np.zeros((M, lda), np.float32); Ap[:, :K] = A
    Bp = np.zeros((K, ldb), np.float32); Bp[:, :N] = B
    live = min(rows, M)
    want = (A[:live].astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    dA = T._to_dev(lib, T._pack(Ap, True)); dB = T._to_dev(lib, T._pack(Bp, True))
    for rep in range(2):
        dC = T._to_dev(lib, np.zeros((M, N), np.float32)); dR = T._to_dev(lib, np.array([rows], np.int32))
        assert lib.rsys_op_gemm_rows(1, dA, dB, dC, M, N, K, lda, ldb, N, 1, 3, dR) == 0, _lib.last_error()
        out = np.empty((M, N), np.float32)
        assert lib.rsys_dev_d2h(out.ctypes.data, dC, out.nbytes) == 0
        lib.rsys_dev_free(dC); lib.rsys_dev_free(dR)
        n += 1
        if not np.array_equal(out[:live], want):
            bad += 1; print("MISMATCH 8m", M, N, K, rows, rep, int((out[:live] != want).sum()))
    lib.rsys_dev_free(dA); lib.rsys_dev_free(dB)
print(f"screened {n} GEMMs, {bad} mismatches")
sys.exit(1 if bad else 0)
