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
print(f"screened {n} GEMMs, {bad} mismatches")
sys.exit(1 if bad else 0)
