"""The production shape's four weight-gradient products (K = 131072 tokens, K-major operands) on the LDS-DMA kernel, one launch
group per shape, for rocprofv3 (--kernel-trace --stats, --pmc FETCH_SIZE): how many bytes does each launch fetch from HBM against
the operands' size?  (tools/pmc_dw_prod.sh)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("RSYS_GEMM_KERNEL_TN", "2")
import bench_gemm as bg
NT = 131072
for (M, N) in [(11264, 2048), (2048, 5632), (4096, 2048), (2048, 2048)]:
    print(f"# operands {(M + N) * NT * 2 / 1e9:.2f} GB, output {M * N * 4 / 1e6:.0f} MB")
    bg.run(M, N, NT, True, True, c_f32=True, splitk=8, reps=3)
