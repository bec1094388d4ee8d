"""What the trunk's four weight-gradient products would cost on K-CONTIGUOUS operand copies (row-major split-K LDS-DMA kernel,
gemm8p_kernel<false, true>) against what they cost today on K-major operands (128x128 register-staged kernel, ds_read_b64_tr_b16):
the GEMM-side bound of "producers emit a transposed second output" (VERDICT r1 item 4).  Shapes at cfg-3, K = 65 536 tokens."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["RSYS_GEMM_KERNEL_TN"] = "1"
os.environ["RSYS_GEMM_KERNEL_NT_SPLITK"] = "2"
import bench_gemm as bg
NT = 65536
for rep in range(2):
    for (name, M, N, sk) in [("dW13", 2816, 512, 8), ("dW2", 512, 1408, 16), ("dWqkv", 1024, 512, 16), ("dWo", 512, 512, 32)]:
        print(name, "K-major 128x128:", end=" "); bg.run(M, N, NT, True, True, c_f32=True, splitk=sk, reps=8)
        print(name, "row-major 256x256 split-K:", end=" "); bg.run(M, N, NT, False, False, c_f32=True, splitk=8, reps=8)
