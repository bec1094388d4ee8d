"""Calibration only (never a dependency of the product path): what PyTorch's own bf16 / fp8 matmul (hipBLASLt / rocBLAS behind
torch.mm and torch._scaled_mm) reaches on the trunk's GEMM shapes, plain bf16 output, beside this package's kernels on the same shapes."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
NT = 65536
shapes = [(NT, 1024, 512), (NT, 512, 512), (NT, 2816, 512), (NT, 512, 1408), (NT, 1408, 512), (NT, 512, 2816), (NT, 512, 1024), (8192, 8192, 8192)]
dev = "cuda"
def t_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); B = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    ms = t_ms(lambda: torch.mm(A, B.t()))
    line = f"M={M:6d} N={N:5d} K={K:5d}  torch.mm bf16 {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s"
    try:
        A8 = A.to(torch.float8_e4m3fn); B8 = B.to(torch.float8_e4m3fn)
        one = torch.tensor(1.0, device=dev)
        ms8 = t_ms(lambda: torch._scaled_mm(A8, B8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
        line += f" | torch._scaled_mm e4m3 {ms8*1e3:8.1f} us {2.0*M*N*K/ms8/1e9:7.1f} TF/s"
    except Exception as e:
        line += f" | _scaled_mm: {type(e).__name__}"
    print(line, flush=True)
    del A, B
