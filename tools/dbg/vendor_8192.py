import torch
for (M, N, K) in [(8192, 8192, 8192), (65536, 512, 2816)]:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(4): torch.mm(A, B.t())
    torch.cuda.synchronize()
