import torch
for (M,N,K) in [(8192,8192,8192),(65536,512,2816),(65536,2816,512),(65536,512,512)]:
    A=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); B=torch.randn(N,K,device="cuda",dtype=torch.bfloat16)
    for _ in range(3): torch.mm(A,B.t())
    torch.cuda.synchronize()
    A8=A.to(torch.float8_e4m3fn); B8=B.to(torch.float8_e4m3fn); one=torch.tensor(1.0,device="cuda")
    for _ in range(3): torch._scaled_mm(A8,B8.t(),scale_a=one,scale_b=one,out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
