"""Achievable HBM bandwidth on this device with library kernels (torch): copy (read+write), fill (write), sum (read)."""
import time, torch
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
xf = x.view(torch.float32); yf = y.view(torch.float32)
print("copy  (1 GiB read + 1 GiB write): %.2f TB/s" % (2 * x.numel() / t(lambda: y.copy_(x)) / 1e12))
print("fill  (1 GiB write):              %.2f TB/s" % (x.numel() / t(lambda: x.fill_(1)) / 1e12))
print("sum   (1 GiB read, f32):          %.2f TB/s" % (x.numel() / t(lambda: xf.sum()) / 1e12))
print("axpy  (2 reads + 1 write, f32):   %.2f TB/s" % (3 * x.numel() / t(lambda: torch.add(xf, yf, out=yf)) / 1e12))
