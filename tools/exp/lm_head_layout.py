import torch, torch.nn.functional as F
dev = torch.device("cuda", 0)
x = torch.randn(256, 4096, device=dev, dtype=torch.float16)
W = torch.randn(4096, 128256, device=dev, dtype=torch.float16) * 0.02
Wt = W.t().contiguous()
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
print("matmul [K,N]   us", round(t(lambda: torch.matmul(x, W)), 1))
print("linear [N,K]   us", round(t(lambda: F.linear(x, Wt)), 1))
print("equal", torch.allclose(torch.matmul(x, W), F.linear(x, Wt), atol=1e-2, rtol=1e-2))
