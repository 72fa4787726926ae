import torch
M=N=K=8192
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
for _ in range(10): torch.matmul(a, w.t())
torch.cuda.synchronize()
