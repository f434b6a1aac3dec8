import torch
dev = torch.device("cuda:0")
for (M, N, K) in ((12800, 5120, 1088), (12800, 1024, 5120), (12800, 1024, 1024), (12800, 2048, 1024), (12800, 1024, 2048)):
    A = torch.randn(M, K, device=dev).half(); Bt = torch.randn(N, K, device=dev).half(); C = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3): torch.matmul(A, Bt.t(), out=C)
torch.cuda.synchronize()
