import torch
for M, N, K in [(24500, 512, 4608), (24500, 2048, 512), (24500, 512, 2048), (98000, 512, 1024), (6400, 1024, 1024), (50432, 4096, 1024), (98000, 256, 2304)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3): torch.matmul(A, W.T)
    torch.cuda.synchronize()
