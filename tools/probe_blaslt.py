import torch
for M, N, K in [(32768, 3072, 768), (32768, 768, 768), (32768, 768, 3072), (32768, 2304, 768)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        C = torch.matmul(A, B.t())
    torch.cuda.synchronize()
