import sys, os, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from stonkgs_amd import _hip as hip
import test_gemm_gpu as T
hip.lib()
M, N, K = 256, 256, 768
A = torch.zeros(M, K, device="cuda"); A[torch.arange(M), torch.arange(M)] = 1.0
A = A.to(torch.bfloat16)
for name, Bf in (("col", (torch.arange(N, device="cuda") % 128).float()[:, None].expand(N, K)),
                 ("row", (torch.arange(K, device="cuda") % 128).float()[None, :].expand(N, K))):
    B = Bf.contiguous().to(torch.bfloat16)
    ref = A.float() @ B.float().t()
    C = torch.full((M, N), -1.0, device="cuda", dtype=torch.bfloat16)
    T._gemm(hip, A, B, torch.bfloat16, flags=hip.EPI_OUT_BF16 | (1 << 20), C=C)
    torch.cuda.synchronize()
    bad = (C.float() != ref).nonzero()
    print(name, "bad", len(bad))
    for i in range(0, min(len(bad), 24)):
        m, n = bad[i].tolist()
        print("  at (m=%d,n=%d) got %s expected %s" % (m, n, C[m, n].item(), ref[m, n].item()))
