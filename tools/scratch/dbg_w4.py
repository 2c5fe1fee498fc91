import sys, os, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from stonkgs_amd import _hip as hip
import test_gemm_gpu as T
hip.lib()
for (M, N, K) in [(256, 384, 768), (256, 512, 768), (512, 384, 768), (1000, 768, 3072)]:
    A, B = T._rand((M, K), seed=1), T._rand((N, K), seed=2)
    ref = A.float() @ B.float().t()
    C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    T._gemm(hip, A, B, torch.bfloat16, flags=hip.EPI_OUT_BF16 | (1 << 20), C=C)
    torch.cuda.synchronize()
    bad = ((C.float() - ref).abs() > 0.5)
    print(M, N, K, "bad", bad.sum().item())
    if bad.any():
        idx = bad.nonzero()
        rows = idx[:, 0].unique(); cols = idx[:, 1].unique()
        print(" rows", rows[:40].tolist(), len(rows)); print(" cols", cols[:80].tolist(), len(cols))
        print(" vals", C[idx[0, 0], idx[0, 1]].item(), ref[idx[0, 0], idx[0, 1]].item())
