"""Launch the four-wave 256x256 GEMM (and the vendor library) on one large shape (target of rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

hip.lib()
M = N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, 0, 0, 0, 0, 0, 0,
             1.0, 1, 0, 0, 0.0, 0, hip.GEMM_WAVE4, hip.stream_ptr())
torch.cuda.synchronize()
for _ in range(3):
    torch.matmul(A, B.t())
torch.cuda.synchronize()
