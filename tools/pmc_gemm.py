"""Launch a few epilogue variants of the 256x256 GEMM once each (target of rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

hip.lib()
M, N, K = 32768, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
for fl in (0, hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT, hip.EPI_GELU_BWD, hip.EPI_BIAS | hip.EPI_RESID):
    for _ in range(3):
        hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, fl,
                 hip.ptr(bias), hip.ptr(res), N, hip.ptr(aux), N, 1.0, 1, 0, 0, 0.1, 7, hip.GEMM_WAVE8, hip.stream_ptr())
    torch.cuda.synchronize()
