"""Epilogue cost of the GEMM kernels on the FFN-up shape (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stonkgs_amd import _hip as hip  # noqa: E402
from bench_kernels import timeit  # noqa: E402

hip.lib()
for M, N, K in [(32768, 3072, 768), (32768, 2304, 768), (16384, 3072, 768), (32768, 768, 3072), (32768, 768, 768)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    cases = {"plain": 0,  "bias": hip.EPI_BIAS, "bias+gelu": hip.EPI_BIAS | hip.EPI_GELU,
             "bias+gelu+save": hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT,
             "bias+gelu+save'": hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT | hip.EPI_AUX_GRAD,
             "gelu_bwd": hip.EPI_GELU_BWD, "mul_aux": hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD, "bias+resid": hip.EPI_BIAS | hip.EPI_RESID,
             "bias+drop+resid": hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT}
    for kname, dbg in (("v1", hip.GEMM_TILE128), ("v2", hip.GEMM_WAVE8), ("w4", hip.GEMM_WAVE4)):
        for cname, fl in cases.items():
            def f():
                hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, fl,
                         hip.ptr(bias), hip.ptr(res), N, hip.ptr(aux), N, 1.0, 1, 0, 0, 0.1, 7, dbg, hip.stream_ptr())
            t = timeit(f, iters=10)
            print(f"{M}x{N}x{K} {kname} {cname}: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.0f} TF/s", flush=True)
