"""The N = 768 launches of the step on the four-wave kernel: 256x256 against 256x192 tiles (us per launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K, fl, name in ((32768, 768, 3072, hip.EPI_RESID, "dgrad FFN-up + residual"),
                          (32768, 768, 2304, hip.EPI_RESID, "dgrad QKV + residual"),
                          (32768, 768, 3072, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT, "FFN-down forward"),
                          (32768, 768, 768, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT, "attention-output forward"),
                          (16384, 768, 3072, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT, "backbone FFN-down forward"),
                          (16384, 768, 768, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT, "backbone attention-output forward"),
                          (32768, 768, 768, 0, "attention-output dgrad (plain)"),
                          (32768, 2304, 768, hip.EPI_BIAS, "fused QKV forward")):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    for kname, kern in (("128x128", hip.GEMM_TILE128), ("eight-wave 256x256", hip.GEMM_WAVE8), ("four-wave 256x256", hip.GEMM_WAVE4),
                        ("four-wave 256x192", hip.GEMM_WAVE4_192)):
        def f():
            hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, fl, hip.ptr(bias), hip.ptr(res), N,
                     0, 0, 1.0, 1, 0, 0, 0.1, 7, kern, hip.stream_ptr())
        try:
            t = timeit(f)
        except hip.StonkHipError:
            continue
        print(f"{name} {M}x{N}x{K} {kname}: {t:.1f} us  {2 * M * N * K / t / 1e6:.0f} TFLOP/s", flush=True)
