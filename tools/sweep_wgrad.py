"""Split-K sweep for the wgrad-shaped GEMMs of the step (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stonkgs_amd import _hip as hip  # noqa: E402
from bench_kernels import timeit  # noqa: E402

hip.lib()
K = 32768
for Mo, No in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    A = torch.randn(Mo, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(No, K, device="cuda").to(torch.bfloat16)
    C = torch.zeros(Mo, No, device="cuda")
    for name, dbg in (("v1", hip.EPI_DEBUG_V1), ("v2", hip.EPI_DEBUG_V2)):
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            def f():
                hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), No, Mo, No, K,
                         hip.EPI_OUT_F32_ATOMIC | dbg, 0, 0, 0, 0, 0, 1.0, sk, 0, 0, 0.0, 0, hip.stream_ptr())
            t = timeit(f, iters=10)
            print(f"wgrad {Mo}x{No} {name} split={sk}: {t*1e6:.1f} us {2*Mo*No*K/t/1e12:.0f} TF/s", flush=True)
