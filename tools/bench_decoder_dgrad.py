"""Label-sparse entity-decoder dgrad: dHs[cnt x 768] += dlogits[cnt x 175104] . W^T-copy[768 x 175104]^T (fp32 atomics, split-K),
128x128 kernel against the 256x256 ones (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stonkgs_amd import _hip as hip  # noqa: E402
from bench_kernels import timeit  # noqa: E402

hip.lib()
cap, cnt_v, N, K = 16384, 2432, 768, 175104
A = torch.zeros(cnt_v + 256, K, device="cuda", dtype=torch.bfloat16)
A[:cnt_v].normal_()
B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
C = torch.zeros(cap, N, device="cuda")
cnt = torch.tensor([cnt_v], device="cuda", dtype=torch.int32)
ref = None
for name, dbg, sks in (("128x128", hip.GEMM_TILE128, (16, 32)), ("256x256 eight-wave", hip.GEMM_WAVE8, (4, 8, 9, 12)),
                       ("256x256 four-wave", hip.GEMM_WAVE4, (8, 9))):
    for sk in sks:
        def f():
            return hip.lib().stonk_gemm_nt_bf16(hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, cap, N, K,
                                                hip.EPI_OUT_F32_ATOMIC, 0, 0, 0, 0, 0, 1.0, sk, hip.ptr(cnt), 0, 0.0, 0,
                                                dbg, hip.stream_ptr())
        rc = f()
        if rc != 0:
            print(f"{name} split {sk}: refused ({rc})", flush=True)
            continue
        t = timeit(f, iters=5)
        C.zero_()
        f()
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        err = float((C - ref).abs().max())
        print(f"decoder dgrad {name} split {sk}: {t*1e6:.1f} us {2*cnt_v*N*K/t/1e12:.0f} TF/s  max|diff| {err:.3g} "
              f"(scale {float(ref.abs().max()):.3g})", flush=True)
