"""Weight-gradient (TN) kernels on the shapes of the step: 128x128 tiles (forced, split sweep) vs persistent 256x256."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stonkgs_amd import _hip as hip  # noqa: E402
from bench_kernels import timeit  # noqa: E402

hip.lib()
T = 32768
for Mo, No in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    dY = torch.randn(T, Mo, device="cuda").to(torch.bfloat16)
    X = torch.randn(T, No, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(Mo, No, device="cuda")
    db = torch.zeros(Mo, device="cuda")
    for sk in (3, 4, 6, 12, -1, 0):
        def f():
            hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, hip.ptr(db), Mo, No, T, 1.0,
                     sk, 0, hip.stream_ptr())
        t = timeit(f, iters=10)
        name = f"128x128 split={sk}" if sk > 0 else ("256x256 four-wave" if sk == 0 else "256x256 eight-wave")
        print(f"wgrad_tn {Mo}x{No} {name}: {t*1e6:.1f} us {2*Mo*No*T/t/1e12:.0f} TF/s", flush=True)
