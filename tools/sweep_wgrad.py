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

# label-sparse entity-decoder weight gradient: dW[175104 x 768] += dlogits[cnt x 175104]^T . hs[cnt x 768], token count on
# the device (2432 of a 16 384-row capacity), 5.7 GB operand extent
if os.environ.get("STONK_SWEEP_DECODER", "1") == "1":
    Mo, No, cap, cnt_v = 175104, 768, 16384, 2432
    dY = torch.zeros(cnt_v + 64, Mo, device="cuda", dtype=torch.bfloat16)   # only the live rows (+ one K tile) are touched
    dY[:cnt_v].normal_()
    X = torch.randn(cap, No, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(Mo, No, device="cuda")
    cnt = torch.tensor([cnt_v], device="cuda", dtype=torch.int32)
    for sk, name in ((1, "128x128"), (0, "256x256 four-wave"), (-160, "256x256 four-wave on 160 CUs")):
        def f():
            hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, 0, Mo, No, cap, 1.0, sk,
                     hip.ptr(cnt), hip.stream_ptr())
        t = timeit(f, iters=5)
        print(f"wgrad_tn decoder {Mo}x{No}, {cnt_v} live tokens, {name}: {t*1e6:.1f} us {2*Mo*No*cnt_v/t/1e12:.0f} TF/s", flush=True)
    ref = torch.zeros(Mo, No, device="cuda")
    dW.zero_()
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(ref), No, 0, Mo, No, cap, 1.0, 1, hip.ptr(cnt),
             hip.stream_ptr())
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, 0, Mo, No, cap, 1.0, 0, hip.ptr(cnt),
             hip.stream_ptr())
    print("decoder four-wave vs 128x128 max |diff|:", float((dW - ref).abs().max()), "scale", float(ref.abs().max()), flush=True)
