"""The entity decoder's weight gradient alone (dW[175104, 768] += dlogits[T, 175104]^T . h[T, 768], T = 2432 labelled rows by
a device-side count, 2052 unsplit 256x256 tiles): us per launch on all CUs and on the 160-CU share the step gives it, and a
check against torch on a slice. Run under STONK_HIP_LIB=ab_ref/libstonk_hip.so for the other build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

M, N, T, CAP = 175104, 768, 2432, 2560
g = torch.Generator(device="cuda").manual_seed(0)
dy = (torch.randn(CAP, M, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
x = torch.randn(CAP, N, device="cuda", generator=g).to(torch.bfloat16)
cnt = torch.tensor([T], dtype=torch.int32, device="cuda")
dW = torch.zeros(M, N, device="cuda")


def run(split):
    hip.call("stonk_gemm_tn_bf16", dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dW.data_ptr(), dW.stride(0), 0,
             M, N, CAP, 1.0, split, cnt.data_ptr(), hip.stream_ptr())


run(0)
torch.cuda.synchronize()
ref = dy[:T, :1024].float().t() @ x[:T].float()
err = float((dW[:1024] - ref).abs().max() / ref.abs().max())
run(0)
torch.cuda.synchronize()
err2 = float((dW[:1024] - 2 * ref).abs().max() / ref.abs().max())   # accumulates: += semantics
print(f"max rel err first launch {err:.2e}, after a second (accumulating) launch {err2:.2e}", flush=True)
for split, name in ((0, "all CUs"), (-160, "160 CUs")):
    for _ in range(2):
        run(split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(split)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name}: {t:.0f} us  {2.0 * M * N * T / t / 1e6:.0f} TFLOP/s", flush=True)
