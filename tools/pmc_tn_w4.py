"""Launch the four-wave weight-gradient kernel on the three shapes it takes in the step (target of the rocprofv3 --pmc
passes behind bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, MI355X_MICROARCH.md "HBM")."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

hip.lib()
T = 32768
bufs = []
for Mo, No in [(2304, 768), (3072, 768), (768, 3072)]:
    dY = torch.randn(T, Mo, device="cuda").to(torch.bfloat16)
    X = torch.randn(T, No, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(Mo, No, device="cuda")
    db = torch.zeros(Mo, device="cuda")
    bufs.append((Mo, No, dY, X, dW, db))
# interleave the shapes as the step does, and sweep 400 MB between launches so no operand is still in the 256 MiB
# Infinity Cache from the previous launch of the same shape (in the step, dY was just written and is partly resident)
junk = torch.empty(400 << 20, dtype=torch.uint8, device="cuda")
for rep in range(4):
    for Mo, No, dY, X, dW, db in bufs:
        junk.add_(1)
        hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, hip.ptr(db), Mo, No, T, 1.0, 0, 0,
                 hip.stream_ptr())
torch.cuda.synchronize()
print("done", flush=True)
