"""Launch the attention kernels a few times at the bench shape (target of rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

hip.lib()
B, S, NH = 64, 512, 12
H = NH * 64
qkv = torch.randn(B * S, 3 * H, device="cuda").to(torch.bfloat16)
dout = torch.randn(B * S, H, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.long, device="cuda")
mask[:, 200:256] = 0
out = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, NH, S, device="cuda")
delta = torch.empty(B, NH, S, device="cuda")
dqkv = torch.empty_like(qkv)
for p in (0.0, 0.1):
    for _ in range(3):
        hip.call("stonk_attention_fwd", hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask), 0, 0,
                 hip.ptr(out), H, hip.ptr(lse), B, NH, S, 64, 0.125, p, 1, hip.stream_ptr())
        hip.call("stonk_attention_bwd", hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask), 0, 0,
                 hip.ptr(out), H, hip.ptr(dout), H, hip.ptr(lse), hip.ptr(delta), hip.ptr(dqkv), hip.ptr(dqkv) + 2 * H,
                 3 * H, hip.ptr(dqkv) + 4 * H, B, NH, S, 64, 0.125, p, 1, hip.stream_ptr())
    torch.cuda.synchronize()
