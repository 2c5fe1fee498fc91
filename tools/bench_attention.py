"""Attention kernels alone at the step's shape (B 64, S 512, 12 heads, key-padding mask, dropout 0.1): forward and
backward (dQ kernel + dK/dV kernel), us per launch and algorithmic TFLOP/s (forward 4*S*S*64 per head, backward
10*S*S*64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

B, S, NH = int(os.environ.get("B", 64)), int(os.environ.get("S", 512)), int(os.environ.get("NH", 12))
H = NH * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * S, 3 * H, device="cuda", generator=g) * 1.0).to(torch.bfloat16)
dout = torch.randn(B * S, H, device="cuda", generator=g).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.long, device="cuda")
for b in range(B):
    mask[b, 32 + 3 * b: S // 2] = 0
out = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, NH, S, device="cuda")
delta = torch.empty(B, NH, S, device="cuda")
dqkv = torch.zeros_like(qkv)
p, seed = float(os.environ.get("P", 0.1)), 3


def fwd():
    hip.call("stonk_attention_fwd", hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask), 0, 0,
             hip.ptr(out), H, hip.ptr(lse), B, NH, S, 64, 0.125, p, seed, hip.stream_ptr())


def bwd(name):
    hip.call(name, hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask), 0, 0, hip.ptr(out), H,
             hip.ptr(dout), H, hip.ptr(lse), hip.ptr(delta), hip.ptr(dqkv), hip.ptr(dqkv) + 2 * H, 3 * H,
             hip.ptr(dqkv) + 4 * H, B, NH, S, 64, 0.125, p, seed, hip.stream_ptr())


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


fl = B * NH * S * S * 64
t = timeit(fwd)
print(f"forward: {t:.1f} us  {4 * fl / t / 1e6:.0f} TFLOP/s", flush=True)
t = timeit(lambda: bwd("stonk_attention_bwd"))
print(f"backward: {t:.1f} us  {10 * fl / t / 1e6:.0f} TFLOP/s (algorithmic)", flush=True)
