"""Micro-benchmarks of individual C-ABI kernels on one MI355X (development aid, not the judged bench)."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    shapes = [(32768, 2304, 768), (32768, 768, 768), (32768, 3072, 768), (32768, 768, 3072), (16384, 2304, 768),
              (4096, 4096, 4096), (8192, 8192, 8192)]
    for M, N, K in shapes:
        A = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for name, dbg in (("v1_128", hip.GEMM_TILE128), ("v2_256", hip.GEMM_WAVE8), ("w4_256", hip.GEMM_WAVE4)):
            def f():
                hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, 0, 0, 0, 0, 0, 0,
                         1.0, 1, 0, 0, 0.0, 0, dbg, hip.stream_ptr())
            t = timeit(f)
            print(f"gemm {M}x{N}x{K} {name}: {t*1e6:.1f} us  {2*M*N*K/t/1e12:.1f} TF/s", flush=True)
        t = timeit(lambda: torch.matmul(A, B.t()))
        print(f"gemm {M}x{N}x{K} torch(hipBLASLt): {t*1e6:.1f} us  {2*M*N*K/t/1e12:.1f} TF/s", flush=True)
    # wgrad-shaped: small output, long K, split-K atomics
    for Mo, No, K, sk, dbg in [(768, 768, 32768, 16, hip.GEMM_TILE128), (768, 768, 32768, 28, hip.GEMM_WAVE8),
                               (3072, 768, 32768, 5, hip.GEMM_TILE128), (3072, 768, 32768, 7, hip.GEMM_WAVE8),
                               (3072, 768, 32768, 14, hip.GEMM_WAVE8), (2304, 768, 32768, 6, hip.GEMM_TILE128),
                               (2304, 768, 32768, 9, hip.GEMM_WAVE8), (2304, 768, 32768, 19, hip.GEMM_WAVE8)]:
        A = torch.randn(Mo, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(No, K, device="cuda").to(torch.bfloat16)
        C = torch.zeros(Mo, No, device="cuda")
        def f():
            hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), No, Mo, No, K,
                     hip.EPI_OUT_F32_ATOMIC, 0, 0, 0, 0, 0, 1.0, sk, 0, 0, 0.0, 0, dbg, hip.stream_ptr())
        t = timeit(f)
        print(f"wgrad {Mo}x{No}x{K} splitk={sk} {'v2' if dbg == hip.GEMM_WAVE8 else 'v1'}: {t*1e6:.1f} us  "
              f"{2*Mo*No*K/t/1e12:.1f} TF/s", flush=True)


def bench_ln():
    rows, H = 32768, 768
    x = torch.randn(rows, H, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(H, device="cuda")
    b = torch.zeros(H, device="cuda")
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    t = timeit(lambda: hip.call("stonk_layernorm_fwd", hip.ptr(x), hip.ptr(g), hip.ptr(b), hip.ptr(y), hip.ptr(mean),
                                hip.ptr(rstd), rows, H, 1e-12, 0, 0.0, 0, hip.stream_ptr()))
    print(f"layernorm_fwd {rows}x{H}: {t*1e6:.1f} us  {rows*H*4/t/1e9:.0f} GB/s", flush=True)
    dx = torch.empty_like(x)
    dg = torch.zeros(H, device="cuda")
    db = torch.zeros(H, device="cuda")
    LN_WS = torch.empty(1024 * 2 * H, device="cuda")
    t = timeit(lambda: hip.call("stonk_layernorm_bwd", hip.ptr(y), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(g),
                                hip.ptr(dx), 0, hip.ptr(dg), hip.ptr(db), rows, H, 0, 0.0, 0, 0.0, 0, hip.ptr(LN_WS), LN_WS.numel(), hip.stream_ptr()))
    print(f"layernorm_bwd {rows}x{H}: {t*1e6:.1f} us  {rows*H*6/t/1e9:.0f} GB/s", flush=True)
    dxd = torch.empty_like(x)
    t = timeit(lambda: hip.call("stonk_layernorm_bwd", hip.ptr(y), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(g),
                                hip.ptr(dx), hip.ptr(dxd), hip.ptr(dg), hip.ptr(db), rows, H, 0, 0.0, 0, 0.1, 5, hip.ptr(LN_WS),
                                LN_WS.numel(), hip.stream_ptr()))
    print(f"layernorm_bwd + dropped copy {rows}x{H}: {t*1e6:.1f} us  {rows*H*8/t/1e9:.0f} GB/s", flush=True)




def bench_xent():
    cap, N = 16384, 175094
    npad = (N + 127) // 128 * 128
    for cnt_v in (2432, 2048, 1500):
        logits = torch.randn(cnt_v + 64, npad, device="cuda")
        dl = torch.empty(cnt_v + 64, npad, device="cuda", dtype=torch.bfloat16)
        tg = torch.randint(0, N, (cap,), device="cuda", dtype=torch.int32)
        cnt = torch.tensor([cnt_v], device="cuda", dtype=torch.int32)
        acc = torch.zeros(1, device="cuda")
        err = torch.zeros(1, device="cuda", dtype=torch.int32)
        t = timeit(lambda: hip.call("stonk_softmax_xent_fwd_bwd", hip.ptr(logits), npad, N, npad, hip.ptr(tg), hip.ptr(cnt),
                                    hip.ptr(acc), hip.ptr(dl), npad, 1.0, cnt_v + 64, hip.ptr(err), hip.stream_ptr()), iters=10)
        print(f"softmax_xent {cnt_v} rows x {N}: {t*1e6:.1f} us  {cnt_v*npad*10/t/1e9:.0f} GB/s (2 fp32 reads + bf16 write)",
              flush=True)


def bench_attn():
    B, S, NH = 64, 512, 12
    H = NH * 64
    qkv = (torch.randn(B * S, 3 * H, device="cuda")).to(torch.bfloat16)
    dout = torch.randn(B * S, H, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, S, dtype=torch.long, device="cuda")
    mask[:, 200:256] = 0
    out = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, NH, S, device="cuda")
    delta = torch.empty(B, NH, S, device="cuda")
    dqkv = torch.empty_like(qkv)
    for p in (0.0, 0.1):
        def f():
            hip.call("stonk_attention_fwd", hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H,
                     hip.ptr(mask), 0, 0, hip.ptr(out), H, hip.ptr(lse), B, NH, S, 64, 0.125, p, 1, hip.stream_ptr())
        t = timeit(f)
        fl = 4.0 * B * NH * S * S * 64
        print(f"attn_fwd B{B} S{S} NH{NH} p={p}: {t*1e6:.1f} us  {fl/t/1e12:.1f} TF/s", flush=True)
        def g():
            hip.call("stonk_attention_bwd", hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H,
                     hip.ptr(mask), 0, 0, hip.ptr(out), H, hip.ptr(dout), H, hip.ptr(lse), hip.ptr(delta), hip.ptr(dqkv),
                     hip.ptr(dqkv) + 2 * H, 3 * H, hip.ptr(dqkv) + 4 * H, B, NH, S, 64, 0.125, p, 1, hip.stream_ptr())
        t = timeit(g)
        print(f"attn_bwd B{B} S{S} NH{NH} p={p}: {t*1e6:.1f} us  {2.5*fl/t/1e12:.1f} TF/s (algorithmic 10*B*NH*S^2*D)",
              flush=True)


def bench_data():
    """On-device batch assembly + dynamic masking (f1) at the bench batch size."""
    import numpy as np
    from stonkgs_amd.data import DeviceBatcher
    B, half, V, K = 64, 256, 28996, 175094
    rng = np.random.RandomState(0)
    text = torch.from_numpy(rng.randint(1000, V, (B, half))).cuda()
    att = torch.ones(B, half, dtype=torch.long, device="cuda")
    walks = torch.from_numpy(rng.randint(0, K, (K, half // 2 - 1)))
    src = torch.from_numpy(rng.randint(0, K, B)).cuda()
    tgt = torch.from_numpy(rng.randint(0, K, B)).cuda()
    bat = DeviceBatcher(walks, V, K, seed=1)
    t = timeit(lambda: bat(text, att, src, tgt, 3), iters=50)
    print(f"device batcher B={B}: {t*1e6:.1f} us per batch (assemble + mask, incl. output allocation)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "ln"]
    hip.lib()
    for w in which:
        globals()["bench_" + w]()
