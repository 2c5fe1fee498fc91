"""The label-sparse decoders' two NT launches ALONE, at the benchmark's sizes (capacity 16 384 rows, a device-side count of
labelled rows; entity vocabulary 175 104 padded, text 29 056): forward logits (fp16 output) and dgrad (fp32 atomics over a
split contraction), each on round 3's kernel and on the written-out four-wave kernel, interleaved.
  python tools/decoder_probe.py > gpurun_out/decoder_probe.log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

H, CAP = 768, 16384
WAVE8, ASM4, ASM4_192 = hip.GEMM_WAVE8, hip.GEMM_ASM4, hip.GEMM_ASM4_192


def gemm(A, B, C, M, N, K, flags, split_k, m_dev, kernel):
    hip.call("stonk_gemm_nt_bf16", A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), M, N, K,
             flags, 0, 0, 0, 0, 0, 1.0, split_k, m_dev.data_ptr(), 0, 0.0, 0, kernel, hip.stream_ptr())


def timeit(fns, reps=12):
    """interleaved: one launch of every variant per repetition; mean and min us per variant"""
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for _ in fns]
    for f in fns:
        f()
    torch.cuda.synchronize()
    for r in range(reps):
        for i, f in enumerate(fns):
            ev[i][r][0].record()
            f()
            ev[i][r][1].record()
    torch.cuda.synchronize()
    out = []
    for i in range(len(fns)):
        t = [a.elapsed_time(b) * 1e3 for a, b in ev[i]]
        out.append((sum(t) / len(t), min(t)))
    return out


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, npad, cnt in (("entity", 175104, 2458), ("text", 29056, 2458)):
        m_dev = torch.tensor([cnt], device=dev, dtype=torch.int32)
        hs = (torch.randn(CAP, H, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        W = (torch.randn(npad, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        Wt = W.t().contiguous()
        logits = torch.empty(CAP, npad, device=dev, dtype=torch.float16)
        dl = torch.empty(CAP, npad, device=dev, dtype=torch.bfloat16)
        dl[:cnt] = (torch.randn(cnt, npad, device=dev, generator=g) * 0.01).to(torch.bfloat16)
        dhs = torch.zeros(CAP, H, device=dev)
        gflop = 2.0 * cnt * npad * H / 1e9
        fwd = timeit([lambda: gemm(hs, W, logits, CAP, npad, H, hip.EPI_OUT_F16, 1, m_dev, WAVE8),
                      lambda: gemm(hs, W, logits, CAP, npad, H, hip.EPI_OUT_F16, 1, m_dev, ASM4),
                      lambda: gemm(hs, W, logits, CAP, npad, H, hip.EPI_OUT_F16, 1, m_dev, 0)])
        print(f"{name:7s} forward  {cnt} x {npad} x {H} ({gflop:.0f} GFLOP, {cnt * npad * 2 / 1e6:.0f} MB of fp16 logits): "
              + "  ".join(f"{n} {m:7.1f} us ({gflop / m * 1e-3:.2f} PF/s, min {mn:.1f})" for n, (m, mn) in zip(("wave8", "a4_256", "auto"), fwd)), flush=True)
        variants = [("wave8/8", WAVE8, 8), ("a4_192/64", ASM4_192, 64), ("a4_192/8", ASM4_192, 8)]
        bwd = timeit([(lambda k=k, s=s: gemm(dl, Wt, dhs, CAP, H, npad, hip.EPI_OUT_F32_ATOMIC, s, m_dev, k)) for _, k, s in variants])
        print(f"{name:7s} dgrad    {cnt} x {H} x {npad}: "
              + "  ".join(f"{n} {m:7.1f} us ({gflop / m * 1e-3:.2f} PF/s, min {mn:.1f})" for (n, _, _), (m, mn) in zip(variants, bwd)), flush=True)
        # agreement of the two kernels on the same operands
        a = torch.zeros(CAP, H, device=dev)
        b = torch.zeros(CAP, H, device=dev)
        gemm(dl, Wt, a, CAP, H, npad, hip.EPI_OUT_F32_ATOMIC, 8, m_dev, WAVE8)
        gemm(dl, Wt, b, CAP, H, npad, hip.EPI_OUT_F32_ATOMIC, 64, m_dev, ASM4_192)
        print(f"        dgrad max |a4 - wave8| = {(a - b).abs().max().item():.3e} (max |wave8| {a.abs().max().item():.3e})", flush=True)
        del hs, W, Wt, logits, dl, dhs, a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
