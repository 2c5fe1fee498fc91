"""GPU probe of the written-out weight-gradient kernel (gemm_tn_a4.hip) against the compiled four-wave one it replaces:
parity on the step's shapes, then interleaved timing - on all CUs (split_k = 0 / -2) and held to 160 CUs' worth of workgroups
(-160 / -2160), as the training step launches it on its second stream."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402


def tn(dY, X, dW, db, sk, alpha=1.0, k_dev=None):
    T, Mo = dY.shape
    No = X.shape[1]
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), dY.stride(0), hip.ptr(X), X.stride(0), hip.ptr(dW), dW.stride(0), hip.ptr(db), Mo, No, T,
             alpha, sk, hip.ptr(k_dev), hip.stream_ptr())


def main():
    hip.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    T = 26432
    shapes = [("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072)]
    ok = True
    for name, Mo, No in shapes:
        dY = (torch.randn(T, Mo, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        X = (torch.randn(T, No, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        ref = dY.float().t() @ X.float()
        refb = dY.float().sum(0)
        for sk in (0, -160):
            dW = torch.zeros(Mo, No, device="cuda")
            db = torch.zeros(Mo, device="cuda")
            tn(dY, X, dW, db, sk)
            torch.cuda.synchronize()
            e = float((dW - ref).norm() / ref.norm())
            eb = float((db - refb).norm() / refb.norm())
            bad = e > 1e-4 or eb > 1e-4
            ok &= not bad
            print(f"{name} {Mo}x{No} T{T} split_k {sk}: rel dW {e:.2e} db {eb:.2e}{'  <-- FAIL' if bad else ''}", flush=True)
        if not ok:
            break
        arms = {"a4": 0, "w4": -2, "a4@160": -160, "w4@160": -2160}
        res = {k: [] for k in arms}
        dW = torch.zeros(Mo, No, device="cuda")
        db = torch.zeros(Mo, device="cuda")
        for k, sk in arms.items():
            for _ in range(3):
                tn(dY, X, dW, db, sk)
        torch.cuda.synchronize()
        for rnd in range(7):
            for k, sk in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    tn(dY, X, dW, db, sk)
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        fl = 2.0 * Mo * No * T
        print(f"{name:9s} {Mo}x{No}: " + "  ".join(f"{k} {statistics.median(v):7.1f} us ({fl / statistics.median(v) / 1e6:5.0f} TF/s)"
                                                 for k, v in res.items()), flush=True)
    print("TN PROBE", "PASSED" if ok else "FAILED", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
