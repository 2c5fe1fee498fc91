"""GPU probe of the written-out four-wave NT kernel (gemm_a4.hip): parity against torch fp32 on the same bf16 inputs for
every epilogue instance and both tile widths, a race screen (bit-identical repeats), then interleaved timing against the
compiled four-wave kernel and the vendor library on the training step's shapes (random operands, real epilogues).
    python tools/a4_probe.py [check] [time] [T]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402


def rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


def gemm(A, B, flags=0, bias=None, resid=None, aux=None, kernel=0, drop_p=0.0, seed=0, C=None, m_dev=None, alpha=1.0):
    M, K = A.shape
    N = B.shape[0]
    if C is None:
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    hip.call("stonk_gemm_nt_bf16", hip.ptr(A), A.stride(0), hip.ptr(B), B.stride(0), hip.ptr(C), C.stride(0), M, N, K,
             flags, hip.ptr(bias), hip.ptr(resid), 0 if resid is None else resid.stride(0), hip.ptr(aux),
             0 if aux is None else aux.stride(0), alpha, 1, hip.ptr(m_dev), 0, drop_p, seed, kernel, hip.stream_ptr())
    return C


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def check():
    ok = True
    for kern, name in ((hip.GEMM_ASM4, "a4_256"), (hip.GEMM_ASM4_192, "a4_192")):
        for (M, N, K) in ((256, 768, 128), (512, 768, 256), (1000, 768, 768), (2050, 1536, 3072), (26411 // 8 * 8, 768, 768),
                          (4096, 2304, 768)):
            if kern == hip.GEMM_ASM4_192 and N % 192:
                continue
            A, B = rand((M, K), 1.0, 1), rand((N, K), 0.05, 2)
            ref = A.float() @ B.float().t()
            C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
            gemm(A, B, kernel=kern, C=C)
            torch.cuda.synchronize()
            e = rel(C, ref)
            bad = e > 6e-3
            ok &= not bad
            print(f"{name} plain {M}x{N}x{K}: rel {e:.2e}{'  <-- FAIL' if bad else ''}", flush=True)
            if bad:
                d = (C.float() - ref).abs()
                rows = (d.max(dim=1).values > 0.1).nonzero().flatten()
                cols = (d.max(dim=0).values > 0.1).nonzero().flatten()
                print("   bad rows", rows[:16].tolist(), "n", len(rows), "| bad cols", cols[:16].tolist(), "n", len(cols))
        # epilogues
        M, N, K = 1000, 768, 256
        A, B = rand((M, K), 0.5, 3), rand((N, K), 0.5, 4)
        bias = torch.randn(N, device="cuda")
        resid = rand((M, N), 1.0, 5)
        pre = A.float() @ B.float().t() + bias
        cases = [("bias", hip.EPI_BIAS, dict(bias=bias), pre),
                 ("resid", hip.EPI_RESID, dict(resid=resid), pre - bias + resid.float()),
                 ("bias+resid", hip.EPI_BIAS | hip.EPI_RESID, dict(bias=bias, resid=resid), pre + resid.float())]
        if kern == hip.GEMM_ASM4:
            u = rand((M, N), 1.0, 6)
            uf = u.float().requires_grad_(True)
            (gp,) = torch.autograd.grad(torch.nn.functional.gelu(uf).sum(), uf)
            cases += [("bias+gelu", hip.EPI_BIAS | hip.EPI_GELU, dict(bias=bias), torch.nn.functional.gelu(pre)),
                      ("gelu_bwd", hip.EPI_GELU_BWD, dict(aux=u), (pre - bias) * gp),
                      ("gelu_bwd+auxgrad", hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD, dict(aux=u), (pre - bias) * u.float())]
        for cname, fl, kw, want in cases:
            out = gemm(A, B, flags=fl, kernel=kern, **kw)
            e = rel(out, want)
            bad = e > 8e-3
            ok &= not bad
            print(f"{name} {cname}: rel {e:.2e}{'  <-- FAIL' if bad else ''}", flush=True)
        if kern == hip.GEMM_ASM4:
            for ag in (0, hip.EPI_AUX_GRAD):
                aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                out = gemm(A, B, flags=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT | ag, kernel=kern, bias=bias, aux=aux)
                pf = pre.clone().requires_grad_(True)
                (gpre,) = torch.autograd.grad(torch.nn.functional.gelu(pf).sum(), pf)
                e1, e2 = rel(out, torch.nn.functional.gelu(pre)), rel(aux, gpre if ag else pre)
                bad = e1 > 8e-3 or e2 > 8e-3
                ok &= not bad
                print(f"{name} bias+gelu+save{'+auxgrad' if ag else ''}: rel out {e1:.2e} aux {e2:.2e}{'  <-- FAIL' if bad else ''}", flush=True)
        # bias + dropout + residual: same mask as the 128x128 kernel
        A4, B4 = rand((4096, 3072), 0.5, 37), rand((768, 3072), 0.05, 38)
        bias = torch.randn(768, device="cuda")
        resid = rand((4096, 768), 1.0, 39)
        fl = hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT
        out = gemm(A4, B4, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=77, kernel=kern)
        out_v1 = gemm(A4, B4, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=77, kernel=hip.GEMM_TILE128)
        e = rel(out, out_v1)
        bad = e > 8e-3
        ok &= not bad
        print(f"{name} bias+dropout+resid vs 128x128 kernel: rel {e:.2e}{'  <-- FAIL' if bad else ''}", flush=True)
        # device-side row count, ragged last tile
        M, N, K = 16384, 768, 1024
        A, B = rand((M, K), 0.5, 31), rand((N, K), 0.5, 32)
        m_dev = torch.tensor([2432 - 5], device="cuda", dtype=torch.int32)
        C = torch.full((M, N), -7.0, device="cuda", dtype=torch.bfloat16)
        gemm(A, B, kernel=kern, C=C, m_dev=m_dev)
        e = rel(C[:2427], A[:2427].float() @ B.float().t())
        bad = e > 6e-3 or not bool((C[2427:] == -7.0).all())
        ok &= not bad
        print(f"{name} device rows: rel {e:.2e}, untouched tail {bool((C[2427:] == -7.0).all())}{'  <-- FAIL' if bad else ''}", flush=True)
        # race screen: many tiles per workgroup, repeated launches bit-identical
        A3, B3 = rand((26432, 768), 1.0, 35), rand((3072 if kern == hip.GEMM_ASM4 else 2304, 768), 0.05, 36)
        first = gemm(A3, B3, kernel=kern)
        e = rel(first, A3.float() @ B3.float().t())
        same = all(torch.equal(gemm(A3, B3, kernel=kern), first) for _ in range(30))
        bad = e > 6e-3 or not same
        ok &= not bad
        print(f"{name} race screen: rel {e:.2e}, 30 repeats identical {same}{'  <-- FAIL' if bad else ''}", flush=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def timing(T):
    import statistics
    shapes = [("qkv", T, 2304, 768, hip.EPI_BIAS), ("attn_out", T, 768, 768, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT),
              ("ffn_up", T, 3072, 768, hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT | hip.EPI_AUX_GRAD),
              ("ffn_down", T, 768, 3072, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT),
              ("dgrad_gelu", T, 3072, 768, hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD), ("dgrad_resid", T, 768, 3072, hip.EPI_RESID),
              ("dgrad_qkv", T, 768, 2304, hip.EPI_RESID), ("plain_ffn_up", T, 3072, 768, 0), ("plain_ffn_down", T, 768, 3072, 0),
              ("plain_qkv", T, 2304, 768, 0), ("plain_768", T, 768, 768, 0), ("8192^3", 8192, 8192, 8192, 0)]
    for name, M, N, K, fl in shapes:
        A, B = rand((M, K), 1.0, 1), rand((N, K), 0.05, 2)
        bias = torch.randn(N, device="cuda")
        side = rand((M, N), 1.0, 3)
        aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(bias=bias if fl & hip.EPI_BIAS else None, resid=side if fl & hip.EPI_RESID else None,
                  aux=(side if fl & hip.EPI_GELU_BWD else aux) if fl & (hip.EPI_GELU_BWD | hip.EPI_SAVE_PREACT) else None,
                  drop_p=0.1 if fl & hip.EPI_DROPOUT else 0.0, seed=5)
        arms = {"w4_256": lambda: gemm(A, B, flags=fl, kernel=hip.GEMM_WAVE4, C=C, **kw),
                "a4_256": lambda: gemm(A, B, flags=fl, kernel=hip.GEMM_ASM4, C=C, **kw)}
        if N % 192 == 0 and fl in (0, hip.EPI_BIAS, hip.EPI_RESID, hip.EPI_BIAS | hip.EPI_RESID, hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT):
            arms["w4_192"] = lambda: gemm(A, B, flags=fl, kernel=hip.GEMM_WAVE4_192, C=C, **kw)
            arms["a4_192"] = lambda: gemm(A, B, flags=fl, kernel=hip.GEMM_ASM4_192, C=C, **kw)
        arms["auto"] = lambda: gemm(A, B, flags=fl, kernel=hip.GEMM_AUTO, C=C, **kw)
        if fl == 0:
            arms["vendor"] = lambda: torch.matmul(A, B.t(), out=C)
        res = {k: [] for k in arms}
        for k, f in arms.items():
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        for rnd in range(7):
            for k, f in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        fl_g = 2.0 * M * N * K
        print(f"{name:14s} {M}x{N}x{K} flags {fl:3d}: " + "  ".join(
            f"{k} {statistics.median(v):7.1f} us ({fl_g / statistics.median(v) / 1e6:5.0f} TF/s, min {min(v):.1f})" for k, v in res.items()),
            flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or ["check", "time"]
    hip.lib()
    ok = True
    if "check" in args:
        ok = check()
    if "time" in args and ok:
        T = next((int(a) for a in args if a.isdigit()), 26432)
        timing(T)
    sys.exit(0 if ok else 1)
