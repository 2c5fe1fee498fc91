"""Timing experiments on the four-wave GEMM (STONK_W4_VAR builds): one subprocess per variant."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stonkgs_amd import _hip as hip
    from bench_kernels import timeit
    hip.lib()
    for M, N, K in [(8192, 8192, 8192), (32768, 3072, 768), (32768, 768, 3072)]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: hip.call("stonk_gemm_nt_bf16", hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K,
                                    hip.EPI_DEBUG_W4, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0.0, 0, hip.stream_ptr()))
        print(f"var {os.environ.get('STONK_W4_VAR', '0')}: {M}x{N}x{K} {t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF/s", flush=True)
else:
    for v in sys.argv[1:] or ["0", "1", "2", "3", "4", "8", "12", "13"]:
        env = dict(os.environ, STONK_W4_VAR=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
