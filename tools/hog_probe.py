"""How long does a persistent one-workgroup-per-CU kernel take when some CUs are held by another stream's kernel (as RCCL's
collectives hold them during data-parallel backward)? Builds tools/hog_kernel.hip on the GPU box, then times the
weight-gradient kernel and the persistent NT kernel alone and beside a 500 us, 32-workgroup hog."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from stonkgs_amd import _hip as hip  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libhog.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools/hog_kernel.hip"), "-o", so])
hog = ctypes.CDLL(so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
hip.lib()
T = 32768
dY = torch.randn(T, 3072, device="cuda").to(torch.bfloat16)
X = torch.randn(T, 768, device="cuda").to(torch.bfloat16)
dW = torch.zeros(3072, 768, device="cuda")
W = (torch.randn(3072, 768, device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty(T, 3072, device="cuda", dtype=torch.bfloat16)
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()


def wgrad(sk):
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), 3072, hip.ptr(X), 768, hip.ptr(dW), 768, 0, 3072, 768, T, 1.0, sk, 0,
             hip.stream_ptr())


def nt(dbg):
    hip.call("stonk_gemm_nt_bf16", hip.ptr(X), 768, hip.ptr(W), 768, hip.ptr(C), 3072, T, 3072, 768, 0, 0, 0, 0, 0, 0, 1.0, 1,
             0, 0, 0.0, 0, dbg, hip.stream_ptr())


def timed(fn, with_hog, n_hog=32, usec=500):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        if with_hog:
            with torch.cuda.stream(side):
                hog.hog_launch(n_hog, usec, sink.data_ptr(), side.cuda_stream)
            # give the hog time to become resident
            e = torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, fn in (("weight gradient, four-wave 256x256, all CUs", lambda: wgrad(0)),
                 ("weight gradient, four-wave 256x256, 160 CUs", lambda: wgrad(-160)),
                 ("weight gradient, 128x128 split 12", lambda: wgrad(12)),
                 ("NT 32768x3072x768, persistent 256x256", lambda: nt(hip.GEMM_WAVE8)),
                 ("NT 32768x3072x768, four-wave persistent", lambda: nt(hip.GEMM_WAVE4)),
                 ("NT 32768x3072x768, four-wave DISPATCHED (one item per workgroup)", lambda: nt(hip.GEMM_DISPATCHED)),
                 ("NT 32768x3072x768, four-wave DISPATCHED2 (two items per workgroup)", lambda: nt(hip.GEMM_DISPATCHED2)),
                 ("NT 32768x3072x768, 128x128", lambda: nt(hip.GEMM_TILE128))):
    for _ in range(2):
        fn()
    a = timed(fn, False)
    b = timed(fn, True)
    c = timed(fn, True, n_hog=8)
    print(f"{name}: alone {a:.0f} us, beside a 32-workgroup 500 us hog {b:.0f} us, beside an 8-workgroup hog {c:.0f} us", flush=True)
