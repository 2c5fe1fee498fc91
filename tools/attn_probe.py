"""The three attention kernels ALONE in the training step's form - packed rows (B 64 sequences of 256 entity rows + a text
of 32..256 rows, 12 heads, dropout 0.1, every row's mask word set) - this build against another build of the library
(ab_ref/libstonk_hip.so, tools/build_ref_lib.sh), launches interleaved; and the two builds' results against each other
(with dropout off, where both must agree to rounding; with dropout on when SAME_MASK=1, i.e. the generator is unchanged).
  python tools/attn_probe.py > gpurun_out/attn_probe.log
LIBS=name=path,... adds further builds to the timing (the ablation variants -DSTONK_ATTN_ABLATE_LOADS / _BARRIER)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stonkgs_amd import _hip as hip  # noqa: E402

B, S, NH = 64, 512, 12
H = NH * 64
DELTA, DQ, DKV = 1, 2, 4


def load_ref(path=None):
    path = path or os.path.join(ROOT, "ab_ref", "libstonk_hip.so")
    if not os.path.exists(path):
        return None
    h = C.CDLL(path)
    for name in ("stonk_attention_fwd", "stonk_attention_bwd", "stonk_attention_bwd_phases"):
        fn = getattr(h, name)
        fn.argtypes = hip._SIGNATURES[name]
        fn.restype = C.c_int
    return h


def main():
    new, ref = hip.lib(), load_ref()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    lens = 256 + torch.randint(32, 257, (B,), generator=torch.Generator().manual_seed(5))
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = int(cu[-1])
    cu_d = cu.to(dev)
    qkv = torch.randn(T, 3 * H, device=dev, generator=g).to(torch.bfloat16)
    dout = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    mask = torch.ones(T, dtype=torch.long, device=dev)
    st = hip.stream_ptr()

    class Bufs:
        def __init__(self):
            self.out = torch.zeros(T, H, device=dev, dtype=torch.bfloat16)
            self.lse = torch.zeros(B, NH, S, device=dev)
            self.delta = torch.zeros(B, NH, S, device=dev)
            self.dqkv = torch.zeros(T, 3 * H, device=dev, dtype=torch.bfloat16)

    def fwd(lib, bf, p):
        rc = lib.stonk_attention_fwd(hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask), hip.ptr(cu_d), 0,
                                     hip.ptr(bf.out), H, hip.ptr(bf.lse), B, NH, S, 64, 0.125, p, 3, st)
        assert rc == 0, rc

    def bwd(lib, bf, p, phases):
        rc = lib.stonk_attention_bwd_phases(phases, hip.ptr(qkv), hip.ptr(qkv) + 2 * H, hip.ptr(qkv) + 4 * H, 3 * H, hip.ptr(mask),
                                            hip.ptr(cu_d), 0, hip.ptr(bf.out), H, hip.ptr(dout), H, hip.ptr(bf.lse), hip.ptr(bf.delta),
                                            hip.ptr(bf.dqkv), hip.ptr(bf.dqkv) + 2 * H, 3 * H, hip.ptr(bf.dqkv) + 4 * H, B, NH, S, 64,
                                            0.125, p, 3, st)
        assert rc == 0, rc

    libs = [("new", new)] + ([("ref", ref)] if ref is not None else [])
    # further builds, timing only (ablation variants: LIBS=name=path,name=path)
    for item in filter(None, os.environ.get("LIBS", "").split(",")):
        name, _, path = item.partition("=")
        libs.append((name, load_ref(os.path.join(ROOT, path))))
    bufs = {n: Bufs() for n, _ in libs}
    # agreement
    for p in (0.0, 0.1):
        for n, lib in libs:
            fwd(lib, bufs[n], p)
            bwd(lib, bufs[n], p, DELTA)
            bwd(lib, bufs[n], p, DQ)
            bwd(lib, bufs[n], p, DKV)
        torch.cuda.synchronize()
        if ref is not None and (p == 0.0 or os.environ.get("SAME_MASK") == "1"):
            a, b = bufs["new"], bufs["ref"]
            print(f"p = {p}: max |new - ref|  out {float((a.out.float() - b.out.float()).abs().max()):.3e}  lse "
                  f"{float((a.lse - b.lse).abs().max()):.3e}  dqkv {float((a.dqkv.float() - b.dqkv.float()).abs().max()):.3e} "
                  f"(max |dqkv| {float(b.dqkv.float().abs().max()):.3e})  bitwise out {torch.equal(a.out, b.out)} dqkv {torch.equal(a.dqkv, b.dqkv)}",
                  flush=True)
    # timing, interleaved
    reps = 15
    what = [("forward", lambda lib, bf: fwd(lib, bf, 0.1)), ("dq", lambda lib, bf: bwd(lib, bf, 0.1, DQ)),
            ("dkv", lambda lib, bf: bwd(lib, bf, 0.1, DKV)), ("forward_p0", lambda lib, bf: fwd(lib, bf, 0.0))]
    for wname, fn in what:
        ev = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for n, _ in libs}
        for n, lib in libs:
            fn(lib, bufs[n])
        torch.cuda.synchronize()
        for r in range(reps):
            for n, lib in libs:
                ev[n][r][0].record()
                fn(lib, bufs[n])
                ev[n][r][1].record()
        torch.cuda.synchronize()
        line = f"{wname:10s}"
        for n, _ in libs:
            t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[n])
            line += f"  {n} {t[len(t) // 2]:7.1f} us (min {t[0]:.1f})"
        print(line, flush=True)
    print(f"rows {T} of {B * S}", flush=True)


if __name__ == "__main__":
    main()
