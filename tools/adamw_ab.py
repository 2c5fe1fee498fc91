"""A/B of AdamW kernel variants (bytes in flight per thread, non-temporal accesses): `build` here, `run` on the GPU box."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SW = os.path.join(ROOT, "ab_ref", "adamw")
VARIANTS = {"u1": ["-DSTONK_ADAMW_UNROLL=1", "-DSTONK_ADAMW_NT=0"], "u1_nt": ["-DSTONK_ADAMW_UNROLL=1", "-DSTONK_ADAMW_NT=1"],
            "u2": ["-DSTONK_ADAMW_UNROLL=2", "-DSTONK_ADAMW_NT=0"], "u2_nt": ["-DSTONK_ADAMW_UNROLL=2", "-DSTONK_ADAMW_NT=1"],
            "u4": ["-DSTONK_ADAMW_UNROLL=4", "-DSTONK_ADAMW_NT=0"], "u4_nt": ["-DSTONK_ADAMW_UNROLL=4", "-DSTONK_ADAMW_NT=1"]}
if os.environ.get("ADAMW_AB") == "blocks":   # grid caps instead (the gradient-norm pass runs 5.4 TB/s on 256 blocks, 1.7 on 4096)
    VARIANTS = {f"b{b}": [f"-DSTONK_ADAMW_BLOCKS={b}"] for b in (256, 512, 1024, 2048, 4096)}
    VARIANTS.update({f"b{b}_u4": [f"-DSTONK_ADAMW_BLOCKS={b}", "-DSTONK_ADAMW_UNROLL=4"] for b in (256, 512)})


def build():
    os.makedirs(SW, exist_ok=True)
    csrc = os.path.join(ROOT, "stonkgs_amd", "csrc")
    others = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "optim.o"]
    for name, defs in VARIANTS.items():
        obj = os.path.join(SW, name + ".o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + csrc, "-Wno-unused-result", "-ffp-contract=fast"] + defs +
                              ["-c", os.path.join(csrc, "optim.hip"), "-o", obj])
        so = os.path.join(SW, f"libstonk_{name}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
        print("built", os.path.relpath(so, ROOT))


def run():
    import torch
    n = 243_400_000 // 1024 * 1024
    p = torch.randn(n, device="cuda") * 0.02
    g = torch.randn(n, device="cuda") * 1e-3
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    pb = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    gn = torch.ones(1, device="cuda")
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    st = torch.cuda.current_stream().cuda_stream
    libs = {k: C.CDLL(os.path.join(SW, f"libstonk_{k}.so")) for k in VARIANTS if os.path.exists(os.path.join(SW, f"libstonk_{k}.so"))}
    res = {k: [] for k in libs}

    def call(lib):
        rc = lib.stonk_adamw_step(vp(p.data_ptr()), vp(g.data_ptr()), vp(m.data_ptr()), vp(v.data_ptr()), vp(pb.data_ptr()), i64(n),
                                  f32(1e-4), f32(0.9), f32(0.999), f32(1e-8), f32(0.0), f32(0.1), f32(0.001), vp(gn.data_ptr()),
                                  f32(1.0), f32(1.0), vp(0), i32(0), i64(0), vp(st))
        assert rc == 0, rc
    for rnd in range(6):
        for k, lib in libs.items():
            g.normal_(0, 1e-3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(lib)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) * 1e3)
    for k, t in res.items():
        med = statistics.median(t)
        print(f"adamw {k}: {med:.0f} us  {n * 34 / med / 1e6:.2f} TB/s (34 B/param)", flush=True)


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
