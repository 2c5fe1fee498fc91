"""Schedule sweep of the written-out K loop (gemm_a4.hip): `build` (CPU, here) writes one include file and one library per
schedule variant under build/sweep/; `run` (GPU box) loads each library and times the same launches interleaved.
    python tools/a4_sweep.py build && gpurun -- python tools/a4_sweep.py run
"""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
SW = os.path.join(ROOT, "ab_ref", "sweep")   # (build/ does not travel to the GPU box; ab_ref/ does and is git-ignored)

VARIANTS = {
    "v0_base": {},
    "v9_dma6": {8: dict(b1_gap=20, dma_step=6, b2_gap=30), 6: dict(b1_gap=18, dma_step=4, b2_gap=30)},
    "a3_base_linsrc": {"defs": ["-DSTONK_A4_LINEAR_SRC"]},
    "a4_dma6_linsrc": {8: dict(b1_gap=20, dma_step=6, b2_gap=30), 6: dict(b1_gap=18, dma_step=4, b2_gap=30), "defs": ["-DSTONK_A4_LINEAR_SRC"]},
    "a0_nodma": {8: dict(ablate=("dma",)), 6: dict(ablate=("dma",))},
    "a2_mfma_only": {8: dict(ablate=("dma", "reads")), 6: dict(ablate=("dma", "reads"))},
}


def build():
    import gen_gemm_a4 as gen
    os.makedirs(SW, exist_ok=True)
    csrc = os.path.join(ROOT, "stonkgs_amd", "csrc")
    others = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "gemm_a4.o"]
    procs = []
    for name, ov in VARIANTS.items():
        inc = os.path.join(SW, name + ".inc")
        try:
            gen.generate(inc, ov)
        except (AssertionError, IndexError) as e:
            print(name, "schedule does not fit:", e)
            continue
        obj = os.path.join(SW, name + ".o")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
               "-Wno-unused-result", "-Wno-inline-asm", "-ffp-contract=fast", f'-DSTONK_A4_LOOP_INC="{inc}"'] + ov.get("defs", []) + ["-c",
               os.path.join(csrc, "gemm_a4.hip"), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
        if len(procs) % 4 == 0:
            for _, _, p in procs[-4:]:
                p.wait()
    for name, obj, p in procs:
        assert p.wait() == 0, name
        so = os.path.join(SW, f"libstonk_{name}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
        print("built", os.path.relpath(so, ROOT))


def run():
    import torch
    T = 26432
    shapes = [("ffn_up", T, 3072, 768, 7), ("qkv", T, 2304, 768, 7), ("ffn_down", T, 768, 3072, 8), ("768", T, 768, 768, 8),
              ("8192^3", 8192, 8192, 8192, 7)]
    libs = {}
    for name in VARIANTS:
        so = os.path.join(SW, f"libstonk_{name}.so")
        if os.path.exists(so):
            lib = C.CDLL(so)
            lib.stonk_gemm_nt_bf16.restype = C.c_int
            libs[name] = lib
    vp, i64, i32, f32, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint32
    st = torch.cuda.current_stream().cuda_stream
    for sname, M, N, K, kern in shapes:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = None

        def call(lib):
            rc = lib.stonk_gemm_nt_bf16(vp(A.data_ptr()), i64(K), vp(B.data_ptr()), i64(K), vp(Cm.data_ptr()), i64(N), i32(M),
                                        i32(N), i32(K), i32(0), vp(0), vp(0), i64(0), vp(0), i64(0), f32(1.0), i32(1), vp(0), vp(0),
                                        f32(0.0), u32(0), i32(kern), vp(st))
            assert rc == 0, rc
        res = {k: [] for k in libs}
        for k, lib in libs.items():
            for _ in range(3):
                call(lib)
            torch.cuda.synchronize()
            if ref is None:
                ref = Cm.clone()
            elif not k.startswith("a"):   # (the ablated variants compute garbage)
                assert torch.equal(ref, Cm), (k, sname, "differs from the first variant")
        for rnd in range(7):
            for k, lib in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call(lib)
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"{sname} {M}x{N}x{K} kernel {kern}: " + "  ".join(f"{k} {statistics.median(v):.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
