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
    "n1_b1_18": {8: dict(b1_gap=18, dma_step=6, b2_gap=30)},
    "n2_b2_40": {8: dict(b1_gap=20, dma_step=6, b2_gap=40), 6: dict(b1_gap=18, dma_step=4, b2_gap=36)},
    "n3_192_dma5": {6: dict(b1_gap=16, dma_step=5, b2_gap=28)},
    "t1_dma5": {"tn": dict(dma_step=5)},
    "t2_dma4": {"tn": dict(dma_step=4)},
    "t3_b2_40": {"tn": dict(b2_gap=40)},
    "t4_b2_20": {"tn": dict(b2_gap=20)},
    "t5_b1_18": {"tn": dict(b1_gap=18)},
}


def build():
    import gen_gemm_a4 as gen
    os.makedirs(SW, exist_ok=True)
    csrc = os.path.join(ROOT, "stonkgs_amd", "csrc")
    others = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f not in ("gemm_a4.o", "gemm_tn_a4.o")]
    procs = []
    for name, ov in VARIANTS.items():
        inc = os.path.join(SW, name + ".inc")
        try:
            gen.generate(inc, ov)
        except (AssertionError, IndexError) as e:
            print(name, "schedule does not fit:", e)
            continue
        objs = []
        for src in ("gemm_a4", "gemm_tn_a4"):
            obj = os.path.join(SW, f"{name}_{src}.o")
            objs.append(obj)
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                   "-Wno-unused-result", "-Wno-inline-asm", "-ffp-contract=fast", f'-DSTONK_A4_LOOP_INC="{inc}"'] + ov.get("defs", []) + ["-c",
                   os.path.join(csrc, src + ".hip"), "-o", obj]
            procs.append((name, subprocess.Popen(cmd)))
            if len(procs) % 6 == 0:
                for _, p in procs[-6:]:
                    p.wait()
        VARIANTS[name]["_objs"] = objs
    for name, p in procs:
        assert p.wait() == 0, name
    for name, ov in VARIANTS.items():
        if "_objs" not in ov:
            continue
        so = os.path.join(SW, f"libstonk_{name}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + ov["_objs"] + others + ["-ldl"])
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
    # the weight gradient (gemm_tn_a4.hip), held to 160 CUs' worth of workgroups as the step launches it
    for sname, Mo, No in (("tn_ffn_up", 3072, 768), ("tn_ffn_down", 768, 3072), ("tn_qkv", 2304, 768)):
        g = torch.Generator(device="cuda").manual_seed(2)
        dY = (torch.randn(T, Mo, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        X = (torch.randn(T, No, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        dW = torch.zeros(Mo, No, device="cuda")
        db = torch.zeros(Mo, device="cuda")

        def call_tn(lib):
            rc = lib.stonk_gemm_tn_bf16(vp(dY.data_ptr()), i64(Mo), vp(X.data_ptr()), i64(No), vp(dW.data_ptr()), i64(No), vp(db.data_ptr()),
                                        i32(Mo), i32(No), i32(T), f32(1.0), i32(-160), vp(0), vp(st))
            assert rc == 0, rc
        res = {k: [] for k in libs}
        for k, lib in libs.items():
            for _ in range(3):
                call_tn(lib)
        torch.cuda.synchronize()
        for rnd in range(7):
            for k, lib in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call_tn(lib)
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"{sname} {Mo}x{No} T{T} @160: " + "  ".join(f"{k} {statistics.median(v):.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
