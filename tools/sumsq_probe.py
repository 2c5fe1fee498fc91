"""The gradient-norm pass alone (sum of squares of 243 M fp32 = 974 MB, fixed summation order): this build against builds
with other grid caps (-DSTONK_SUMSQ_BLOCKS=...), interleaved.  LIBS=name=path,... python tools/sumsq_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stonkgs_amd import _hip as hip  # noqa: E402

n = 243430144
g = torch.randn(n, device="cuda") * 1e-3
libs = [("built", hip.lib())]
for item in filter(None, os.environ.get("LIBS", "").split(",")):
    name, _, path = item.partition("=")
    h = C.CDLL(os.path.join(ROOT, path))
    h.stonk_sumsq_f32.argtypes = hip._SIGNATURES["stonk_sumsq_f32"]
    h.stonk_sumsq_f32.restype = C.c_int
    h.stonk_sumsq_workspace_floats.restype = C.c_int64
    libs.append((name, h))
ws = {nm: torch.zeros(int(h.stonk_sumsq_workspace_floats()), device="cuda") for nm, h in libs}
out = {nm: torch.zeros(1, device="cuda") for nm, _ in libs}
st = hip.stream_ptr()
ref = float((g.double() ** 2).sum())
for nm, h in libs:
    assert h.stonk_sumsq_f32(g.data_ptr(), n, out[nm].data_ptr(), ws[nm].data_ptr(), ws[nm].numel(), st) == 0
torch.cuda.synchronize()
reps = 15
ev = {nm: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for nm, _ in libs}
for r in range(reps):
    for nm, h in libs:
        ev[nm][r][0].record()
        h.stonk_sumsq_f32(g.data_ptr(), n, out[nm].data_ptr(), ws[nm].data_ptr(), ws[nm].numel(), st)
        ev[nm][r][1].record()
torch.cuda.synchronize()
for nm, _ in libs:
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[nm])
    print(f"{nm:8s} {t[len(t) // 2]:7.1f} us (min {t[0]:.1f})  {n * 4 / t[len(t) // 2] / 1e6:.2f} TB/s   rel err of the first sum {abs(float(out[nm]) / (reps + 1) - ref) / ref:.1e}", flush=True)
