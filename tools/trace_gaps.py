"""Where the GPU is idle inside a training step: from a rocprofv3 --kernel-trace CSV, the union of all kernels' [start, end)
intervals over the last steps (an AdamW launch ends a step), the gaps between them by size and by the kernel that
follows, and per-stream busy time.  python tools/trace_gaps.py kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
ends = [e for s, e, n, q in ev if "adamw_kernel" in n]
if len(ends) < 3:
    sys.exit("fewer than three steps in the trace")
t0, t1 = ends[-3], ends[-1]          # the last two whole steps
steps = 2
win = [x for x in ev if x[0] >= t0 and x[1] <= t1]
busy = 0
cur_s, cur_e = None, None
gaps = []
for s, e, n, q in win:
    if cur_e is None:
        cur_s, cur_e = s, e
        gaps.append((s - t0, n))
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"span {span / steps / 1e6:.3f} ms per step, some kernel running {busy / steps / 1e6:.3f} ms, idle {(span - busy) / steps / 1e6:.3f} ms "
      f"({len(win) / steps:.0f} kernels per step)")
hist = collections.Counter()
tot = collections.Counter()
for g, n in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist[b] += 1
    tot[b] += g
for b in ("<2us", "<5us", "<10us", "<50us", ">=50us"):
    print(f"  gaps {b:7s}: {hist[b] / steps:7.1f} per step, {tot[b] / steps / 1e3:8.1f} us per step")
byk = collections.Counter()
for g, n in gaps:
    byk[n[:70]] += g
print("idle time by the kernel that follows the gap (us per step):")
for n, g in byk.most_common(15):
    print(f"  {g / steps / 1e3:8.1f}  {n}")
perq = collections.Counter()
for s, e, n, q in win:
    perq[q] += e - s
print("kernel time per queue / stream (ms per step):", {q: round(v / steps / 1e6, 3) for q, v in perq.items()})
# the framework's own small launches (fills, copies): which, how large, on which queue, after which kernel
print("ATen / runtime launches in one step (queue, us, grid, previous kernel on that queue):")
last_on_q = {}
step_lo = ends[-2]
for s, e, n, q in win:
    if s >= step_lo and ("at::native" in n or "rocclr" in n):
        r = next(r for r in rows if int(r["Start_Timestamp"]) == s and r["Kernel_Name"] == n)
        print(f"  q{q} {(e - s) / 1e3:7.1f} us grid {r.get('Grid_Size', '?'):>10s}  {n[:60]:60s} after {last_on_q.get(q, '-')[:50]}")
    last_on_q[q] = n
