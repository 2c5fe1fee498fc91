"""stonk_label_compact alone on 16 384 labels (15 % labelled): result against torch, us per launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip
n, half, S = 16384, 256, 512
g = torch.Generator(device="cuda").manual_seed(0)
labels = torch.full((n,), -100, device="cuda", dtype=torch.long)
m = torch.rand(n, device="cuda", generator=g) < 0.15
labels[m] = torch.randint(0, 1000, (int(m.sum()),), device="cuda", generator=g)
rows = torch.zeros(n, device="cuda", dtype=torch.int32); tg = torch.zeros(n, device="cuda", dtype=torch.int32); cnt = torch.zeros(1, device="cuda", dtype=torch.int32)
def run(): hip.call("stonk_label_compact", hip.ptr(labels), n, half, S, half, hip.ptr(rows), hip.ptr(tg), hip.ptr(cnt), 0, hip.stream_ptr())
run(); torch.cuda.synchronize()
idx = torch.nonzero(m).flatten()
exp_rows = (idx // half) * S + half + idx % half
assert int(cnt) == idx.numel() and torch.equal(rows[:int(cnt)].long(), exp_rows) and torch.equal(tg[:int(cnt)].long(), labels[m])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"label_compact: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch, count {int(cnt)}")
