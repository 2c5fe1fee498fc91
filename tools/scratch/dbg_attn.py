import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from stonkgs_amd import _hip as hip
import test_attention_gpu as T
hip.lib()
B, S, NH = 2, 256, 2
qkv, dout, mask = T._inputs(B, S, NH, 21, True)
for p in (0.0, 0.1):
    o1, l1 = T._run_fwd(hip, qkv, mask, B, S, NH, p, 5)
    o2, l2 = T._run_fwd(hip, qkv, mask, B, S, NH, p, 5)
    torch.cuda.synchronize()
    d = (o1 != o2)
    print("p", p, "nan", torch.isnan(o1).sum().item(), "diff", d.sum().item(), "of", d.numel())
    if d.any():
        idx = d.nonzero()
        print(idx[:20].tolist(), idx[-5:].tolist())
        rows = idx[:, 0].unique()
        print("rows", rows[:40].tolist(), len(rows))
        cols = idx[:, 1].unique()
        print("cols", cols[:70].tolist(), len(cols))
        print((o1.float() - o2.float()).abs().max().item())
    print("lse diff", (l1 != l2).sum().item())
