"""Why does the four-wave NT kernel win alone and not in the step? Forward-only encode (frozen backbone + encoder, no
backward, no second stream) with QKV / bias+GELU FFN-up on the eight-wave or the four-wave kernel, interleaved; then the
same with per-kernel HIP events around the QKV launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402

cfg = STonKGsConfig()
model = STonKGsForPreTraining(cfg, seed=0)
model.eval()
dev = model.device
b = {k: v.to(dev) for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1).items()}
args = (b["input_ids"], b["attention_mask"], b["token_type_ids"])
for _ in range(3):
    model.encode(*args)
torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(6):
    for val in (True, False):
        model.engine.kernel_for = {"qkv": hip.GEMM_WAVE4, "ffn_up": hip.GEMM_WAVE4} if val else {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.encode(*args)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 10 * 1e3)
for val in (True, False):
    r = sorted(res[val])
    print(f"forward-only encode, four-wave QKV/FFN-up={val}: median {r[len(r)//2]:.3f} ms  min {r[0]:.3f}  max {r[-1]:.3f}", flush=True)
