"""Forward-only encoder (embedding extraction) at 12L / 768h, batch 64: every position a row
against the packed layout of `encode(..., pooled_only=True)`; ms per batch and sequences per second."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402

cfg = STonKGsConfig(kg_vocab_size=4096)
model = STonKGsForPreTraining(cfg, seed=0)
model.eval()
B = 64
batches = [{k: v.cuda() for k, v in synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=70 + i).items()}
           for i in range(4)]
for pooled_only in (False, True, False, True):
    for i in range(3):
        model.encode(batches[i]["input_ids"], batches[i]["attention_mask"], batches[i]["token_type_ids"], pooled_only=pooled_only)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for i in range(n):
        b = batches[i % 4]
        model.encode(b["input_ids"], b["attention_mask"], b["token_type_ids"], pooled_only=pooled_only)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"encode pooled_only={pooled_only}: {dt * 1e3:.2f} ms per batch of {B}, {B / dt:.0f} sequences/s", flush=True)
