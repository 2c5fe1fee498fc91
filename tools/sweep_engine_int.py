"""Interleaved sweep of an integer Engine attribute inside one process (e.g. the CU share of the side-stream weight
gradients): python tools/sweep_engine_int.py tn_cus 128 160 192 224"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402
from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments  # noqa: E402

attr, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
cfg = STonKGsConfig()
model = STonKGsForPreTraining(cfg, seed=0)
tr = Trainer(model, TrainingArguments(per_device_train_batch_size=64, max_steps=10000))
dev = model.device
batches = [{k: v.to(dev) for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1234 + i).items()}
           for i in range(4)]
for i in range(5):
    tr.training_step(model, batches[i % 4])
torch.cuda.synchronize()
res = {v: [] for v in values}
for rnd in range(5):
    for v in values:
        setattr(model.engine, attr, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            tr.training_step(model, batches[i % 4], next_inputs=batches[(i + 1) % 4])
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in values:
    r = sorted(res[v])
    print(f"{attr}={v}: median {r[len(r) // 2]:.2f} ms  min {r[0]:.2f}  max {r[-1]:.2f}", flush=True)
