"""Per-step GPU time of a run in the bench's form (four batches cycling, each step told the next batch): an event after
every training step, the differences between consecutive events. Shows how many steps the step time takes to settle -
bench.py times steps W .. W + K - 1.  python tools/step_series.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402
from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = STonKGsConfig()
model = STonKGsForPreTraining(cfg, seed=0)
tr = Trainer(model, TrainingArguments(per_device_train_batch_size=64, max_steps=200, learning_rate=1e-4))
dev = model.device
batches = [{k: v.to(dev) for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1234 + i).items()}
           for i in range(4)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    tr.training_step(model, batches[i % 4], next_inputs=batches[(i + 1) % 4])
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
    if i == 4:   # (as bench.py: a barrier between warm-up and the timed region)
        torch.cuda.synchronize()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print(f"steps {i:2d}-{min(n, i + 10) - 1:2d}: GPU " + " ".join(f"{x:6.2f}" for x in ms[i:i + 10]) + "   host " + " ".join(f"{x:5.1f}" for x in host[i:i + 10]), flush=True)
print(f"mean of steps 5-24: {sum(ms[5:25]) / 20:.2f} ms; of steps 8-47: {sum(ms[8:48]) / 40:.2f} ms")
