"""Where a step's time goes at its boundary (no profiler attached): Engine.marks collects host times and events on the main
stream at four points - the host about to wait for the row plan, the plan known, the first encoder launch, the end of
backward - and this prints, per step, the GPU time from the end of one step's backward to the start of the next step's
encoder forward (the optimizer's sumsq + AdamW + transposes are 2.3 ms of it by construction; anything beyond is a bubble),
how long the host waited for the plan, how late the host was with the first encoder launch (was the optimizer's
"parameters final" event already complete when the host got there?), and the GPU time of forward and backward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402
from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments  # noqa: E402

cfg = STonKGsConfig()
model = STonKGsForPreTraining(cfg, seed=0)
tr = Trainer(model, TrainingArguments(per_device_train_batch_size=64, max_steps=10000))
for kv in sys.argv[1:]:
    where, _, rest = kv.partition(".")
    name, _, value = rest.partition("=")
    setattr(model.engine if where == "engine" else tr.args, name, int(value))
dev = model.device
batches = [{k: v.to(dev) for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1234 + i).items()}
           for i in range(4)]
for i in range(8):
    tr.training_step(model, batches[i % 4], next_inputs=batches[(i + 1) % 4])
torch.cuda.synchronize()
model.engine.marks = []
t0 = time.perf_counter()
n = 12
for i in range(n):
    tr.training_step(model, batches[i % 4], next_inputs=batches[(i + 1) % 4])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
marks, model.engine.marks = model.engine.marks, None
steps = [marks[i:i + 4] for i in range(0, len(marks), 4)]
print(f"{wall:.2f} ms per step; per step: boundary = backward_end(prev) -> encoder_fwd_begin [GPU ms], plan wait [host ms], "
      "host plan_known -> encoder launch [ms], params final before the host launched?, fwd+heads+bwd [GPU ms]")
for k in range(1, len(steps)):
    prev, cur = steps[k - 1], steps[k]
    names = [m[0] for m in cur]
    assert names == ["plan_wait", "plan_known", "encoder_fwd_begin", "backward_end"], names
    boundary = prev[3][2].elapsed_time(cur[2][2])
    work = cur[2][2].elapsed_time(cur[3][2])
    print(f"  step {k}: boundary {boundary:.2f}  plan wait {(cur[1][1] - cur[0][1]) * 1e3:.2f}  host to launch "
          f"{(cur[2][1] - cur[1][1]) * 1e3:.2f}  params final first: {cur[2][3]}  work {work:.2f}")
