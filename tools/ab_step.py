"""A/B of engine switches inside ONE process (interleaved rounds; run-to-run noise on the boxes is +-4 %)."""
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
dev = model.device
batches = [{k: v.to(dev) for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1234 + i).items()}
           for i in range(4)]
attr = sys.argv[1] if len(sys.argv) > 1 else "overlap_wgrad"
for fixed in sys.argv[2:]:   # further arguments pin switches for the whole run: engine.NAME=0/1 or args.NAME=0/1
    where, _, rest = fixed.partition(".")
    name, _, value = rest.partition("=")
    pinned = bool(int(value)) if value in ("0", "1") else int(value)   # (0 / 1: a switch; anything else: an integer attribute)
    setattr(model.engine if where == "engine" else tr.args, name, pinned)
    print(f"pinned {where}.{name} = {pinned}", flush=True)
for i in range(5):
    tr.training_step(model, batches[i % 4])
torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(6):
    for val in (True, False):
        if attr.startswith("env:"):   # an environment switch read by the process under test: True = variable set
            name, _, value = attr[4:].partition("=")   # env:NAME or env:NAME=VALUE
            if val:
                os.environ[name] = value or "1"
            else:
                os.environ.pop(name, None)
        elif attr.startswith("args:"):   # a TrainingArguments switch
            setattr(tr.args, attr[5:], val)
        elif attr.startswith("int:"):   # int:NAME=VALUE - an integer engine attribute: True = VALUE, False = what it was
            name, _, value = attr[4:].partition("=")
            if not hasattr(model, "_ab_orig"):
                model._ab_orig = getattr(model.engine, name)
            setattr(model.engine, name, int(value) if val else model._ab_orig)
        elif attr.startswith("kernel:"):   # kernel:SITE=K[,SITE=K]: pin NT kernels (Engine.kernel_for) against the library's choice
            model.engine.kernel_for = ({k: int(v) for k, v in (kv.split("=") for kv in attr[7:].split(","))} if val else {})
        else:
            setattr(model.engine, attr, val)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            tr.training_step(model, batches[i % 4], next_inputs=batches[(i + 1) % 4])
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 10 * 1e3)
for val in (True, False):
    r = sorted(res[val])
    print(f"{attr}={val}: median {r[len(r)//2]:.2f} ms  min {r[0]:.2f}  max {r[-1]:.2f}", flush=True)
