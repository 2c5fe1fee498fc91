"""N data-parallel ranks of the training step, checked against each other and against one process:
  * backend "nccl" (= RCCL, default when there are at least as many GPUs as ranks): one rank per GPU, the production path;
  * backend "gloo" (STONK_DIST_BACKEND=gloo): all ranks on GPU 0 - a rehearsal of the multi-GPU step logic on a one-GPU box
    (bucketed all-reduce issued under the weight-gradient stream, optimizer waiting for it, the dynamically scheduled
    dgrad kernel while communication holds CUs).
Launch: tests/test_dist_gpu.py starts the ranks itself (stonkgs_amd/launch.py: RANK / WORLD_SIZE / MASTER_* in the environment,
one session per rank, logs in files); `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29533 tools/dp_check.py` works as well.
A rank that stalls says where: every phase is announced on stderr, rendezvous and collectives carry a 120-s limit (an
error, not a silent wait), and after STONK_DP_CHECK_DUMP_AFTER seconds (default 200) every thread's Python stack is written
to stderr - the launcher keeps the ranks' output in files, so it survives the kill that follows.
Checks: (A) the synchronised GRADIENT buffer - every bucket after `finish()`, before any optimizer touches it - equals the
sum of the ranks' single-process gradients to fp32 round-off (1e-5 of the largest gradient: a bucket boundary off by a few
thousand elements leaves them at half their value); (B) after two training steps both ranks hold bitwise-identical
parameters, and they match a single-process run that accumulates the two ranks' batches (DDP semantics: mean over ranks
of per-rank mean losses) within what Adam makes of fp32 round-off. STONK_DP_SHARD=1 runs the same with the optimizer
sharded (reduce-scatter into the owned piece of every bucket, sharded AdamW, all-gather of the parameters)."""
import faulthandler
import os
import sys
import time
from datetime import timedelta

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd.config import STonKGsConfig  # noqa: E402
from stonkgs_amd.data import synthetic_batch  # noqa: E402
from stonkgs_amd.stonkgs_model import STonKGsForPreTraining  # noqa: E402
from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments  # noqa: E402


def build(seed=0):
    cfg = STonKGsConfig(vocab_size=2048, kg_vocab_size=640, num_hidden_layers=2, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(5)
    table = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    return cfg, STonKGsForPreTraining(cfg, kg_embeddings=table, seed=seed)


T0 = time.time()


def say(msg):
    print(f"[dp_check rank {os.environ.get('RANK', '?')} +{time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    faulthandler.dump_traceback_later(float(os.environ.get("STONK_DP_CHECK_DUMP_AFTER", "200")), exit=False)
    backend = os.environ.get("STONK_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    limit = timedelta(seconds=120)
    say(f"rendezvous ({backend}, {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')})")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit)
    else:
        torch.cuda.set_device(0)
        dist.init_process_group(backend, timeout=limit)
    rank, world = dist.get_rank(), dist.get_world_size()
    say("process group up; building the model")
    B = 32
    shard = os.environ.get("STONK_DP_SHARD") == "1"
    cfg, model = build()
    tr = Trainer(model, TrainingArguments(learning_rate=1e-3, max_steps=10, per_device_train_batch_size=B, ddp_bucket_mb=8,
                                          shard_optimizer=shard))
    assert tr.world == world and model.engine.comm_overlap == (world > 1) and tr.sync.shard == (shard and world > 1)
    # ---- (A) the gradient buffer after the collectives, against the sum of single-process gradients
    model.train()
    model.forward_backward(synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=90 + rank),
                           on_segment_done=tr.sync.on_segment_done)
    tr.sync.finish()
    torch.cuda.synchronize()
    got = model._store.grad.detach().clone()
    model._store.grad.zero_()
    ref = torch.zeros_like(got)
    if rank == 0:
        _, solo = build()
        solo.train()
        solo.engine.comm_overlap = False
        for r in range(world):
            solo.forward_backward(synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=90 + r))
        solo.engine.join_wgrad()
        torch.cuda.synchronize()
        ref.copy_(solo._store.grad)
        del solo
    dist.broadcast(ref, src=0)
    spans = tr.sync.owned_spans() or [(0, got.numel())]
    gmax = float(ref.abs().max())
    worst = max(float((got[lo:hi] - ref[lo:hi]).abs().max()) for lo, hi in spans)
    say(f"gradient buffer after the collectives: max |dp - sum of single-process gradients| = {worst:.3e} "
        f"(largest gradient {gmax:.3e}; {len(spans)} span(s) checked)")
    grads_ok = worst <= 1e-5 * gmax
    flag = torch.tensor([1.0 if grads_ok else 0.0])
    dist.all_reduce(flag.to(got.device) if backend == "nccl" else flag, op=dist.ReduceOp.MIN)
    assert grads_ok, (worst, gmax)
    # ---- (B) two training steps
    losses = []
    for step in range(2):
        b = synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=100 + 10 * step + rank)
        losses.append(float(tr.training_step(model, b)))
        say(f"step {step} done, loss {losses[-1]:.4f}")
    model.engine.check_errors()
    model.engine.wait_params()   # the optimizer runs on its own stream
    flat = model._store.data.detach().clone()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    gathered = [g.to(flat.device) for g in gathered]
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0 and not same:
        diff = (gathered[0] != gathered[1]).nonzero().flatten()
        print("differing elements:", diff.numel(), "first", diff[:5].tolist(), "last", diff[-5:].tolist(), flush=True)
        st = model._store
        for name, (off, shape, pshape) in st.index.items():
            n = 1
            for d in pshape:
                n *= d
            cnt = int(((diff >= off) & (diff < off + n)).sum())
            if cnt:
                print(f"  {name}: {cnt} of {n} differ, max {float((gathered[0][off:off+n]-gathered[1][off:off+n]).abs().max()):.3e}", flush=True)
        print("buckets:", tr.sync.buckets, flush=True)
    say(f"replicas compared: identical = {same}")
    if rank == 0:
        # single process, the same 2 x 2 batches with gradient accumulation over the "ranks"
        cfg2, ref = build()
        tr2 = Trainer(ref, TrainingArguments(learning_rate=1e-3, max_steps=10, per_device_train_batch_size=B,
                                             gradient_accumulation_steps=world, ddp_force_collectives=False))
        tr2.world, tr2.sync.world, tr2.sync.active, tr2.sync.shard = 1, 1, False, False
        ref.engine.comm_overlap = False
        for step in range(2):
            for r in range(world):
                tr2.training_step(ref, synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=100 + 10 * step + r))
        ref.engine.wait_params()
        d = (ref._store.data - flat).abs().max().item()
        scale = (ref._store.data.abs().max().item())
        print(f"ranks identical: {same}; losses {losses}; max |dp - accumulated| = {d:.3e} (param scale {scale:.2f})", flush=True)
        assert same and d < 2e-3, (same, d)
        print(f"DP{world} OK ({backend}{', sharded optimizer' if shard else ''})", flush=True)
    say("final barrier")
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
