#!/usr/bin/env python
"""Benchmark of the STonKGs pre-training hot path on MI355X (BASELINE.json metric: text-triple pairs/sec, whole node).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step of STonKGsForPreTraining on one synthetic batch per GPU (BASELINE config 2:
12L/768h, V=28 996, K=175 094, per-GPU batch 64, seq 256 text + 256 entity, dropout 0.1 live incl. the frozen
backbone, bf16 MFMA compute / fp32 master weights): frozen backbone forward, KG gather, encoder forward, heads,
3 x cross-entropy, full backward, gradient all-reduce (N > 1), global-norm clip, AdamW, weight refresh.
Inputs are resident in HBM before the timed region. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# gemm_tn_w4_kernel, average of its 37 launches per step: 337.7 MB fetched (FETCH_SIZE doubled, the gfx950 correction of
# MI355X_MICROARCH.md) + 61.1 MB of float atomics written (profiles/r01_final_pmc_traffic.csv)
TN_W4_TRAFFIC_BYTES = 398.8e6
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
# algorithmic GFLOP per text-triple pair, 12L/768/S=512, label-sparse decoders
GFLOP_PER_PAIR_STEP = 373.4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config 2: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may actually use (affinity mask and cgroup quota, not the machine's core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference's HuggingFace path, pinned by tests/test_oracle_golden.py) timed
    on this host's cores: full training steps at the SAME model shape on a bounded sample (B = 2)."""
    from oracle import stonkgs_oracle as orc
    from stonkgs_amd.data import synthetic_batch

    ocfg = orc.OracleConfig(vocab_size=cfg.vocab_size, kg_vocab_size=cfg.kg_vocab_size, hidden_size=cfg.hidden_size,
                            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                            intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings)
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads, building fp32 weights")
    sd = orc.init_state_dict(ocfg, seed=0, bf16_exact=False)
    table = torch.randn(ocfg.kg_vocab_size + 3, ocfg.hidden_size) * 0.3
    B = 2
    batch = synthetic_batch(B, ocfg.vocab_size, ocfg.kg_vocab_size, ocfg.max_position_embeddings, seed=4321)
    state = orc.AdamState()
    tw = time.time()
    orc.train_step(sd, ocfg, table, batch, state)  # warm-up (allocations, oneDNN primitive caches)
    log(f"cpu_baseline: warm-up step {time.time() - tw:.1f} s")
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 < seconds_budget and n < 8):
        orc.train_step(sd, ocfg, table, batch, state)
        n += 1
        log(f"cpu_baseline: step {n} at {time.time() - t0:.1f} s")
    dt = time.time() - t0
    return {"value": B * n / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full fp32 training steps of the CPU oracle at the same model shape, batch {B} "
                      f"(seq 512, V={ocfg.vocab_size}, K={ocfg.kg_vocab_size}), torch {torch.__version__}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist

    # one rank per GPU over RCCL; STONK_DIST_BACKEND=gloo lets several ranks share one card (a rehearsal of the N > 1
    # control flow on a one-GPU box - never a measurement)
    backend = os.environ.get("STONK_DIST_BACKEND", "nccl")
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend)
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    log(f"rank {rank}/{world}: building model")
    cfg = STonKGsConfig()  # 12L / 768h / 12 heads / 3072 / 512 positions / V 28996 / K 175094, dropout 0.1
    model = STonKGsForPreTraining(cfg, seed=0)  # same seed on every rank: replicas start identical (as DDP broadcasts)
    trainer = Trainer(model, TrainingArguments(per_device_train_batch_size=args.batch, max_steps=200, learning_rate=1e-4))
    dev = model.device
    batches = [{k: v.to(dev) for k, v in synthetic_batch(args.batch, cfg.vocab_size, cfg.kg_vocab_size,
                                                         cfg.max_position_embeddings, seed=1234 + rank * 100 + i).items()}
               for i in range(4)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("model + batches ready; warm-up")
    loss = None
    for i in range(args.warmup):
        loss = trainer.training_step(model, batches[i % len(batches)])
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss {float(loss):.4f}")
    barrier()
    log("timed region")
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.training_step(model, batches[i % len(batches)])
    barrier()
    dt = time.perf_counter() - t0
    model.engine.check_errors()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)
    log(f"timed: {dt / args.steps * 1e3:.2f} ms/step, loss {final_loss:.4f}")

    roofline = None
    gemm_all = None
    if not args.no_roofline:
        # dominant kernel of the step = gemm_tn_w4_kernel (weight + bias gradients of the FFN and fused-QKV linears and the
        # entity decoder's weight gradient, 37 launches/step; profiles/ has the rocprofv3 kernel-trace of the same command): per-launch HIP events on the
        # launch stream over two extra steps; achieved = algorithmic FLOPs (2 * M' * N' * tokens per launch) / summed
        # durations
        # (every rank runs the two steps - they contain the gradient all-reduce - and times its own launches; rank 0 reports)
        from stonkgs_amd.engine import GemmTimer

        model.engine.gemm_timer = GemmTimer()
        for i in range(2):
            trainer.training_step(model, batches[i % len(batches)])
        torch.cuda.synchronize()
        s = model.engine.gemm_timer.summarize("tn_w4")
        a = model.engine.gemm_timer.summarize(None)
        model.engine.gemm_timer = None
        ach = s["flops"] / s["seconds"] / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_tn_w4_kernel (bf16 MFMA 32x32x16, 256x256 tiles over 64-token steps, four "
                                               "waves, transposed LDS reads, split-K fp32 atomics)",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": TN_W4_TRAFFIC_BYTES,
                    "traffic_note": "bytes per launch at the L2's memory side = 2 x FETCH_SIZE + WRITE_SIZE (KiB, separate "
                                    "rocprofv3 --pmc passes over this command, profiles/r01_final_pmc_traffic.csv) against "
                                    "274 MB of operands + output per launch (36 encoder launches of 243 MB, the entity decoder's 1.39 GB); a recorded constant - counters "
                                    "cannot be read from inside the bench",
                    "launches_per_step": s["launches"] // 2,
                    "avg_launch_us": round(s["seconds"] / s["launches"] * 1e6, 1),
                    "avg_launch_gflop": round(s["flops"] / s["launches"] / 1e9, 2)}
        gemm_all = {"kernels": "gemm_tn_w4_kernel + gemm_tn_kernel + gemm_nt_kernel + gemm256_kernel", "launches_per_step": a["launches"] // 2,
                    "achieved_tflops": round(a["flops"] / a["seconds"] / 1e12, 1),
                    "frac": round(a["flops"] / a["seconds"] / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "ms_per_step": round(a["seconds"] / 2 * 1e3, 2)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / dt
        out = {"metric": "text-triple pairs/sec (whole node), seq_len=512 hidden=768, 1/2/4/8 MI355X", "value": round(value, 2),
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "STonKGs-150k pretraining step (12L/768h, V=28996, K=175094), per-GPU batch "
                                      f"{args.batch}, seq 256 text + 256 entity, dropout 0.1, AdamW lr 1e-4",
                          "global_batch": args.batch * world, "seq_len": 512, "parallelism": f"dp{world}"},
               "final_loss": round(final_loss, 4),
               "step_mfma_frac": round(value * GFLOP_PER_PAIR_STEP / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
               "roofline": roofline, "all_gemm": gemm_all}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
