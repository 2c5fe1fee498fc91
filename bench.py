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

# HBM-side bytes per launch of the dominant kernel, from separate rocprofv3 --pmc passes over this command:
# gemm_tn_a4_kernel (12L/768, average of its 46 launches per step on the unpadded rows): 250.0 MB fetched (FETCH_SIZE
# doubled, the gfx950 correction of MI355X_MICROARCH.md) + 60.0 MB of float atomics written (WRITE_SIZE),
# profiles/r04_pmc_traffic.csv; algorithmic: 193.5 MB.
# (round 3, gemm_tn_w4_kernel: 242.7 + 58.6 = 301.3 MB)
TRAFFIC_BYTES = {("150k", "tn_a4"): 310.0e6}
KERNEL_NOTES = {
    "tn_a4": "gemm_tn_a4_kernel (weight + bias gradients: bf16 MFMA 16x16x32, 256x256 tiles over 64-token steps, four waves, "
             "LDS-DMA operands, transposed LDS reads, a written-out K loop, split-K fp32 buffer atomics)",
    "tn": "gemm_tn_kernel (weight + bias gradients on 128x128 tiles)",
    "nt": "NT forward / dgrad GEMMs (gemm_a4_kernel - written-out four-wave loop -, gemm256_kernel, gemm_nt_kernel)",
}
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
# algorithmic GFLOP per text-triple pair, 12L/768/S=512, label-sparse decoders
GFLOP_PER_PAIR_STEP = 373.4
# BASELINE.json configs[1]/[2] (the headline) and configs[3] (24L/1024h: no reference counterpart - the reference can only
# build BioBERT's 12L/768; frozen backbone and node2vec table are instantiated at width 1024)
MODELS = {
    "150k": dict(batch=64, cfg={}, name="STonKGs-150k pretraining step (12L/768h, V=28996, K=175094)"),
    "24L1024": dict(batch=64, cfg=dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                       intermediate_size=4096),
                    name="STonKGs 24L/1024h synthetic scale-up pretraining step (V=28996, K=175094, frozen backbone and "
                         "table at width 1024)"),
}


def encoder_gflop_per_pair(cfg) -> float:
    """Forward + backward GFLOP per pair of the attention+FFN path of the trainable encoder (the path the 40 %-of-MFMA-peak
    target of BASELINE.json refers to; 12L/768: 289.9)."""
    H, I, L, S = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.max_position_embeddings
    return 3 * L * (2 * S * (4 * H * H + 2 * H * I) + 4 * S * S * H) / 1e9


def gflop_per_pair(cfg) -> float:
    """Algorithmic GFLOP of one training step per text-triple pair (SURVEY section 8d's accounting: 1 MAC = 2 FLOP,
    backward = 2 x forward for trainable blocks, frozen backbone forward only at S/2, label-sparse decoders on
    int(half * 0.15) labelled rows per half). 12L/768: 373.4."""
    H, I, L, S = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.max_position_embeddings
    half = S // 2

    def enc(seq):   # forward FLOPs of one layer on `seq` tokens
        return 2 * seq * (4 * H * H + 2 * H * I) + 4 * seq * seq * H

    lab = int(half * 0.15)
    f = 3 * L * enc(S) + L * enc(half) + 3 * 2 * S * H * H + 3 * 2 * lab * H * (cfg.vocab_size + cfg.kg_vocab_size)
    return f / 1e9


def executed_fraction(rows_executed):
    """(linear, attention, read): share of the padded encoder's row-proportional and rows^2-proportional work that the
    unpadded encoder actually ran, and the share of rows the LAST layer's feed-forward block, the pooler and the head
    transform ran on (Engine.rows_executed)."""
    r = rows_executed
    return (r[0] / r[1], r[3] / r[4], r[5] / r[1], r[6] / r[4]) if r[1] else (1.0, 1.0, 1.0, 1.0)


def gflop_per_pair_executed(cfg, lin: float, att: float, rdf: float, att_last: float = None) -> dict:
    """gflop_per_pair / encoder_gflop_per_pair with the trainable encoder's and the head transform's terms scaled to the
    rows that ran (frozen backbone and label-sparse decoders are unaffected)."""
    H, I, L, S = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.max_position_embeddings
    half = S // 2
    proj, ffn, attn = 2 * S * 4 * H * H, 2 * S * 2 * H * I, 4 * S * S * H
    att_last = att if att_last is None else att_last
    pruned_attn = att_last < att          # the last layer's output projection then runs on the read rows as well
    last_proj = lin * proj * 0.75 + (rdf if pruned_attn else lin) * proj * 0.25    # QKV on every row | output projection
    enc = 3 * ((L - 1) * (lin * (proj + ffn) + att * attn) + last_proj + rdf * ffn + att_last * attn)
    bb = L * (2 * half * (4 * H * H + 2 * H * I) + 4 * half * half * H)
    lab = int(half * 0.15)
    heads = rdf * 3 * 2 * S * H * H + 3 * 2 * lab * H * (cfg.vocab_size + cfg.kg_vocab_size)
    return {"step": (enc + bb + heads) / 1e9, "encoder": enc / 1e9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64, BASELINE config 2)")
    ap.add_argument("--model", choices=sorted(MODELS), default="150k",
                    help="150k = BASELINE config 2/3 (12L/768h, the headline); 24L1024 = config 4 (synthetic scale-up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-alone", action="store_true",
                    help="two more instrumented steps WITHOUT the second stream: the dominant kernel's own efficiency as a "
                         "labelled extra field (off by default, so that a rocprofv3 summary of the default command holds "
                         "in-step launches only and reproduces roofline.frac)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = MODELS[args.model]["batch"]
    return args


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


def cpu_baseline(cfg, seconds_budget=72.0):
    """The oracle (CPU restatement of the reference's HuggingFace path, pinned by tests/test_oracle_golden.py) timed
    on this host's cores: full training steps at the SAME model shape on a bounded sample - batch 8 as SURVEY section 8d
    specifies, at all the threads this process may use and, as a second point, at 8 threads."""
    from oracle import stonkgs_oracle as orc
    from stonkgs_amd.data import synthetic_batch

    ocfg = orc.OracleConfig(vocab_size=cfg.vocab_size, kg_vocab_size=cfg.kg_vocab_size, hidden_size=cfg.hidden_size,
                            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                            intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings)
    cores = host_cores()
    log(f"cpu_baseline: {cores} threads available, building fp32 weights")
    sd = orc.init_state_dict(ocfg, seed=0, bf16_exact=False)
    table = torch.randn(ocfg.kg_vocab_size + 3, ocfg.hidden_size) * 0.3
    B = 8
    batch = synthetic_batch(B, ocfg.vocab_size, ocfg.kg_vocab_size, ocfg.max_position_embeddings, seed=4321)
    state = orc.AdamState()

    def timed(threads, budget, max_steps, warm):
        torch.set_num_threads(threads)
        for i in range(warm):   # warm-up (allocations, oneDNN primitive caches)
            tw = time.time()
            orc.train_step(sd, ocfg, table, batch, state)
            log(f"cpu_baseline[{threads} threads]: warm-up step {i + 1} {time.time() - tw:.1f} s")
        n, t0 = 0, time.time()
        while n < 1 or (time.time() - t0 < budget and n < max_steps):
            orc.train_step(sd, ocfg, table, batch, state)
            n += 1
            log(f"cpu_baseline[{threads} threads]: step {n} at {time.time() - t0:.1f} s")
        return n, time.time() - t0

    n, dt = timed(cores, seconds_budget * 0.55, 5, 3)   # three warm-up steps, five timed
    out = {"value": B * n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"{n} full fp32 training steps (forward, backward, clip, AdamW) of the CPU oracle at the same model "
                     f"shape, batch {B} (seq 512, V={ocfg.vocab_size}, K={ocfg.kg_vocab_size}), after three warm-up steps; "
                     f"torch {torch.__version__}, mkldnn {torch.backends.mkldnn.is_available()}, "
                     f"mkl {torch.backends.mkl.is_available()}"}
    if cores > 8:
        n8, dt8 = timed(8, seconds_budget * 0.2, 2, 1)
        out["at_8_threads"] = {"value": B * n8 / dt8, "cores": 8, "steps": n8}
    return out


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 outside a launcher: start N ranks (one per GPU, RCCL) as CHILD processes this
    parent owns (stonkgs_amd/launch.py: explicit RANK / WORLD_SIZE environment, one session per rank, a wall limit after
    which every rank is stopped and the exit code is non-zero) and relay rank 0's JSON line. The parent does no GPU work
    and is never replaced: it only counts devices and waits."""
    from stonkgs_amd.launch import run_ranks

    n_dev = torch.cuda.device_count()   # (a device count only; the ranks are fresh processes either way)
    if n_dev < args.gpus and os.environ.get("STONK_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible", file=sys.stderr, flush=True)
        return 2
    # generous: model build + warm-up + timed steps + instrumented steps; a hung rank ends the job here, not never
    limit = float(os.environ.get("STONK_BENCH_WALL_LIMIT", 600 + 3.0 * (args.steps + args.warmup)))
    log(f"spawning {args.gpus} ranks (wall limit {limit:.0f} s)")
    res = run_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout=limit, relay=True)
    if res.returncode != 0:
        why = "wall limit reached, ranks stopped" if res.timed_out else f"rank exit codes {res.codes}"
        print(f"bench.py: {args.gpus}-rank job failed ({why})\n{res.tail(1500)}", file=sys.stderr, flush=True)
    return res.returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # never report a job of a different size than the one asked for
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    import torch.distributed as dist

    # one rank per GPU over RCCL; STONK_DIST_BACKEND=gloo lets several ranks share one card (a rehearsal of the N > 1
    # control flow on a one-GPU box - never a measurement)
    backend = os.environ.get("STONK_DIST_BACKEND", "nccl")
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from datetime import timedelta

        limit = timedelta(minutes=5)   # a rank that never arrives becomes an error on the others, not an endless wait
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    log(f"rank {rank}/{world}: building model {args.model}")
    spec = MODELS[args.model]
    cfg = STonKGsConfig(**spec["cfg"])  # default: 12L / 768h / 12 heads / 3072 / 512 positions / V 28996 / K 175094, dropout 0.1
    model = STonKGsForPreTraining(cfg, seed=0)  # same seed on every rank: replicas start identical (as DDP broadcasts)
    trainer = Trainer(model, TrainingArguments(per_device_train_batch_size=args.batch, max_steps=200, learning_rate=1e-4))
    if world > 1 and (trainer.world != world or trainer.sync.world != world):
        raise RuntimeError("gradient synchronizer does not see every rank")
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
    nb = len(batches)
    for i in range(args.warmup):   # (every step is told the next batch, as Trainer.train does: its frozen-backbone forward
        loss = trainer.training_step(model, batches[i % nb], next_inputs=batches[(i + 1) % nb])   # is queued a step ahead)
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss {float(loss):.4f}")
    barrier()
    log("timed region")
    t0 = time.perf_counter()
    for i in range(args.steps):   # (each timed step contains ONE backbone forward: the next batch's)
        loss = trainer.training_step(model, batches[(args.warmup + i) % nb], next_inputs=batches[(args.warmup + i + 1) % nb])
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
    encoder_path = None
    if not args.no_roofline:
        # dominant kernel of the step (rocprofv3 kernel-trace of this same command, profiles/): the weight-gradient GEMM.
        # Every GEMM launch of two extra steps is bracketed by HIP events ON THE STREAM IT IS LAUNCHED ON, in the launch
        # configuration the timed steps use (weight gradients on the second stream with their CU share, beside the dgrad
        # chain): achieved = algorithmic FLOPs (2 * M' * N' * tokens per launch) / summed durations, so the figure is
        # the in-step one and can be recomputed from the rocprofv3 average of the same kernel
        # (every rank runs the two steps - they contain the gradient all-reduce - and times its own launches; rank 0 reports)
        from stonkgs_amd.engine import GemmTimer

        model.engine.gemm_timer = GemmTimer()
        for i in range(2):
            trainer.training_step(model, batches[i % len(batches)])
        torch.cuda.synchronize()
        timer = model.engine.gemm_timer
        # (--roofline-alone) the same launches ALONE (serial order: no second stream, every CU theirs) in two more steps - a
        # labelled second figure, never the headline: in the step the kernel shares the chip with the dgrad chain
        alone = None
        if args.roofline_alone:
            model.engine.gemm_timer = GemmTimer()
            model.engine.overlap_wgrad = False
            for i in range(2):
                trainer.training_step(model, batches[i % len(batches)])
            torch.cuda.synchronize()
            alone = model.engine.gemm_timer
            model.engine.overlap_wgrad = True
        model.engine.gemm_timer = None
        kinds = timer.kinds()
        # the dominant SINGLE kernel of the rocprofv3 summary ("nt" lumps several forward / dgrad kernels together)
        single = [k for k in kinds if k != "nt"] or kinds
        dom = max(single, key=lambda k: timer.summarize(k)["seconds"])
        s = timer.summarize(dom)
        a = timer.summarize(None)
        ach = s["flops"] / s["seconds"] / 1e12
        roofline = {"bound": "mfma", "kernel": KERNEL_NOTES.get(dom, dom),
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": TRAFFIC_BYTES.get((args.model, dom)),
                    "traffic_note": "bytes per launch at the L2's memory side = 2 x FETCH_SIZE + WRITE_SIZE (KiB, separate "
                                    "rocprofv3 --pmc passes over this command, profiles/) - a recorded constant: counters "
                                    "cannot be read from inside the bench; null when no pass was taken for this kernel",
                    "timing": "HIP events around every launch on the stream the step launches it on (second stream, "
                              "CU share as in the timed steps), two instrumented steps after the timed region",
                    "launches_per_step": s["launches"] // 2,
                    "alone": None if alone is None else {
                        "what": "the same launches without the second stream (serial order, all 256 CUs): the kernel's own "
                                "efficiency, not what the step achieves",
                        "frac": round(alone.summarize(dom)["flops"] / alone.summarize(dom)["seconds"] / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "avg_launch_us": round(alone.summarize(dom)["seconds"] / max(1, alone.summarize(dom)["launches"]) * 1e6, 1)},
                    "avg_launch_us": round(s["seconds"] / s["launches"] * 1e6, 1),
                    "avg_launch_gflop": round(s["flops"] / s["launches"] / 1e9, 2),
                    "by_kernel": {k: {"launches_per_step": timer.summarize(k)["launches"] // 2,
                                      "ms_per_step": round(timer.summarize(k)["seconds"] / 2 * 1e3, 3),
                                      "frac": round(timer.summarize(k)["flops"] / max(timer.summarize(k)["seconds"], 1e-12)
                                                    / 1e12 / PEAK_BF16_TFLOPS, 4)} for k in kinds}}
        gemm_all = {"kernels": "every GEMM launch of the step (NT forward / dgrad and TN weight-gradient kernels)",
                    "launches_per_step": a["launches"] // 2,
                    "achieved_tflops": round(a["flops"] / a["seconds"] / 1e12, 1),
                    "frac": round(a["flops"] / a["seconds"] / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "ms_per_step": round(a["seconds"] / 2 * 1e3, 2)}
        # the attention+FFN path (BASELINE.json's 40 % target): stream-order spans around the trainable encoder's forward
        # and backward (the backward span ends once that span's weight gradients on the second stream are done; the
        # decoders' weight gradients still running on that stream at its start are inside it, so this errs low)
        enc_s = (timer.span_seconds("encoder_fwd") + timer.span_seconds("encoder_bwd")) / 2
        lin, att, rdf, att_last = executed_fraction(model.engine.rows_executed)
        enc_gf = gflop_per_pair_executed(cfg, lin, att, rdf, att_last)["encoder"]
        enc_tf = enc_gf * args.batch / 1e3 / enc_s
        encoder_path = {"what": "trainable encoder forward + backward (QKV, attention, projections, FFN, LayerNorm, "
                                "weight gradients), event spans on the main stream over two instrumented steps; FLOPs of "
                                "the rows that ran (padding rows nothing reads are dropped), not of the padded batch",
                        "gflop_per_pair": round(enc_gf, 1), "gflop_per_pair_padded": round(encoder_gflop_per_pair(cfg), 1),
                        "ms_per_step": round(enc_s * 1e3, 2),
                        "achieved_tflops": round(enc_tf, 1), "frac": round(enc_tf / PEAK_BF16_TFLOPS, 4)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / dt
        lin, att, rdf, att_last = executed_fraction(model.engine.rows_executed)
        gfl = gflop_per_pair_executed(cfg, lin, att, rdf, att_last)["step"]   # the FLOPs the step executed
        out = {"metric": "text-triple pairs/sec (whole node), seq_len=512 hidden=768, 1/2/4/8 MI355X", "value": round(value, 2),
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"{spec['name']}, per-GPU batch "
                                      f"{args.batch}, seq 256 text + 256 entity, dropout 0.1, AdamW lr 1e-4",
                          "global_batch": args.batch * world, "seq_len": 512, "parallelism": f"dp{world}"},
               "final_loss": round(final_loss, 4),
               "gflop_per_pair": round(gfl, 1), "gflop_per_pair_padded": round(gflop_per_pair(cfg), 1),
               "rows_executed": {"linear": round(lin, 4), "attention": round(att, 4), "last_ffn_and_heads": round(rdf, 4),
                                 "last_attention": round(att_last, 4),
                                 "what": "share of the padded encoder's rows (and of its rows^2 per sequence) the unpadded "
                                         "encoder ran: positions that are neither live keys, nor labelled, nor position 0 "
                                         "are dropped - no loss term or gradient reads them; the last layer's feed-forward "
                                         "block, the pooler and the head transform run on the labelled rows + position 0"},
               "step_mfma_frac": round(value * gfl / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
               "roofline": roofline, "all_gemm": gemm_all,
               "encoder_path": encoder_path}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
