"""Row (e) on hardware: the data-parallel step with real collectives.

* one-rank RCCL group with the collectives forced on: the nccl backend, the all-reduce issued under the weight-gradient
  stream, the optimizer stream waiting for it and the kernel routing that goes with communication all execute on a one-GPU
  box; a sum over one rank leaves the gradients as they are, so the result must equal the plain single-process step;
* two ranks sharing GPU 0 over gloo (`tools/dp_check.py`): replicas bitwise identical, equal to a single process that
  accumulates the two ranks' batches;
* two ranks over RCCL, one per GPU - the production path (`bench.py --gpus 2` runs it too): skipped on a one-GPU box.
Reference behaviour: DDP through HF Trainer, ref:src/stonkgs/models/stonkgs_pretraining.py:215-223."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ranks(n, script, *args, env=None, timeout=300):
    """Start `n` ranks of a script through stonkgs_amd/launch.py: the test owns every rank's pid (each rank is the leader
    of its own session and is signalled directly: SIGTERM, then SIGKILL - nothing is left behind on the GPU, and no
    launcher sits in between that a kill would orphan its workers from), their output goes to files and comes back whole,
    also when the wall limit ends the job."""
    from stonkgs_amd.launch import run_ranks

    return run_ranks(n, [sys.executable, os.path.join(ROOT, script), *args], timeout=timeout, env=env)


def test_rccl_path_executes_in_a_one_rank_group(hip):
    import torch.distributed as dist

    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg = STonKGsConfig(vocab_size=2048, kg_vocab_size=640, num_hidden_layers=2, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    table = torch.randn(640, cfg.hidden_size, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 0.3
    B = 32   # T = 16 384 tokens: the launches take the kernels the full-size step takes
    batches = [synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=100 + i) for i in range(2)]

    init = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)._store.data.detach().clone()

    def run(force, shard=False):
        model = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)
        tr = Trainer(model, TrainingArguments(learning_rate=1e-3, max_steps=10, per_device_train_batch_size=B,
                                              ddp_bucket_mb=8, ddp_force_collectives=force, shard_optimizer=shard))
        model.engine.comm_overlap = True        # same kernel routing in both arms
        losses = [float(tr.training_step(model, b)) for b in batches]
        model.engine.check_errors()
        model.engine.wait_params()
        torch.cuda.synchronize()
        return losses, model._store.data.detach().clone(), tr

    from datetime import timedelta

    from stonkgs_amd.launch import free_port

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()),
                            timeout=timedelta(seconds=120))
    try:
        l1, p1, tr1 = run(True)
        assert tr1.sync.active and len(tr1.sync.buckets) >= 3 and dist.get_backend() == "nccl"
        l0, p0, tr0 = run(False)
        assert not tr0.sync.active
        # the sharded optimizer over RCCL in the one-rank group: reduce_scatter_tensor into the (whole-bucket) owned piece,
        # per-piece AdamW, all_gather_into_tensor in place, bf16 mirror rebuilt from the gathered parameters
        l2, p2, tr2 = run(True, shard=True)
        assert tr2.sync.shard and tr2.optimizer.spans == [(lo, hi) for lo, hi in tr2.sync.buckets]
    finally:
        dist.destroy_process_group()
    # (the second step's loss sits behind one Adam step of lr 1e-3 on gradients whose fp32 atomics reorder from run to run:
    # two plain runs differ by 1-2e-5 of it; a wrong bucket moves it by percents)
    assert l1 == pytest.approx(l0, rel=5e-5)
    # fp32 atomics reorder addends from run to run, and Adam turns last-bit differences of near-zero gradients into +-lr
    # steps: compare the two-step displacement of the parameters, and bound the largest difference by two lr-sized steps
    d = (p1 - p0).abs()
    moved1, moved0 = p1 - init, p0 - init
    cos = torch.nn.functional.cosine_similarity(moved1.flatten(), moved0.flatten(), dim=0).item()
    assert float(d.max()) <= 2.1e-3 and cos > 0.999, (float(d.max()), cos)
    assert l2 == pytest.approx(l0, rel=5e-5)
    cos2 = torch.nn.functional.cosine_similarity((p2 - init).flatten(), moved0.flatten(), dim=0).item()
    assert float((p2 - p0).abs().max()) <= 2.1e-3 and cos2 > 0.999, (float((p2 - p0).abs().max()), cos2)


def test_stonk_comm_c_abi_in_a_one_rank_group(hip):
    """`stonk_comm_*` (RCCL on a library-owned stream, event hand-off) on the one-GPU box: a one-rank
    communicator - the collectives execute, a sum over one rank leaves the values as they are. Checked: the comm stream is
    ordered BEHIND the producer stream (the all-reduce sees what a kernel queued just before it wrote), the consumer stream
    behind the comm stream, reduce-scatter / all-gather into the rank's own slice in place, both dtypes; and the training
    step on this backend (all-reduce and sharded) equals the plain step."""
    from stonkgs_amd.comm import StonkComm
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    comm = StonkComm(0, 1, torch.cuda.current_device())
    assert comm.stream_ptr not in (0, torch.cuda.current_stream().cuda_stream)
    n = 64 << 20
    x = torch.zeros(n, device="cuda")
    big = torch.randn(8192, 8192, device="cuda")
    for _ in range(3):
        big = big @ big * 1e-4                       # keeps the producer stream busy: the fill below is queued, not done
    x.fill_(3.0)
    comm.all_reduce(x)
    comm.wait()
    y = x * 2
    torch.cuda.synchronize()
    assert float(y.min()) == 6.0 and float(y.max()) == 6.0
    h = torch.arange(4096, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    comm.all_reduce(h)
    recv = torch.empty(4096, device="cuda")
    send = torch.arange(4096, device="cuda", dtype=torch.float32)
    comm.reduce_scatter(recv, send)
    comm.reduce_scatter(send[:4096], send)           # in place: the rank's own slice of the input
    out = torch.zeros(4096, device="cuda")
    comm.all_gather(out, recv)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(out, send) and torch.equal(h.float(), torch.arange(4096, device="cuda").to(torch.bfloat16).float())
    comm.close()

    cfg = STonKGsConfig(vocab_size=2048, kg_vocab_size=640, num_hidden_layers=2, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    table = torch.randn(640, cfg.hidden_size, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 0.3
    B = 32
    batches = [synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=100 + i) for i in range(2)]
    init = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)._store.data.detach().clone()

    def run(backend, force, shard=False):
        model = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)
        tr = Trainer(model, TrainingArguments(learning_rate=1e-3, max_steps=10, per_device_train_batch_size=B, ddp_bucket_mb=8,
                                              ddp_force_collectives=force, shard_optimizer=shard, comm_backend=backend))
        model.engine.comm_overlap = True
        losses = [float(tr.training_step(model, b)) for b in batches]
        model.engine.check_errors()
        model.engine.wait_params()
        torch.cuda.synchronize()
        p = model._store.data.detach().clone()
        if tr.comm is not None:
            tr.comm.close()
        return losses, p, tr

    l0, p0, tr0 = run("torch", False)
    l1, p1, tr1 = run("stonk", True)
    assert tr1.sync.active and tr1.sync.comm is not None and len(tr1.sync.buckets) >= 3 and not tr0.sync.active
    l2, p2, tr2 = run("stonk", True, shard=True)
    assert tr2.sync.shard
    for l, p in ((l1, p1), (l2, p2)):
        assert l == pytest.approx(l0, rel=5e-5)
        cos = torch.nn.functional.cosine_similarity((p - init).flatten(), (p0 - init).flatten(), dim=0).item()
        assert float((p - p0).abs().max()) <= 2.1e-3 and cos > 0.999, (float((p - p0).abs().max()), cos)


def test_bucket_grad_norm_equals_the_one_pass_norm(hip):
    """TrainingArguments.bucket_grad_norm (round 4): sum(g^2) taken bucket by bucket as backward finalises them - on the
    weight-gradient stream - must clip exactly as the one pass over the whole buffer did: same losses, same parameters after
    two steps (up to the reordering of fp32 atomics that two runs of either show)."""
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg = STonKGsConfig(vocab_size=2048, kg_vocab_size=640, num_hidden_layers=2, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    table = torch.randn(640, cfg.hidden_size, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 0.3
    B = 32
    batches = [synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=100 + i) for i in range(2)]
    init = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)._store.data.detach().clone()
    res = []
    for flag in (True, False):
        model = STonKGsForPreTraining(cfg, kg_embeddings=table, seed=0)
        tr = Trainer(model, TrainingArguments(learning_rate=1e-3, max_steps=10, per_device_train_batch_size=B, ddp_bucket_mb=8,
                                              bucket_grad_norm=flag))
        assert (tr.sync._norm_parts is not None) == flag
        losses, norms = [], []
        for b in batches:
            losses.append(float(tr.training_step(model, b)))
            model.engine.wait_params()
            norms.append(float(tr.optimizer.gnorm_sq.sqrt()))
        model.engine.check_errors()
        torch.cuda.synchronize()
        res.append((losses, norms, model._store.data.detach().clone()))
    (l1, n1, p1), (l0, n0, p0) = res
    assert n1[0] > 1.0                                            # the clip is active: the norm matters
    # step 1: the same gradients, two summation orders; step 2 sits behind one Adam step on gradients whose fp32 atomics
    # reorder from run to run (two runs of EITHER form differ by a few 1e-5 there)
    assert n1[0] == pytest.approx(n0[0], rel=5e-6) and n1[1] == pytest.approx(n0[1], rel=2e-4) and l1 == pytest.approx(l0, rel=5e-5)
    cos = torch.nn.functional.cosine_similarity((p1 - init).flatten(), (p0 - init).flatten(), dim=0).item()
    assert float((p1 - p0).abs().max()) <= 2.1e-3 and cos > 0.999


def test_two_ranks_on_one_gpu_over_gloo(hip):
    r = _ranks(2, "tools/dp_check.py", env={"STONK_DIST_BACKEND": "gloo"})
    assert r.returncode == 0 and "DP2 OK (gloo)" in r.stdout[0], r.tail()
    assert "gradient buffer after the collectives" in r.stderr[0]      # (check A ran: bucket boundaries)


def test_two_ranks_on_one_gpu_with_a_sharded_optimizer(hip):
    """TrainingArguments.shard_optimizer (ZeRO-2, as the reference can run it: ref:stonkgs_pretraining.py:174-175): each rank
    reduces into and updates its half of every bucket, the parameters are all-gathered; replicas bitwise identical and
    equal to the single-process run, as without sharding."""
    r = _ranks(2, "tools/dp_check.py", env={"STONK_DIST_BACKEND": "gloo", "STONK_DP_SHARD": "1"})
    assert r.returncode == 0 and "DP2 OK (gloo, sharded optimizer)" in r.stdout[0], r.tail()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_over_rccl(hip):
    r = _ranks(2, "tools/dp_check.py")
    assert r.returncode == 0 and "DP2 OK (nccl)" in r.stdout[0], r.tail()
    # the sharded optimizer's REAL multi-rank path - in-place reduce_scatter_tensor into a slice of its own input, in-place
    # all_gather_into_tensor, ownership of piece r of every bucket for r > 0 - has run over gloo and in one-rank groups only:
    # the first box with two GPUs exercises it here
    r = _ranks(2, "tools/dp_check.py", env={"STONK_DP_SHARD": "1"})
    assert r.returncode == 0 and "DP2 OK (nccl, sharded optimizer)" in r.stdout[0], r.tail()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_gpus_2_is_a_two_rank_rccl_job(hip):
    """`python bench.py --gpus 2` (no launcher) starts two ranks itself and reports them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-roofline"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, STONK_BENCH_WALL_LIMIT="600"))   # (bench.py stops its own ranks at its limit)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["config"]["parallelism"] == "dp2"


def test_bench_refuses_more_ranks_than_gpus(hip):
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "visible" in r.stderr
