"""CPU, world_size 2 over gloo: the data-parallel gradient path (bucketed all-reduce of the flat gradient buffer in
backward-completion order, averaging folded into the optimizer scale) and the dataset sharding of the Trainer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stonkgs_amd.stonkgs_pretraining import GradSynchronizer, plan_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5000
        grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
        segments = {"a": 1200, "b": 1300, "c": 3000, "d": n}
        sync = GradSynchronizer(grad, segments, bucket_mb=1000 * 4 / (1 << 20))  # 1000-element buckets
        assert sync.buckets == plan_buckets([1200, 1300, 3000, n], 1000)
        # backward reports segments front to back; everything before a reported end must be reducible at once
        sync.on_segment_done("a")
        launched_after_a = len(sync._works)
        sync.on_segment_done("b")
        sync.on_segment_done("c")
        scale = sync.finish()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(grad, expect) and scale == 1.0 / world and launched_after_a == 1
        # second step reuses the synchronizer
        grad.fill_(float(rank))
        for name in ("a", "b", "c", "d"):
            sync.on_segment_done(name)
        sync.finish()
        ok = ok and torch.equal(grad, torch.full((n,), float(sum(range(world)))))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_a_noop():
    grad = torch.ones(10)
    s = GradSynchronizer(grad, {"x": 10})
    s.on_segment_done("x")
    assert s.finish() == 1.0 and torch.equal(grad, torch.ones(10))


# ---- the rank launcher (stonkgs_amd/launch.py): owned pids, bounded run time, output that survives a kill
def _probe(mode, **kw):
    import sys

    from stonkgs_amd.launch import run_ranks

    here = os.path.dirname(os.path.abspath(__file__))
    return run_ranks(2, [sys.executable, os.path.join(here, "_rank_probe.py"), mode], **kw)


def test_launcher_runs_two_gloo_ranks():
    r = _probe("allreduce", timeout=120)
    assert r.returncode == 0 and r.codes == [0, 0], r.tail()
    assert all("sum 3" in o for o in r.stdout)
    sids = {o.split("sid ")[1].split()[0] for o in r.stdout}
    assert len(sids) == 2 and str(os.getsid(0)) not in sids      # every rank leads a session of its own


def test_launcher_deadline_stops_every_rank_and_keeps_their_output():
    import time

    t0 = time.time()
    r = _probe("hang", timeout=3)
    assert r.timed_out and r.returncode == 124 and time.time() - t0 < 30
    assert all("about to hang" in e for e in r.stderr)           # what the ranks wrote before the kill came back
    pids = [int(o.split("pid ")[1].split()[0]) for o in r.stdout]
    pids.append(int(r.stdout[1].split("grandchild ")[1].split()[0]))
    time.sleep(0.5)
    for pid in pids:                                             # ranks AND the grandchild are gone
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid


def test_launcher_stops_the_peers_of_a_dead_rank():
    import time

    t0 = time.time()
    r = _probe("die", timeout=120, peer_grace=1.0)
    assert r.returncode == 7 and not r.timed_out and r.codes[0] == 7 and r.codes[1] != 0
    assert time.time() - t0 < 30


def _pids_gone(pids, wait=5.0):
    import time

    t_end = time.time() + wait
    while True:
        alive = []
        for pid in pids:
            try:
                os.kill(pid, 0)
                if open(f"/proc/{pid}/stat").read().split()[2] != "Z":
                    alive.append(pid)
            except (ProcessLookupError, FileNotFoundError):
                pass
        if not alive or time.time() > t_end:
            return alive
        time.sleep(0.1)


@pytest.mark.parametrize("sig", ["TERM", "KILL"])
def test_ranks_do_not_outlive_the_launcher(tmp_path, sig):
    """An outer `timeout` / harness limit stops the LAUNCHER, not the ranks (each leads a session of its own): SIGTERM is
    turned into a stop of every rank's group by the launcher's handler, SIGKILL - which nothing can handle - reaches the
    ranks through PR_SET_PDEATHSIG. Either way no rank keeps running (and keeps its GPU)."""
    import signal
    import subprocess
    import sys
    import time

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    pidfile = tmp_path / "pids"
    code = (f"import sys; sys.path.insert(0, {root!r})\n"
            "from stonkgs_amd.launch import run_ranks\n"
            f"run_ranks(2, [sys.executable, {os.path.join(here, '_rank_probe.py')!r}, 'hang_pidfile', {str(pidfile)!r}], timeout=300)\n")
    launcher = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    t_end = time.time() + 60
    while time.time() < t_end and not (pidfile.exists() and len(pidfile.read_text().split()) >= 3):
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) >= 3, "two ranks and rank 1's grandchild should have reported"
    launcher.send_signal(signal.SIGTERM if sig == "TERM" else signal.SIGKILL)
    rc = launcher.wait(timeout=30)
    assert rc != 0
    if sig == "TERM":
        assert rc == 128 + signal.SIGTERM and b"ranks stopped" in launcher.stderr.read()
        assert not _pids_gone(pids), "ranks (or the grandchild) survived a SIGTERMed launcher"
    else:
        assert not _pids_gone(pids[:2]), "ranks survived a SIGKILLed launcher"     # (the parent-death signal is per child)
        for pid in pids[2:]:
            try:
                os.kill(pid, signal.SIGKILL)
            except ProcessLookupError:
                pass


# ---- optimizer sharding (TrainingArguments.shard_optimizer): reduce into the owned piece of every bucket, update it,
# all-gather the parameters - world 2 over gloo
def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 8192
        segments = {"a": 2048, "b": 2560, "c": 6144, "d": n}          # tensor boundaries: multiples of 256
        for step in range(2):
            g_local = [torch.arange(n, dtype=torch.float32) * (r + 1) + step for r in range(world)]
            grad = g_local[rank].clone()
            sync = GradSynchronizer(grad, segments, bucket_mb=2048 * 4 / (1 << 20), shard=True)
            assert sync.shard and sync.buckets == plan_buckets([2048, 2560, 6144, n], 2048)
            spans = sync.owned_spans()
            assert len(spans) == len(sync.buckets)
            for (lo, hi), (blo, bhi) in zip(spans, sync.buckets):
                piece = (bhi - blo) // world
                assert (lo, hi) == (blo + rank * piece, blo + (rank + 1) * piece)
            for name in ("a", "b", "c"):
                sync.on_segment_done(name)
            scale = sync.finish()
            total = sum(g_local)
            ok = scale == 1.0 / world
            for lo, hi in spans:                                       # the owned pieces hold the sum over the ranks
                ok = ok and torch.equal(grad[lo:hi], total[lo:hi])
            # grad-norm: squares of the owned pieces, summed over the ranks = the whole buffer's
            sq = torch.stack([(grad[lo:hi].double() ** 2).sum() for lo, hi in spans]).sum().reshape(1)
            sync.all_reduce_scalar(sq)
            ok = ok and abs(float(sq) - float((total.double() ** 2).sum())) <= 1e-9 * float(sq)
            # "optimizer": every rank updates ITS pieces of the parameters, then the pieces are gathered
            params = torch.full((n,), 5.0)
            for lo, hi in spans:
                params[lo:hi] -= 0.5 * scale * grad[lo:hi]
            sync.gather_params(params)
            ok = ok and torch.equal(params, 5.0 - 0.5 * scale * total)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_optimizer_collectives_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_sharding_refuses_a_world_size_that_cuts_unaligned_pieces(monkeypatch):
    import torch.distributed as d

    monkeypatch.setattr(d, "is_initialized", lambda: True)
    monkeypatch.setattr(d, "get_world_size", lambda group=None: 3)
    monkeypatch.setattr(d, "get_rank", lambda group=None: 0)
    with pytest.raises(ValueError, match="world size 3"):
        GradSynchronizer(torch.zeros(4096), {"a": 2048, "b": 4096}, bucket_mb=1e-9, shard=True)
