"""One process per GPU on one node: start the ranks of a data-parallel job, own their pids, bound their run time.

The reference leaves this to whatever launches HF ``Trainer`` (ref:src/stonkgs/models/stonkgs_pretraining.py:215-223: DDP
"if launched distributed"). Here the launcher is explicit, because two things a generic one does not give are needed:

* the caller OWNS every rank's pid. ``torch.distributed.run`` starts its workers in sessions of their own, so killing the
  launcher's process group reaches the launcher only and a SIGKILLed launcher cannot reap them: a hung rank keeps its GPU.
  `run_ranks` starts each rank itself (``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` in
  its environment, the contract ``torch.distributed``'s ``env://`` rendezvous reads), each in its own session, and on a
  deadline - or when one rank dies and the others would wait for it in a collective - sends SIGTERM, then SIGKILL, to every
  rank's process group;
* every rank's stdout / stderr goes to a FILE (no pipe that a dead reader can block, nothing lost on a timeout); rank 0's
  stderr is relayed while the job runs, its stdout (the one JSON line of bench.py) when it ends.

* the ranks do not outlive the launcher: SIGTERM / SIGHUP / SIGINT to the launcher (an outer ``timeout``, a harness step
  limit, a closed terminal) stop every rank's process group before it exits non-zero, and a launcher that is SIGKILLed -
  which no handler sees - takes its ranks with it through PR_SET_PDEATHSIG (each rank asks the kernel for SIGKILL when
  its parent dies).

Nothing here touches a GPU: the parent may start ranks before or without initialising HIP, and never replaces itself.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank: int, world: int, port: int, extra: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of one rank. 127.0.0.1 everywhere: a container's hostname need not resolve, and a resolver that fails
    SLOWLY turns every lookup into a stall - `GLOO_SOCKET_IFNAME=lo` keeps gloo from looking its hostname up at all
    (RCCL on one node talks over xGMI / shared memory and needs no interface)."""
    e = dict(os.environ)
    e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
             MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.setdefault("GLOO_SOCKET_IFNAME", "lo")
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    e.update(extra or {})
    return e


@dataclass
class RankResult:
    returncode: int                      # 0, the first failing rank's code, or 124 on the deadline
    timed_out: bool = False
    stdout: List[str] = field(default_factory=list)   # per rank
    stderr: List[str] = field(default_factory=list)
    codes: List[Optional[int]] = field(default_factory=list)

    def tail(self, n: int = 3000) -> str:
        return "\n".join(f"--- rank {r} (exit {self.codes[r]}) ---\n{self.stdout[r][-n:]}\n{self.stderr[r][-n:]}"
                         for r in range(len(self.codes)))


def _signal_group(p: subprocess.Popen, sig: int) -> None:
    try:
        os.killpg(p.pid, sig)            # the rank is the leader of its own session / process group
    except (ProcessLookupError, PermissionError):
        pass


def _group_alive(p: subprocess.Popen) -> bool:
    try:
        os.killpg(p.pid, 0)
        return True
    except (ProcessLookupError, PermissionError):
        return False


def stop_ranks(procs: Sequence[subprocess.Popen], grace: float = 5.0) -> None:
    """SIGTERM every rank's process group, wait `grace` seconds, SIGKILL what is left, reap everything. EVERY rank's group
    is signalled, also that of a rank whose leader has exited: what it started (a data-loader worker, a helper that still
    holds the GPU) is in that group, and a group id cannot be reused while the group has members."""
    for p in procs:
        _signal_group(p, signal.SIGTERM)
    t_end = time.time() + grace
    while time.time() < t_end and any(p.poll() is None or _group_alive(p) for p in procs):
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None or _group_alive(p):
            _signal_group(p, signal.SIGKILL)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:   # (unkillable: a process stuck in the driver; nothing more a parent can do)
            pass


class _Stopped(BaseException):
    """Raised by the launcher's signal handlers into `run_ranks`, whose `finally` stops the ranks."""

    def __init__(self, signum: int):
        super().__init__(signum)
        self.signum = signum


def _die_with_parent() -> None:
    """preexec_fn of a rank: SIGKILL from the kernel when the launcher dies (prctl(PR_SET_PDEATHSIG) - Linux)."""
    try:
        import ctypes

        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG = 1
    except Exception:
        pass


def run_ranks(world: int, argv: Sequence[str], timeout: float, env: Optional[Dict[str, str]] = None,
              relay: bool = False, peer_grace: float = 20.0) -> RankResult:
    """Run `argv` (e.g. [sys.executable, "bench.py", ...]) as `world` ranks and wait for them.

    * `timeout` seconds for the whole job; on expiry every rank is stopped and the result says so (returncode 124);
    * a rank that exits non-zero stops the others after `peer_grace` seconds (they would wait for it in a collective);
    * `relay`: copy rank 0's stderr to this process's stderr while the job runs and its stdout to stdout at the end.
    """
    port = free_port()
    logdir = tempfile.mkdtemp(prefix="stonk_ranks_")
    files, procs = [], []
    # SIGTERM / SIGHUP / SIGINT to the launcher become an exception here, so that the `finally` below runs (handlers can
    # only be installed from the main thread; elsewhere the caller keeps that responsibility)
    handled = {}
    import threading

    if threading.current_thread() is threading.main_thread():
        def _raise(signum, _frame):
            raise _Stopped(signum)

        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            handled[sig] = signal.signal(sig, _raise)
    try:
        for r in range(world):
            fo = open(os.path.join(logdir, f"rank{r}.out"), "w+b")
            fe = open(os.path.join(logdir, f"rank{r}.err"), "w+b")
            files.append((fo, fe))
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, env), stdout=fo, stderr=fe,
                                          stdin=subprocess.DEVNULL, start_new_session=True, preexec_fn=_die_with_parent))
        t_end = time.time() + timeout
        relayed = 0
        failed_at = None
        timed_out = False
        while any(p.poll() is None for p in procs):
            now = time.time()
            if relay:
                relayed = _relay(files[0][1], relayed, sys.stderr)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad and failed_at is None:
                failed_at = now
            if now > t_end or (failed_at is not None and now - failed_at > peer_grace):
                timed_out = now > t_end
                stop_ranks(procs)
                break
            time.sleep(0.2)
        stop_ranks(procs, grace=0.0)         # (reaps; everything has exited by now)
        if relay:
            _relay(files[0][1], relayed, sys.stderr)
        outs, errs = [], []
        for fo, fe in files:
            for f, dst in ((fo, outs), (fe, errs)):
                f.flush()
                f.seek(0)
                dst.append(f.read().decode("utf-8", "replace"))
        codes = [p.returncode for p in procs]
        rc = 124 if timed_out else next((c for c in codes if c), 0)
        if relay and outs:
            sys.stdout.write(outs[0])
            sys.stdout.flush()
        return RankResult(rc, timed_out, outs, errs, codes)
    except _Stopped as stop:
        for sig in handled:                  # (a second signal while the ranks are being stopped must not interrupt that)
            signal.signal(sig, signal.SIG_IGN)
        stop_ranks(procs)
        sys.stderr.write(f"[launch] signal {stop.signum}: {world} ranks stopped\n")
        sys.stderr.flush()
        raise SystemExit(128 + stop.signum)
    finally:
        stop_ranks(procs, grace=0.0)
        for sig, old in handled.items():
            signal.signal(sig, old)
        for fo, fe in files:
            fo.close()
            fe.close()
        import shutil

        shutil.rmtree(logdir, ignore_errors=True)


def _relay(f, pos: int, dst) -> int:
    f.flush()
    size = os.fstat(f.fileno()).st_size
    if size > pos:
        data = os.pread(f.fileno(), size - pos, pos)
        dst.write(data.decode("utf-8", "replace"))
        dst.flush()
    return size
