"""Rank body for tests/test_dp_cpu.py::test_launcher_*: what it does is chosen by argv[1]."""
import os
import sys
import time

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
print(f"rank {rank} of {world} pid {os.getpid()} sid {os.getsid(0)}", flush=True)
if mode == "allreduce":
    from datetime import timedelta

    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", timeout=timedelta(seconds=60))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print(f"sum {t.item():.0f}", flush=True)
    dist.destroy_process_group()
elif mode == "hang":
    print("about to hang", file=sys.stderr, flush=True)
    if rank == 1:   # a grandchild in the rank's own session: the group kill must reach it too
        import subprocess

        child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
        print(f"grandchild {child.pid}", flush=True)
    time.sleep(600)
elif mode == "hang_pidfile":   # as "hang", reporting the pids through a file (the launcher's own output is not read)
    pids = [os.getpid()]
    if rank == 1:
        import subprocess

        pids.append(subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"]).pid)
    with open(sys.argv[2], "a") as f:
        f.write(" ".join(map(str, pids)) + "\n")
    time.sleep(600)
elif mode == "die":
    if rank == 0:
        sys.exit(7)
    time.sleep(600)
