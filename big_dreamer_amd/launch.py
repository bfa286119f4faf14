"""One process per GPU: spawn the ranks of a single-node data-parallel run.

``bench.py --gpus N`` (and ``src/main.py world_size=N``) started WITHOUT a launcher call :func:`spawn_ranks`, which
starts N fresh child processes of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set
-- what ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would have set -- and waits for them.  The
parent never touches HIP (a process that has initialised the GPU must not fork/exec workers on this pool); rank 0's
stdout is passed through, the other ranks' stdout goes to stderr, and the first failing rank takes the others down.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def launched_by_torchrun() -> bool:
    """True inside a rank (torch.distributed.run, or spawn_ranks below): the rendezvous variables are set."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))
    return env


def _stop(procs: Sequence[subprocess.Popen], grace: float = 15.0) -> None:
    """Terminate, wait for, then kill exactly the processes we started (never a pattern)."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.terminate()
        except OSError:
            pass
    t_end = time.monotonic() + grace
    for p in live:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


class _Interrupted(Exception):
    def __init__(self, signum: int):
        self.signum = signum


def spawn_ranks(argv: Sequence[str], world: int, timeout: Optional[float] = None,
                extra_env: Optional[Dict[str, str]] = None) -> int:
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "8"]) as `world` ranks.
    Returns 0 if every rank exited 0, else the first non-zero exit code (the remaining ranks are terminated); 124 after
    `timeout` seconds; 128 + signal when the launcher itself is told to stop.  However the call ends -- a failing rank,
    the timeout, SIGTERM / SIGINT / SIGHUP delivered to the launcher (Ctrl-C, `timeout -k`, a harness limit), an
    exception -- no rank process is left behind holding a GPU."""
    import signal
    assert world >= 1
    port = free_port()
    procs: List[subprocess.Popen] = []

    def on_signal(signum, _frame):
        raise _Interrupted(signum)

    handled = (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)
    old = {}
    for sg in handled:
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:          # not the main thread: the caller's handlers stay, the finally below still cleans up
            pass
    rc = 0
    try:
        for r in range(world):
            env = rank_env(r, world, port)
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else sys.stderr))
        t0 = time.monotonic()
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                print(f"[launch] ranks still running after {timeout:.0f} s: terminating them", file=sys.stderr, flush=True)
                rc = 124
                break
            if live:
                time.sleep(0.05)
    except _Interrupted as e:
        print(f"[launch] signal {e.signum}: terminating the rank processes", file=sys.stderr, flush=True)
        rc = 128 + e.signum
    finally:
        _stop(procs)
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc
