"""CPU: the self-launcher behind `python bench.py --gpus N` (big_dreamer_amd/launch.py) -- N fresh rank processes with
the torch.distributed.run environment, rank 0's stdout passed through, first failure terminates the rest."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = ("import sys; sys.path.insert(0, %r); from big_dreamer_amd import launch; "
          "sys.exit(launch.spawn_ranks([sys.executable, %r], 2, timeout=120))"
          % (ROOT, os.path.join(ROOT, "tests", "launch_worker.py")))


def test_two_gloo_ranks_rendezvous_and_rank0_owns_stdout():
    out = subprocess.run([sys.executable, "-c", DRIVER], capture_output=True, text=True, timeout=180, cwd=ROOT,
                         env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    assert "LAUNCH_OK world=2 sum=3 addr=127.0.0.1" in out.stdout
    assert "rank 0 stdout local_rank=0" in out.stdout
    # the other rank's stdout is routed to stderr, so the parent's stdout carries exactly one rank's lines (one JSON line)
    assert "rank 1 stdout" not in out.stdout and "rank 1 stdout local_rank=1" in out.stderr


def test_failing_rank_terminates_the_others_and_sets_the_exit_code():
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", DRIVER], capture_output=True, text=True, timeout=120, cwd=ROOT,
                         env=dict(os.environ, LAUNCH_FAIL_RANK="1", OMP_NUM_THREADS="1"))
    assert out.returncode == 3
    assert time.time() - t0 < 45, "the surviving rank was not terminated"


def test_parent_of_a_multi_gpu_bench_never_initialises_hip():
    """`python bench.py --gpus 2` without a launcher becomes the launcher before any torch.cuda call: on this GPU-less
    host the RANKS fail (no device), the parent reports their exit code instead of crashing itself."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and out.stdout.strip() == ""
        assert "Traceback" in out.stderr          # from a rank, relayed


def _pids_alive(pids):
    alive = []
    for pid in pids:
        try:
            os.kill(pid, 0)
            # a zombie still answers kill(0): read its state
            with open(f"/proc/{pid}/stat") as f:
                if f.read().split(")")[-1].split()[0] != "Z":
                    alive.append(pid)
        except (OSError, IOError):
            pass
    return alive


def test_a_terminated_launcher_takes_its_ranks_down():
    """SIGTERM to the launcher (a harness limit, `timeout -k`, Ctrl-C) must not orphan rank processes that hold GPUs:
    the parent terminates exactly the Popen objects it started and exits 128 + signal."""
    import signal
    worker = ("import os, sys, time; print('PID', os.getpid(), flush=True); time.sleep(120)")
    driver = ("import sys; sys.path.insert(0, %r); from big_dreamer_amd import launch; "
              "sys.exit(launch.spawn_ranks([sys.executable, '-c', %r], 2, timeout=300))" % (ROOT, worker))
    p = subprocess.Popen([sys.executable, "-c", driver], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    pids = []
    line = p.stdout.readline()                    # rank 0's stdout is passed through
    assert line.startswith("PID"), line
    pids.append(int(line.split()[1]))
    time.sleep(1.0)                               # rank 1 has started by now as well (its PID goes to stderr)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    pids += [int(l.split()[1]) for l in err.splitlines() if l.startswith("PID")]
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    assert len(pids) == 2, (pids, err[-2000:])
    time.sleep(0.2)
    assert _pids_alive(pids) == [], "rank processes survived their launcher"


def test_launch_timeout_terminates_hung_ranks():
    worker = "import time; time.sleep(120)"
    driver = ("import sys; sys.path.insert(0, %r); from big_dreamer_amd import launch; "
              "sys.exit(launch.spawn_ranks([sys.executable, '-c', %r], 2, timeout=2))" % (ROOT, worker))
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 124 and time.time() - t0 < 60
    assert "terminating them" in out.stderr
