"""Worker of tests/test_launch_cpu.py: one rank started by big_dreamer_amd.launch.spawn_ranks (gloo, CPU)."""
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("LAUNCH_FAIL_RANK") == str(rank):
        sys.exit(3)                       # dies before the rendezvous: the launcher must take the others down
    if "LAUNCH_FAIL_RANK" in os.environ:
        time.sleep(60)                    # would out-live the test if the launcher did not terminate it
        sys.exit(0)
    dist.init_process_group("gloo")       # MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE from the launcher
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print(f"rank {rank} stdout local_rank={os.environ['LOCAL_RANK']}", flush=True)
    if rank == 0:
        print(f"LAUNCH_OK world={world} sum={t.item():.0f} addr={os.environ['MASTER_ADDR']}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
