"""Data-parallel rules of the training step: the replay-batch dimension is sharded over
ranks (one process per GPU), weights are replicated, and three gradient all-reduces per step (world model,
actor, critic) run over RCCL/xGMI through ``torch.distributed`` (backend "nccl" is RCCL on ROCm).

Exactness with respect to one big batch:
  * every loss is a mean over the GLOBAL batch, so kernels scale local gradients by 1/global_count and the
    all-reduce is a plain SUM -- no extra division pass over the gradient buffer;
  * the balanced-KL free-nats clamp (src/dreamer.py:134-141) acts on a batch MEAN, which is not shard-additive:
    the local KL sums are all-reduced (one float) BEFORE the clamp decision;
  * clip_grad_norm_ runs on the reduced gradients, identically on every rank.
Small messages (0.67-4.9 MB) are latency-bound on xGMI, hence one flat fp32 bucket per optimiser.

This module holds only host logic (no kernels), so it is also exercised on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class DataParallel:
    def __init__(self, world_size: int = 1, rank: int = 0, process_group=None, groups: Optional[dict] = None):
        """`groups` may map "model" / "actor" / "critic" to separate process groups.  The engine's cross-step pipeline
        issues the three gradient all-reduces from three HIP streams; on ONE communicator they would execute in issue
        order (model k, actor k, critic k, model k+1, ...), so the world model's all-reduce of step k+1 would wait
        for the low-priority critic update of step k.  One RCCL communicator per optimiser removes that coupling."""
        assert world_size >= 1 and 0 <= rank < world_size
        self.world_size, self.rank, self.pg = world_size, rank, process_group
        self.groups = dict(groups or {})
        # Rehearsal on one GPU: issue the collectives even with a single rank (a sum over one rank is the identity), so the
        # stream ordering around the RCCL calls can be checked where only one device is at hand (BD_FORCE_DP=1).
        self.force = False

    @staticmethod
    def make_phase_groups(backend: Optional[str] = None, parent=None) -> dict:
        """Three process groups over the ranks of `parent` (default: all ranks).  Collective: every rank of the default group
        calls it, once, in the same order (torch.distributed.new_group)."""
        ranks = torch.distributed.get_process_group_ranks(parent) if parent is not None else None
        return {k: torch.distributed.new_group(ranks=ranks, backend=backend) for k in ("model", "actor", "critic")}

    # ---- scaling --------------------------------------------------------------------------------------
    def mean_grad_scale(self, local_count: int) -> float:
        """d(mean over the global batch)/d(sum over local elements): 1 / (local_count * world_size)."""
        return 1.0 / (local_count * self.world_size)

    # ---- collectives -----------------------------------------------------------------------------------
    def allreduce_sum_(self, t: torch.Tensor, key: Optional[str] = None) -> torch.Tensor:
        if self.world_size > 1 or self.force:
            pg = self.groups.get(key, self.pg)
            if t.is_cuda and torch.distributed.get_backend(pg) == "gloo":
                # test transport only (several ranks sharing one GPU): stage through the host
                h = t.detach().cpu()
                torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.SUM, group=pg)
                t.copy_(h)
            else:
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=pg)
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size > 1:
            torch.distributed.broadcast(t, src=src, group=self.pg)
        return t

    # ---- sharding ----------------------------------------------------------------------------------------
    def shard_batch(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Columns [r*B/W, (r+1)*B/W) of a time-major global batch (dim 1 of every (L, B, ...) tensor)."""
        out = {}
        for k, v in batch.items():
            B = v.shape[1]
            assert B % self.world_size == 0, f"global batch {B} not divisible by world size {self.world_size}"
            b = B // self.world_size
            out[k] = v[:, self.rank * b:(self.rank + 1) * b].contiguous()
        return out

    def shard_rows(self, x: torch.Tensor, T: int, B: int) -> torch.Tensor:
        """Rows of a (..., T*B, F) tensor (row = t*B + b, the imagination's flattened start states) that belong
        to this rank's batch columns."""
        b = B // self.world_size
        lead = x.shape[:-2]
        v = x.reshape(*lead, T, B, x.shape[-1])[..., :, self.rank * b:(self.rank + 1) * b, :]
        return v.reshape(*lead, T * b, x.shape[-1]).contiguous()
