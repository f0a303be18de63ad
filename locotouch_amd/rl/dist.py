"""One-process-per-GPU data parallelism over RCCL (backend "nccl" on ROCm) / gloo (CPU tests).

The reference is single-GPU; the data-parallel design is new: envs shard across ranks, every rank
holds a full policy replica, and the only collectives on the training path are
  1. one flat gradient all-reduce (mean) per optimizer step - 687 513 fp32 = 2.75 MB, latency-bound on xGMI,
  2. a 1-float KL all-reduce before the adaptive-LR decision (otherwise ranks would diverge),
  3. a 3-float advantage-moment all-reduce once per iteration,
  4. a parameter broadcast at start / after load.
No env state ever crosses GPUs.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


class Dist:
    def __init__(self, rank: int = 0, world_size: int = 1, local_rank: int = 0, initialized_here: bool = False, backend: str = "nccl"):
        self.rank, self.world_size, self.local_rank = rank, world_size, local_rank
        self._initialized_here = initialized_here
        self.backend = backend

    @staticmethod
    def from_env(backend: str | None = None) -> "Dist":
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return Dist()
        rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:  # LT_DIST_BACKEND=gloo: rehearsal of the multi-rank path with several ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("LT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        here = False
        if not td.is_initialized():
            td.init_process_group(backend=backend, rank=rank, world_size=world)
            here = True
        return Dist(rank, world, local, here, backend)

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def barrier(self) -> None:
        if self.world_size > 1:
            if self.backend == "nccl":
                td.barrier(device_ids=[self.local_rank])  # (no device guess from the rank: this process's GPU is LOCAL_RANK)
            else:
                td.barrier()

    def _collective(self, t: torch.Tensor, fn) -> torch.Tensor:
        if self.backend == "gloo" and t.is_cuda:  # gloo rehearsal on a GPU box: stage through the host
            h = t.detach().cpu()
            fn(h)
            t.copy_(h)
        else:
            fn(t)
        return t

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            self._collective(t, lambda x: td.all_reduce(x, op=td.ReduceOp.SUM))
        return t

    def all_reduce_mean_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            self._collective(t, lambda x: td.all_reduce(x, op=td.ReduceOp.SUM))
            t.div_(self.world_size)
        return t

    def all_reduce_mean_begin(self, t: torch.Tensor):
        """Start `all_reduce_mean_` and return a handle for `all_reduce_mean_end` (RCCL: asynchronous - its latency hides under
        whatever the caller enqueues before the end call; gloo rehearsal / one rank: done here)."""
        if self.world_size > 1 and self.backend == "nccl":
            return (t, td.all_reduce(t, op=td.ReduceOp.SUM, async_op=True))
        self.all_reduce_mean_(t)
        return (t, None)

    def all_reduce_mean_end(self, handle) -> torch.Tensor:
        t, work = handle
        if work is not None:
            work.wait()
            t.div_(self.world_size)
        return t

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            self._collective(t, lambda x: td.all_reduce(x, op=td.ReduceOp.MAX))
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size > 1:
            self._collective(t, lambda x: td.broadcast(x, src=src))
        return t

    def shutdown(self) -> None:
        if self._initialized_here and td.is_initialized():
            td.destroy_process_group()
