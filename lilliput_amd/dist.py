"""Multi-GPU plumbing of the batched path: one process per GPU, images sharded by rank, no data-path collective.

The images of a batch are independent, so N ranks simply take N disjoint shards; torch.distributed is
used only for the barrier around the timed region and for the max-over-ranks of the elapsed time (backend "nccl" = RCCL
on the GPUs, "gloo" in the CPU tests). bench.py and tests/test_multi_rank.py both go through this module.
"""
import os
import time


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend, device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist

    def shard(self, n_items):
        """Indices of the global item list this rank owns: contiguous, sizes differing by at most one."""
        base, extra = divmod(n_items, self.world)
        lo = self.rank * base + min(self.rank, extra)
        return range(lo, lo + base + (1 if self.rank < extra else 0))

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                import torch

                torch.cuda.synchronize()

    def reduce(self, value, op="max"):
        """max / sum of a python float over the ranks (every rank gets the result)."""
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, step, steps, warmup):
        """The bench contract: `warmup` untimed steps, then exactly `steps` steps bracketed by barrier (+ device
        synchronisation) on both sides; returns the MAX elapsed seconds over the ranks."""
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.time()
        for _ in range(steps):
            step()
        self.barrier()
        return self.reduce(time.time() - t0, "max")

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
