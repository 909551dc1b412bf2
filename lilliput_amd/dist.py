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
        # under torchrun (RANK set) the process group is brought up even for ONE rank, so that the RCCL path of the multi-GPU runs
        # (backend "nccl": barrier, all-reduce of the elapsed time, all-gather of the counters) is the path a 1-GPU run takes too
        if self.world > 1 or (os.environ.get("RANK") is not None and os.environ.get("MASTER_ADDR")):
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

    def all_gather_ints(self, values):
        """values: list of python ints of this rank -> list (one entry per rank) of lists."""
        if self.dist is None:
            return [list(values)]
        import torch

        mine = torch.tensor(list(values), dtype=torch.int64, device=self.device)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[int(v) for v in t.tolist()] for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class WorkQueue:
    """Chunk queue of a multi-GPU job with work stealing, synchronised with one tiny collective per epoch.

    Chunks 0..n-1 are units of independent images. Rank r starts with a contiguous range; the ranks advance
    in epochs: in every epoch each rank works through its range for one time slice (a fast rank finishes several chunks, a
    slow one at least one) and then calls sync(), an all-gather of the ranks' (next, end) pairs -- 16 bytes per rank,
    latency bound; no image data ever crosses the interconnect. When a rank has run dry, the richest rank hands it the upper
    half of what it still owns. Every rank evaluates the same deterministic rule on the same gathered state, so ownership
    never needs a second message. Use `q.run(process, slice_s)`.
    """

    def __init__(self, ranks, n_chunks):
        self.r = ranks
        sh = ranks.shard(n_chunks)
        self.next, self.end = sh.start, sh.stop
        self.state = None
        self.sync()

    def take(self):
        if self.next >= self.end:
            return None
        c = self.next
        self.next += 1
        return c

    def busy(self):
        return any(e > n for n, e in self.state)

    def run(self, process, slice_s=0.05):
        """Process every chunk of the job exactly once across the ranks; returns the chunks this rank handled."""
        mine = []
        while self.busy():
            t_end = time.time() + slice_s
            while True:
                c = self.take()
                if c is None:
                    break
                process(c)
                mine.append(c)
                if time.time() >= t_end:
                    break
            self.sync()
        return mine

    def sync(self):
        state = [tuple(v) for v in self.r.all_gather_ints([self.next, self.end])]
        # deterministic rebalancing, identical on every rank: repeatedly give the first idle rank the upper half of the richest range
        state = [list(v) for v in state]
        while True:
            idle = [i for i, (n, e) in enumerate(state) if e <= n]
            rich = max(range(len(state)), key=lambda i: (state[i][1] - state[i][0], -i))
            left = state[rich][1] - state[rich][0]
            if not idle or left < 2:
                break
            give = left // 2
            thief = idle[0]
            state[thief] = [state[rich][1] - give, state[rich][1]]
            state[rich][1] -= give
        self.next, self.end = state[self.r.rank]
        self.state = [tuple(v) for v in state]


def transform_queue(ranks, batch, sources, width, height, chunk=64, slice_s=0.05, **kw):
    """Mixed-size firehose (BASELINE configs[4] shape): `sources` is the same list on every rank (host memory); chunks of
    `chunk` images are claimed through a WorkQueue and pushed through this rank's Batch (upload + device path + download per
    chunk). Returns {global index: BatchResult} for the images this rank processed."""
    n_chunks = (len(sources) + chunk - 1) // chunk
    out = {}

    def process(c):
        lo = c * chunk
        for i, r in enumerate(batch.transform(sources[lo:lo + chunk], width, height, **kw)):
            out[lo + i] = r

    WorkQueue(ranks, n_chunks).run(process, slice_s)
    return out
