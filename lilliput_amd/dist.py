"""Multi-GPU plumbing of the batched path when it runs as ONE PROCESS PER GPU: images sharded by rank, no data-path collective.

The images of a batch are independent, so N ranks simply take N disjoint shards; what the ranks need from each
other is a barrier around the timed region, the max-over-ranks of the elapsed time and a few gathered counters. Three ways to get it:
  * backend "nccl" (the default under torchrun, which is how the driver launches N > 1): torch.distributed over RCCL;
  * backend "gloo": torch.distributed on the CPU (the CPU tests);
  * backend "file": NO PyTorch at all -- the ranks of one node meet in a directory (one small file per rank and phase). What
    `bench.py --gpus N --ranks` uses for the processes it spawns itself, and what "nccl" falls back to when torch or RCCL cannot be
    brought up (say so: Ranks.backend).
(One process driving every GPU of the node -- lilliput_hip_node_*, what a cgo service links and what `bench.py --gpus N` runs by
default -- needs none of this: its chunk queue is a host atomic, lp_batch.cpp.)
bench.py and tests/test_multi_rank.py both go through this module.
"""
import os
import struct
import time


class _FileGroup:
    """Barrier / all-gather of a few numbers among the ranks of ONE node through a shared directory: rank r publishes
    <dir>/<phase>.<r> (written under another name, then renamed: a reader never sees half a file) and polls for the others'."""

    def __init__(self, rank, world, path, timeout_s=600.0):
        self.rank, self.world, self.path, self.timeout_s, self.phase = rank, world, path, timeout_s, 0
        os.makedirs(path, exist_ok=True)

    def all_gather(self, payload):
        ph, self.phase = self.phase, self.phase + 1
        mine = os.path.join(self.path, "%d.%d" % (ph, self.rank))
        with open(mine + ".tmp", "wb") as f:
            f.write(payload)
        os.replace(mine + ".tmp", mine)
        out, t_end, nap = [], time.time() + self.timeout_s, 0.0002
        for r in range(self.world):
            p = os.path.join(self.path, "%d.%d" % (ph, r))
            while True:
                try:
                    with open(p, "rb") as f:
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    if time.time() > t_end:
                        raise TimeoutError("rank %d never reached phase %d of %s" % (r, ph, self.path))
                    time.sleep(nap)
                    nap = min(nap * 1.5, 0.005)
        if ph >= 2:  # everybody has read phase ph - 2 (they published ph - 1 after reading it): this rank's old file can go
            try:
                os.remove(os.path.join(self.path, "%d.%d" % (ph - 2, self.rank)))
            except OSError:
                pass
        return out

    def close(self):
        # leaving: a rank may only remove what nobody will read again. Everybody says "bye" and, having read everybody's, "done";
        # rank 0 waits for the "done"s and clears the directory.
        self.all_gather(b"bye")
        if self.rank != 0:
            with open(os.path.join(self.path, "done.%d.tmp" % self.rank), "wb") as f:
                f.write(b"1")
            os.replace(os.path.join(self.path, "done.%d.tmp" % self.rank), os.path.join(self.path, "done.%d" % self.rank))
            return
        t_end = time.time() + self.timeout_s
        while any(not os.path.exists(os.path.join(self.path, "done.%d" % r)) for r in range(1, self.world)) and time.time() < t_end:
            time.sleep(0.001)
        for name in os.listdir(self.path):
            try:
                os.remove(os.path.join(self.path, name))
            except OSError:
                pass
        try:
            os.rmdir(self.path)
        except OSError:
            pass


def _rendezvous_dir():
    d = os.environ.get("LILLIPUT_BENCH_RDV")
    if d:
        return d
    # under torchrun every worker has the same parent (the elastic agent): its pid, its START TIME (a recycled pid is not the same run:
    # phase files a crashed earlier run left behind must not be read as this run's) and the master port name this run
    ppid, born = os.getppid(), "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            born = f.read().rsplit(")", 1)[1].split()[19]   # field 22: starttime in clock ticks since boot
    except (OSError, IndexError):
        pass
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "lilliput_rdv_%s_%d_%s" % (os.environ.get("MASTER_PORT", "0"), ppid, born))


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        self.files = None
        self.backend = "none"
        # under torchrun (RANK set) the process group is brought up even for ONE rank, so that the RCCL path of the multi-GPU runs
        # (backend "nccl": barrier, all-reduce of the elapsed time, all-gather of the counters) is the path a 1-GPU run takes too
        if self.world > 1 or (os.environ.get("RANK") is not None and os.environ.get("MASTER_ADDR")):
            backend = backend or "nccl"
            if backend != "file" and self.world > 1:
                # The choice between torch.distributed and the file rendezvous is made by ALL ranks together, before anybody enters
                # init_process_group: a rank that cannot import torch (or sees no GPU for "nccl") would otherwise meet through files while
                # the others wait for it in the process group until its timeout. Every rank says whether it can; one "no" sends all to files.
                can = True
                try:
                    import torch

                    can = backend != "nccl" or torch.cuda.is_available()
                except Exception:
                    can = False
                pre = _FileGroup(self.rank, self.world, _rendezvous_dir() + "_pre")
                votes = pre.all_gather(b"1" if can else b"0")
                pre.close()
                if any(v != b"1" for v in votes):
                    import sys

                    print("[lilliput_amd.dist] rank %d: backend %s is not available on every rank (%s): all ranks meet through files" % (
                        self.rank, backend, "".join(v.decode() for v in votes)), file=sys.stderr, flush=True)
                    backend = "file"
            if backend != "file":
                try:
                    import torch
                    import torch.distributed as dist

                    if backend == "nccl":
                        torch.cuda.set_device(self.local_rank)
                        self.device = torch.device("cuda", self.local_rank)
                        dist.init_process_group(backend, device_id=self.device)
                    else:
                        self.device = torch.device("cpu")
                        dist.init_process_group(backend)
                    self.dist = dist
                    self.backend = backend
                except Exception as e:  # no torch, no RCCL, a rendezvous that does not come up: the ranks still only need a barrier
                    import sys

                    print("[lilliput_amd.dist] rank %d: backend %s unavailable (%r): meeting the other ranks through files instead" % (self.rank, backend, e), file=sys.stderr, flush=True)
                    self.dist, self.device, backend = None, None, "file"
            if backend == "file":
                self.files = _FileGroup(self.rank, self.world, _rendezvous_dir())
                self.backend = "file"

    def shard(self, n_items):
        """Indices of the global item list this rank owns: contiguous, sizes differing by at most one."""
        base, extra = divmod(n_items, self.world)
        lo = self.rank * base + min(self.rank, extra)
        return range(lo, lo + base + (1 if self.rank < extra else 0))

    def barrier(self):
        if self.files is not None:
            self.files.all_gather(b"b")  # (the product's calls are synchronous: when transform returns the device is idle)
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                import torch

                torch.cuda.synchronize()

    def reduce(self, value, op="max"):
        """max / sum of a python float over the ranks (every rank gets the result)."""
        if self.files is not None:
            vals = [struct.unpack("<d", b)[0] for b in self.files.all_gather(struct.pack("<d", float(value)))]
            return max(vals) if op == "max" else sum(vals)
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
        if self.files is not None:
            vals = list(values)
            return [list(struct.unpack("<%dq" % len(vals), b)) for b in self.files.all_gather(struct.pack("<%dq" % len(vals), *[int(v) for v in vals]))]
        if self.dist is None:
            return [list(values)]
        import torch

        mine = torch.tensor(list(values), dtype=torch.int64, device=self.device)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[int(v) for v in t.tolist()] for t in out]

    def close(self):
        if self.files is not None:
            self.files.close()
            self.files = None
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
