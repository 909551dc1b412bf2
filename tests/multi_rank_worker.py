"""Worker of tests/test_multi_rank.py: run under torch.distributed.run with the gloo backend (CPU), or as plain processes that meet
through files (LILLIPUT_BENCH_BACKEND=file: no PyTorch in the process)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lilliput_amd.dist import Ranks, WorkQueue  # noqa: E402


def main():
    out_dir, n_items = sys.argv[1], int(sys.argv[2])
    r = Ranks(backend=os.environ.get("LILLIPUT_BENCH_BACKEND", "gloo"))
    if os.environ.get("LILLIPUT_BENCH_BACKEND") == "file":
        assert r.backend == "file" and "torch" not in sys.modules
    mine = list(r.shard(n_items))
    # the "work": a digest per owned item (stands for one image through the device path)
    digests = {i: hashlib.sha256(b"item-%d" % i).hexdigest()[:8] for i in mine}
    calls = []

    def step():
        calls.append(time.time())
        time.sleep(0.05 * (r.rank + 1))  # rank 1 is the slow one: the reported time must be ITS time

    elapsed = r.timed(step, steps=3, warmup=1)
    total = r.reduce(len(mine), "sum")
    # work-stealing queue: rank 1 is 4x slower per chunk, so rank 0 must end up processing chunks of rank 1's initial range
    n_chunks = 16
    q = WorkQueue(r, n_chunks)
    epochs = [0]
    real_sync = q.sync

    def counted_sync():
        epochs[0] += 1
        real_sync()

    q.sync = counted_sync
    t0 = time.time()
    done = q.run(lambda c: time.sleep(0.02 * (4 if r.rank == 1 else 1)), slice_s=0.1)
    epochs = epochs[0]
    queue_s = time.time() - t0
    with open(os.path.join(out_dir, "rank%d.json" % r.rank), "w") as f:
        json.dump({"rank": r.rank, "world": r.world, "items": mine, "digests": digests, "elapsed": elapsed, "steps_run": len(calls), "total": total, "queue_done": done, "queue_epochs": epochs, "queue_s": queue_s, "backend": r.backend}, f)
    r.close()


if __name__ == "__main__":
    main()
