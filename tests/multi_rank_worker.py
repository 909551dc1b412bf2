"""Worker of tests/test_multi_rank.py: run under torch.distributed.run with the gloo backend (CPU)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lilliput_amd.dist import Ranks  # noqa: E402


def main():
    out_dir, n_items = sys.argv[1], int(sys.argv[2])
    r = Ranks(backend="gloo")
    mine = list(r.shard(n_items))
    # the "work": a digest per owned item (stands for one image through the device path)
    digests = {i: hashlib.sha256(b"item-%d" % i).hexdigest()[:8] for i in mine}
    calls = []

    def step():
        calls.append(time.time())
        time.sleep(0.05 * (r.rank + 1))  # rank 1 is the slow one: the reported time must be ITS time

    elapsed = r.timed(step, steps=3, warmup=1)
    total = r.reduce(len(mine), "sum")
    with open(os.path.join(out_dir, "rank%d.json" % r.rank), "w") as f:
        json.dump({"rank": r.rank, "world": r.world, "items": mine, "digests": digests, "elapsed": elapsed, "steps_run": len(calls), "total": total}, f)
    r.close()


if __name__ == "__main__":
    main()
