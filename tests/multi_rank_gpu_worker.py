"""Worker of tests/test_multi_rank.py::test_two_ranks_share_one_gpu_through_the_queue (GPU): two processes (gloo for the queue
state, both on GPU 0) push real images through their own Batch via lilliput_amd.dist.transform_queue."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lilliput_amd.dist import Ranks, transform_queue  # noqa: E402


def main():
    out_dir = sys.argv[1]
    import lilliput_amd as la

    r = Ranks(backend="gloo")
    fix = os.path.join(ROOT, "tests", "golden", "inputs")
    names = sorted(n for n in os.listdir(fix) if n.endswith(".jpg"))
    sources = [open(os.path.join(fix, n), "rb").read() for n in names] * 3
    b = la.Batch(0)
    got = transform_queue(r, b, sources, 64, 64, chunk=2, slice_s=0.02, quality=85)
    b.close()
    with open(os.path.join(out_dir, "gpu_rank%d.json" % r.rank), "w") as f:
        json.dump({"n": len(sources), "items": {str(i): [v.status, hashlib.sha256(v.data).hexdigest()] for i, v in got.items()}}, f)
    r.close()


if __name__ == "__main__":
    main()
