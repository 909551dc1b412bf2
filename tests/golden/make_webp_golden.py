"""Records the reference library's answers (oracle/_ref/librefwebp.so = libwebp 1.5.0 + mux + demux of /root/reference/deps) for the
WebP decoder cases of tests/test_webp.py -> tests/golden/webp_golden.json. Run in the build container (needs the reference mount)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
import test_webp as T  # noqa: E402

cases = dict(T.fixtures())
cases.update(T.mutations(7, 400))
gold = {"decode": {n: T.ref_digest(oracle, d) for n, d in cases.items()}}
json.dump(gold, open(T.GOLD, "w"), indent=0, sort_keys=True)
print(len(gold["decode"]), "cases;", sum(1 for v in gold["decode"].values() if v == "none"), "rejected by the reference")
