"""Writes tests/golden/gif_golden.json from the REFERENCE's giflib 5.2.2 + the restated compositing (oracle/_ref/librefgif.so):
"host" = digest of per-frame metadata + colour indices + animation info, "canvas" = digest of the composited BGRA canvases.
Run in the build container (needs /root/reference to build oracle/_ref)."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gif_cases
from oracle import oracle as O

assert O.ref_gif() is not None, "build oracle/_ref first (make -C oracle)"
cases = dict(gif_cases.fixtures()); cases.update(gif_cases.hand_cases()); cases.update(gif_cases.fuzz_cases(41, 500))
host, canvas = {}, {}
for k, v in cases.items():
    r = O.ref_gif_frames(v)
    if r is None:
        host[k] = canvas[k] = "none"
        continue
    h = hashlib.sha1(); c = hashlib.sha1()
    for cv, meta, idx in r[2]:
        h.update(np.array(meta[:10], dtype=np.int32).tobytes()); h.update(idx.tobytes()); c.update(cv.tobytes())
    host[k] = "%dx%d:%d:%d:%s:%s" % (r[0], r[1], len(r[2]), r[3], ",".join(map(str, O.ref_gif_info(v))), h.hexdigest()[:16])
    canvas[k] = "%dx%d:%d:%d:%s" % (r[0], r[1], len(r[2]), r[3], c.hexdigest()[:16])
json.dump({"host": host, "canvas": canvas}, open(os.path.join(ROOT, "tests", "golden", "gif_golden.json"), "w"), indent=0, sort_keys=True)
print(len(host), "cases")
