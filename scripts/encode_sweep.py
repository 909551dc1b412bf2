"""Ad-hoc randomised sweep of the JPEG encoder through the C ABI: bitstream identical to the oracle's libjpeg restatement."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_gpu_parity as P

L = la.lib()
rng = np.random.default_rng(int(sys.argv[1]))
n = int(sys.argv[2]); bad = 0; t0 = time.time()
for it in range(n):
    cn = int(rng.choice([1, 3, 3, 4]))
    h, w = (int(rng.integers(1, 30)), int(rng.integers(1, 30))) if rng.random() < 0.3 else (int(rng.integers(1, 500)), int(rng.integers(1, 500)))
    kind = rng.random()
    if kind < 0.4: px = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
    elif kind < 0.6: px = np.full((h, w, cn), int(rng.integers(0, 256)), np.uint8)
    else:
        y, x = np.mgrid[0:h, 0:w]
        px = np.clip(np.stack([128 + 100 * np.sin(x / 9.0 + c) + 30 * np.cos(y / 5.0) for c in range(cn)], -1) + rng.normal(0, 10, (h, w, cn)), 0, 255).astype(np.uint8)
    q = int(rng.choice([1, 10, 30, 50, 75, 85, 95, 100]))
    got = P._abi_encode(L, px[:, :, 0] if cn == 1 else px, q)
    exp = O.jpeg_encode(px[:, :, 0] if cn == 1 else px, q)
    if got != exp:
        bad += 1
        if bad < 10: print("ENCODE MISMATCH", (h, w, cn), q, len(got) if got else None, len(exp))
print("checked", n, "bad", bad, "in %.1fs" % (time.time() - t0))
