"""Ad-hoc randomised sweep of opencv_mat_resize / crop / orientation through the C ABI against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_gpu_parity as P

L = la.lib()
rng = np.random.default_rng(int(sys.argv[1]))
n = int(sys.argv[2])
bad = worst = 0
t0 = time.time()
for it in range(n):
    cn = int(rng.choice([1, 3, 4]))
    sh, sw = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    if rng.random() < 0.3:  # integer scale
        k = int(rng.integers(1, 17)); dh, dw = max(1, sh // k), max(1, sw // k); sh, sw = dh * k, dw * k
    else:
        dh, dw = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    src = rng.integers(0, 256, (sh, sw, cn), dtype=np.uint8)
    crop = None
    if rng.random() < 0.4 and sh > 4 and sw > 4:
        x, y = int(rng.integers(0, sw // 2)), int(rng.integers(0, sh // 2))
        crop = (x, y, int(rng.integers(1, sw - x + 1)), int(rng.integers(1, sh - y + 1)))
    ref_src = src if crop is None else np.ascontiguousarray(src[crop[1]:crop[1] + crop[3], crop[0]:crop[0] + crop[2]])
    exp, _ = O.resize_area(ref_src, dw, dh)
    got = P._abi_resize(L, src, dw, dh, crop=crop)
    d = int(np.abs(got.astype(int) - exp.reshape(got.shape).astype(int)).max())
    worst = max(worst, d)
    if d > 1:
        bad += 1
        if bad < 10: print("RESIZE MISMATCH", (sh, sw, cn), (dh, dw), crop, d)
    if it % 4 == 0:
        o = int(rng.integers(1, 9))
        m = P.Mat(L, src); L.opencv_mat_orientation_transform(o, m.h); g2 = m.array(); m.release()
        if not np.array_equal(g2, O.orientation_transform(src, o).reshape(g2.shape)):
            bad += 1; print("ORIENT MISMATCH", (sh, sw, cn), o)
print("checked", n, "bad", bad, "worst |diff|", worst, "in %.1fs" % (time.time() - t0))
