"""Ad-hoc sweep: random JPEGs with random EXIF orientation through the batch Transform (Fit / Resize / NoResize, normalize on/off,
integer and fractional scales) against the oracle's restatement of ops.go."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_gpu_sweep as T, test_gpu_parity as P

b = la.Batch(0)
rng = np.random.default_rng(int(sys.argv[1])); bad = tot = 0; t0 = time.time()
for rep in range(int(sys.argv[2])):
    cases = [(i, d, P._with_exif_orientation(data, int(rng.integers(1, 9)))) for i, d, data in T._cases(int(rng.integers(1 << 30)), 40)]
    for method, om in ((la.ImageOpsFit, O.FIT), (la.ImageOpsResize, O.RESIZE), (la.ImageOpsNoResize, O.NO_RESIZE)):
        for normalize in (False, True):
            if rng.random() < 0.5:
                tw, th = int(rng.integers(1, 300)), int(rng.integers(1, 300))
            else:
                tw = th = int(rng.choice([8, 16, 32, 64]))
            res = b.transform([c[2] for c in cases], tw, th, method=method, normalize=normalize, quality=80, dst_cap=4 << 20)
            for (i, desc, data), r in zip(cases, res):
                tot += 1
                info = O.jpeg_info(data)
                exp = O.jpeg_encode(O.transform_static(O.jpeg_decode(data), info["orientation"], tw, th, om, normalize), 80)
                ok = r.status == 0 and r.data == exp
                if not ok and r.status == 0:
                    a, c = O.jpeg_decode(r.data), O.jpeg_decode(exp)
                    ok = a.shape == c.shape and np.abs(a.astype(int) - c.astype(int)).max() <= 8
                if not ok:
                    bad += 1
                    if bad < 12: print("MISMATCH", desc, "orient", info["orientation"], (tw, th), "method", method, "norm", normalize, "status", r.status)
print("checked", tot, "bad", bad, "%.1fs" % (time.time() - t0))
