"""Ad-hoc large randomised sweep (not part of the test suite): device decode and batch Transform against the oracle."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_gpu_sweep as T

seed0, nseeds, per = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = la.Batch(0)
bad = 0; tot = 0; t0 = time.time()
for seed in range(seed0, seed0 + nseeds):
    cases = list(T._cases(seed, per, big=len(sys.argv) > 4))
    for S, Cc in ((0, 0), (256, 64), (64, 32)):
        b.set_subsequence(S, Cc)
        for i, desc, data in cases:
            tot += 1
            try:
                got, _ = b.decode_jpeg(data)
                exp = O.jpeg_decode(data)
                ok = got.shape == exp.shape and np.array_equal(got, exp)
            except Exception as e:
                ok = False
            if not ok:
                bad += 1
                if bad < 10: print("DECODE MISMATCH", seed, i, desc, S); open("gpurun_out/bad_%d_%d.jpg" % (seed, i), "wb").write(data)
    b.set_subsequence(0, 0)
    for tw, th in ((48, 48), (33, 77)):
        res = b.transform([c[2] for c in cases], tw, th, quality=80)
        for (i, desc, data), r in zip(cases, res):
            tot += 1
            exp = O.transform_jpeg_thumbnail(data, tw, th, 80)
            if r.status != 0 or r.data != exp:
                a, c = (O.jpeg_decode(r.data) if r.status == 0 else None), O.jpeg_decode(exp)
                if a is None or a.shape != c.shape or np.abs(a.astype(int) - c.astype(int)).max() > 8:
                    bad += 1
                    if bad < 10: print("TRANSFORM MISMATCH", seed, i, desc, (tw, th), r.status)
print("checked", tot, "bad", bad, "in %.1fs" % (time.time() - t0))
