"""Round 5, one-off: the randomised GPU sweeps of tests/test_gpu_sweep.py with other seeds and more cases (integer scales through every
resample kernel; mixed batches at four target sizes incl. big sources), against the oracle. Prints a summary; exit code 1 on any difference."""
import io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_gpu_sweep as T
import test_gpu_parity as P
from PIL import Image

batch = la.Batch(0)
bad = []
n = 0
for seed in (int(s) for s in (sys.argv[1:] or ["1", "2", "3"])):
    rng = np.random.default_rng(seed)
    for it in range(160):
        s = int(rng.choice([2, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 24, 31, 32, 33, 34, 40, 64, 66]))
        lim = max(2, min(90, 1400 // s))
        tw, th = int(rng.integers(1, lim)), int(rng.integers(1, lim))
        extra = int(rng.integers(0, 40))
        wide = rng.random() < 0.5
        ow, oh = (tw * s + (extra if wide else 0), th * s + (0 if wide else extra))
        o = int(rng.integers(1, 9))
        w, h = (oh, ow) if o >= 5 else (ow, oh)
        gray = rng.random() < 0.12
        img = T._image(rng, h, w, gray)
        kw = {"quality": int(rng.choice([40, 85, 92, 100]))}
        if not gray:
            kw["subsampling"] = int(rng.choice([0, 1, 2, 2]))
        if rng.random() < 0.2:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        d = P._with_exif_orientation(buf.getvalue(), o)
        for norm in (False, True):
            r = batch.transform([d], tw, th, normalize=norm, quality=85)[0]
            frame = O.transform_static(O.jpeg_decode(d), o, tw, th, O.FIT, norm)
            n += 1
            if r.status != 0 or (r.width, r.height) != (frame.shape[1], frame.shape[0]) or r.data != O.jpeg_encode(frame, 85):
                bad.append(("int", seed, it, s, (w, h), (tw, th), o, norm, gray, kw, r.status))
    # mixed batches (every route in one launch), small and big sources
    for big in (False, True):
        cases = list(T._cases(1000 + seed, 64 if big else 128, big=big))
        for tw, th, q in ((64, 64, 85), (37, 91, 70), (128, 128, 90), (16, 16, 50), (256, 256, 85)):
            res = batch.transform([c[2] for c in cases], tw, th, quality=q)
            for (i, desc, data), r in zip(cases, res):
                n += 1
                if r.status != 0:
                    bad.append(("mix", seed, i, desc, (tw, th), "status %d" % r.status)); continue
                exp = O.transform_jpeg_thumbnail(data, tw, th, q)
                if r.data != exp:
                    a, b = O.jpeg_decode(r.data), O.jpeg_decode(exp)
                    if a.shape != b.shape or np.abs(a.astype(int) - b.astype(int)).max() > 8:
                        bad.append(("mix", seed, i, desc, (tw, th), a.shape, b.shape))
    print("seed", seed, "cases so far", n, "bad", len(bad), flush=True)
print("TOTAL", n, "bad", len(bad))
for b in bad[:20]: print(b)
sys.exit(1 if bad else 0)
