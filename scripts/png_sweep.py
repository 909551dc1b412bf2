"""Ad-hoc sweep: larger generated PNGs (several 64-row bands, long rows, Adam7) through the device decoder against libpng (_ref)."""
import os, sys, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import png_cases, test_png

L = la.lib()
rnd = random.Random(int(sys.argv[1])); n = int(sys.argv[2]); bad = 0; t0 = time.time()
for it in range(n):
    ct = rnd.choice([0, 2, 3, 4, 6]); depth = rnd.choice({0: (1, 2, 4, 8, 16), 2: (8, 16), 3: (1, 2, 4, 8), 4: (8, 16), 6: (8, 16)}[ct])
    w, h = rnd.randrange(1, 260), rnd.randrange(1, 330)
    data, _ = png_cases.make_png(w, h, ct, depth, rnd, interlace=rnd.random() < 0.4, smooth=rnd.random() < 0.5, idat_split=rnd.choice([0, 0, 3, 17]))
    ref, mine = O.ref_png_decode(data), test_png._decode(L, data)
    if ref is None or mine is None or not np.array_equal(ref, mine):
        bad += 1; print("MISMATCH", (w, h, ct, depth), ref is None, mine is None)
print("checked", n, "bad", bad, "%.1fs" % (time.time() - t0))
