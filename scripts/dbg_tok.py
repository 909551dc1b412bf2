import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la
from oracle import oracle as O
O.lib()
b = la.Batch(0)
d = "tests/golden/inputs"
for name in ("coast.jpg", "sunrise.jpg", "firefox-gray.jpg", "ferry_sunset.jpg", "large-sunrise.jpg"):
    data = open(os.path.join(d, name), "rb").read()
    info = O.jpeg_info(data)
    for S in (0, 256, 1024, 4096, 16384):
        b.set_subsequence(S, 0)
        for c in range(info["ncomp"]):
            got = b.decode_jpeg_coefs(data, c).astype(int); exp = O.jpeg_decode_coefs(data, c).astype(int)
            if np.array_equal(got, exp):
                continue
            bad = np.argwhere(got != exp)
            blocks = sorted({(int(y), int(x)) for y, x, _ in bad})
            pos = sorted({int(k) for _, _, k in bad})
            print(name, "S", S, "comp", c, "shape", got.shape, "bad blocks", len(blocks), "of", got.shape[0] * got.shape[1], "first", blocks[:6], "positions", pos[:12])
            y, x = blocks[0]
            print("   got", got[y, x][:16].tolist(), "\n   exp", exp[y, x][:16].tolist())
    print(name, "checked")
