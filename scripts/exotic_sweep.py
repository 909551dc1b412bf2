"""Ad-hoc sweep (not part of the test suite): JPEGs written by the reference's own libjpeg-turbo with random component counts, sampling
factors, scan structures and colour spaces -> device decode + Transform against the oracle and the reference library.
usage: python scripts/exotic_sweep.py <seed> <files>   (needs oracle/_ref/libref.so)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lilliput_amd as la
from oracle import oracle as O

R = O.ref()
assert R is not None, "reference library not built"
R.ref_jpeg_encode_ex.restype = C.c_long
rng = np.random.default_rng(int(sys.argv[1]))
n_files = int(sys.argv[2])
SAMP3 = [(1, 1, 1, 1, 1, 1), (2, 1, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1), (1, 2, 1, 1, 1, 1), (4, 1, 1, 1, 1, 1), (4, 2, 1, 1, 1, 1), (2, 2, 2, 1, 1, 1), (1, 1, 2, 2, 2, 2),
         (3, 1, 1, 1, 1, 1), (1, 4, 1, 1, 1, 1), (2, 4, 1, 1, 1, 1), (2, 2, 1, 2, 2, 1), (1, 1, 1, 1, 2, 2), (2, 1, 1, 2, 1, 1), (1, 3, 1, 1, 1, 1), (4, 1, 2, 1, 1, 1)]
SAMP4 = SAMP3[:4]
OPTS = [0, 1, 2, 3, 4, 8, 16, 17, 33, 16 + 64, 16 + 33]
L = la.lib()
b = la.Batch(0)
bad = tot = 0
t0 = time.time()
files = []
for it in range(n_files):
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
    nc = int(rng.choice([1, 3, 3, 3, 4, 4]))
    y, x = np.mgrid[0:h, 0:w]
    px = np.clip(np.stack([128 + 95 * np.sin(x / rng.uniform(4, 30) + k) + 35 * np.cos(y / rng.uniform(4, 30) - k) for k in range(4)], -1) + rng.normal(0, rng.uniform(0, 20), (h, w, 4)), 0, 255).astype(np.uint8)
    px = np.ascontiguousarray(px[:, :, 0] if nc == 1 else px[:, :, :nc])
    samp = (1, 1, 1, 1, 1, 1) if nc == 1 else SAMP3[int(rng.integers(len(SAMP3)))] if nc == 3 else SAMP4[int(rng.integers(4))]
    opt = OPTS[int(rng.integers(len(OPTS)))]
    if nc == 1 and opt & (64 | 32):
        opt &= ~(64 | 32)
    if nc == 4 and opt & 64:
        opt &= ~64
    mode = int(rng.integers(0, 4))
    buf = np.zeros(h * w * 4 + 65536, np.uint8)
    n = R.ref_jpeg_encode_ex(px.ctypes.data_as(C.c_void_p), w, h, nc, mode, (C.c_int * 6)(*samp), int(rng.integers(5, 100)), 1, int(rng.choice([0, 0, 1, 7])), opt,
                             buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    if n <= 0:
        continue
    files.append(((it, h, w, nc, samp, opt, mode), buf[:n].tobytes()))
for dev_mode in (0, 1):
    L.lilliput_hip_set_progressive_entropy(dev_mode)
    for desc, data in files:
        tot += 1
        ref = O.ref_jpeg_decode(data)
        exp = O.jpeg_decode(data)
        got, _ = b.decode_jpeg(data)
        if not (got.shape == exp.shape == ref.shape and np.array_equal(got, exp) and np.array_equal(exp, ref)):
            bad += 1
            print("DECODE MISMATCH", desc, dev_mode, "oracle==ref", exp.shape == ref.shape and np.array_equal(exp, ref))
    res = b.transform([f[1] for f in files], 40, 33, quality=80)
    for (desc, data), r in zip(files, res):
        tot += 1
        e = O.transform_jpeg_thumbnail(data, 40, 33, 80)
        if r.status != 0 or r.data != e:
            a, c = (O.jpeg_decode(r.data) if r.status == 0 else None), O.jpeg_decode(e)
            if a is None or a.shape != c.shape or np.abs(a.astype(int) - c.astype(int)).max() > 8:
                bad += 1
                print("TRANSFORM MISMATCH", desc, dev_mode, r.status)
L.lilliput_hip_set_progressive_entropy(0)
print("files", len(files), "checked", tot, "bad", bad, "in %.1fs" % (time.time() - t0))
