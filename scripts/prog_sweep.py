"""Ad-hoc large randomised sweep of progressive sources and of the progressive / PNG writers (not part of the test suite).
usage: python scripts/prog_sweep.py <first seed> <seeds> <files per seed>"""
import ctypes as C
import io
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import test_progressive as TP
import test_png_output as TPO

seed0, nseeds, per = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = la.lib()
b = la.Batch(0)
bad = tot = 0
t0 = time.time()
for seed in range(seed0, seed0 + nseeds):
    cases = list(TP._cases(seed, per, lo=1, hi=900 if seed % 3 == 0 else 260))
    for mode in (0, 1):
        if mode == 1 and seed % 3 == 0:
            continue  # device lanes take long on the big files
        L.lilliput_hip_set_progressive_entropy(mode)
        for i, desc, data in cases:
            tot += 1
            exp = O.jpeg_decode(data)
            got, _ = b.decode_jpeg(data)
            if got.shape != exp.shape or not np.array_equal(got, exp):
                bad += 1
                print("DECODE MISMATCH", seed, i, desc, mode)
        res = b.transform([c[2] for c in cases], 57, 41, quality=80)
        for (i, desc, data), r in zip(cases, res):
            tot += 1
            exp = O.transform_jpeg_thumbnail(data, 57, 41, 80)
            if r.status != 0 or r.data != exp:
                a, c = (O.jpeg_decode(r.data) if r.status == 0 else None), O.jpeg_decode(exp)
                if a is None or a.shape != c.shape or np.abs(a.astype(int) - c.astype(int)).max() > 8:
                    bad += 1
                    print("TRANSFORM MISMATCH", seed, i, desc, mode, r.status)
    L.lilliput_hip_set_progressive_entropy(0)
print("progressive sources: checked", tot, "bad", bad, "in %.1fs" % (time.time() - t0))

# writers: progressive JPEG and PNG output against the reference libraries (when present) on fresh random pixels
bad = tot = 0
if O.ref() is not None and O.ref_png() is not None:
    import test_progressive_output as TPJ
    rng = np.random.default_rng(seed0)
    for it in range(per * nseeds):
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        cn = int(rng.choice([1, 3, 4]))
        px = TPO._pixels(1000 + it, h, w, cn)
        level = int(rng.choice([-1, 0, 2, 5, 7, 9]))
        tot += 1
        got = TPO._abi_png(L, px, level)
        if O.png_filtered_stream(got)[:2] != O.png_filtered_stream(O.ref_png_encode(px, level))[:2]:
            bad += 1; print("PNG MISMATCH", it, h, w, cn, level)
        if cn != 4:
            q = int(rng.choice([1, 30, 60, 85, 95, 100]))
            rgb = px if cn == 1 else np.ascontiguousarray(px[:, :, ::-1])
            src = L.opencv_mat_create_from_data(w, h, 0 if cn == 1 else 16, px.ctypes.data_as(C.c_void_p), C.c_size_t(px.size))
            outbuf = np.zeros(h * w * 3 + 65536, np.uint8)
            dst = L.opencv_mat_create_empty_from_data(outbuf.size, outbuf.ctypes.data_as(C.c_void_p))
            enc = L.opencv_encoder_create(b".jpeg", dst)
            ok = L.opencv_encoder_write(enc, src, (C.c_int * 4)(1, q, 2, 1), C.c_size_t(4))
            out = outbuf[: L.opencv_mat_get_height(dst)].tobytes()
            L.opencv_encoder_release(enc); L.opencv_mat_release(src); L.opencv_mat_release(dst)
            tot += 1
            if not ok or out != TPJ._ref_encode(O, rgb, q, True):
                bad += 1; print("PROGRESSIVE OUTPUT MISMATCH", it, h, w, cn, q)
    print("writers: checked", tot, "bad", bad)

# threads: progressive sources in, progressive / PNG out, several Python threads on one device
errs = []
def worker(k):
    try:
        ops = la.ImageOps(2048)
        for i, desc, data in TP._cases(500 + k, 12, lo=30, hi=300):
            exp = O.transform_jpeg_thumbnail(data, 64, 64, 85)
            for ft, eo in ((".jpeg", {la.JpegQuality: 85}), (".jpeg", {la.JpegQuality: 85, la.JpegProgressive: 1}), (".png", {la.PngCompression: 5})):
                d = la.Decoder(data)
                out = ops.Transform(d, la.ImageOptions(ft, 64, 64, la.ImageOpsFit, False, eo))
                d.Close()
                if ft == ".jpeg" and not eo.get(la.JpegProgressive):
                    a, c = O.jpeg_decode(out), O.jpeg_decode(exp)
                    if a.shape != c.shape or np.abs(a.astype(int) - c.astype(int)).max() > 8:
                        errs.append((k, i, desc, "jpeg"))
                elif ft == ".jpeg":
                    if O.jpeg_decode(out).shape != O.jpeg_decode(exp).shape:
                        errs.append((k, i, desc, "progressive"))
                elif O.png_filtered_stream(out)[0][:2] != O.jpeg_decode(exp).shape[1::-1]:
                    errs.append((k, i, desc, "png"))
        ops.Close()
    except Exception as e:  # noqa
        errs.append((k, repr(e)))
ths = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
t0 = time.time()
[t.start() for t in ths]; [t.join() for t in ths]
print("threads: 8 x 36 transforms, errors", len(errs), errs[:4], "in %.1fs" % (time.time() - t0))
