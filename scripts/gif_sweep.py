"""Ad-hoc sweep: random animated GIFs (partial frames that may hang off the canvas, every disposal mode, transparency, local
palettes, interlace) -> device compositing against the reference render, and GIF -> GIF against the reference writer."""
import os, sys, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import gif_cases as G, test_gif as TG

L = TG.G.__wrapped__(la.lib()) if hasattr(TG.G, "__wrapped__") else None
rnd = random.Random(int(sys.argv[1])); n = int(sys.argv[2]); bad = 0; t0 = time.time()
import ctypes as C
L = la.lib()
L.giflib_decoder_create.restype = C.c_void_p; L.giflib_decoder_create.argtypes = [C.c_void_p]
for nm in ("get_width", "get_height", "get_num_frames", "get_frame_width", "get_frame_height", "get_prev_frame_delay", "get_prev_frame_disposal", "decode_frame_header", "skip_frame", "release"):
    getattr(L, "giflib_decoder_" + nm).argtypes = [C.c_void_p]
L.giflib_decoder_release.restype = None
L.giflib_decoder_decode_frame.argtypes = [C.c_void_p, C.c_void_p]; L.giflib_decoder_decode_frame.restype = C.c_bool
for it in range(n):
    sw, sh = rnd.randrange(1, 120), rnd.randrange(1, 120)
    bits = rnd.randrange(1, 9); ncol = 1 << bits
    pal = bytes(rnd.randrange(256) for _ in range(3 * ncol))
    recs = []
    if rnd.random() < 0.5: recs.append(b"\x21\xff\x0bNETSCAPE2.0\x03\x01" + bytes([rnd.randrange(256), 0]) + b"\x00")
    for f in range(rnd.randrange(1, 6)):
        if rnd.random() < 0.85:
            recs.append(G.gce(rnd.randrange(0, 4), rnd.randrange(0, 30), rnd.randrange(ncol) if rnd.random() < 0.5 else None))
        l, t = rnd.randrange(0, sw + 3), rnd.randrange(0, sh + 3)
        w, h = rnd.randrange(1, sw + 5), rnd.randrange(1, sh + 5)
        if rnd.random() < 0.3: l, t, w, h = 0, 0, sw, sh
        local = bytes(rnd.randrange(256) for _ in range(3 * ncol)) if rnd.random() < 0.3 else None
        px = bytes(rnd.randrange(ncol) for _ in range(w * h)) if rnd.random() < 0.5 else bytes(((x // 3 + y // 2) % ncol) for y in range(h) for x in range(w))
        recs.append(G.image(l, t, w, h, px, min_code=max(2, bits), interlace=rnd.random() < 0.3, local=local))
    data = G.gif(sw, sh, recs, palette=pal, bg=rnd.randrange(ncol))
    ref, mine = O.ref_gif_frames(data), TG.device_frames(L, data)
    ok = (ref is None) == (mine is None)
    if ok and ref is not None:
        ok = ref[:2] == mine[:2] and ref[3] == mine[3] and len(ref[2]) == len(mine[2]) and all(np.array_equal(a[0], b[0]) for a, b in zip(mine[2], ref[2]))
    if not ok:
        bad += 1; print("DECODE MISMATCH", it, (sw, sh)); open("gpurun_out/badgif_%d.gif" % it, "wb").write(data); continue
    if ref is not None and ref[3] == 1 and ref[2]:
        tw, th = max(1, sw * 2 // 3), max(1, sh * 2 // 3)
        exp = O.ref_gif_transcode(data, lambda c: O.transform_static(c, 1, tw, th, O.FIT, False))
        try:
            got = TG._transform(data, FileType=".gif", Width=tw, Height=th, ResizeMethod=la.ImageOpsFit)
        except la.LilliputError as e:
            got = None
        if got != exp:
            bad += 1; print("ENCODE MISMATCH", it, (sw, sh), None if got is None else len(got), None if exp is None else len(exp)); open("gpurun_out/badgif_%d.gif" % it, "wb").write(data)
print("checked", n, "bad", bad, "%.1fs" % (time.time() - t0))
