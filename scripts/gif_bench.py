"""Animated path timing on one GPU: GIF -> decode + composite + Fit per frame (ImageOps.Transform into the raw frame sink),
next to the CPU checker (reference giflib + restated compositing + INTER_AREA restatement, one core)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
import gif_cases
from oracle import oracle as O

def synthetic(side, frames, seed=0):
    rng = np.random.default_rng(seed)
    base = (rng.integers(0, 8, (side // 8, side // 8)).repeat(8, 0).repeat(8, 1)).astype(np.uint8)
    recs = []
    for f in range(frames):
        img = np.roll(base, f * 3, axis=1)
        recs += [gif_cases.gce(1, 4), gif_cases.image(0, 0, side, side, img.tobytes())]
    return gif_cases.gif(side, side, recs)

cases = {"no-loop.gif 128x128 x44": gif_cases.fixtures()["no-loop.gif"], "synthetic 512x512 x12": synthetic(512, 12), "synthetic 1024x1024 x6": synthetic(1024, 6)}
ops = la.ImageOps(2048)
for name, data in cases.items():
    nfr = la.Decoder(data).AnimationInfo()[1]
    opt = la.ImageOptions(".bgra-frames", 128, 128, la.ImageOpsFit, EncodeTimeout=60 * 10**9)
    for rep in range(3):
        t = time.perf_counter()
        d = la.Decoder(data); out = ops.Transform(d, opt, dst_cap=64 << 20); d.Close()
        dt = time.perf_counter() - t
    t = time.perf_counter()
    ref = O.ref_gif_frames(data)
    for canvas, _, _ in ref[2]:
        O.transform_static(canvas, 1, 128, 128, O.FIT, False)
    ct = time.perf_counter() - t
    print("%s: device path %.2f ms (%.0f frames/s); CPU checker %.2f ms (%.0f frames/s)" % (name, dt * 1e3, nfr / dt, ct * 1e3, nfr / ct))
ops.Close()
