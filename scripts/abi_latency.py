"""Latency of the one-image drop-in path (Decoder + ImageOps.Transform through the opencv_* ABI) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lilliput_amd as la
from lilliput_amd import synth

cases = {"large-sunrise.jpg 1300x1942 -> 256x256": open(os.path.join(ROOT, "tests/golden/inputs/large-sunrise.jpg"), "rb").read(),
         "synthetic 4096x4096 -> 256x256": synth.synth_jpeg(0, 4096)}
ops = la.ImageOps(8192)
for name, data in cases.items():
    for n in (3, 20):
        t = time.time()
        for _ in range(n):
            d = la.Decoder(data)
            out = ops.Transform(d, la.ImageOptions(".jpeg", 256, 256, la.ImageOpsFit, False, {la.JpegQuality: 85}))
            d.Close()
        dt = (time.time() - t) / n
    print("%s: %.2f ms per Transform (%d bytes out)" % (name, dt * 1e3, len(out)))
ops.Close()
