"""8192 x 8192 (the largest frame an ImageOps(8192) holds): batch Transform and the one-image ABI against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lilliput_amd as la
from lilliput_amd import synth
from oracle import oracle as O
for (w, h) in ((8192, 8192), (8191, 5003), (3001, 8192)):
    t = time.time(); data = synth.synth_jpeg(3, 8192, width=w, height=h); print("synth", (w, h), len(data), "%.1fs" % (time.time() - t), flush=True)
    b = la.Batch(0)
    for tw, th in ((256, 256), (300, 200)):
        t = time.time(); r = b.transform([data], tw, th, quality=85)[0]; dt = time.time() - t
        exp = O.transform_jpeg_thumbnail(data, tw, th, 85)
        same = r.status == 0 and r.data == exp
        if not same and r.status == 0:
            a, c = O.jpeg_decode(r.data), O.jpeg_decode(exp); same = "within %d" % np.abs(a.astype(int) - c.astype(int)).max()
        print("  batch", (tw, th), "status", r.status, "equal:", same, "%.0f ms" % (dt * 1e3), flush=True)
    b.close()
    ops = la.ImageOps(8192); d = la.Decoder(data)
    t = time.time(); out = ops.Transform(d, la.ImageOptions(".jpeg", 256, 256, la.ImageOpsFit, False, {la.JpegQuality: 85}, EncodeTimeout=10**11)); dt = time.time() - t
    print("  one-image ABI equal:", out == O.transform_jpeg_thumbnail(data, 256, 256, 85), "%.0f ms" % (dt * 1e3), flush=True)
    d.Close(); ops.Close()
