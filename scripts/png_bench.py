"""PNG source timing on one GPU: opencv_decoder_read_data (host chunk walk + inflate, device filter reversal + expansion) and
PNG -> 256x256 JPEG through ImageOps.Transform, next to the reference's libpng + zlib-ng on one core (oracle/_ref)."""
import ctypes as C, os, struct, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
import png_cases
from lilliput_amd import synth
from oracle import oracle as O

def big_png(side, rgba=False):
    rgb = synth.synth_rgb(1, side)                      # photo-like content of the JPEG workload
    px = np.dstack([rgb, np.full((side, side, 1), 255, np.uint8)]) if rgba else rgb
    flat = px.reshape(side, -1).astype(np.int16)
    up = np.vstack([flat[:1], flat[1:] - flat[:-1]]).astype(np.uint8)   # filter type 2 (Up) on every row but the first
    raw = np.hstack([np.r_[np.uint8(0), np.full(side - 1, 2, np.uint8)].reshape(-1, 1), up]).tobytes()
    ih = png_cases.chunk(b"IHDR", struct.pack(">IIBBBBB", side, side, 8, 6 if rgba else 2, 0, 0, 0))
    return png_cases.SIG + ih + png_cases.chunk(b"IDAT", zlib.compress(raw, 6)) + png_cases.chunk(b"IEND", b"")

L = la.lib()
L.lilliput_hip_png_inflate_check.restype = C.c_long
L.lilliput_hip_png_inflate_check.argtypes = [C.c_char_p, C.c_size_t]
ops = la.ImageOps(4096)
for name, data in (("ferry_sunset.png 800x297 RGB", png_cases.fixtures()["ferry_sunset.png"]), ("synthetic 2048x2048 RGB", big_png(2048)), ("synthetic 4096x4096 RGBA", big_png(4096, True))):
    t = time.perf_counter(); n = L.lilliput_hip_png_inflate_check(data, len(data)); th = time.perf_counter() - t
    for rep in range(3):
        t = time.perf_counter()
        d = la.Decoder(data); out = ops.Transform(d, la.ImageOptions(".jpeg", 256, 256, la.ImageOpsFit, False, {la.JpegQuality: 85}, EncodeTimeout=10**10)); d.Close()
        tt = time.perf_counter() - t
    t = time.perf_counter(); ref = O.ref_png_decode(data) if O.ref_png() else None; tr = time.perf_counter() - t
    print("%s (%d KB, %.1f MB inflated): Transform -> JPEG %.2f ms, of which host walk + inflate %.2f ms; reference libpng decode alone %.2f ms" % (name, len(data) // 1024, n / 1e6, tt * 1e3, th * 1e3, tr * 1e3))
ops.Close()
