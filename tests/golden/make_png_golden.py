"""Writes tests/golden/png_golden.json from the REFERENCE's libpng 1.6.47 + zlib-ng driven like cv::PngDecoder
(oracle/_ref/librefpng.so): per case "none" (rejected) or "<w>x<h>x<channels>:<sha1 of the pixels>".
Run in the build container (needs /root/reference to build oracle/_ref)."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import png_cases
from oracle import oracle as O

assert O.ref_png() is not None, "build oracle/_ref first (make -C oracle)"
cases = dict(png_cases.fixtures()); cases.update(png_cases.generated()); cases.update(png_cases.fuzz(21, 1500))
gold = {}
for k, v in cases.items():
    px = O.ref_png_decode(v)
    gold[k] = "none" if px is None else "%dx%dx%d:%s" % (px.shape[1], px.shape[0], px.shape[2], hashlib.sha1(px.tobytes()).hexdigest()[:16])
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "png_golden.json"), "w"), indent=0, sort_keys=True)
print(len(gold), "cases,", sum(1 for v in gold.values() if v != "none"), "decodable")
