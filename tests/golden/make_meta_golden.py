"""Writes tests/golden/meta_golden.json: for every case of tests/meta_cases.py the answer of the REFERENCE's own prebuilt
libjpeg-turbo 3.1.0 / libpng 1.6.47 (through oracle/_ref/librefmeta.so, built from /root/reference/deps by oracle/Makefile),
as "<length>:<sha1 of the bytes>". Run in the build container (the reference is not present on the GPU box)."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import meta_cases
from oracle import oracle as O

assert O.ref_meta() is not None, "build oracle/_ref first (make -C oracle)"
fn = {"jpeg_icc": O.ref_jpeg_icc, "png_icc": O.ref_png_icc, "png_cicp": lambda d: O.ref_png_cicp(d) or b""}
gold = {}
for kind, name, data in meta_cases.all_cases():
    r = fn[kind](data)
    gold["%s/%s" % (kind, name)] = "%d:%s" % (len(r), hashlib.sha1(r).hexdigest()[:16])
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "meta_golden.json"), "w"), indent=0, sort_keys=True)
print(len(gold), "cases;", sum(1 for v in gold.values() if not v.startswith("0:")), "with a non-empty answer")
