"""Regenerates tests/golden/bmp_digests.json: what the REFERENCE's cv::BmpDecoder (oracle/_ref/librefbmp.so: OpenCV 4.11's own
grfmt_bmp.cpp.o out of the reference's libopencv_imgcodecs.a) answers for every generated BMP of tests/bmp_cases.py and 1 500 damaged
variants -- "WxHxC:sha256-of-pixels", "header" (readHeader refuses) or "data" (readData fails). Run in the build container
(needs /root/reference for oracle/_ref): python tests/golden/make_bmp_digests.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bmp_cases  # noqa: E402
from oracle import oracle  # noqa: E402


def digest(data):
    px, err = oracle.ref_bmp_decode(data)
    if px is None:
        return "header" if err == 1 else "big" if err == -1 else "data"  # "big": beyond the test's output buffer, undecided
    return "%dx%dx%d:%s" % (px.shape[1], px.shape[0], px.shape[2], hashlib.sha256(px.tobytes()).hexdigest()[:24])


if __name__ == "__main__":
    assert oracle.ref_bmp() is not None, "oracle/_ref/librefbmp.so is not built"
    cases = dict(bmp_cases.generated())
    cases.update(bmp_cases.fuzz(41, 1500))
    out = {k: digest(v) for k, v in sorted(cases.items())}
    json.dump(out, open(os.path.join(HERE, "bmp_digests.json"), "w"), indent=0, sort_keys=True)
    print(len(out), "cases,", sum(1 for v in out.values() if ":" in v), "decoded")
