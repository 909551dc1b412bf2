"""Regenerates tests/golden/pxm_digests.json: what the REFERENCE's cv::PxMDecoder (oracle/_ref/librefpxm.so: OpenCV 4.11's own
grfmt_pxm.cpp.o out of the reference's libopencv_imgcodecs.a) answers for every generated file of tests/pxm_cases.py and 2 000 damaged
variants -- "tT:WxHxC:sha256-of-pixels" (T = the decoder's own type), "header" (readHeader refuses), "data:tT:WxHxC:sha256 of what was
written before readData failed". Run in the build container (needs /root/reference for oracle/_ref): python tests/golden/make_pxm_digests.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pxm_cases  # noqa: E402
from oracle import oracle  # noqa: E402


def digest(px, err, typ):
    if px is None:
        return "header" if err == 1 else "big"  # "big": beyond the test's output buffer, undecided
    return "%st%d:%dx%dx%d:%s" % ("data:" if err else "", typ, px.shape[1], px.shape[0], px.shape[2], hashlib.sha256(px.tobytes()).hexdigest()[:24])


def all_cases():
    cases = dict(pxm_cases.generated())
    cases.update(pxm_cases.fuzz(43, 2000))
    return cases


if __name__ == "__main__":
    assert oracle.ref_pxm() is not None, "oracle/_ref/librefpxm.so is not built"
    out = {k: digest(*oracle.ref_pxm_decode(v)) for k, v in sorted(all_cases().items())}
    json.dump(out, open(os.path.join(HERE, "pxm_digests.json"), "w"), indent=0, sort_keys=True)
    kinds = {}
    for v in out.values():
        kinds[v.split(":")[0][:4]] = kinds.get(v.split(":")[0][:4], 0) + 1
    print(len(out), "cases", kinds)
