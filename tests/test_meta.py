"""Colour-metadata readers of the opencv.hpp ABI (host-side container walks, no GPU needed): lp_abi_meta.cpp against
(1) answers of the reference's own libjpeg-turbo / libpng recorded in tests/golden/meta_golden.json and
(2) the same libraries live, when oracle/_ref/librefmeta.so is present."""
import ctypes as C
import hashlib
import json
import os

import pytest
from conftest import fresh_seed

import meta_cases


@pytest.fixture(scope="module")
def readers(hip_lib):
    L = hip_lib
    for f in (L.opencv_decoder_get_jpeg_icc, L.opencv_decoder_get_png_icc):
        f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.opencv_decoder_get_png_cicp.argtypes = [C.c_char_p, C.c_size_t] + [C.POINTER(C.c_uint8)] * 4
    L.opencv_png_insert_cicp.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t] + [C.c_uint8] * 4
    L.opencv_png_insert_cicp.restype = C.c_size_t

    def icc(fn, data, cap=1 << 16):
        out = C.create_string_buffer(cap)
        n = fn(data, len(data), out, cap)
        return out.raw[:n]

    def cicp(data):
        v = [C.c_uint8(0) for _ in range(4)]
        return bytes(x.value for x in v) if L.opencv_decoder_get_png_cicp(data, len(data), *[C.byref(x) for x in v]) else b""

    return {"jpeg_icc": lambda d: icc(L.opencv_decoder_get_jpeg_icc, d), "png_icc": lambda d: icc(L.opencv_decoder_get_png_icc, d), "png_cicp": cicp, "lib": L}


def test_readers_match_recorded_answers_of_the_reference_libraries(readers):
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meta_golden.json")))
    cases = meta_cases.all_cases()
    assert len(cases) == len(gold)
    bad = []
    for kind, name, data in cases:
        r = readers[kind](data)
        if "%d:%s" % (len(r), hashlib.sha1(r).hexdigest()[:16]) != gold["%s/%s" % (kind, name)]:
            bad.append((kind, name, len(r), gold["%s/%s" % (kind, name)]))
    assert not bad, bad[:10]


def test_readers_match_the_reference_libraries_live(readers, oracle):
    if oracle.ref_meta() is None:
        pytest.skip("oracle/_ref/librefmeta.so not built (needs /root/reference)")
    ref = {"jpeg_icc": oracle.ref_jpeg_icc, "png_icc": oracle.ref_png_icc, "png_cicp": lambda d: oracle.ref_png_cicp(d) or b""}
    for kind, name, data in meta_cases.hand_cases() + meta_cases.fuzz_cases(fresh_seed(303), 400):
        assert readers[kind](data) == ref[kind](data), (kind, name)


def test_jpeg_icc_real_profile_and_capacity(readers, fixture_bytes):
    L = readers["lib"]
    data = fixture_bytes["ferry_sunset.jpg"]
    prof = readers["jpeg_icc"](data)
    assert len(prof) == 536 and prof[36:40] == b"acsp"       # the Display P3 profile embedded in the fixture
    out = C.create_string_buffer(535)
    assert L.opencv_decoder_get_jpeg_icc(data, len(data), out, 535) == 0   # does not fit: 0, nothing partial (opencv.cpp:283)
    # the same profile split over three APP2 chunks written out of order comes back whole
    bare = fixture_bytes["field.jpg"]
    parts = [prof[:200], prof[200:400], prof[400:]]
    segs = [meta_cases.app2(i + 1, 3, parts[i]) for i in (2, 0, 1)]
    assert readers["jpeg_icc"](bare[:2] + b"".join(segs) + bare[2:]) == prof


def test_png_insert_cicp_round_trip(readers):
    L = readers["lib"]
    src = meta_cases.png([meta_cases.chunk(b"gAMA", b"\0\0\xb1\x8f")])
    buf = C.create_string_buffer(src, len(src) + 16)
    n = L.opencv_png_insert_cicp(buf, len(src), len(src) + 16, 9, 16, 0, 1)
    assert n == len(src) + 16
    out = buf.raw[:n]
    assert out[:33] == src[:33] and out[33 + 16 :] == src[33:] and out[37:41] == b"cICP"   # directly after IHDR, rest untouched
    assert readers["png_cicp"](out) == bytes([9, 16, 0, 1])
    # no room, not a PNG, IHDR not first: untouched, old length back (opencv.cpp:421-440)
    assert L.opencv_png_insert_cicp(buf, len(src), len(src) + 15, 9, 16, 0, 1) == len(src)
    jb = C.create_string_buffer(b"\xff\xd8\xff\xe0" + b"\0" * 60, 128)
    assert L.opencv_png_insert_cicp(jb, 64, 128, 9, 16, 0, 1) == 64
    odd = src[:12] + b"IHDS" + src[16:]
    ob = C.create_string_buffer(odd, len(odd) + 16)
    assert L.opencv_png_insert_cicp(ob, len(odd), len(odd) + 16, 9, 16, 0, 1) == len(odd) and ob.raw[: len(odd)] == odd


def test_reference_icc_presence_cases_through_the_decoder_api(fixture_bytes):
    """opencv_test.go:222-262 TestICC: Decoder.ICC() is non-empty for the ferry_sunset fixtures that embed a profile and empty for
    the ones that do not (host-side readers: no GPU needed)."""
    import lilliput_amd as la
    import png_cases

    png = png_cases.fixtures()
    no_icc = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs_misc", "ferry_sunset_no_icc.jpg"), "rb").read()
    for data, want in ((fixture_bytes["ferry_sunset.jpg"], True), (no_icc, False), (png["ferry_sunset.png"], True), (png["firefox.png"], False)):
        d = la.Decoder(data)
        icc = d.ICC()
        d.Close()
        assert (len(icc) > 0) == want
        if want:
            assert icc[36:40] == b"acsp"
