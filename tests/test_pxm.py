"""PBM / PGM / PPM sources (cv::PxMDecoder behind opencv_decoder_create in the reference, opencv.cpp:99-171): the product's decoder
(lilliput_amd/csrc/lp_pxm.cpp) against the reference's own OpenCV object code (oracle/_ref/librefpxm.so) -- live where it is built,
through recorded answers (tests/golden/pxm_digests.json, written by tests/golden/make_pxm_digests.py) everywhere -- and such files
through the opencv_decoder_* entry points and the device path."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import pxm_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "pxm_digests.json")))


def _cases():
    cases = dict(pxm_cases.generated())
    cases.update(pxm_cases.fuzz(43, 2000))
    return cases


def _mine(L, data):
    L.lilliput_hip_pxm_decode.restype = C.c_int
    L.lilliput_hip_pxm_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    w, h, t = C.c_int(), C.c_int(), C.c_int()
    cap = 1 << 26
    out = np.zeros(cap, np.uint8)
    r = L.lilliput_hip_pxm_decode(data, len(data), C.byref(w), C.byref(h), C.byref(t), out.ctypes.data_as(C.c_void_p), cap)
    if r in (0, 2):
        cn = (t.value >> 3) + 1
        return out[: w.value * h.value * cn].reshape(h.value, w.value, cn).copy(), (None if r == 0 else 2), t.value
    return None, r, None


def _digest(px, err, typ):
    if px is None:
        return "header" if err == 1 else "big"  # "big": beyond the test's output buffer, undecided
    return "%st%d:%dx%dx%d:%s" % ("data:" if err else "", typ, px.shape[1], px.shape[0], px.shape[2], hashlib.sha256(px.tobytes()).hexdigest()[:24])


def test_every_variant_and_2000_damaged_files_decode_like_the_recorded_reference(hip_lib):
    """Verdict, the decoder's own type (8- or 16-bit, 1 or 3 channels), the pixels -- and, where readData gives up, the rows it had
    written by then."""
    cases = _cases()
    assert set(cases) == set(GOLD)
    bad = [(k, GOLD[k], _digest(*_mine(hip_lib, v))) for k, v in cases.items() if _digest(*_mine(hip_lib, v)) != GOLD[k]]
    assert not bad, bad[:10]
    assert sum(1 for v in GOLD.values() if v.startswith("t")) >= 1000 and sum(1 for v in GOLD.values() if v.startswith("data")) >= 500
    assert {v.split(":")[0] for v in GOLD.values() if v.startswith("t")} == {"t0", "t2", "t16", "t18"}


def test_recorded_answers_are_the_reference_decoders(oracle):
    if oracle.ref_pxm() is None:
        pytest.skip("oracle/_ref/librefpxm.so not built")
    cases = _cases()
    for k in list(cases)[::5]:
        assert _digest(*oracle.ref_pxm_decode(cases[k])) == GOLD[k], k


def test_damaged_files_decode_like_the_reference_decoder_live(hip_lib, oracle):
    """The same comparison live, on damaged files of a seed that LILLIPUT_FUZZ_SEED_OFFSET moves (conftest.fresh_seed)."""
    if oracle.ref_pxm() is None:
        pytest.skip("oracle/_ref/librefpxm.so not built")
    from conftest import fresh_seed

    cases = dict(pxm_cases.fuzz(fresh_seed(143), 600))
    bad = [(k, _digest(*oracle.ref_pxm_decode(v)), _digest(*_mine(hip_lib, v))) for k, v in cases.items() if _digest(*_mine(hip_lib, v)) != _digest(*oracle.ref_pxm_decode(v))]
    assert not bad, bad[:10]


def test_sample_arithmetic_by_hand(hip_lib):
    """The few rules, on files small enough to check by eye: bitmaps 1 -> 0 and 0 -> 255; ASCII samples clamped to the announced range and
    scaled to 0..255; raw samples copied; 16-bit raw samples give their upper byte; RGB stored as BGR."""
    px, err, t = _mine(hip_lib, b"P1\n4 1\n1 0 1 1\n")
    assert err is None and t == 0 and px[0, :, 0].tolist() == [0, 255, 0, 0]
    px, err, t = _mine(hip_lib, b"P4\n10 1\n\xa5\x40")
    assert err is None and px[0, :, 0].tolist() == [0, 255, 0, 255, 255, 0, 255, 0, 255, 0]
    px, err, t = _mine(hip_lib, b"P2\n4 1\n15\n0 15 16 7\n")
    assert err is None and px[0, :, 0].tolist() == [0, 255, 255, 7 * 255 // 15]
    px, err, t = _mine(hip_lib, b"P5\n3 1\n15\n\x00\x0f\x07")
    assert err is None and px[0, :, 0].tolist() == [0, 15, 7]
    px, err, t = _mine(hip_lib, b"P5\n2 1\n1000\n\x03\xe8\x01\x02")
    assert err is None and t == 2 and px[0, :, 0].tolist() == [3, 1]
    px, err, t = _mine(hip_lib, b"P6\n1 1\n255\n\x0a\x14\x1e")
    assert err is None and t == 16 and px[0, 0].tolist() == [30, 20, 10]
    px, err, t = _mine(hip_lib, b"P3\n1 1\n255\n10 20 30\n")
    assert err is None and px[0, 0].tolist() == [30, 20, 10]


def _abi(L):
    L.opencv_mat_create_from_data.restype = C.c_void_p
    L.opencv_mat_create_from_data.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.opencv_mat_release.argtypes = [C.c_void_p]
    L.opencv_decoder_create.restype = C.c_void_p
    L.opencv_decoder_create.argtypes = [C.c_void_p]
    L.opencv_decoder_release.argtypes = [C.c_void_p]
    L.opencv_decoder_read_header.restype = C.c_bool
    L.opencv_decoder_read_header.argtypes = [C.c_void_p]
    L.opencv_decoder_read_data.restype = C.c_bool
    L.opencv_decoder_read_data.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("opencv_decoder_get_width", "opencv_decoder_get_height", "opencv_decoder_get_pixel_type", "opencv_decoder_get_orientation"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p]
    L.opencv_decoder_get_description.restype = C.c_char_p
    L.opencv_decoder_get_description.argtypes = [C.c_void_p]
    L.opencv_mat_create.restype = C.c_void_p
    L.opencv_mat_create.argtypes = [C.c_int, C.c_int, C.c_int]
    return L


def test_symbol_is_exported_and_declared(hip_lib):
    assert hasattr(hip_lib, "lilliput_hip_pxm_decode")
    assert "lilliput_hip_pxm_decode" in open(os.path.join(ROOT, "include", "lilliput_hip.h")).read()


@pytest.mark.gpu
def test_decoder_entry_points_on_pxm_files(hip_lib, oracle):
    """opencv_decoder_create .. read_data (opencv.cpp:99-171) on one file of each kind: description "PXM" (what the Go layer keys the
    file type on, opencv.go:190-215), the decoder's own type, orientation 1, the pixels of cv::PxMDecoder."""
    L = _abi(hip_lib)
    for kind, mv in ((1, 1), (2, 255), (3, 255), (4, 1), (5, 255), (6, 255), (5, 1023), (6, 4095), (2, 100)):
        data = pxm_cases.make_pxm(kind, 67, 41, mv, seed=kind * 7 + mv)
        px, err, typ = _mine(hip_lib, data)
        assert err is None
        if oracle.ref_pxm() is not None:
            rpx, rerr, rtyp = oracle.ref_pxm_decode(data)
            assert rerr is None and rtyp == typ and np.array_equal(rpx, px)
        buf = np.frombuffer(data, np.uint8).copy()
        src = L.opencv_mat_create_from_data(len(data), 1, 0, buf.ctypes.data_as(C.c_void_p), len(data))
        d = L.opencv_decoder_create(src)
        assert d
        assert L.opencv_decoder_get_description(d) == b"PXM"
        assert L.opencv_decoder_read_header(d)
        assert (L.opencv_decoder_get_width(d), L.opencv_decoder_get_height(d)) == (67, 41)
        assert L.opencv_decoder_get_pixel_type(d) == typ and L.opencv_decoder_get_orientation(d) == 1
        cn = (typ >> 3) + 1
        out = np.zeros((41, 67, cn), np.uint8)
        dst = L.opencv_mat_create_from_data(67, 41, (cn - 1) << 3, out.ctypes.data_as(C.c_void_p), out.size)
        assert L.opencv_decoder_read_data(d, dst)
        assert np.array_equal(out, px)
        L.opencv_mat_release(dst)
        L.opencv_decoder_release(d)
        L.opencv_mat_release(src)
    # a file whose samples end early: header accepted, read_data refused (the reference's ErrReadData)
    data = pxm_cases.make_pxm(6, 67, 41, 255, seed=5)[:-9]
    buf = np.frombuffer(data, np.uint8).copy()
    src = L.opencv_mat_create_from_data(len(data), 1, 0, buf.ctypes.data_as(C.c_void_p), len(data))
    d = L.opencv_decoder_create(src)
    assert d and L.opencv_decoder_read_header(d)
    out = np.zeros((41, 67, 3), np.uint8)
    dst = L.opencv_mat_create_from_data(67, 41, 16, out.ctypes.data_as(C.c_void_p), out.size)
    assert not L.opencv_decoder_read_data(d, dst)
    L.opencv_mat_release(dst)
    L.opencv_decoder_release(d)
    L.opencv_mat_release(src)


def test_set_source_points_the_decoder_at_another_buffer(hip_lib):
    """opencv_decoder_set_source (opencv.hpp:68: declared, never defined by the reference): cv::ImageDecoder::setSource's meaning -- the
    decoder chosen at create() reads the new buffer, no signature check, the header is read again. Host-decoded formats only: no device."""
    L = _abi(hip_lib)
    L.opencv_decoder_set_source.restype = C.c_bool
    L.opencv_decoder_set_source.argtypes = [C.c_void_p, C.c_void_p]
    a, b = pxm_cases.make_pxm(5, 31, 17, 255, seed=1), pxm_cases.make_pxm(6, 12, 9, 255, seed=2)
    ba, bb = np.frombuffer(a, np.uint8).copy(), np.frombuffer(b, np.uint8).copy()
    ma = L.opencv_mat_create_from_data(len(a), 1, 0, ba.ctypes.data_as(C.c_void_p), len(a))
    mb = L.opencv_mat_create_from_data(len(b), 1, 0, bb.ctypes.data_as(C.c_void_p), len(b))
    d = L.opencv_decoder_create(ma)
    assert d and L.opencv_decoder_read_header(d) and (L.opencv_decoder_get_width(d), L.opencv_decoder_get_height(d)) == (31, 17)
    assert L.opencv_decoder_set_source(d, mb)
    assert L.opencv_decoder_read_header(d) and (L.opencv_decoder_get_width(d), L.opencv_decoder_get_height(d)) == (12, 9)
    assert L.opencv_decoder_get_pixel_type(d) == 16
    assert not L.opencv_decoder_set_source(d, None) and not L.opencv_decoder_set_source(None, mb)
    jpeg = np.frombuffer(b"\xff\xd8\xff\xe0 not a pxm file", np.uint8).copy()
    mj = L.opencv_mat_create_from_data(jpeg.size, 1, 0, jpeg.ctypes.data_as(C.c_void_p), jpeg.size)
    assert L.opencv_decoder_set_source(d, mj) and not L.opencv_decoder_read_header(d)  # another format: fails at the header, like OpenCV
    L.opencv_decoder_release(d)
    for m in (ma, mb, mj):
        L.opencv_mat_release(m)


@pytest.mark.gpu
def test_pxm_sources_through_transform(hip_lib, oracle):
    """PPM / PGM / PBM -> ImageOps.Transform -> JPEG on the device path, one image at a time and as items of a batch, against the
    reference CPU path fed by the reference's own decoder."""
    import lilliput_amd as la

    if oracle.ref_pxm() is None:
        pytest.skip("oracle/_ref/librefpxm.so not built")
    sources = [pxm_cases.make_pxm(kind, w, h, mv, seed=w + kind)
               for kind, w, h, mv in ((6, 640, 480, 255), (5, 333, 517, 255), (4, 512, 384, 1), (3, 200, 120, 255), (2, 301, 200, 15), (1, 160, 90, 1), (6, 256, 256, 1023), (5, 400, 300, 4095))]
    ops = la.ImageOps(8192)
    for data in sources:
        want = oracle.transform_any_to_jpeg(data, 128, 128, 85)
        opts = la.ImageOptions(FileType=".jpeg", Width=128, Height=128, ResizeMethod=la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
        d = la.Decoder(data)
        got = ops.Transform(d, opts, 1 << 20)
        d.Close()
        assert bytes(got) == want
    ops.Close()
    b = la.Batch(0)
    res = b.transform(sources, 128, 128, quality=85)
    for data, r in zip(sources, res):
        assert r.status == 0 and bytes(r.data) == oracle.transform_any_to_jpeg(data, 128, 128, 85)
    b.close()
