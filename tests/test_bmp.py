"""BMP sources (cv::BmpDecoder behind opencv_decoder_create in the reference, opencv.cpp:99-171): the product's decoder
(lilliput_amd/csrc/lp_bmp.cpp) against the reference's own OpenCV object code (oracle/_ref/librefbmp.so) -- live where it is built,
through recorded answers (tests/golden/bmp_digests.json, written by tests/golden/make_bmp_digests.py) everywhere -- and BMP files through
the device path."""
import ctypes as C
import hashlib
import json
import os
import random
import struct

import numpy as np
import pytest
from conftest import fresh_seed

import bmp_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "bmp_digests.json")))


def _mine(L, data):
    L.lilliput_hip_bmp_decode.restype = C.c_int
    L.lilliput_hip_bmp_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    w, h, cn = C.c_int(), C.c_int(), C.c_int()
    cap = 1 << 26
    out = np.zeros(cap, np.uint8)
    r = L.lilliput_hip_bmp_decode(data, len(data), C.byref(w), C.byref(h), C.byref(cn), out.ctypes.data_as(C.c_void_p), cap)
    if r == 0:
        return out[: w.value * h.value * cn.value].reshape(h.value, w.value, cn.value).copy(), None
    return None, r


def _digest(px, err):
    if px is None:
        return "header" if err == 1 else "big" if err == -1 else "data"  # "big": beyond the test's output buffer, undecided
    return "%dx%dx%d:%s" % (px.shape[1], px.shape[0], px.shape[2], hashlib.sha256(px.tobytes()).hexdigest()[:24])


def test_every_variant_and_1500_damaged_files_decode_like_the_recorded_reference(hip_lib):
    cases = dict(bmp_cases.generated())
    cases.update(bmp_cases.fuzz(41, 1500))
    assert set(cases) == set(GOLD)
    bad = [(k, GOLD[k], _digest(*_mine(hip_lib, v))) for k, v in cases.items() if _digest(*_mine(hip_lib, v)) != GOLD[k]]
    assert not bad, bad[:10]
    assert sum(1 for v in GOLD.values() if ":" in v) >= 700


def test_recorded_answers_are_the_reference_decoders(oracle):
    if oracle.ref_bmp() is None:
        pytest.skip("oracle/_ref/librefbmp.so not built")
    cases = dict(bmp_cases.generated())
    cases.update(bmp_cases.fuzz(41, 1500))
    for k in list(cases)[::7]:
        assert _digest(*oracle.ref_bmp_decode(cases[k])) == GOLD[k], k


def _rle_file(w, h, bpp, stream, grey=False, topdown=False, seed=1):
    rnd = random.Random(seed)
    n = 1 << bpp
    pal = b"".join(bytes([i * 255 // (n - 1)] * 3 + [0]) for i in range(n)) if grey else b"".join(bytes([rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), 0]) for _ in range(n))
    dib = struct.pack("<IiiHHIIiiII", 40, w, -h if topdown else h, 1, bpp, 2 if bpp == 4 else 1, len(stream), 0, 0, 0, 0)
    off = 14 + 40 + len(pal)
    return struct.pack("<2sIHHI", b"BM", off + len(stream), 0, 0, off) + dib + pal + stream


def test_arbitrary_rle_streams_and_bit_field_masks_live(hip_lib, oracle):
    """The corners that were found by probing the reference's decoder, kept as a live differential test: random RLE4 / RLE8 streams
    (end-of-bitmap before the last row, deltas, runs that leave their row, short data) and 32-bit bit fields with arbitrary masks."""
    if oracle.ref_bmp() is None:
        pytest.skip("oracle/_ref/librefbmp.so not built")
    rnd = random.Random(fresh_seed(123))
    for t in range(4000):
        bpp = rnd.choice((4, 8))
        w, h = rnd.randrange(1, 40), rnd.randrange(1, 9)
        stream = bytes(rnd.choice((0, 0, 1, 2, 3, 4, 5, w & 255, rnd.randrange(256))) for _ in range(rnd.randrange(2, 60)))
        d = _rle_file(w, h, bpp, stream, grey=rnd.random() < 0.3, topdown=rnd.random() < 0.3, seed=t)
        assert _digest(*_mine(hip_lib, d)) == _digest(*oracle.ref_bmp_decode(d)), (bpp, w, h, stream.hex())
    for t in range(400):
        masks = []
        for c in range(4):
            nb = rnd.randrange(1, 20)
            m = (rnd.getrandbits(nb) | 1 | (1 << (nb - 1))) << rnd.randrange(0, 32 - nb)
            masks.append(m if c < 3 or rnd.random() < 0.7 else 0)
        w, h = rnd.randrange(1, 9), rnd.randrange(1, 4)
        px = bytes(rnd.getrandbits(8) for _ in range(w * h * 4))
        dib = struct.pack("<IiiHHIIiiII", 108, w, h, 1, 32, 3, len(px), 0, 0, 0, 0) + struct.pack("<IIII", *masks) + bytes(108 - 56)
        d = struct.pack("<2sIHHI", b"BM", 14 + 108 + len(px), 0, 0, 14 + 108) + dib + px
        assert _digest(*_mine(hip_lib, d)) == _digest(*oracle.ref_bmp_decode(d)), [hex(m) for m in masks]


def _decode_abi(L, data):
    arr = np.frombuffer(data, np.uint8).copy()
    em = L.opencv_mat_create_from_data(len(data), 1, 0, arr.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)))
    dec = L.opencv_decoder_create(em)
    out = None
    desc = None
    if dec:
        L.opencv_decoder_get_description.restype = C.c_char_p
        L.opencv_decoder_get_description.argtypes = [C.c_void_p]
        desc = L.opencv_decoder_get_description(dec)
        if L.opencv_decoder_read_header(dec):
            w, h, t = L.opencv_decoder_get_width(dec), L.opencv_decoder_get_height(dec), L.opencv_decoder_get_pixel_type(dec)
            cn = (t >> 3) + 1
            buf = np.zeros(w * h * cn, dtype=np.uint8)
            m = L.opencv_mat_create_from_data(w, h, t, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
            if L.opencv_decoder_read_data(dec, m):
                out = buf.reshape(h, w, cn).copy()
            assert L.opencv_decoder_get_orientation(dec) == 1
            L.opencv_mat_release(m)
        L.opencv_decoder_release(dec)
    L.opencv_mat_release(em)
    return out, desc


def test_row_pitch_that_overflows_int_is_refused_not_thrown(hip_lib):
    """A 200-byte file claiming 2^26 .. 2^30 pixels per row at 32 bits per pixel: cv::BmpDecoder's 32-bit row pitch goes negative (a
    failed allocation there, the decode refused); here the pitch is computed in 64 bits and the decode refused -- no exception may
    unwind through opencv_decoder_read_data (std::length_error reached terminate())."""
    L = hip_lib
    L.lilliput_hip_bmp_decode.restype = C.c_int
    L.lilliput_hip_bmp_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    out = np.zeros(1 << 20, np.uint8)
    for width in (1 << 26, (1 << 26) + 3, 1 << 27, (1 << 27) + 5, 1 << 29, (1 << 31) - 1):
        for bpp in (24, 32):
            head = b"BM" + struct.pack("<IHHI", 54 + 146, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, width, 1, 1, bpp, 0, 0, 0, 0, 0, 0)
            data = head + bytes(146)
            w, h, cn = C.c_int(), C.c_int(), C.c_int()
            # cap = SIZE_MAX / 2: the entry point's own size test passes, the row reader must refuse before it writes anything
            r = L.lilliput_hip_bmp_decode(data, len(data), C.byref(w), C.byref(h), C.byref(cn), out.ctypes.data_as(C.c_void_p), (1 << 62))
            assert r in (1, 2), (width, bpp, r)
    assert not out.any()


def test_opencv_decoder_abi_serves_bmp_files(hip_lib):
    """opencv_decoder_create .. read_data on BMP buffers: "BMP" as the description (what cv::ImageDecoder::getDescription answers in the
    reference build), the decoder's own Mat type, the recorded pixels. No device involved: the rows are unpacked on the host."""
    L = hip_lib
    for name, data in bmp_cases.generated().items():
        px, desc = _decode_abi(L, data)
        assert desc == b"BMP"
        assert _digest(px, 2) == GOLD[name], name


@pytest.mark.gpu
def test_bmp_sources_through_transform(hip_lib, oracle):
    """BMP -> ImageOps.Transform -> JPEG on the device path, one image at a time and as items of a batch, against the reference CPU
    path fed by the reference's own BMP decoder."""
    import lilliput_amd as la

    if oracle.ref_bmp() is None:
        pytest.skip("oracle/_ref/librefbmp.so not built")
    rng = np.random.default_rng(4)
    sources = []
    for w, h, bpp, kw in ((640, 480, 24, {}), (333, 517, 32, {}), (512, 512, 8, {"grey": True}), (400, 300, 8, {"compression": 1}), (301, 200, 4, {}), (256, 256, 16, {}),
                          (200, 120, 32, {"compression": 3, "header": 108})):
        sources.append(bmp_cases.make_bmp(w, h, bpp, seed=int(rng.integers(1 << 20)), **kw))
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
