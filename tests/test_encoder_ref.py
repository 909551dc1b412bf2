"""The JPEG ENCODER pinned to the reference's own object code: cv::JpegEncoder::write out of the reference's libopencv_imgcodecs.a
(grfmt_jpeg.cpp.o, the class cv::ImageEncoder(".jpeg", dst) holds behind opencv_encoder_create / opencv_encoder_write,
/root/reference/opencv.cpp:173-194) over the reference's libjpeg.a, driven by oracle/ref_jpegcv_driver.cpp. Until round 6 the encoder's
call sequence (quality clamping, which IMWRITE_JPEG_* keys it reads, what it does with 1 / 3 / 4 channels, progressive mode, the
destination Mat's growth) was restated from the survey's disassembly notes; here the restatement (oracle.jpeg_encode) and the product
(opencv_encoder_write through the C ABI) are held against the class itself, byte for byte."""
import ctypes as C

import numpy as np
import pytest

JPEG_QUALITY, JPEG_PROGRESSIVE, PNG_COMPRESSION, WEBP_QUALITY = 1, 2, 16, 64   # opencv.go:43-48 / opencv.hpp:33-36
SIZES = [(1, 1), (1, 7), (7, 1), (8, 8), (9, 8), (15, 17), (16, 16), (17, 16), (33, 47), (64, 48), (100, 75)]   # (w, h)
QUALITIES = [-1, 0, 1, 2, 50, 85, 95, 100, 101, 1000]


def _frame(rng, w, h, ch, kind):
    if kind == 0:   # noise
        px = rng.integers(0, 256, (h, w, ch))
    elif kind == 1:  # smooth
        y, x = np.mgrid[0:h, 0:w]
        px = np.stack([(x * 5 + y * 3 + 40 * c) % 256 for c in range(ch)], axis=-1)
    else:            # saturated patches
        px = (rng.integers(0, 2, (h // 4 + 1, w // 4 + 1, ch)) * 255).repeat(4, 0).repeat(4, 1)[:h, :w]
    return np.ascontiguousarray(px.astype(np.uint8))


def _need_ref(oracle):
    if oracle.ref_cvjpeg() is None:
        pytest.skip("oracle/_ref/librefjpegcv.so not built (the reference's archives are absent)")


def test_restated_encoder_equals_the_reference_class(oracle):
    """oracle.jpeg_encode(px, q) == cv::JpegEncoder::write(px, {IMWRITE_JPEG_QUALITY: q}) for gray / BGR / BGRA frames down to 1 x 1,
    for every quality the Go layer can pass (the class clamps to 0..100; libjpeg's jpeg_set_quality turns 0 into 1)."""
    _need_ref(oracle)
    rng = np.random.default_rng(61)
    n = 0
    for (w, h) in SIZES:
        for ch in (1, 3, 4):
            for k, q in enumerate(QUALITIES):
                px = _frame(rng, w, h, ch, (n + k) % 3)
                ref = oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, q))
                assert ref is not None and ref[:2] == b"\xff\xd8" and ref[-2:] == b"\xff\xd9", (w, h, ch, q)
                assert oracle.jpeg_encode(px, q) == ref, (w, h, ch, q)
                n += 1
    assert n == len(SIZES) * 3 * len(QUALITIES)


def test_reference_class_defaults_and_foreign_keys(oracle):
    """No parameters = quality 95 (cv::JpegEncoder's default); the keys the other encoders own (PngCompression, WebpQuality -- opencv.go
    passes the caller's whole EncodeOptions map) change nothing; a later JpegQuality pair wins over an earlier one."""
    _need_ref(oracle)
    rng = np.random.default_rng(62)
    px = _frame(rng, 45, 31, 3, 1)
    base = oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 85))
    assert oracle.ref_cv_jpeg_encode(px, ()) == oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 95)) == oracle.jpeg_encode(px, 95)
    assert oracle.ref_cv_jpeg_encode(px, (PNG_COMPRESSION, 9, JPEG_QUALITY, 85, WEBP_QUALITY, 10)) == base
    assert oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 20, JPEG_QUALITY, 85)) == base
    assert oracle.ref_cv_jpeg_encode(px, (JPEG_PROGRESSIVE, 0, JPEG_QUALITY, 85)) == base
    prog = oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 85, JPEG_PROGRESSIVE, 1))
    assert prog != base and b"\xff\xc2" in prog[:700] and np.array_equal(oracle.jpeg_decode(prog), oracle.jpeg_decode(base))   # same coefficients, SOF2


def test_reference_class_moves_a_too_small_destination(oracle):
    """The overflow rule opencv.go:890-895 relies on: a destination whose capacity the result exceeds ends up with ANOTHER data pointer."""
    _need_ref(oracle)
    rng = np.random.default_rng(63)
    px = _frame(rng, 64, 64, 3, 0)
    full = oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90))
    assert oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90), cap=len(full)) == full
    assert oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90), cap=len(full) - 1) == ("moved", len(full))
    assert oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90), cap=100) == ("moved", len(full))


def _product_encode(L, px, params, cap=1 << 20):
    h, w = px.shape[:2]
    ch = 1 if px.ndim == 2 else px.shape[2]
    for f in ("opencv_mat_create_from_data", "opencv_mat_create_empty_from_data", "opencv_encoder_create", "opencv_mat_get_data"):
        getattr(L, f).restype = C.c_void_p
    for f in ("opencv_mat_release", "opencv_encoder_release"):
        getattr(L, f).restype = None
    L.opencv_encoder_write.restype = C.c_bool
    src = np.ascontiguousarray(px)
    m = L.opencv_mat_create_from_data(C.c_int(w), C.c_int(h), C.c_int((ch - 1) << 3), C.c_void_p(src.ctypes.data), C.c_size_t(src.size))
    out = np.zeros(max(cap, 1), np.uint8)
    dm = L.opencv_mat_create_empty_from_data(C.c_int(cap), C.c_void_p(out.ctypes.data))
    e = L.opencv_encoder_create(b".jpeg", C.c_void_p(dm))
    par = (C.c_int * max(len(params), 1))(*params)
    ok = L.opencv_encoder_write(C.c_void_p(e), C.c_void_p(m), par, C.c_size_t(len(params)))
    res = None
    if ok:
        n = L.opencv_mat_get_height(C.c_void_p(dm))
        p = L.opencv_mat_get_data(C.c_void_p(dm))
        res = bytes(out[:n]) if p == out.ctypes.data else ("moved", n)
    L.opencv_encoder_release(C.c_void_p(e))
    L.opencv_mat_release(C.c_void_p(dm))
    L.opencv_mat_release(C.c_void_p(m))
    return res


@pytest.mark.gpu
def test_product_encoder_equals_the_reference_class(hip_lib, oracle):
    """opencv_encoder_write (the device encoder behind the C ABI) against cv::JpegEncoder::write of the reference, byte for byte: gray /
    BGR / BGRA, sizes down to 1 x 1, every quality, progressive mode, no parameters, foreign keys, a destination that is too small."""
    _need_ref(oracle)
    rng = np.random.default_rng(64)
    n = 0
    for (w, h) in SIZES:
        for ch in (1, 3, 4):
            for k, q in enumerate(QUALITIES):
                if (n + k) % 3 and (w, h) not in ((1, 1), (17, 16), (100, 75)):   # a third of the grid, the corner sizes in full
                    continue
                px = _frame(rng, w, h, ch, (n + k) % 3)
                if ch == 1:
                    px = px[:, :, 0]
                assert _product_encode(hip_lib, px, (JPEG_QUALITY, q)) == oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, q)), (w, h, ch, q)
            n += 1
    for (w, h, ch) in ((45, 31, 3), (16, 16, 1), (33, 9, 4), (1, 1, 3)):
        px = _frame(rng, w, h, ch, 1)
        if ch == 1:
            px = px[:, :, 0]
        for params in ((), (PNG_COMPRESSION, 9, JPEG_QUALITY, 85, WEBP_QUALITY, 10), (JPEG_QUALITY, 20, JPEG_QUALITY, 85), (JPEG_QUALITY, 85, JPEG_PROGRESSIVE, 1),
                       (JPEG_PROGRESSIVE, 1), (JPEG_PROGRESSIVE, 0, JPEG_QUALITY, 70), (JPEG_PROGRESSIVE, 7, JPEG_QUALITY, 101)):
            assert _product_encode(hip_lib, px, params) == oracle.ref_cv_jpeg_encode(px, params), (w, h, ch, params)
    px = _frame(rng, 64, 64, 3, 0)
    full = oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90))
    for cap in (len(full), len(full) - 1, 100):
        assert _product_encode(hip_lib, px, (JPEG_QUALITY, 90), cap=cap) == oracle.ref_cv_jpeg_encode(px, (JPEG_QUALITY, 90), cap=cap), cap
