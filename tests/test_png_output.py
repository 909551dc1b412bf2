"""PNG OUTPUT -- opencv_encoder_write behind FileType ".png" (opencv.cpp:185-194 -> cv::PngEncoder::write -> libpng 1.6.47). The
per-row filter choice (libpng's smallest-sum heuristic, or SUB only when no compression level is given) and the filtering run on
the device, deflate on the host. Bar: the same chunk layout and the same filtered scanlines, byte for byte, as the reference's libpng
driven like OpenCV drives it (oracle/ref_png_driver.c ref_png_encode_like_opencv), and -- PNG being lossless -- the same pixels
back through the reference's decoder. The compressed bytes themselves differ: the reference links zlib-ng, this library zlib."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "png_output_golden.json")


def _pixels(seed, h, w, cn):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    kind = seed % 4
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 4)).astype(float)
    elif kind == 1:
        img = np.stack([128 + 100 * np.sin(x / 9.0 + c) * np.cos(y / 7.0 - c) for c in range(4)], -1)
    elif kind == 2:
        img = np.stack([(x * 3 + y * 5 + 40 * c) % 256 for c in range(4)], -1).astype(float)
    else:
        img = np.stack([128 + 60 * np.sin(x / 23.0 + c) + 50 * np.cos(y / 31.0) for c in range(4)], -1) + rng.normal(0, 3, (h, w, 4))
        img[:, : w // 2] = np.round(img[:, : w // 2] / 32) * 32  # flat areas: ties between filters
    px = np.clip(img, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(px[:, :, 0] if cn == 1 else px[:, :, :cn])


def _cases():
    rng = np.random.default_rng(4)
    for it in range(40):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        if it % 9 == 0:
            h = 1
        if it % 11 == 0:
            w = 1
        yield it, h, w, int(rng.choice([1, 3, 4])), int(rng.choice([-1, -1, 0, 1, 3, 6, 7, 9]))
    yield 100, 256, 256, 3, 7
    yield 101, 300, 255, 4, -1


def _abi_png(L, px, level):
    px = np.ascontiguousarray(px)
    h, w = px.shape[:2]
    cn = 1 if px.ndim == 2 else px.shape[2]
    src = L.opencv_mat_create_from_data(w, h, {1: 0, 3: 16, 4: 24}[cn], px.ctypes.data_as(C.c_void_p), C.c_size_t(px.size))
    outbuf = np.zeros(px.size * 2 + 65536, np.uint8)
    dst = L.opencv_mat_create_empty_from_data(outbuf.size, outbuf.ctypes.data_as(C.c_void_p))
    enc = L.opencv_encoder_create(b".png", dst)
    assert enc
    opts = (C.c_int * 2)(16, level)
    assert L.opencv_encoder_write(enc, src, opts if level >= 0 else None, C.c_size_t(2 if level >= 0 else 0))
    assert L.opencv_mat_get_data(dst) == outbuf.ctypes.data
    got = outbuf[: L.opencv_mat_get_height(dst)].tobytes()
    L.opencv_encoder_release(enc)
    L.opencv_mat_release(src)
    L.opencv_mat_release(dst)
    return got


def test_golden_digests_are_the_reference_librarys(oracle):
    """The recorded digests (IHDR + inflated filtered stream + chunk names) come from the reference's libpng."""
    if oracle.ref_png() is None:
        pytest.skip("oracle/_ref/librefpng.so not built (needs /root/reference)")
    gold = json.load(open(GOLD))
    for it, h, w, cn, level in _cases():
        px = _pixels(it, h, w, cn)
        ihdr, stream, names = oracle.png_filtered_stream(oracle.ref_png_encode(px, level))
        assert "%s|%s|%s" % (ihdr, hashlib.sha1(stream).hexdigest()[:20], ",".join(names if len(stream) < 8000 else names[:1] + names[-1:])) == gold[str(it)], it
        assert np.array_equal(oracle.ref_png_decode(oracle.ref_png_encode(px, level)).reshape(px.shape), px)


@pytest.mark.gpu
def test_png_output_rows_and_chunks_equal_libpngs(hip_lib, oracle):
    import lilliput_amd as la

    gold = json.load(open(GOLD))
    L = la.lib()
    for it, h, w, cn, level in _cases():
        px = _pixels(it, h, w, cn)
        got = _abi_png(L, px, level)
        ihdr, stream, names = oracle.png_filtered_stream(got)
        assert "%s|%s|%s" % (ihdr, hashlib.sha1(stream).hexdigest()[:20], ",".join(names if len(stream) < 8000 else names[:1] + names[-1:])) == gold[str(it)], (it, h, w, cn, level)
        if oracle.ref_png() is not None:
            assert np.array_equal(oracle.ref_png_decode(got).reshape(px.shape), px), it
            assert stream == oracle.png_filtered_stream(oracle.ref_png_encode(px, level))[1], it


@pytest.mark.gpu
def test_transform_to_png(hip_lib, oracle, fixture_bytes):
    """ImageOps.Transform(FileType ".png", PngCompression 7) from JPEG and PNG sources: the PNG holds exactly the pixels of the
    resized frame (compared through the product's own PNG decoder and the oracle's JPEG path), an SDR cICP chunk of a PNG source
    is carried over, and the result round-trips through Transform again."""
    import lilliput_amd as la

    ops = la.ImageOps(2048)
    data = fixture_bytes["ferry_sunset.jpg"]
    d = la.Decoder(data)
    png = ops.Transform(d, la.ImageOptions(".png", 100, 100, la.ImageOpsFit, False, {la.PngCompression: 7}))
    d.Close()
    ihdr, stream, names = oracle.png_filtered_stream(png)
    assert ihdr[:5] == (100, 100, 8, 2, 0) and names[0] == "IHDR" and names[-1] == "IEND"
    d = la.Decoder(data)
    raw = ops.Transform(d, la.ImageOptions(".bgra-frames", 100, 100, la.ImageOpsFit, False, {}, EncodeTimeout=30 * 10**9))
    d.Close()
    frame = la.parse_raw_frames(raw)[0][0]
    d = la.Decoder(png)
    back = ops.Transform(d, la.ImageOptions(".bgra-frames", 100, 100, la.ImageOpsFit, False, {}, EncodeTimeout=30 * 10**9))
    d.Close()
    assert np.array_equal(la.parse_raw_frames(back)[0][0], frame)
    # an SDR cICP of a PNG source travels to the PNG output (ops.go applyOutputCICP)
    import struct
    import zlib

    src_px = _pixels(3, 40, 50, 3)
    src_png = _abi_png(la.lib(), src_px, 6)
    cicp = struct.pack(">I4s4B", 4, b"cICP", 1, 13, 0, 1)
    cicp += struct.pack(">I", zlib.crc32(cicp[4:]))
    src_png = src_png[:33] + cicp + src_png[33:]
    d = la.Decoder(src_png)
    out = ops.Transform(d, la.ImageOptions(".png", 20, 20, la.ImageOpsFit, False, {la.PngCompression: 3}))
    d.Close()
    names = oracle.png_filtered_stream(out)[2]
    assert names[:2] == ["IHDR", "cICP"], names
    assert out[33 + 8 : 33 + 12] == bytes([1, 13, 0, 1])
    ops.Close()
