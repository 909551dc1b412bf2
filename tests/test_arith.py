"""Arithmetic-coded JPEG sources (SOF9 sequential, SOF10 progressive): the reference's libjpeg-turbo decodes them (jdarith.c behind
opencv_decoder_read_data, /root/reference/opencv.cpp:166-171); the product decodes their QM-coded scans on host threads
(lilliput_amd/csrc/lp_arith_host.cpp) into the coefficient arena and takes over on the device at the IDCT, like the progressive
Huffman sources. Fixtures: written by the reference's own compressor (tests/golden/make_arith_jpegs.py), answers recorded from its
decoder (tests/golden/arith_golden.json); compared live as well wherever oracle/_ref/libref.so is built."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "inputs_arith")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "arith_golden.json")))


def _files():
    return {n: open(os.path.join(DIR, n), "rb").read() for n in sorted(os.listdir(DIR))}


def _host_coefs(L, data, comp, with_rc=False):
    """The coefficients the product's host threads decode (test access, no device): [block row][block column][64] or None.
    with_rc: (rc, coefficients) -- rc -2 = the scan decoders report that the reference's decoder fails on this file (out of data, an
    unknown marker behind a scan)."""
    a = np.frombuffer(bytes(data), np.uint8)
    out = np.zeros(1 << 22, np.int16)
    bw, bh = C.c_int(), C.c_int()
    rc = L.lilliput_hip_progressive_coefs_host(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(comp), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                                               C.byref(bw), C.byref(bh), C.c_int(2))
    if rc not in (0, -2):
        return (rc, None) if with_rc else None
    co = out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64).copy()
    return (rc, co) if with_rc else co


def test_fixture_set_is_what_the_generator_wrote():
    assert set(_files()) == set(GOLD) and len(GOLD) >= 20
    assert any(b"\xff\xc9" in d[:700] for d in _files().values()) and any(b"\xff\xca" in d[:700] for d in _files().values())  # SOF9 and SOF10
    assert any(b"\xff\xcc" in d[:700] for d in _files().values())  # DAC segments: conditioning other than the defaults


def test_qm_decoder_coefficients_equal_the_reference_librarys(hip_lib, oracle):
    """Every component of every fixture: the quantised coefficients of the product's QM decoder against the recorded digests of
    jpeg_read_coefficients of the reference's libjpeg.a -- and against the library itself when it is built here."""
    ref = oracle.ref() is not None
    for name, data in _files().items():
        for comp, want in enumerate(GOLD[name]["coefs"]):
            mine = _host_coefs(hip_lib, data, comp)
            assert mine is not None, (name, comp)
            assert hashlib.sha1(np.ascontiguousarray(mine).tobytes()).hexdigest()[:16] == want, (name, comp)
            if ref:
                assert np.array_equal(mine.ravel(), oracle.ref_jpeg_decode_coefs(data, comp).ravel()), (name, comp)


def test_header_verdicts_agree_on_arithmetic_files(hip_lib, oracle):
    """The oracle's header walk accepts SOF9 / SOF10 like libjpeg and the product do (its entropy decode of them stays with the real
    library); a progressive arithmetic scan with impossible parameters is refused by both."""
    from test_host_logic import _header_verdict

    for name, data in _files().items():
        assert _header_verdict(hip_lib, data), name
        info = oracle.jpeg_info(data)
        assert info["width"] > 0
        with pytest.raises(ValueError):
            oracle.jpeg_decode(data)  # LO_ERR_UNSUPPORTED: the restatement does not decode QM-coded data
    data = bytearray(_files()["arith_prog_simple_420.jpg"])
    sos = data.find(b"\xff\xda")
    ns = data[sos + 4]
    data[sos + 5 + 2 * ns + 1] = 5  # Se = 5 in a DC scan: JERR_BAD_PROGRESSION
    assert not _header_verdict(hip_lib, bytes(data))
    with pytest.raises(ValueError):
        oracle.jpeg_info(bytes(data))


def test_damaged_arithmetic_streams_decode_like_the_reference_library(hip_lib, oracle):
    """Bit flips, byte substitutions and truncation inside the entropy-coded data. The verdict is the reference's own decoder's
    (cv::JpegDecoder over its libjpeg.a, oracle/_ref/librefjpegcv.so): a stream that runs out of bytes FAILS there (jdarith.c get_byte
    cannot suspend under OpenCV's source manager), a damaged one decodes with a warning. Wherever it returns an image the coefficients
    are the library's. A damaged restart marker sends both through jpeg_resync_to_restart's rules (lp_jbits.h); a byte pair that only
    looks like a marker (code below 0xC0) inside a scan with a restart interval is read past by both (lp_jpeg_parse.cpp), and a
    multi-scan file whose scan ends on one is refused by both."""
    if oracle.ref() is None or oracle.ref_cvjpeg() is None:
        pytest.skip("oracle/_ref not built")
    rnd = random.Random(5)
    files = {n: d for n, d in _files().items() if "big" not in n}
    names = sorted(files)
    same = failed = 0
    odd = []
    for it in range(400):
        n = rnd.choice(names)
        d = bytearray(files[n])
        last_sos = d.rfind(b"\xff\xda")
        lo = last_sos + 6 + 2 * d[last_sos + 4] + 3  # inside the last scan's entropy-coded data
        if lo >= len(d) - 2:
            continue
        q = rnd.randrange(lo, len(d) - 2)
        mode = rnd.randrange(4)
        if mode == 0:
            d[q] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            d = d[:q]
        elif mode == 2:
            d = d[:q] + b"\xff\xd9"  # cut short and closed: zero bytes from the marker on, a warning, an image
        else:
            d[q] = rnd.randrange(256)
        d = bytes(d)
        cv = oracle.ref_cv_jpeg_decode(d)
        rc, mine = _host_coefs(hip_lib, d, 0, with_rc=True)
        if (cv is None) != (rc != 0):
            odd.append((it, n, mode, "verdict", cv is None, rc))
            continue
        if cv is None:
            failed += 1
            continue
        try:
            ref = oracle.ref_jpeg_decode_coefs(d, 0)
        except Exception:
            continue  # (the coefficient interface reads on to EOI and may stumble there; the verdict above stands)
        if np.array_equal(mine.ravel(), ref.ravel()):
            same += 1
        else:
            odd.append((it, n, mode, "differs"))
    assert not odd, (same, failed, odd[:8])
    assert same >= 250 and failed >= 60, (same, failed)


@pytest.mark.gpu
def test_arithmetic_sources_on_the_device(batch, oracle):
    """Decode (host QM decoder -> device IDCT, upsampling, colour) to the reference's pixels, and NewDecoder -> ImageOps.Transform ->
    JPEG to the reference CPU path's bytes, one image at a time and as items of a batch next to Huffman-coded ones."""
    import lilliput_amd as la

    files = _files()
    for name, data in files.items():
        got, _ = batch.decode_jpeg(data)
        want = GOLD[name]["pixels"]
        assert "%dx%dx%d:%s" % (got.shape[0], got.shape[1], got.shape[2], hashlib.sha1(got.tobytes()).hexdigest()[:16]) == want, name
        if oracle.ref() is not None:
            assert np.array_equal(got, oracle.ref_jpeg_decode(data)), name
    if oracle.ref() is None:
        return
    ops = la.ImageOps(1024)
    picks = ["arith_seq_420.jpg", "arith_prog_deep_420.jpg", "arith_seq_big_420.jpg", "arith_seq_gray.jpg", "arith_seq_444_dri3.jpg", "arith_prog_ycck.jpg"]
    for name in picks:
        d = la.Decoder(files[name])
        assert d.Description() == "JPEG"
        out = ops.Transform(d, la.ImageOptions(".jpeg", 32, 24, la.ImageOpsFit, False, {la.JpegQuality: 85}))
        d.Close()
        exp = oracle.transform_jpeg_thumbnail(files[name], 32, 24, 85, use_ref=True)
        assert _same_or_one_lsb(la, ops, oracle, files[name], out, exp, 32, 24), name
    huff = open(os.path.join(ROOT, "tests", "golden", "inputs", "coast.jpg"), "rb").read()
    items = [files[n] for n in picks] + [huff]
    res = batch.transform(items, 32, 24, quality=85)
    for it, r in zip(items, res):
        assert r.status == 0
        exp = oracle.transform_jpeg_thumbnail(it, 32, 24, 85, use_ref=True)
        assert _same_or_one_lsb(la, ops, oracle, it, r.data, exp, 32, 24)
    ops.Close()


def _same_or_one_lsb(la, ops, oracle, data, out, exp, w, h):
    """The reference path's bytes, or (fractional INTER_AREA: float taps, north_star's +-1 LSB) a pre-encode frame within one LSB of the
    oracle's that the output encodes byte-exactly."""
    if out == exp:
        return True
    d = la.Decoder(data)
    frame = la.parse_raw_frames(ops.Transform(d, la.ImageOptions(".bgra-frames", w, h, la.ImageOpsFit, False, {})))[0][0]
    d.Close()
    ref = oracle.transform_static(oracle.ref_jpeg_decode(data), oracle.jpeg_info(data)["orientation"], w, h, oracle.FIT, False)
    if frame.shape != ref.shape or np.abs(frame.astype(int) - ref.astype(int)).max() > 1:
        return False
    return out == oracle.ref_jpeg_encode(frame if frame.shape[2] > 1 else frame[:, :, 0], 85)
