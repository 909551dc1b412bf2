"""Damaged, cut and padded JPEG streams: the product against the reference's OWN decoder.

The truth is oracle.ref_cv_jpeg_decode: cv::JpegDecoder's object code out of the reference's libopencv_imgcodecs.a linked with the
reference's libjpeg.a (oracle/ref_jpegcv_driver.cpp) -- what opencv_decoder_read_header + opencv_decoder_read_data
(/root/reference/opencv.cpp:126-171) answer for a buffer. Two things it settles that jpeg_mem_src (oracle.ref_jpeg_decode) cannot:
  * a stream that RUNS OUT OF BYTES fails (OpenCV's source manager suspends libjpeg; readData returns false, the Go layer reports
    ErrDecodingFailed, opencv.go:828-831), while a stream that stops AT a marker decodes with a warning: zero bits, grey MCUs;
  * libjpeg reads up to eight bytes ahead of the bits it needs, so whether the last MCUs of a buffer without EOI still decode depends
    on where its refills fell (lilliput_amd/csrc/lp_jbits.h restates the holding register for that).
Restart markers with wrong numbers, missing or surplus ones, byte pairs that only look like markers, real markers inside the data:
jdmarker.c jpeg_resync_to_restart / next_marker, restated in lp_jbits.h. The device decoder must recognise every stream it would
decode differently (state error bits, lp_types.h) and hand it to the serial route.
CPU half: the serial route's coefficients and verdicts, and the device's lane logic through tests/emu. GPU half: pixels and
thumbnails through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import jpeg_damage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_refs(oracle):
    if oracle.ref() is None or oracle.ref_cvjpeg() is None:
        pytest.skip("oracle/_ref (the reference's libjpeg.a / cv::JpegDecoder drivers) not built")


def _host_coefs(L, data, comp, force):
    """Coefficients of the serial route (host threads; test access, no device). rc 0, -2 = the reference's decoder fails on this file,
    -1 = the header walk refuses it."""
    a = np.frombuffer(bytes(data), np.uint8)
    out = np.zeros(1 << 23 if len(data) > (1 << 17) else 1 << 21, np.int16)
    bw, bh = C.c_int(), C.c_int()
    rc = L.lilliput_hip_progressive_coefs_host(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(comp), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                                               C.byref(bw), C.byref(bh), C.c_int(-1 if force else 1))
    if rc not in (0, -2):
        return rc, None
    return rc, out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64).copy()


def _check_serial(hip_lib, oracle, cases, force):
    verdict_bad, coef_bad, ok, failed, unchecked = [], [], 0, 0, 0
    for tag, data in cases:
        cv = oracle.ref_cv_jpeg_decode(data)
        rc, _ = _host_coefs(hip_lib, data, 0, force)
        if (cv is None) != (rc != 0):
            verdict_bad.append((tag, cv is None, rc))
            continue
        if cv is None:
            failed += 1
            continue
        try:  # jpeg_read_coefficients reads on to EOI and may stumble there where read_data has long returned: then the verdict stands alone
            want = [oracle.ref_jpeg_decode_coefs(data, c) for c in range(3 if cv.shape[2] == 3 else 1)]
        except ValueError:
            unchecked += 1
            continue
        if all(np.array_equal(_host_coefs(hip_lib, data, c, force)[1], w) for c, w in enumerate(want)):
            ok += 1
        else:
            coef_bad.append(tag)
    return verdict_bad, coef_bad, ok, failed, unchecked


def test_reference_decoder_fails_what_runs_out_of_bytes_and_paints_grey_what_stops_at_a_marker(oracle, fixture_bytes):
    _need_refs(oracle)
    for name in ("sunrise.jpg", "ferry_sunset.jpg", "large-sunrise.jpg"):
        data = fixture_bytes[name]
        assert np.array_equal(oracle.ref_cv_jpeg_decode(data), oracle.ref_jpeg_decode(data)), name
        for frac in (0.4, 0.9, 0.999):
            cut = data[: int(len(data) * frac)]
            assert oracle.ref_cv_jpeg_decode(cut) is None, (name, frac)
            closed = oracle.ref_cv_jpeg_decode(cut + b"\xff\xd9")
            assert closed is not None and np.array_equal(closed, oracle.ref_jpeg_decode(cut)), (name, frac)  # jpeg_mem_src fakes that EOI
        assert oracle.ref_cv_jpeg_decode(data[:-2]) is None, name  # even the EOI alone: the read-ahead of the last MCUs finds no byte


def test_serial_route_equals_the_reference_decoder_on_damaged_baseline_streams(hip_lib, oracle):
    _need_refs(oracle)
    cases = jpeg_damage.cases(5)
    verdict_bad, coef_bad, ok, failed, unchecked = _check_serial(hip_lib, oracle, cases, True)
    assert not verdict_bad and not coef_bad, (verdict_bad[:6], coef_bad[:6])
    assert len(cases) >= 600 and ok >= 400 and failed >= 100, (len(cases), ok, failed, unchecked)


def test_serial_route_equals_the_reference_decoder_on_damaged_progressive_streams(hip_lib, oracle):
    _need_refs(oracle)
    cases = jpeg_damage.cases(6, per_base=80, progressive=True)
    verdict_bad, coef_bad, ok, failed, unchecked = _check_serial(hip_lib, oracle, cases, False)
    assert not verdict_bad and not coef_bad, (verdict_bad[:6], coef_bad[:6])
    assert len(cases) >= 300 and ok >= 80 and failed >= 60, (len(cases), ok, failed, unchecked)


def test_end_of_buffer_rule_on_real_files(hip_lib, oracle, fixture_bytes):
    """Files large enough for libjpeg's fast Huffman path (more than 512 bytes per block of the MCU in the buffer), so that the hand-over
    from decode_mcu_fast's refill pattern to decode_mcu_slow's is part of what decides the verdicts at the end of the buffer."""
    _need_refs(oracle)
    n = 0
    for name in ("large-sunrise.jpg", "ferry_sunset.jpg", "coast.jpg", "firefox-gray.jpg"):
        base = fixture_bytes[name]
        pattern_ref, pattern_ours = "", ""
        for tag, data in jpeg_damage.tails(base, upto=10):
            cv = oracle.ref_cv_jpeg_decode(data)
            rc, _ = _host_coefs(hip_lib, data, 0, True)
            pattern_ref += "1" if cv is not None else "0"
            pattern_ours += "1" if rc == 0 else "0"
            n += 1
        assert pattern_ref == pattern_ours, (name, pattern_ref, pattern_ours)
        assert "1" in pattern_ref and "0" in pattern_ref, name
    assert n == 120


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libemu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(d, "emu_huff.cpp"), os.path.join(ROOT, "lilliput_amd", "csrc", "lp_jpeg_parse.cpp")], check=True)
    return C.CDLL(so)


def _emu_coefs(emu, data, S, comp):
    a = np.frombuffer(bytes(data), np.uint8)
    out = np.zeros(1 << 21, np.int16)
    bw, bh, rounds, nsub, hits = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = emu.emu_decode_coefs(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_uint32(S), C.c_uint32(64), C.c_int(comp), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                              C.byref(bw), C.byref(bh), C.byref(rounds), C.byref(nsub), C.byref(hits))
    return (rc, None) if rc else (0, out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64).copy())


def test_device_lane_logic_hands_over_every_stream_it_would_decode_differently(emu, oracle):
    """tests/emu runs the kernels' lane logic (lp_huff_core.h, lp_unstuff_core.h) with the device's error bits: whatever it ACCEPTS it
    must decode to the reference's coefficients; everything else goes to the serial route (tested above). The restart checks at work:
    marker numbers and count (k_unstuff_*), blocks per interval and boundaries inside a block (lp_write_pass), a stream that ends
    inside a block."""
    _need_refs(oracle)
    rng = np.random.default_rng(11)
    accepted = rejected = 0
    wrong = []
    for tag, data in jpeg_damage.cases(7, per_base=70):
        S = int(rng.choice([256, 512, 1024, 4096]))
        rc, got = _emu_coefs(emu, data, S, 0)
        if rc:
            assert rc in (-11, -13, -16, -17, -1, -2, -3), (tag, rc)  # marker in the data, short of blocks, irregular restarts, serial route by the parser, header verdicts
            rejected += 1
            continue
        accepted += 1
        cv = oracle.ref_cv_jpeg_decode(data)
        if cv is None:
            wrong.append((tag, "reference fails"))
            continue
        for c in range(3 if cv.shape[2] == 3 else 1):
            if not np.array_equal(oracle.ref_jpeg_decode_coefs(data, c), _emu_coefs(emu, data, S, c)[1]):
                wrong.append((tag, c))
                break
    assert not wrong, wrong[:8]
    assert accepted >= 150 and rejected >= 300, (accepted, rejected)


def test_simd_idct_restatement_equals_the_reference_decoder_on_hostile_coefficients(oracle):
    """oracle/jpeg_oracle.c lo_idct_islow_simd -- libjpeg-turbo's x86-64 SIMD jpeg_idct_islow restated (16-bit lanes: wrapping
    dequantisation and sums, saturating packs, the in0 << 2 shortcut of a block without AC coefficients) -- against the reference's
    own decoder: damaged files at quality 1-5 (quantisers up to 255) produce dequantised values far outside 16 bits. The restatement's
    back half on the library's own coefficients must give the library's pixels on every file, where the C arithmetic does not."""
    import io

    from PIL import Image

    from lilliput_amd import synth

    _need_refs(oracle)
    rng = np.random.default_rng(3)
    rgb = synth.synth_rgb(7, 256)[:96, :128]
    files = [d for _, d in jpeg_damage.cases(5, per_base=20)]
    for q in (1, 2, 3, 5):
        for sub in (0, 2):
            b = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(rgb)).save(b, "JPEG", quality=q, subsampling=sub)
            files += [jpeg_damage.damage(b.getvalue(), rng, (0, 2, 9)[it % 3]) for it in range(30)]
    n = c_differs = 0
    L = oracle.lib()
    for d in files:
        cv = oracle.ref_cv_jpeg_decode(d)
        if cv is None:
            continue
        try:
            co = [oracle.ref_jpeg_decode_coefs(d, c) for c in range(3 if cv.shape[2] == 3 else 1)]
        except ValueError:
            continue
        n += 1
        L.lo_set_idct_simd(0)
        try:
            c_differs += not np.array_equal(oracle.jpeg_pixels_from_coefs(d, co), cv)
        finally:
            L.lo_set_idct_simd(1)
        assert np.array_equal(oracle.jpeg_pixels_from_coefs(d, co), cv)
    assert n >= 300 and c_differs >= 40, (n, c_differs)


# ------------------------------------------------------------------------------------------------ GPU half
def _hostile(seed=3):
    """Damaged quality-1..5 files: quantisers up to 255, dequantised coefficients far outside 16 bits."""
    import io

    from PIL import Image

    from lilliput_amd import synth

    rng = np.random.default_rng(seed)
    rgb = synth.synth_rgb(7, 256)[:96, :128]
    out = []
    for q in (1, 2, 3, 5):
        for sub in (0, 2):
            b = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(rgb)).save(b, "JPEG", quality=q, subsampling=sub)
            out += [("q%d/%d/%d" % (q, sub, it), jpeg_damage.damage(b.getvalue(), rng, (0, 2, 9)[it % 3])) for it in range(30)]
            b = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(rgb)).save(b, "JPEG", quality=q, subsampling=sub, progressive=True)
            out += [("q%dp/%d/%d" % (q, sub, it), jpeg_damage.damage(b.getvalue(), rng, (0, 9)[it % 2])) for it in range(10)]
    return out


@pytest.mark.gpu
def test_hostile_coefficients_decode_like_the_reference_decoder_on_the_device(batch, oracle):
    """k_idct computes libjpeg-turbo's SIMD lane arithmetic: on damaged quality-1..5 files (a third of which the C arithmetic gets
    differently) the product's pixels are the reference decoder's, baseline (fast path with its fall-back) and progressive (exact path)."""
    import lilliput_amd

    _need_refs(oracle)
    bad, ok, failed = [], 0, 0
    for tag, data in _hostile():
        cv = oracle.ref_cv_jpeg_decode(data)
        try:
            px, _ = batch.decode_jpeg(data)
        except lilliput_amd.LilliputError as e:
            if cv is not None or e.code not in (1, 2):
                bad.append((tag, "fails with", e.code, cv is None))
            else:
                failed += 1
            continue
        if cv is None:
            bad.append((tag, "decodes, the reference fails"))
        elif np.array_equal(px, cv):
            ok += 1
        else:
            bad.append((tag, "pixels differ", int((px != cv).any(axis=2).sum())))
    assert not bad, (len(bad), bad[:8])
    assert ok >= 250, (ok, failed)



def _reference_pixels(oracle, data):
    """(pixels of the reference's decoder or None, None). Until round 5's end a second answer was accepted where damaged bits produce
    dequantised coefficients a real image cannot have -- libjpeg-turbo's SIMD IDCT (x86-64: 16-bit lanes, wrapping adds, saturating
    packs) and its C IDCT then differ; k_idct now computes the SIMD routine's lane arithmetic (lp_kernels_decode.hip idct_*_exact, the
    fast path's conditions), so the reference's pixels are the only answer."""
    return oracle.ref_cv_jpeg_decode(data), None


@pytest.mark.gpu
def test_damaged_streams_decode_like_the_reference_decoder_on_the_device(batch, oracle):
    """>= 600 damaged / cut / padded baseline streams through lilliput_hip_decode_jpeg (what opencv_decoder_read_data does): the
    reference's pixels, or ErrDecodingFailed exactly where the reference fails."""
    import lilliput_amd

    _need_refs(oracle)
    cases = jpeg_damage.cases(5)
    bad, ok, failed, c_arith = [], 0, 0, 0
    for tag, data in cases:
        cv, alt = _reference_pixels(oracle, data)
        try:
            px, _ = batch.decode_jpeg(data)
        except lilliput_amd.LilliputError as e:
            if cv is not None or e.code not in (1, 2):
                bad.append((tag, "fails with", e.code, cv is None))
            else:
                failed += 1
            continue
        if cv is None:
            bad.append((tag, "decodes, the reference fails"))
        elif np.array_equal(px, cv):
            ok += 1
        elif alt is not None and np.array_equal(px, alt):
            c_arith += 1
        else:
            bad.append((tag, "pixels differ", int((px != cv).any(axis=2).sum())))
    assert not bad, bad[:8]
    assert len(cases) >= 600 and ok >= 400 and failed >= 100 and c_arith <= ok // 20, (ok, failed, c_arith)
    # the engine is healthy afterwards
    base = jpeg_damage.bases()[0][1]
    assert np.array_equal(batch.decode_jpeg(base)[0], oracle.ref_cv_jpeg_decode(base))


@pytest.mark.gpu
def test_damaged_progressive_streams_decode_like_the_reference_decoder_on_the_device(batch, oracle):
    import lilliput_amd

    _need_refs(oracle)
    bad, ok, failed = [], 0, 0
    for tag, data in jpeg_damage.cases(6, per_base=40, progressive=True):
        cv, alt = _reference_pixels(oracle, data)
        try:
            px, _ = batch.decode_jpeg(data)
        except lilliput_amd.LilliputError as e:
            if cv is not None or e.code not in (1, 2):
                bad.append((tag, "fails with", e.code, cv is None))
            else:
                failed += 1
            continue
        if cv is None or not (np.array_equal(px, cv) or (alt is not None and np.array_equal(px, alt))):
            bad.append((tag, "decodes" if cv is None else "pixels differ"))
        else:
            ok += 1
    assert not bad, bad[:8]
    assert ok >= 40 and failed >= 30, (ok, failed)


@pytest.mark.gpu
def test_damaged_streams_as_items_of_a_batch(batch, oracle):
    """The same streams as items of ONE lilliput_hip_batch_transform call next to intact ones: every status and every thumbnail byte
    is what the reference CPU path gives (cv::JpegDecoder -> Fit -> libjpeg encode), twice (nothing depends on stale device memory)."""
    _need_refs(oracle)
    cases = jpeg_damage.cases(9, per_base=40)
    items = [d for _, d in cases]
    want = []
    for _, d in cases:
        cv, alt = _reference_pixels(oracle, d)
        want.append(None if cv is None else [oracle.jpeg_encode(oracle.transform_static(x, 1, 64, 64, oracle.FIT, False), 80) for x in (cv, alt) if x is not None])
    for _ in range(2):
        res = batch.transform(items, 64, 64, quality=80)
        bad = []
        for (tag, _), r, w in zip(cases, res, want):
            if w is None:
                if r.status not in (1, 2):
                    bad.append((tag, "status", r.status, "the reference fails"))
            elif r.status != 0 or r.data not in w:
                bad.append((tag, "status", r.status, "bytes differ" if r.status == 0 else ""))
        assert not bad, bad[:8]
    assert sum(w is not None for w in want) >= 250 and sum(w is None for w in want) >= 80
