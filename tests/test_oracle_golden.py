"""Pins the oracle (CPU restatement) against the committed golden vectors: pixels produced by the reference's
own libjpeg-turbo 3.1.0, and the ThumbHash known answers of /root/reference/thumbhash_test.go:63-81.
Runs on CPU."""
import base64
import hashlib
import os

import numpy as np
import pytest


def test_decode_matches_reference_libjpeg_pixels(oracle, golden, fixture_bytes):
    for name, g in golden.items():
        px = oracle.jpeg_decode(fixture_bytes[name])
        assert px.shape == (g["height"], g["width"], g["channels"]), name
        assert hashlib.sha256(px.tobytes()).hexdigest() == g["pixels_sha256"], name
        assert oracle.jpeg_info(fixture_bytes[name])["orientation"] == g["orientation"], name


def test_thumbhash_known_answers(oracle, golden, fixture_bytes):
    """thumbhash_test.go: Transform(NoResize, NormalizeOrientation) -> .thumbhash; pins decode + orientation."""
    checked = 0
    for name, g in golden.items():
        if not g["thumbhash_b64"]:
            continue
        data = fixture_bytes[name]
        info = oracle.jpeg_info(data)
        frame = oracle.transform_static(oracle.jpeg_decode(data), info["orientation"], info["width"], info["height"], oracle.NO_RESIZE, True)
        assert base64.b64encode(oracle.thumbhash(frame)).decode() == g["thumbhash_b64"], name
        checked += 1
    assert checked == 9


def test_thumbnail_bytes_match_reference_path(oracle, golden, fixture_bytes):
    """configs[0]: Fit 256x256 q85 through the restatement == through the reference's libjpeg (golden hash)."""
    for name, g in golden.items():
        out = oracle.transform_jpeg_thumbnail(fixture_bytes[name], 256, 256, 85)
        assert len(out) == g["thumb256_q85_len"], name
        assert hashlib.sha256(out).hexdigest() == g["thumb256_q85_sha256"], name


def test_restatement_vs_reference_library_when_present(oracle, fixture_bytes):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    for name, data in fixture_bytes.items():
        assert np.array_equal(oracle.jpeg_decode(data), oracle.ref_jpeg_decode(data)), name
        info = oracle.jpeg_info(data)
        for c in range(info["ncomp"]):
            assert np.array_equal(oracle.jpeg_decode_coefs(data, c), oracle.ref_jpeg_decode_coefs(data, c)), (name, c)
    px = oracle.jpeg_decode(fixture_bytes["large-sunrise.jpg"])
    for crop in (px[:256, :256], px[100:343, 50:300], px[:17, :33], px[:1, :1], px[:243, :250], px[:200, :123, 1]):
        for q in (85, 50, 95, 10, 100):
            assert oracle.jpeg_encode(crop, q) == oracle.ref_jpeg_encode(crop, q)


def test_area_resize_branches_and_exact_cases(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    d, br = oracle.resize_area(src, 96, 64)
    assert br == 0 and np.array_equal(d, src)
    d, br = oracle.resize_area(src, 48, 32)  # 2x2 fast path: (a+b+c+d+2)>>2
    assert br == 1
    s = src.astype(np.int32)
    exp = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(d, exp.astype(np.uint8))
    d, br = oracle.resize_area(src, 24, 16)  # 4x4: round-half-even of sum/16
    assert br == 1
    sums = s.reshape(16, 4, 24, 4, 3).sum(axis=(1, 3))
    assert np.array_equal(d, np.rint(sums.astype(np.float32) * np.float32(1 / 16)).astype(np.uint8))
    d, br = oracle.resize_area(src, 50, 30)
    assert br == 2
    const = np.full((37, 53, 3), 77, np.uint8)
    d, br = oracle.resize_area(const, 20, 10)
    assert br == 2 and np.all(np.abs(d.astype(int) - 77) <= 0)
    d, br = oracle.resize_area(src, 200, 32)
    assert br == 3


def test_orientation_is_a_permutation(oracle):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    exp = {1: src, 2: src[:, ::-1], 3: src[::-1, ::-1], 4: src[::-1], 5: src.transpose(1, 0, 2), 6: src.transpose(1, 0, 2)[:, ::-1],
           7: src.transpose(1, 0, 2)[::-1, ::-1], 8: src.transpose(1, 0, 2)[::-1]}
    for o, e in exp.items():
        assert np.array_equal(oracle.orientation_transform(src, o), e), o


def test_go_control_logic(oracle):
    # ops.go:243-255
    assert oracle.calculate_expected_size(1300, 1942, 256, 256) == (256, 256)
    assert oracle.calculate_expected_size(800, 297, 512, 512) == (297, 297)
    assert oracle.calculate_expected_size(100, 75, 400, 300) == (100, 75)
    assert oracle.calculate_expected_size(100, 75, 50, 300) == (50, 300)
    # SURVEY.md App. A geometries (opencv.go:331-363)
    assert oracle.fit_crop_rect(4096, 4096, 256, 256) == (0, 0, 4096, 4096)
    assert oracle.fit_crop_rect(1300, 1942, 256, 256) == (0, 321, 1300, 1300)
    assert oracle.fit_crop_rect(800, 297, 297, 297) == (251, 0, 297, 297)
    assert oracle.fit_crop_rect(480, 270, 128, 128) == (105, 0, 270, 270)
    assert oracle.fit_crop_rect(8192, 6144, 256, 256) == (1024, 0, 6144, 6144)


def test_restatement_vs_reference_library_on_generated_files(oracle):
    """Randomised sources (every Pillow encoder setting, tiny and skinny images, restart intervals, optimised tables) and
    progressive files: decode bit-exact against the reference's libjpeg-turbo; random pixels: encode byte-identical."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    import io

    from PIL import Image

    import test_gpu_sweep as T

    for seed in (11, 12, 13):
        for i, desc, data in T._cases(seed, 120):
            assert np.array_equal(oracle.jpeg_decode(data), oracle.ref_jpeg_decode(data)), (seed, i, desc)
    rng = np.random.default_rng(1)
    for sub in (2, 1, 0):
        for w in range(1, 10):
            for h in (1, 2, 3, 9):
                b = io.BytesIO()
                Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(b, "JPEG", quality=90, subsampling=sub)
                assert np.array_equal(oracle.jpeg_decode(b.getvalue()), oracle.ref_jpeg_decode(b.getvalue())), (sub, w, h)
    for (h, w, sub, q, extra) in ((64, 64, 2, 85, {}), (123, 77, 2, 90, {}), (200, 333, 0, 75, {}), (97, 150, 1, 95, {"optimize": True}), (64, 80, 2, 85, {"restart_marker_rows": 1})):
        b = io.BytesIO()
        y, x = np.mgrid[0:h, 0:w]
        px = np.clip(np.stack([128 + 90 * np.sin(x / 17.0 + c) + 30 * np.cos(y / 9.0) for c in range(3)], -1) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(px).save(b, "JPEG", quality=q, progressive=True, subsampling=sub, **extra)
        assert np.array_equal(oracle.jpeg_decode(b.getvalue()), oracle.ref_jpeg_decode(b.getvalue())), ("progressive", h, w, sub)
    for it in range(200):
        cn = int(rng.choice([1, 3, 4]))
        px = rng.integers(0, 256, (int(rng.integers(1, 200)), int(rng.integers(1, 200)), cn), dtype=np.uint8)
        q = int(rng.choice([1, 10, 50, 85, 100]))
        src = px[:, :, 0] if cn == 1 else px
        assert oracle.jpeg_encode(src, q) == oracle.ref_jpeg_encode(src, q), (px.shape, q)


def test_restatement_vs_reference_library_on_truncated_baseline_files(oracle, fixture_bytes):
    """jdhuff.c decode_mcu's end-of-data rule for baseline files -- the MCU at hand is finished on zero bits, the following ones are
    left untouched (flat grey) until a restart marker is found again -- restated in jpeg_oracle.c decode_coefs: pixels and coefficients
    against the reference's own libjpeg-turbo on truncated fixtures (with and without restart intervals, colour and grey)."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built")
    n = 0
    for name in ("sunrise.jpg", "ferry_sunset.jpg", "firefox-gray.jpg", "coast.jpg", "large-sunrise.jpg"):
        data = fixture_bytes[name]
        for frac in (0.1, 0.3, 0.5, 0.8, 0.95, 0.999):
            cut = data[: int(len(data) * frac)]
            try:
                want = oracle.ref_jpeg_decode(cut)
            except Exception:
                continue   # cut inside the headers: no image either way
            assert np.array_equal(oracle.jpeg_decode(cut), want), (name, frac)
            for c in range(oracle.jpeg_info(cut)["ncomp"]):
                assert np.array_equal(oracle.jpeg_decode_coefs(cut, c), oracle.ref_jpeg_decode_coefs(cut, c)), (name, frac, c)
            n += 1
    assert n >= 25


def test_restatement_vs_reference_library_on_damaged_progressive_files(oracle):
    """Bit-flipped progressive files (entropy data, scan headers, the tables between scans): the oracle accepts and rejects exactly
    what the reference's libjpeg-turbo does and decodes the same coefficients -- the end of a scan's data (insufficient_data), codes
    that match no table entry (17 bits, a zero symbol), coefficient indices past 63, missing tables (no Annex-K fallback in
    jdphuff.c), duplicate SOI / SOF, DAC contents. Pixels are compared where the coefficients stay in range: libjpeg's SIMD IDCT
    wraps 16-bit products of absurd coefficients and smooths blocks of a file that lost a scan; neither is restated."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    import test_progressive as TP

    rng = np.random.default_rng(3)
    n_ok = n_err = n_px = 0
    for i, desc, data in TP._cases(17, 30, lo=16, hi=120):
        sos = data.index(b"\xff\xda")
        for k in range(20):
            d = bytearray(data)
            d[int(rng.integers(sos - 30, len(d) - 2))] ^= 1 << int(rng.integers(0, 8))
            d = bytes(d)
            try:
                mine = [oracle.jpeg_decode_coefs(d, c) for c in range(1 if desc[2] else 3)]
            except Exception:
                mine = None
            try:
                ref = [oracle.ref_jpeg_decode_coefs(d, c) for c in range(1 if desc[2] else 3)]
            except Exception:
                ref = None
            assert (mine is None) == (ref is None), (i, k, desc)
            if mine is None:
                n_err += 1
                continue
            n_ok += 1
            assert all(np.array_equal(a, b) for a, b in zip(mine, ref)), (i, k, desc)
            if max(int(np.abs(a).max()) for a in mine) < 1024:
                a, b = oracle.jpeg_decode(d), oracle.ref_jpeg_decode(d)
                n_px += int(np.array_equal(a, b))
    assert n_ok > 300 and n_err > 20 and n_px > 0.97 * n_ok, (n_ok, n_err, n_px)


def test_cmyk_to_bgr_rule_is_opencvs(oracle):
    """Four-component JPEGs: cv::JpegDecoder asks libjpeg for CMYK rows and runs them through icvCvt_CMYK2BGR_8u_C4C3R. The rule the
    oracle (and the device kernel) states, x -> k - ((255 - x) * k >> 8), against the compiled function out of the reference's own
    libopencv_imgcodecs.a on 200 k random quadruples and every (x, k) pair."""
    cv = oracle.ref_cv()
    if cv is None:
        pytest.skip("oracle/_ref/librefcv.so not built (needs /root/reference)")
    import ctypes as C

    rng = np.random.default_rng(0)
    grid = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).reshape(-1, 2)
    cmyk = np.concatenate([rng.integers(0, 256, (200000, 4)), np.stack([grid[:, 0], 255 - grid[:, 0], grid[:, 0] // 2, grid[:, 1]], -1)]).astype(np.uint8)
    got = np.zeros((len(cmyk), 3), np.uint8)
    cv.ref_cv_cmyk2bgr(cmyk.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), len(cmyk))
    c = cmyk.astype(np.int32)
    k = c[:, 3:4]
    exp = (k - (((255 - c[:, :3]) * k) >> 8))[:, ::-1].astype(np.uint8)
    assert np.array_equal(got, exp)


def test_cpu_path_animated_worker_loop_equals_the_stagewise_reference_path(oracle):
    """bench.py's configs[3] baseline (oracle/cpu_path.c lo_path_transform_anim: giflib + restated compositing / libwebp playback ->
    Fit per frame -> the reference's animation writer, all in C) writes the bytes the stage-by-stage Python path writes."""
    if oracle.ref_gif() is None or oracle.ref_webp() is None:
        pytest.skip("oracle/_ref/librefgif.so / librefwebp.so not built")
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    srcs = [open(os.path.join(gold, "inputs_gif", "party-discord.gif"), "rb").read(), open(os.path.join(gold, "inputs_webp", "big_buck_bunny_720_5s.webp"), "rb").read()]
    want = [oracle.transform_animated_to_webp(d, 128, 128, 75) for d in srcs]
    r = oracle.cpu_path_run(srcs, 128, 128, threads=2, jobs=2, keep=True, webp_quality=75, animated=True)
    assert r["ok"] == 2 and r["frames"] == sum(w[1] for w in want)
    assert [o == w[0] for o, w in zip(r["outputs"], want)] == [True, True]
