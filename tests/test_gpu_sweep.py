"""Randomised sweep of the JPEG hot path: encoder settings Pillow exposes (quality 1-100, every chroma subsampling, optimised
Huffman tables, restart intervals, greyscale, odd sizes, tiny and skinny images) -> device decode and device Transform against
the oracle, bit for bit. Seeds are fixed; the sources are generated on the spot."""
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def _image(rng, h, w, gray):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 90 * np.sin(x / rng.uniform(3, 40) + c) + 40 * np.cos(y / rng.uniform(3, 40) - c) for c in range(3)], -1)
    img = base + rng.normal(0, rng.uniform(0, 25), (h, w, 3))
    if rng.random() < 0.2:
        img[rng.integers(0, h) :, :, :] = rng.choice([0, 255])  # a hard edge: large coefficients
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[:, :, 0] if gray else img


def _cases(seed, n, big=False):
    rng = np.random.default_rng(seed)
    for i in range(n):
        h, w = (int(rng.integers(1, 40)), int(rng.integers(1, 40))) if rng.random() < 0.25 else (int(rng.integers(8, 420)), int(rng.integers(8, 420)))
        if big:
            h, w = int(rng.integers(300, 1700)), int(rng.integers(300, 1700))
        if rng.random() < 0.1:
            h, w = (1, int(rng.integers(1, 300))) if rng.random() < 0.5 else (int(rng.integers(1, 300)), 1)
        gray = rng.random() < 0.15
        kw = {"quality": int(rng.choice([1, 5, 25, 50, 75, 85, 90, 95, 100])), "optimize": bool(rng.random() < 0.4)}
        if not gray:
            kw["subsampling"] = int(rng.choice([0, 1, 2]))
        if rng.random() < (0.6 if big else 0.35):
            if rng.random() < 0.5:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            else:
                kw["restart_marker_blocks"] = int(rng.integers(1, 12))
        buf = io.BytesIO()
        try:
            PIL.fromarray(_image(rng, h, w, gray)).save(buf, "JPEG", **kw)
        except OSError:  # Pillow's encoder buffer is too small for some tiny-image / many-restart combinations
            continue
        yield i, (h, w, gray, kw), buf.getvalue()


@pytest.mark.gpu
def test_random_jpegs_decode_bit_exact(batch, oracle):
    bad = []
    for i, desc, data in _cases(2024, 260):
        exp = oracle.jpeg_decode(data)
        got, _ = batch.decode_jpeg(data)
        if got.shape != exp.shape or not np.array_equal(got, exp):
            bad.append((i, desc))
    assert not bad, bad[:8]


@pytest.mark.gpu
def test_random_jpegs_transform_matches_oracle(batch, oracle):
    """Batch Transform (fused and non-fused resample paths, all resize methods the batch takes) on the same kind of sources."""
    rng = np.random.default_rng(7)
    cases = list(_cases(99, 96))
    bad = []
    for tw, th, q in ((64, 64, 85), (37, 91, 70), (200, 120, 95), (16, 16, 50)):
        res = batch.transform([c[2] for c in cases], tw, th, quality=q)
        for (i, desc, data), r in zip(cases, res):
            if r.status != 0:
                bad.append((i, desc, (tw, th), "status %d" % r.status))
                continue
            exp = oracle.transform_jpeg_thumbnail(data, tw, th, q)
            if r.data != exp:  # bit-identical except where the float area resize may differ by 1 LSB before the encoder
                a, b = oracle.jpeg_decode(r.data), oracle.jpeg_decode(exp)
                if a.shape != b.shape or np.abs(a.astype(int) - b.astype(int)).max() > 8:
                    bad.append((i, desc, (tw, th), a.shape, b.shape, int(np.abs(a.astype(int) - b.astype(int)).max()) if a.shape == b.shape else -1))
    assert not bad, (len(bad), bad[:12])


@pytest.mark.gpu
def test_random_integer_scales_every_kernel_bit_exact(batch, oracle):
    """Integer scales at random: box sizes 2 .. 34, crops that start anywhere (Fit centres them: even, odd, on and off the kernels'
    grids), every sampling and grey, every orientation, with and without normalisation -- whichever kernel takes the op (k_resample_420 /
    _small / _hv1 / _gray, the area walk with unit taps, the general kernel), the bytes are decode -> ExifTransform -> crop ->
    resizeAreaFast_ -> the reference encoder's."""
    import test_gpu_parity as P

    rng = np.random.default_rng(20260923)
    bad = []
    n = 0
    for it in range(140):
        s = int(rng.choice([2, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 16, 20, 24, 32, 34]))
        tw, th = int(rng.integers(1, 40 if s > 8 else 90)), int(rng.integers(1, 40 if s > 8 else 90))
        if rng.random() < 0.3:
            tw = th = int(rng.choice([8, 16, 32, 64]))
        extra = int(rng.integers(0, 24))
        wide = rng.random() < 0.5
        ow, oh = (tw * s + (extra if wide else 0), th * s + (0 if wide else extra))  # the ORIENTED frame: Fit crops `extra` off the longer side
        # Fit keeps the crop an exact multiple only when the aspect ratios work out; check with the oracle's own plan below
        o = int(rng.integers(1, 9))
        w, h = (oh, ow) if o >= 5 else (ow, oh)
        gray = rng.random() < 0.12
        img = _image(rng, h, w, gray)
        kw = {"quality": int(rng.choice([60, 85, 92, 100]))}
        if not gray:
            kw["subsampling"] = int(rng.choice([0, 1, 2, 2]))
        buf = io.BytesIO()
        PIL.fromarray(img).save(buf, "JPEG", **kw)
        d = P._with_exif_orientation(buf.getvalue(), o)
        for norm in (False, True):
            r = batch.transform([d], tw, th, normalize=norm, quality=85)[0]
            frame = oracle.transform_static(oracle.jpeg_decode(d), o, tw, th, oracle.FIT, norm)
            n += 1
            if r.status != 0 or (r.width, r.height) != (frame.shape[1], frame.shape[0]) or r.data != oracle.jpeg_encode(frame, 85):
                bad.append((it, s, (w, h), (tw, th), o, norm, gray, kw, r.status))
    assert not bad, (len(bad), n, bad[:10])


@pytest.mark.gpu
def test_narrow_images_use_plain_chroma_replication(batch, oracle):
    """jdsample.c picks the fancy h2v1 / h2v2 upsamplers only when the chroma plane is more than two samples wide: images up to
    four pixels wide get plain replication (the oracle is checked against the real libjpeg on the same files in
    tests/test_oracle_golden.py)."""
    rng = np.random.default_rng(1)
    for sub in (2, 1, 0):
        for w in range(1, 10):
            for h in (1, 2, 3, 5, 9, 40):
                buf = io.BytesIO()
                PIL.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(buf, "JPEG", quality=90, subsampling=sub)
                data = buf.getvalue()
                got, _ = batch.decode_jpeg(data)
                assert np.array_equal(got, oracle.jpeg_decode(data)), (sub, w, h)
                r = batch.transform([data], max(1, w // 2), max(1, h // 2), quality=90)[0]
                assert r.status == 0 and r.data == oracle.transform_jpeg_thumbnail(data, max(1, w // 2), max(1, h // 2), 90), (sub, w, h)


def _exotic():
    import os

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs_exotic")
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d))}


def test_oracle_decodes_exotic_fixtures_like_the_reference_library(oracle):
    """Files written by the reference's own libjpeg-turbo compressor with settings Pillow cannot produce (4:4:0, RGB-in-JPEG with
    an Adobe marker, YCbCr without JFIF, 16-bit quantisation tables / SOF1, odd restart intervals); the recorded digests are of
    the pixels the reference's libjpeg decodes (tests/golden/make_exotic_jpegs.py)."""
    import hashlib
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exotic_golden.json")))
    for name, data in _exotic().items():
        px = oracle.jpeg_decode(data)
        assert "%dx%dx%d:%s" % (px.shape[0], px.shape[1], px.shape[2], hashlib.sha1(px.tobytes()).hexdigest()[:16]) == gold[name], name


@pytest.mark.gpu
def test_exotic_fixtures_on_device(batch, oracle):
    import hashlib
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exotic_golden.json")))
    files = _exotic()
    for name, data in files.items():
        got, _ = batch.decode_jpeg(data)
        exp = oracle.jpeg_decode(data)
        assert got.shape == exp.shape and np.array_equal(got, exp), name
        assert "%dx%dx%d:%s" % (got.shape[0], got.shape[1], got.shape[2], hashlib.sha1(got.tobytes()).hexdigest()[:16]) == gold[name], name
    import lilliput_amd as la

    ops = la.ImageOps(512)  # the same files through NewDecoder / Header / ImageOps.Transform, as lilliput's callers drive them
    for name in ("cmyk_adobe_420k_dri3.jpg", "ycck_adobe_420.jpg", "seq_two_scans_422.jpg", "prog_deep_420.jpg"):
        d = la.Decoder(files[name])
        h = d.Header()
        exp = oracle.jpeg_decode(files[name])
        assert (h["height"], h["width"]) == exp.shape[:2] and d.Description() == "JPEG", name
        out = ops.Transform(d, la.ImageOptions(".jpeg", 24, 24, la.ImageOpsFit, False, {la.JpegQuality: 85}))
        d.Close()
        assert out == oracle.transform_jpeg_thumbnail(files[name], 24, 24, 85), name
    ops.Close()
    names = list(files)
    for tw, th in ((20, 20), (13, 31)):
        res = batch.transform([files[n] for n in names], tw, th, quality=85)
        for n, r in zip(names, res):
            assert r.status == 0 and r.data == oracle.transform_jpeg_thumbnail(files[n], tw, th, 85), (n, tw, th)
