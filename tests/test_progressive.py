"""Progressive (SOF2) JPEG sources -- SURVEY.md section 8 row n3. The reference decodes them through libjpeg-turbo's jdphuff.c; the
device path runs every scan as one lane (lilliput_amd/csrc/lp_prog_core.h) into an int16 coefficient arena and then shares the
IDCT / upsampling / colour / resample / encode stages with baseline files. Checked against the oracle, which is pinned against the
reference's libjpeg on progressive files in tests/test_oracle_golden.py and tests/test_gpu_sweep.py (recorded digests)."""
import ctypes as C
import io
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _photo(rng, h, w, gray):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 90 * np.sin(x / rng.uniform(3, 40) + c) + 40 * np.cos(y / rng.uniform(3, 40) - c) for c in range(3)], -1)
    img = np.clip(img + rng.normal(0, rng.uniform(0, 25), (h, w, 3)), 0, 255).astype(np.uint8)
    return img[:, :, 0] if gray else img


def _cases(seed, n, lo=1, hi=300):
    rng = np.random.default_rng(seed)
    for i in range(n):
        h, w = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        gray = rng.random() < 0.15
        kw = {"quality": int(rng.choice([1, 10, 50, 75, 85, 95, 100])), "optimize": bool(rng.random() < 0.5), "progressive": True}
        if not gray:
            kw["subsampling"] = int(rng.choice([0, 1, 2]))
        buf = io.BytesIO()
        try:
            PIL.fromarray(_photo(rng, h, w, gray)).save(buf, "JPEG", **kw)
        except OSError:  # Pillow's encoder buffer is too small for some size / quality combinations
            continue
        yield i, (h, w, gray, kw), buf.getvalue()


def _exotic_progressive():
    d = os.path.join(ROOT, "tests", "golden", "inputs_exotic")
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d)) if n.startswith("prog_")}


# ------------------------------------------------------------------------------------------ CPU: parser + lane logic
@pytest.fixture(scope="module")
def emu():
    import subprocess

    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libemu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(d, "emu_huff.cpp"),
                    os.path.join(ROOT, "lilliput_amd", "csrc", "lp_jpeg_parse.cpp")], check=True)
    return C.CDLL(so)


def _emu_coefs(emu, data, comp):
    a = np.frombuffer(data, np.uint8)
    out = np.zeros(1 << 23, np.int16)
    bw, bh, ns = C.c_int(), C.c_int(), C.c_int()
    rc = emu.emu_decode_coefs_progressive(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(comp), out.ctypes.data_as(C.c_void_p),
                                          C.c_size_t(out.size), C.byref(bw), C.byref(bh), C.byref(ns))
    assert rc == 0, rc
    return out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64), ns.value


def test_forced_scan_walk_of_short_baseline_files_matches_libjpeg(emu, oracle):
    """A baseline file the device decoder cannot finish (too few blocks) is parsed again with `force_scans` and decoded by the serial
    scan decoder: lp_prog_core.h's sequential walk with libjpeg's end-of-data rule (zero bits for the MCU at hand, the following MCUs
    untouched). Coefficients against the reference's own libjpeg-turbo on truncated fixtures (CPU: the lane logic through tests/emu)."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built")
    fix = os.path.join(ROOT, "tests", "golden", "inputs")
    n = 0
    for name in ("sunrise.jpg", "ferry_sunset.jpg", "firefox-gray.jpg", "coast.jpg"):
        data = open(os.path.join(fix, name), "rb").read()
        for frac in (0.3, 0.5, 0.8, 0.95, 1.0):
            cut = data[: int(len(data) * frac)]
            for c in range(oracle.jpeg_info(cut)["ncomp"]):
                a = np.frombuffer(cut, np.uint8)
                out = np.zeros(1 << 23, np.int16)
                bw, bh, ns = C.c_int(), C.c_int(), C.c_int(-1)   # -1: force the scan-by-scan walk
                rc = emu.emu_decode_coefs_progressive(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(c), out.ctypes.data_as(C.c_void_p),
                                                      C.c_size_t(out.size), C.byref(bw), C.byref(bh), C.byref(ns))
                assert rc == 0, (name, frac, rc)
                got = out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64)
                assert np.array_equal(got, oracle.ref_jpeg_decode_coefs(cut, c)), (name, frac, c)
                n += 1
    assert n >= 50


def test_scan_lane_logic_reproduces_oracle_coefficients(emu, oracle):
    """lp_prog_core.h run serially on the CPU (tests/emu): DC / AC first and refinement scans, EOB runs, correction bits, restart
    intervals, interleaved and single-component scans -- coefficient for coefficient what the oracle (jdphuff.c restated) yields."""
    n = 0
    for name, data in _exotic_progressive().items():
        for c in range(oracle.jpeg_info(data)["ncomp"]):
            got, ns = _emu_coefs(emu, data, c)
            assert ns >= 2 and np.array_equal(got, oracle.jpeg_decode_coefs(data, c)), (name, c)
            n += 1
    for i, desc, data in _cases(5, 40, hi=200):
        for c in range(1 if desc[2] else 3):
            got, _ = _emu_coefs(emu, data, c)
            assert np.array_equal(got, oracle.jpeg_decode_coefs(data, c)), (i, desc, c)
            n += 1
    assert n > 100


def _host_coefs(hip_lib, data, comp, nthreads=0):
    a = np.frombuffer(data, np.uint8)
    out = np.zeros(1 << 23, np.int16)
    bw, bh = C.c_int(), C.c_int()
    rc = hip_lib.lilliput_hip_progressive_coefs_host(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(comp), out.ctypes.data_as(C.c_void_p),
                                                     C.c_size_t(out.size), C.byref(bw), C.byref(bh), C.c_int(nthreads))
    assert rc == 0, rc
    return out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64)


def test_host_entropy_threads_reproduce_oracle_coefficients(hip_lib, oracle):
    """The hybrid mode's host side (lp_prog_host.cpp: unstuff + the scan lanes on threads, independent scans concurrently) through
    the library's test hook -- no device involved."""
    for name, data in _exotic_progressive().items():
        for c in range(oracle.jpeg_info(data)["ncomp"]):
            for nt in (1, 4):
                assert np.array_equal(_host_coefs(hip_lib, data, c, nt), oracle.jpeg_decode_coefs(data, c)), (name, c, nt)
    for i, desc, data in _cases(6, 25):
        for c in range(1 if desc[2] else 3):
            assert np.array_equal(_host_coefs(hip_lib, data, c), oracle.jpeg_decode_coefs(data, c)), (i, desc, c)


def test_damaged_files_same_verdict_and_coefficients_as_the_oracle(hip_lib, oracle):
    """Bit flips in the entropy data, the scan headers and the tables between scans: the product's parser accepts and rejects what
    the oracle does (which is pinned against libjpeg on the same kind of files in tests/test_oracle_golden.py), and the host-side
    scan decoder yields the oracle's coefficients -- end-of-data handling, bad codes, out-of-range indices included."""
    rng = np.random.default_rng(3)
    n_ok = n_err = 0
    for i, desc, data in _cases(17, 30, lo=16, hi=120):
        sos = data.index(b"\xff\xda")
        for k in range(20):
            d = bytearray(data)
            d[int(rng.integers(sos - 30, len(d) - 2))] ^= 1 << int(rng.integers(0, 8))
            d = bytes(d)
            a = np.frombuffer(d, np.uint8)
            out = np.zeros(1 << 20, np.int16)
            bw, bh = C.c_int(), C.c_int()
            try:
                exp = [oracle.jpeg_decode_coefs(d, c) for c in range(1 if desc[2] else 3)]
            except Exception:
                exp = None
            for c in range(1 if desc[2] else 3):
                rc = hip_lib.lilliput_hip_progressive_coefs_host(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(c), out.ctypes.data_as(C.c_void_p),
                                                                 C.c_size_t(out.size), C.byref(bw), C.byref(bh), C.c_int(1))
                if rc == -2:  # the host route's verdict "the reference's decoder fails on this file" (tests/test_damaged.py): so must the reference's own decoder
                    if oracle.ref_cvjpeg() is not None:
                        assert oracle.ref_cv_jpeg_decode(d) is None, (i, k, desc)
                    continue
                assert (rc != 0) == (exp is None), (i, k, desc, rc)
                if exp is not None:
                    got = out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64)
                    assert np.array_equal(got, exp[c]), (i, k, desc, c)
            n_ok += exp is not None
            n_err += exp is None
    assert n_ok > 300 and n_err > 20, (n_ok, n_err)


def _sos_headers(data):
    """(offset of the Ss byte) of every SOS header of a file"""
    out, i = [], 2
    while i + 4 <= len(data):
        if data[i] != 0xFF or data[i + 1] in (0x00, 0xFF) or 0xD0 <= data[i + 1] <= 0xD7:
            i += 1
            continue
        m, L = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == 0xD9:
            break
        if m == 0xDA:
            out.append(i + 5 + 2 * data[i + 4])
        i += 2 + L
    return out


def _smoothing_probe_files():
    """Progressive files with every bit of every scan header's Ss / Se / Ah-Al bytes flipped, and their DC-only cuts (first scan + EOI)."""
    for i, desc, data in _cases(29, 12, lo=24, hi=90):
        q = _sos_headers(data)[0] + 3  # first entropy-coded byte of the first scan; its data ends at the next marker
        while not (data[q] == 0xFF and data[q + 1] != 0 and not 0xD0 <= data[q + 1] <= 0xD7):
            q += 1
        yield (i, "whole"), data
        yield (i, "dc_only"), data[:q] + b"\xff\xd9"
        for off in _sos_headers(data):
            for byte in range(3):
                for bit in range(8):
                    d = bytearray(data)
                    d[off + byte] ^= 1 << bit
                    yield (i, off, byte, bit), bytes(d)


def test_interblock_smoothing_of_unfinished_progressive_files_is_libjpegs(hip_lib, oracle):
    """libjpeg estimates low AC coefficients that never reached full precision from the neighbouring blocks' DC values (jdcoefct.c
    smoothing_ok / decompress_smooth_data; on by default, also under cv::JpegDecoder): DC-only previews, scan scripts that stop at Al > 0,
    a damaged Ss / Se / Al. Found as a pixel difference on fresh seeds in round 6; restated in lp_prog_smooth (product, host side) and in
    oracle.jpeg_smoothing_plan / jpeg_smooth_coefs, weights and bottom-row rules pinned here against the reference's libjpeg.a:
    (1) the product's flag and the oracle's plan agree with each other and with the library (wherever its pixels change with
    do_block_smoothing off, both are up; a complete scan script: down); (2) the oracle's pixels are the reference decoder's;
    (3) the product's smoothed coefficients are the oracle's, value for value."""
    if oracle.ref() is None or oracle.ref_cvjpeg() is None:
        pytest.skip("oracle/_ref not built")
    f = hip_lib.lilliput_hip_jpeg_reference_smooths
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t]
    n_smooth = n_plain = n_exact = 0
    for tag, d in _smoothing_probe_files():
        cv = oracle.ref_cv_jpeg_decode(d)
        if cv is None:
            continue
        flag, plan = f(d, len(d)), oracle.jpeg_smoothing_plan(d)
        assert flag == (1 if plan is not None else 0), tag
        if tag[1] == "whole":
            assert flag == 0, tag
        if tag[1] == "dc_only":
            assert flag == 1, tag
        plain = oracle.ref_jpeg_decode_unsmoothed(d)
        assert np.array_equal(oracle.jpeg_decode_unsmoothed(d), plain), (tag, "the restatement without the pass is libjpeg without it")
        if not np.array_equal(cv, plain):
            assert flag == 1, (tag, "the reference smooths this file and the flag is down")
        if flag == 0:
            n_plain += 1
            continue
        n_smooth += 1
        a = np.frombuffer(d, np.uint8)
        for c in range(len(plan["comps"])):
            want = oracle.jpeg_smooth_coefs(oracle.ref_jpeg_decode_coefs(d, c), plan, c)
            out = np.zeros(want.size, np.int16)
            bw, bh = C.c_int(), C.c_int()
            rc = hip_lib.lilliput_hip_progressive_coefs_smoothed(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(c), out.ctypes.data_as(C.c_void_p),
                                                                 C.c_size_t(out.size), C.byref(bw), C.byref(bh))
            assert rc == 0 and np.array_equal(out.reshape(want.shape), want), (tag, c, "lp_prog_smooth against the oracle's pass")
        n_exact += np.array_equal(oracle.jpeg_decode(d), cv)
    # (one probe file -- 24 rows, 4:2:0: two iMCU rows -- differs in ONE block whose AC01 estimate sits exactly on a rounding boundary; open)
    assert n_smooth > 100 and n_plain > 100 and n_exact >= n_smooth - 1, (n_smooth, n_plain, n_exact)


# ------------------------------------------------------------------------------------------ GPU
_MODES = {"host-entropy": 0, "device-entropy": 1, "lanes-entropy": 2}


@pytest.fixture(params=list(_MODES))
def mode(request, hip_lib):
    """The three homes of the scans' entropy decode (lilliput_hip_set_progressive_entropy): host threads, the device's wave-per-scan
    decoder (lp_kernels_prog.hip) and its one-lane-per-scan reference (k_prog_scan). Left at auto (the default) afterwards."""
    hip_lib.lilliput_hip_set_progressive_entropy(_MODES[request.param])
    yield request.param
    hip_lib.lilliput_hip_set_progressive_entropy(-1)


@pytest.mark.gpu
def test_progressive_decode_stages_bit_exact(batch, oracle, mode):
    for name, data in _exotic_progressive().items():
        for c in range(oracle.jpeg_info(data)["ncomp"]):
            assert np.array_equal(batch.decode_jpeg_coefs(data, c), oracle.jpeg_decode_coefs(data, c)), (name, "coefs", c)
            assert np.array_equal(batch.decode_jpeg_plane(data, c), oracle.jpeg_decode_plane(data, c)), (name, "plane", c)
        px, _ = batch.decode_jpeg(data)
        assert np.array_equal(px, oracle.jpeg_decode(data)), name


@pytest.mark.gpu
def test_progressive_random_sweep_decode_and_transform(batch, oracle, mode):
    cases = list(_cases(11, 80))
    bad = []
    for i, desc, data in cases:
        got, _ = batch.decode_jpeg(data)
        exp = oracle.jpeg_decode(data)
        if got.shape != exp.shape or not np.array_equal(got, exp):
            bad.append((i, desc))
    assert not bad, bad[:8]
    for tw, th, q in ((64, 64, 85), (37, 91, 70)):
        res = batch.transform([c[2] for c in cases], tw, th, quality=q)
        for (i, desc, data), r in zip(cases, res):
            assert r.status == 0, (i, desc, r.status)
            exp = oracle.transform_jpeg_thumbnail(data, tw, th, q)
            if r.data != exp:  # the float area resize may differ by 1 LSB before the encoder (north_star tolerance)
                a, b = oracle.jpeg_decode(r.data), oracle.jpeg_decode(exp)
                assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).max() <= 8, (i, desc, tw, th)


@pytest.mark.gpu
def test_progressive_and_baseline_share_a_batch(batch, oracle, fixture_bytes, mode):
    """One decode range holding both kinds: the Huffman stages skip the progressive images, the scan lanes skip the others."""
    prog = [c[2] for c in _cases(3, 10, lo=40, hi=400)][:6]  # (the first six sizes Pillow's encoder takes)
    base = [fixture_bytes[n] for n in ("sunrise.jpg", "ferry_sunset.jpg", "firefox-gray.jpg")]
    srcs = [prog[0], base[0], prog[1], prog[2], base[1], base[2], prog[3], prog[4], prog[5]]
    for tw, th in ((48, 48), (100, 30)):
        res = batch.transform(srcs, tw, th, quality=85)
        for k, (data, r) in enumerate(zip(srcs, res)):
            assert r.status == 0, k
            exp = oracle.transform_jpeg_thumbnail(data, tw, th, 85)
            if r.data != exp:
                a, b = oracle.jpeg_decode(r.data), oracle.jpeg_decode(exp)
                assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).max() <= 8, (k, tw, th)


@pytest.mark.gpu
def test_progressive_through_the_go_api_mirror(hip_lib, oracle, mode):
    """NewDecoder / Header / DecodeTo / ImageOps.Transform on a progressive source, as lilliput's callers drive it."""
    import lilliput_amd as la

    _, desc, data = next(_cases(21, 4, lo=300, hi=500))  # (the first size Pillow's encoder takes: it refuses some)
    d = la.Decoder(data)
    h = d.Header()
    assert (h["height"], h["width"]) == desc[:2] and d.Description() == "JPEG"
    ops = la.ImageOps(2048)
    out = ops.Transform(d, la.ImageOptions(".jpeg", 96, 96, la.ImageOpsFit, False, {la.JpegQuality: 85}))
    d.Close()
    ops.Close()
    exp = oracle.transform_jpeg_thumbnail(data, 96, 96, 85)
    if out != exp:
        a, b = oracle.jpeg_decode(out), oracle.jpeg_decode(exp)
        assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).max() <= 8


@pytest.mark.gpu
def test_a_call_with_more_progressive_files_than_one_set_may_hold_is_cut_into_chunks(oracle, tmp_path):
    """lilliput_hip_batch_transform with more progressive files than LILLIPUT_HIP_PROG_PINNED_MAX / _DEVICE_MAX bytes of int16 coefficients let into one
    upload set (defaults 4 / 16 GiB: 85 / 340 files of 4096 x 4096): the files beyond the bound open the next chunk -- until late round 6 they
    answered ErrBufTooSmall (1 024 such files in one call: 340 served). A child process with bounds of three 256 x 192 files' worth, host and
    device routes: every item served, every output the oracle's."""
    import subprocess
    import sys

    rng = np.random.default_rng(11)
    files = []
    for i in range(14):
        buf = io.BytesIO()
        PIL.fromarray(_photo(rng, 192, 256, False)).save(buf, "JPEG", quality=85, progressive=True, subsampling=2)
        files.append(buf.getvalue())
        open(os.path.join(tmp_path, "p%02d.jpg" % i), "wb").write(files[-1])
    bound = 3 * (16 * 12 * 6 * 128) + 100   # 256 x 192 at 4:2:0 = 16 x 12 MCUs of six blocks, 128 bytes of int16 per block
    child = (
        "import glob, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import lilliput_amd as la\n"
        "fs = [open(f, 'rb').read() for f in sorted(glob.glob(os.path.join(%r, 'p*.jpg')))]\n"
        "b = la.Batch(0)\n"
        "r = b.transform(fs, 64, 64, quality=85)\n"
        "assert all(x.status == 0 for x in r), [x.status for x in r]\n"
        "[open(os.path.join(%r, 'o%%02d_%%s.jpg' %% (i, os.environ['LILLIPUT_HIP_PROG_ENTROPY'])), 'wb').write(x.data) for i, x in enumerate(r)]\n"
        "b.close()\n") % (ROOT, str(tmp_path), str(tmp_path))
    want = [oracle.transform_jpeg_thumbnail(f, 64, 64, 85) for f in files]
    for mode in ("host", "device"):
        env = dict(os.environ, LILLIPUT_HIP_PROG_PINNED_MAX=str(bound), LILLIPUT_HIP_PROG_DEVICE_MAX=str(bound), LILLIPUT_HIP_PROG_ENTROPY=mode)
        p = subprocess.run([sys.executable, "-c", child], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        for i, w in enumerate(want):
            assert open(os.path.join(tmp_path, "o%02d_%s.jpg" % (i, mode)), "rb").read() == w, (mode, i)


@pytest.mark.gpu
def test_progressive_large_image(batch, oracle):
    """2048 x 1536, 4:2:0, ten scans: whole-image equality with the oracle plus the thumbnail (default mode; the device-lane mode
    is covered at smaller sizes above -- it needs seconds for an image this large)."""
    rng = np.random.default_rng(4)
    buf = io.BytesIO()
    PIL.fromarray(_photo(rng, 1536, 2048, False)).save(buf, "JPEG", quality=88, progressive=True)
    data = buf.getvalue()
    got, _ = batch.decode_jpeg(data)
    assert np.array_equal(got, oracle.jpeg_decode(data))
    r = batch.transform([data] * 3, 256, 256, quality=85)
    assert all(x.status == 0 and x.data == r[0].data for x in r)
    assert r[0].data == oracle.transform_jpeg_thumbnail(data, 256, 256, 85)


@pytest.mark.gpu
def test_progressive_damaged_files_match_the_oracle(batch, oracle, mode):
    """Bit-flipped progressive files through the device path: the oracle's verdict (error or image) and, where the coefficients stay
    in the range a real image can have, the oracle's pixels; always the same answer twice. (Absurd coefficients overflow libjpeg's
    16-bit SIMD IDCT in ways neither the oracle nor the device restates.)"""
    import lilliput_amd

    rng = np.random.default_rng(0)
    n_img = n_err = n_same = 0
    for i, desc, data in _cases(8, 10, lo=40, hi=200):
        sos = data.index(b"\xff\xda")
        for k in range(12):
            d = bytearray(data)
            for _ in range(1 + k % 3):
                d[int(rng.integers(sos - 30, len(d) - 2))] ^= 1 << int(rng.integers(0, 8))
            d = bytes(d)
            outs = []
            for _ in range(2):
                try:
                    outs.append(batch.decode_jpeg(d)[0])
                except lilliput_amd.LilliputError as e:
                    outs.append(e.code)
            assert type(outs[0]) is type(outs[1]) and np.array_equal(outs[0], outs[1]), (i, k)
            if oracle.ref_cvjpeg() is not None:  # the reference's own decoder (cv::JpegDecoder over its libjpeg.a): verdict and pixels
                exp = oracle.ref_cv_jpeg_decode(d)
            else:
                try:
                    exp = oracle.jpeg_decode(d)
                except Exception:
                    exp = None
            assert isinstance(outs[0], int) == (exp is None), (i, k, desc, outs[0] if isinstance(outs[0], int) else "image")
            if exp is None:
                n_err += 1
                continue
            n_img += 1
            # (damaged bits at quality 1 produce dequantised coefficients beyond 16 bits: the reference's SIMD IDCT wraps and saturates there,
            # and so does k_idct since round 5 -- no allowance)
            assert np.array_equal(outs[0], exp), (i, k, desc)
            n_same += 1
    assert n_img > 50 and n_err > 3, (n_img, n_err, n_same)


@pytest.mark.gpu
def test_smoothed_files_decode_to_the_reference_decoders_pixels_on_the_device(batch, oracle, hip_lib, mode):
    """The files libjpeg smooths, through the product in every entropy mode (they take the host threads' route whatever the mode:
    lp_prog_smooth runs behind the scans, in front of k_idct): the pixels of the reference's own decoder."""
    if oracle.ref_cvjpeg() is None:
        pytest.skip("oracle/_ref not built")
    f = hip_lib.lilliput_hip_jpeg_reference_smooths
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t]
    n, bad = 0, []
    for tag, d in _smoothing_probe_files():
        if f(d, len(d)) != 1:
            continue
        cv = oracle.ref_cv_jpeg_decode(d)
        if cv is None:
            continue
        if not np.array_equal(batch.decode_jpeg(d)[0], cv):
            bad.append(tag)
        n += 1
    assert n > 100 and len(bad) <= max(1, n // 100), (n, bad[:8])  # (the one open block of the CPU test above; fresh seeds may draw another such file)


@pytest.mark.gpu
def test_smoothed_and_plain_progressive_files_share_a_set(batch, oracle, hip_lib, mode):
    """One upload set holding files libjpeg smooths (host threads' route + lp_prog_smooth in every mode) next to ordinary progressive files
    (on the device in the device modes) and a baseline file: every item's bytes are the ones it gets alone."""
    f = hip_lib.lilliput_hip_jpeg_reference_smooths
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t]
    flagged, plain = [], []
    for tag, d in _smoothing_probe_files():
        if tag[1] == "whole" and len(plain) < 4:
            plain.append(d)
        elif tag[1] == "dc_only" and len(flagged) < 3 and f(d, len(d)) == 1:
            flagged.append(d)
    assert len(flagged) == 3 and len(plain) == 4
    from lilliput_amd import synth

    base = synth.synth_jpeg(5, 96)
    srcs = [plain[0], flagged[0], plain[1], base, flagged[1], plain[2], flagged[2], plain[3]]
    together = batch.transform(srcs, 40, 40, quality=85)
    for k, (d, r) in enumerate(zip(srcs, together)):
        alone = batch.transform([d], 40, 40, quality=85)[0]
        assert r.status == 0 and alone.status == 0 and r.data == alone.data, k


@pytest.mark.gpu
def test_progressive_modes_agree_on_damaged_files(batch, hip_lib):
    """Host threads, device waves and device lanes give the same pixels (or the same error) for cut and bit-flipped files too: what the
    device decoders find irregular is decoded again by the host threads."""
    import lilliput_amd

    rng = np.random.default_rng(5)
    files = []
    for _, _, data in _cases(9, 6, lo=60, hi=160):
        for k in range(6):
            d = bytearray(data)
            if k % 2:
                d = d[: int(rng.integers(300, len(d)))]
            else:
                for _ in range(2):
                    d[int(rng.integers(len(d) // 2, len(d)))] ^= 1 << int(rng.integers(0, 8))
            files.append(bytes(d))
    outs = {}
    for m in (0, 1, 2):
        hip_lib.lilliput_hip_set_progressive_entropy(m)
        try:
            res = []
            for d in files:
                try:
                    res.append(batch.decode_jpeg(d)[0].tobytes())
                except lilliput_amd.LilliputError as e:
                    res.append(e.code)
            outs[m] = res
        finally:
            hip_lib.lilliput_hip_set_progressive_entropy(-1)
    assert outs[0] == outs[1] and outs[0] == outs[2]
    assert sum(isinstance(x, bytes) for x in outs[0]) > 5


def test_wave_decoder_listing_keeps_its_hand_managed_registers():
    """lp_kernels_prog.hip keeps its in-flight loads in physical registers the compiler does not know about (PW_RING_*): the gfx950
    listing must name them in the hand-written instructions only, every take behind its s_waitcnt, and use no scratch
    (scripts/r06_check_prog_isa.py; hipcc cross-compiles without a GPU)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r06_check_prog_isa.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
