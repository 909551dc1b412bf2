"""CPU tests of the product's host side: the C ABI library loads and exports every declared symbol, the JPEG
marker parser / table builder and the Go-mirroring control logic agree with the oracle, and the per-lane
Huffman logic (run serially through tests/emu) reproduces the oracle's coefficients."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "lilliput_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:opencv|lilliput)_[a-z0-9_]+)\s*\(", hdr))
    names |= {"CV_INTER_AREA", "CV_INTER_LINEAR", "CV_INTER_CUBIC"}
    assert len(names) > 50
    missing = [n for n in sorted(names) if not hasattr(hip_lib, n)]
    assert not missing, missing
    assert C.c_int.in_dll(hip_lib, "CV_INTER_AREA").value == 3


def test_no_gpu_means_loud_failure_not_fallback(hip_lib):
    import lilliput_amd

    if hip_lib.lilliput_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lilliput_amd.LilliputError):
        lilliput_amd.Batch(0)
    with pytest.raises(lilliput_amd.LilliputError):
        d = lilliput_amd.Decoder(open(os.path.join(ROOT, "tests/golden/inputs/coast.jpg"), "rb").read())
        ops = lilliput_amd.ImageOps(512)
        ops.Transform(d, lilliput_amd.ImageOptions(".jpeg", 32, 32))


def test_control_logic_matches_oracle(hip_lib, oracle):
    rng = np.random.default_rng(0)
    for _ in range(3000):
        ow, oh, rw, rh = [int(x) for x in rng.integers(1, 9000, 4)]
        if rng.random() < 0.3:
            rh = rw
        w, h = C.c_int(), C.c_int()
        hip_lib.lilliput_calculate_expected_size(ow, oh, rw, rh, C.byref(w), C.byref(h))
        assert (w.value, h.value) == oracle.calculate_expected_size(ow, oh, rw, rh)
        l, t, cw, ch = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        hip_lib.lilliput_fit_crop_rect(ow, oh, rw, rh, C.byref(l), C.byref(t), C.byref(cw), C.byref(ch))
        assert (l.value, t.value, cw.value, ch.value) == oracle.fit_crop_rect(ow, oh, rw, rh)


def test_content_length_scanners(hip_lib, fixture_bytes):
    """opencv_test.go:9-220 style cases for detectContentLength / detectAPNG."""
    def cl(b):
        a = np.frombuffer(bytes(b), np.uint8)
        return hip_lib.lilliput_detect_content_length(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size))

    j = fixture_bytes["coast.jpg"]
    assert cl(j) == len(j)
    assert cl(j + b"\x00" * 37) == len(j)          # trailing garbage is cut at EOI
    assert cl(b"\x01\x02\x03") == 3                # neither JPEG nor PNG: everything
    png = bytes([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]) + b"\x00\x00\x00\x00IHDRxxxx" + b"\x00\x00\x00\x00IENDyyyy"
    assert cl(png + b"junk") == len(png)
    a = np.frombuffer(png, np.uint8)
    assert hip_lib.lilliput_detect_apng(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)) == 0
    apng = png[:8] + b"\x00\x00\x00\x00acTLzzzz" + png[8:]
    a = np.frombuffer(apng, np.uint8)
    assert hip_lib.lilliput_detect_apng(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)) == 1


def test_sidecar_build_exports_only_its_own_entry_points():
    """INTEGRATION.md section 1, side-by-side mode: `make sidecar` links the same objects with an export list that keeps the re-declared
    reference symbols (opencv_*, giflib_*, webp_*, thumbhash_*, the color_info functions) local, so the library can sit next to the
    reference's stock shims in one binary."""
    d = os.path.join(ROOT, "lilliput_amd", "csrc")
    subprocess.run(["make", "-C", d, "sidecar"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = os.path.join(ROOT, "lilliput_amd", "liblilliput_hip_sidecar.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert {"lilliput_hip_batch_transform", "lilliput_hip_node_transform", "lilliput_new_decoder", "lilliput_image_ops_transform"} <= names
    clash = [n for n in names if n.startswith(("opencv_", "giflib_", "webp_", "thumbhash_", "tonemap_", "cicp_", "icc_", "is_hdr"))]
    assert not clash, clash
    assert all(n.startswith("lilliput_") for n in names), sorted(n for n in names if not n.startswith("lilliput_"))
    C.CDLL(so)  # loads: every internal reference was bound at link time


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libemu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(d, "emu_huff.cpp"),
                    os.path.join(ROOT, "lilliput_amd", "csrc", "lp_jpeg_parse.cpp")], check=True)
    return C.CDLL(so)


def test_unstuff_classifier_word_arithmetic(emu):
    """lp_unstuff_core.h (the byte classes of k_unstuff_count / k_unstuff_scatter as 4-bytes-at-a-time arithmetic) against the
    byte-by-byte definition: 4 M groups from marker-heavy alphabets, every neighbour and segment-end case."""
    emu.emu_unstuff_classify_check.restype = C.c_long
    assert emu.emu_unstuff_classify_check(C.c_long(4_000_000), C.c_uint32(7)) == 0


def _emu_coefs(emu, data, S, Cc, comp):
    a = np.frombuffer(data, np.uint8)
    cap = 1 << 24
    out = np.empty(cap, np.int16)
    bw, bh, r, ns, hits = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = emu.emu_decode_coefs(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_uint32(S), C.c_uint32(Cc), C.c_int(comp),
                              out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(bw), C.byref(bh), C.byref(r), C.byref(ns), C.byref(hits))
    assert rc == 0, rc
    return out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64), r.value, ns.value


@pytest.mark.parametrize("S,Cc", [(64, 32), (256, 64), (1024, 32), (4096, 128), (16384, 512)])
def test_lane_logic_reproduces_oracle_coefficients(emu, oracle, fixture_bytes, S, Cc):
    """speculate -> verify (worst-case Jacobi sweeps) -> scan -> write, restart intervals included."""
    for name in ("sunrise.jpg", "firefox-gray.jpg", "ferry_sunset.jpg", "large-sunrise.jpg"):
        if name == "large-sunrise.jpg" and S < 256:
            continue
        data = fixture_bytes[name]
        for comp in range(oracle.jpeg_info(data)["ncomp"]):
            got, rounds, nsub = _emu_coefs(emu, data, S, Cc, comp)
            assert np.array_equal(got, oracle.jpeg_decode_coefs(data, comp)), (name, comp)


def test_lane_logic_other_samplings_and_tables(emu, oracle):
    import io

    from PIL import Image

    from lilliput_amd import synth

    rgb = synth.synth_rgb(5, 256)
    for (w, h) in ((256, 256), (251, 133), (17, 9)):
        im = Image.fromarray(np.ascontiguousarray(rgb[:h, :w]))
        for kw in ({"subsampling": 0}, {"subsampling": 1}, {"subsampling": 2}, {"subsampling": 2, "optimize": True},
                   {"subsampling": 2, "restart_marker_blocks": 3}):
            b = io.BytesIO()
            im.save(b, "JPEG", quality=88, **kw)
            data = b.getvalue()
            for comp in range(3):
                got, _, _ = _emu_coefs(emu, data, 256, 64, comp)
                assert np.array_equal(got, oracle.jpeg_decode_coefs(data, comp)), (w, h, kw, comp)


def test_multi_symbol_entries_of_the_counting_passes(emu, oracle, fixture_bytes):
    """LpHuffSet::lutm (lp_build_huff_multi): every entry re-derived here from the one-symbol table by walking the window bit by bit --
    bits consumed, zigzag advance, EOB flag of the group -- for Annex-K and optimised tables, every sampling; a multi-symbol step must
    be exactly the one-symbol steps it stands for (jdhuff.c decode_mcu's loop, symbol by symbol), for every z the lane logic applies
    it at. And the SPEC pass really takes fewer steps with it."""
    import io

    from PIL import Image

    from lilliput_amd import synth

    files = [fixture_bytes["sunrise.jpg"], fixture_bytes["large-sunrise.jpg"], fixture_bytes["firefox-gray.jpg"]]
    rgb = synth.synth_rgb(3, 128)
    for kw in ({"subsampling": 0}, {"subsampling": 2, "optimize": True}, {"subsampling": 1, "optimize": True, "quality": 35}):
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", **({"quality": 90} | kw))
        files.append(b.getvalue())
    groups = 0
    for data in files:
        a = np.frombuffer(data, np.uint8)
        lut = np.zeros((4, 1024), np.uint16)
        lutm = np.zeros((4, 1024), np.uint16)
        bits = C.c_int()
        ncomp = emu.emu_huff_tables(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), lut.ctypes.data_as(C.c_void_p), lutm.ctypes.data_as(C.c_void_p), C.byref(bits))
        assert ncomp > 0 and bits.value == 10
        W = bits.value
        for slot in range(4):
            for i in range(1 << W):
                e1, em = int(lut[slot, i]), int(lutm[slot, i])
                if em == e1:
                    continue
                groups += 1
                assert e1 & 31 and not e1 & 0x8000            # a short first code that does not end the block
                # walk: symbols whose whole code lies inside the window, through the AC table behind this slot (a DC slot: the builder
                # knows which AC slot follows; one of the two must reproduce the entry)
                ok_any = False
                for t in ([slot] if slot >= 2 else [2, 3]):
                    syms = [(e1 & 31, (e1 >> 9) & 15, False)]   # (bits, run, EOB) of every symbol of the group
                    n2, adv2 = e1 & 31, ((e1 >> 9) & 15) + 1
                    while n2 < W:
                        e = int(lut[t, (i << n2) & ((1 << W) - 1)])
                        nb, sz = e & 31, (e >> 5) & 15
                        if nb == 0 or nb - sz > W - n2 or n2 + nb > 31:
                            break
                        if e & 0x8000:
                            syms.append((nb, 0, True))
                            n2 += nb
                            break
                        a2 = ((e >> 9) & 15) + 1
                        if adv2 + a2 > 63:
                            break
                        syms.append((nb, a2 - 1, False))
                        adv2, n2 = adv2 + a2, n2 + nb
                    eob2 = syms[-1][2]
                    want = n2 | ((adv2 if eob2 else adv2 - 1) << 9) | (0x8000 if eob2 else 0)
                    if len(syms) < 2 or want != em:
                        continue
                    ok_any = True
                    # the lane logic (LpLane::step): from every z it can stand at, the group step == its symbols one at a time
                    for z in ([0] if slot < 2 else range(1, 64)):
                        runx = em >> 9
                        if not z + (runx & 63) < 64:
                            continue                            # the lane takes the one-symbol entry instead
                        zg = z + runx + 1
                        g = (em & 31, 0 if zg > 63 else zg, zg > 63)
                        zs, ns, done = z, 0, False
                        for (nb, run, eob) in syms:
                            assert not done, (slot, i, z)      # no symbol of a group is read after the block has ended
                            k = zs + run + (64 if eob else 0)
                            ns, done = ns + nb, k + 1 > 63
                            zs = 0 if done else k + 1
                        assert g == (ns, zs, done), (slot, i, z, g, (ns, zs, done))
                assert ok_any, (slot, i, hex(e1), hex(em))
    assert groups > 1000
    # fewer SPEC steps, same coefficients (the emulation counts first-level lookups of the speculative pass)
    emu.emu_last_spec_steps.restype = C.c_ulonglong
    data = fixture_bytes["large-sunrise.jpg"]
    got, _, nsub = _emu_coefs(emu, data, 4096, 256, 0)
    assert np.array_equal(got, oracle.jpeg_decode_coefs(data, 0))
    steps = emu.emu_last_spec_steps()
    assert steps / nsub < 700, steps / nsub                           # 1 054 steps per subsequence one symbol at a time, 519 with the groups


def test_shapes_a_service_sees_reach_a_thread_per_box_kernel_or_the_area_walk(hip_lib):
    """Regression guard for round 5's finding: every integer scale but 8 / 16 / 32 used to take k_resample_fused's wave per destination
    pixel (30 - 58 us per image; 512 x 512 sources ran at 23 k images/s, now 154 k). lilliput_hip_resample_route answers, without a device,
    which kernel the batch path picks for a shape: YCbCr and grey sources of the sizes a service sees, Fit to a square thumbnail, every
    orientation -- never route 7, and the expected kernel for the common cases."""
    L = hip_lib
    FIT = 1  # ops.go:18-22 ImageOpsFit
    route = lambda w, h, o, nc, hs, vs, tw, th: L.lilliput_hip_resample_route(w, h, o, nc, hs, vs, tw, th, FIT, 0)
    assert route(4096, 4096, 1, 3, 2, 2, 256, 256) == 1      # the headline: 16 x 16 boxes
    assert route(2048, 2048, 6, 3, 2, 2, 256, 256) == 1      # 8 x 8, rotated
    assert route(512, 512, 1, 3, 2, 2, 256, 256) == 2        # 2 x 2
    assert route(1024, 1024, 3, 3, 2, 2, 256, 256) == 2      # 4 x 4, mirrored both ways
    assert route(4096, 4096, 1, 3, 1, 1, 256, 256) == 3      # 4:4:4
    assert route(4096, 4096, 1, 3, 2, 1, 256, 256) == 3      # 4:2:2
    assert route(1024, 1024, 1, 1, 1, 1, 256, 256) == 4      # grey
    assert route(4000, 3000, 1, 3, 2, 2, 256, 256) == 5      # a photograph: fractional scale
    assert route(4032, 3024, 6, 3, 2, 2, 256, 256) == 5
    assert route(768, 768, 1, 3, 2, 2, 256, 256) == 6        # 3 x 3
    assert route(3072, 3072, 8, 3, 2, 2, 256, 256) == 6      # 12 x 12, transposed
    assert route(512, 512, 1, 3, 1, 1, 256, 256) == 6        # 2 x 2 of a 4:4:4 source
    assert route(200, 200, 1, 3, 2, 2, 256, 256) == 0        # no upscale: no resize
    seen = {}
    for nc, hs, vs in ((3, 2, 2), (3, 2, 1), (3, 1, 1), (1, 1, 1)):
        for side in (128, 256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096):
            for extra in (0, 2, 8, 10, 128, 250):                 # the crop Fit takes starts at extra / 2
                for t in (64, 128, 256):
                    if side < t:
                        continue
                    for o in range(1, 9):
                        for wide in (True, False):
                            w, h = (side + extra, side) if wide else (side, side + extra)
                            r = route(w, h, o, nc, hs, vs, t, t)
                            assert r >= 0
                            seen[r] = seen.get(r, 0) + 1
                            if side % t == 0 and side // t <= 34 and side > t:
                                assert r != 7, (w, h, o, nc, hs, vs, t)
    assert all(seen.get(k, 0) for k in (1, 2, 3, 4, 6)), seen


def test_multi_symbol_entries_when_components_share_a_dc_table_but_not_an_ac_table(emu, oracle, fixture_bytes):
    """A scan whose components use ONE DC table with DIFFERENT AC tables (legal, unusual): the DC slot's multi-symbol entries must stay one
    symbol (there is no single AC table behind the slot) while the AC slots keep their groups. The fixture is a file whose SOS selectors
    were rewritten (Cb: DC table 0 with AC table 1), i.e. a stream read with tables it was not written for: the lane logic either
    reproduces libjpeg's coefficients for it or hands the stream over (a negative code; this fixture: the block count does not come out,
    -16) -- never a different answer."""
    data = bytearray(fixture_bytes["sunrise.jpg"])
    sos = data.find(b"\xff\xda")
    assert sos > 0 and data[sos + 4] == 3                      # three components in the scan
    assert data[sos + 5 + 2 * 1 + 1] == 0x11                   # Cb: Td 1, Ta 1
    data[sos + 5 + 2 * 1 + 1] = 0x01                           # -> Td 0, Ta 1
    data = bytes(data)
    a = np.frombuffer(data, np.uint8)
    lut = np.zeros((4, 1024), np.uint16)
    lutm = np.zeros((4, 1024), np.uint16)
    bits = C.c_int()
    assert emu.emu_huff_tables(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), lut.ctypes.data_as(C.c_void_p), lutm.ctypes.data_as(C.c_void_p), C.byref(bits)) == 3
    assert np.array_equal(lut[0], lutm[0])                     # DC slot 0 feeds AC slots 2 and 3: no group may continue into either
    assert not np.array_equal(lut[2], lutm[2])                 # the AC slots keep their groups
    cap = 1 << 22
    out = np.empty(cap, np.int16)
    for comp in range(3):
        bw, bh, r, ns, hits = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = emu.emu_decode_coefs(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_uint32(1024), C.c_uint32(64), C.c_int(comp), out.ctypes.data_as(C.c_void_p),
                                  C.c_size_t(cap), C.byref(bw), C.byref(bh), C.byref(r), C.byref(ns), C.byref(hits))
        if rc < 0:
            continue                                           # handed to the serial route: its business (tests/test_damaged.py)
        try:
            ref = oracle.ref_jpeg_decode_coefs(data, comp) if oracle.ref() is not None else oracle.jpeg_decode_coefs(data, comp)
        except Exception:
            continue
        assert np.array_equal(out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64), ref), comp


def _strip_segments(jpeg, marker):
    """Drop every segment with this marker code between SOI and SOS."""
    out, i = bytearray(jpeg[:2]), 2
    while jpeg[i + 1] != 0xDA:
        L = (jpeg[i + 2] << 8) | jpeg[i + 3]
        if jpeg[i + 1] != marker:
            out += jpeg[i : i + 2 + L]
        i += 2 + L
    return bytes(out + jpeg[i:])


def _header_verdict(L, data):
    """opencv_decoder_create + read_header on the host (no GPU involved): accepted?"""
    arr = np.frombuffer(data, np.uint8).copy()
    em = L.opencv_mat_create_from_data(len(data), 1, 0, arr.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)))
    dec = L.opencv_decoder_create(em)
    ok = bool(dec) and bool(L.opencv_decoder_read_header(dec))
    if dec:
        L.opencv_decoder_release(dec)
    L.opencv_mat_release(em)
    return ok


def test_header_walk_accepts_and_rejects_like_libjpeg(hip_lib, oracle, fixture_bytes):
    """Seeded header mutations: the product parser, the oracle and (when built) the reference's libjpeg-turbo agree on
    accept/reject, and the oracle's pixels equal libjpeg's whenever a mutation leaves the entropy-coded data decodable
    without libjpeg warnings (table-less Motion-JPEG frames, shuffled or garbage-separated segments, bogus APPn lengths)."""
    import random

    rnd = random.Random(11)
    names = ["field.jpg", "coast.jpg", "firefox-gray.jpg", "sunrise.jpg"]
    ref = oracle.ref() is not None
    disagree = []
    for it in range(1200):
        b = fixture_bytes[rnd.choice(names)]
        data = bytearray(b)
        hdr_end = data.find(b"\xff\xda") + 14
        mode = rnd.randrange(3)
        if mode == 0:
            for _ in range(rnd.randrange(1, 3)):
                data[rnd.randrange(hdr_end)] = rnd.randrange(256)
        elif mode == 1:
            data[rnd.randrange(hdr_end)] ^= 1 << rnd.randrange(8)
        else:
            q = rnd.randrange(2, hdr_end)
            data = data[:q] + data[q + rnd.randrange(1, 4) :]
        data = bytes(data)
        mine = _header_verdict(hip_lib, data)
        try:
            oracle.jpeg_info(data)
            orc = True
        except Exception:
            orc = False
        sig = data[:3] == b"\xff\xd8\xff"  # OpenCV picks its JPEG decoder by this signature before libjpeg sees the file
        if mine != (orc and sig):
            disagree.append(("oracle", it))
        if ref:
            try:
                oracle.ref_jpeg_decode(data)
                r = True
            except Exception:
                r = False
            if mine and not r:
                disagree.append(("libjpeg rejects", it))
    assert not disagree, disagree[:10]


def test_table_less_frames_use_annex_k_tables(hip_lib, oracle):
    """jdhuff.c jinit_huff_decoder: Huffman table ids 0/1 without a DHT fall back to the Annex-K tables."""
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    full = oracle.jpeg_encode(px, 80)                # libjpeg's default tables ARE the Annex-K ones
    bare = _strip_segments(full, 0xC4)
    assert len(bare) < len(full) - 400 and _header_verdict(hip_lib, bare)
    assert np.array_equal(oracle.jpeg_decode(bare), oracle.jpeg_decode(full))
    if oracle.ref() is not None:
        assert np.array_equal(oracle.ref_jpeg_decode(bare), oracle.ref_jpeg_decode(full))


def test_lane_logic_random_sweep(emu, oracle):
    """The randomised sources of tests/test_gpu_sweep.py through the CPU emulation of the lane logic (every subsequence size the
    engine picks for small images). Regression: with one-MCU restart intervals a speculating lane used to run over a boundary
    mid-block, jump back to it and visit the same positions twice; a checkpoint of the first visit then spliced the second
    visit's block count in again."""
    import test_gpu_sweep as T

    bad = []
    for seed, n in ((99, 96), (2024, 40)):
        for i, desc, data in T._cases(seed, n):
            info = oracle.jpeg_info(data)
            for S in (256, 1024):
                for comp in range(info["ncomp"]):
                    try:
                        got, _, _ = _emu_coefs(emu, data, S, 64, comp)
                        ok = np.array_equal(got, oracle.jpeg_decode_coefs(data, comp))
                    except AssertionError:
                        ok = False
                    if not ok:
                        bad.append((seed, i, desc, S, comp))
    assert not bad, bad[:8]


def test_nothing_unwinds_through_the_c_abi(hip_lib, fixture_bytes):
    """An exception inside an exported function -- std::bad_alloc from a buffer a header sized, std::length_error -- must come out as the
    entry point's failure value with the reason in lilliput_hip_last_error(), never as an unwind into the caller (a Go process would
    abort). lilliput_hip_test_fault(n) makes the next n guarded allocation sites throw std::bad_alloc (csrc/lp_abi_guard.h); every
    exported function with a body is a function-try-block (the check at the end reads that off the sources)."""
    import re

    L = hip_lib
    L.lilliput_hip_last_error.restype = C.c_char_p
    L.opencv_mat_create.restype = C.c_void_p
    L.opencv_mat_create_from_data.restype = C.c_void_p
    L.opencv_decoder_create.restype = C.c_void_p
    L.giflib_decoder_create.restype = C.c_void_p
    L.webp_decoder_create.restype = C.c_void_p
    L.thumbhash_encoder_create.restype = C.c_void_p
    for f in ("opencv_mat_release", "opencv_decoder_release", "lilliput_decoder_close"):
        getattr(L, f).restype = None
    data = np.frombuffer(fixture_bytes["coast.jpg"], dtype=np.uint8).copy()
    buf = L.opencv_mat_create_from_data(C.c_int(data.size), C.c_int(1), C.c_int(0), C.c_void_p(data.ctypes.data), C.c_size_t(data.size))
    assert buf

    def faulted(call, failure):
        L.lilliput_hip_test_fault(1)
        got = call()
        L.lilliput_hip_test_fault(0)
        assert got == failure or (failure is None and not got), (got, failure)
        assert b"out of memory" in L.lilliput_hip_last_error(), L.lilliput_hip_last_error()

    faulted(lambda: L.opencv_mat_create(C.c_int(8), C.c_int(8), C.c_int(16)), None)
    faulted(lambda: L.opencv_decoder_create(C.c_void_p(buf)), None)
    d = L.opencv_decoder_create(C.c_void_p(buf))
    assert d
    L.opencv_decoder_read_header.restype = C.c_bool
    faulted(lambda: L.opencv_decoder_read_header(C.c_void_p(d)), False)
    assert L.opencv_decoder_read_header(C.c_void_p(d))          # and the decoder is still usable afterwards
    L.opencv_decoder_release(C.c_void_p(d))
    faulted(lambda: L.giflib_decoder_create(C.c_void_p(buf)), None)
    faulted(lambda: L.webp_decoder_create(C.c_void_p(buf)), None)
    out = np.zeros(64, dtype=np.uint8)
    faulted(lambda: L.thumbhash_encoder_create(C.c_void_p(out.ctypes.data), C.c_size_t(out.size)), None)
    dec = C.c_void_p()
    faulted(lambda: L.lilliput_new_decoder(C.c_void_p(data.ctypes.data), C.c_size_t(data.size), C.byref(dec)), 5)  # LILLIPUT_ERR_DEVICE + the text
    assert L.lilliput_new_decoder(C.c_void_p(data.ctypes.data), C.c_size_t(data.size), C.byref(dec)) == 0
    L.lilliput_decoder_close(dec)
    L.opencv_mat_release(C.c_void_p(buf))
    # hostile headers: dimensions that would size gigabytes -- refused by value, no exception reaches the caller either way
    png = bytearray(open(os.path.join(ROOT, "tests/golden/inputs_png", sorted(os.listdir(os.path.join(ROOT, "tests/golden/inputs_png")))[0]), "rb").read())
    png[16:24] = (0x7FFFFFFF).to_bytes(4, "big") * 2
    a = np.frombuffer(bytes(png), dtype=np.uint8).copy()
    if L.lilliput_new_decoder(C.c_void_p(a.ctypes.data), C.c_size_t(a.size), C.byref(dec)) == 0:  # (the size check is DecodeTo's, opencv.go:824-827)
        w, h = C.c_int(), C.c_int()
        L.lilliput_decoder_header(dec, C.byref(w), C.byref(h), None, None, None, None)
        L.lilliput_decoder_close(dec)
    # every exported function that has a body of its own is guarded
    src_dir = os.path.join(ROOT, "lilliput_amd", "csrc")
    import subprocess

    exported = {l.split()[2] for l in subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "lilliput_amd", "liblilliput_hip.so")], capture_output=True, text=True).stdout.splitlines()
                if " T " in l and not l.split()[2].startswith("_Z")}
    unguarded = []
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith(".cpp"):
            continue
        lines = open(os.path.join(src_dir, f)).read().split("\n")
        for i, l in enumerate(lines[:-1]):
            m = re.match(r'^(?:extern "C" )?[A-Za-z_][\w \*:<>]*?[ \*](\w+)\(.*\)\s*(//.*|/\*.*\*/)?\s*$', l)
            if m and m.group(1) in exported and lines[i + 1].startswith("{"):
                unguarded.append((f, m.group(1)))
    assert not unguarded, unguarded
