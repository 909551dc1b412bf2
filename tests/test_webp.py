"""WebP row: the webp.hpp C ABI of the product -- own RIFF container walk and animation writer, VP8 / VP8L
payloads on the host through libwebp -- against the REFERENCE's libwebp 1.5.0 + libwebpmux + libwebpdemux driven like webp.cpp
(oracle/ref_webp_driver.c -> oracle/_ref/librefwebp.so), and through ImageOps.Transform on the device (BASELINE configs[2], [3])."""
import ctypes as C
import hashlib
import json
import os
import random
import struct

import numpy as np
import pytest
from conftest import fresh_seed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "inputs_webp")
GOLD = os.path.join(ROOT, "tests", "golden", "webp_golden.json")


def fixtures():
    return {n: open(os.path.join(FIX, n), "rb").read() for n in sorted(os.listdir(FIX))}


@pytest.fixture(scope="module")
def W(hip_lib):
    L = hip_lib
    L.webp_decoder_create.restype = C.c_void_p
    L.webp_decoder_create.argtypes = [C.c_void_p]
    for n in ("get_width", "get_height", "get_pixel_type", "get_num_frames", "get_total_duration", "get_prev_frame_delay", "get_prev_frame_dispose",
              "get_prev_frame_blend", "get_prev_frame_x_offset", "get_prev_frame_y_offset", "has_more_frames"):
        getattr(L, "webp_decoder_" + n).argtypes = [C.c_void_p]
    L.webp_decoder_get_bg_color.restype = C.c_uint32
    L.webp_decoder_get_bg_color.argtypes = [C.c_void_p]
    L.webp_decoder_get_loop_count.restype = C.c_uint32
    L.webp_decoder_get_loop_count.argtypes = [C.c_void_p]
    L.webp_decoder_get_icc.restype = C.c_size_t
    L.webp_decoder_get_icc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.webp_decoder_release.argtypes = [C.c_void_p]
    L.webp_decoder_advance_frame.argtypes = [C.c_void_p]
    L.webp_decoder_decode.restype = C.c_bool
    L.webp_decoder_decode.argtypes = [C.c_void_p, C.c_void_p]
    L.webp_encoder_create.restype = C.c_void_p
    L.webp_encoder_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]
    L.webp_encoder_write.restype = C.c_size_t
    L.webp_encoder_write.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.webp_encoder_flush.restype = C.c_size_t
    L.webp_encoder_flush.argtypes = [C.c_void_p]
    L.webp_encoder_release.argtypes = [C.c_void_p]
    return L


def product_decode(L, data):
    """The decoder half of webp.hpp driven like webpDecoder (webp.go:27-167), host only: (info dict, [(frame, meta)]) or None."""
    src = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    m = L.opencv_mat_create_from_data(src.size, 1, 0, src.ctypes.data, src.size)
    d = L.webp_decoder_create(m)
    if not d:
        L.opencv_mat_release(m)
        return None
    icc = np.zeros(1 << 20, dtype=np.uint8)
    info = {"width": L.webp_decoder_get_width(d), "height": L.webp_decoder_get_height(d), "has_alpha": 1 if L.webp_decoder_get_pixel_type(d) == 24 else 0,
            "num_frames": L.webp_decoder_get_num_frames(d), "total_duration": L.webp_decoder_get_total_duration(d), "bgcolor": L.webp_decoder_get_bg_color(d),
            "loop_count": L.webp_decoder_get_loop_count(d), "icc_len": L.webp_decoder_get_icc(d, icc.ctypes.data, icc.size)}
    frames = []
    cn = 4 if info["has_alpha"] else 3
    buf = np.zeros(info["width"] * info["height"] * 4 + 16, dtype=np.uint8)
    while True:
        fm = L.opencv_mat_create_from_data(info["width"], info["height"], 24 if cn == 4 else 16, buf.ctypes.data, buf.size)  # Framebuffer.resizeMat
        ok = L.webp_decoder_decode(d, fm)
        if ok:
            w, h = L.opencv_mat_get_width(fm), L.opencv_mat_get_height(fm)
            px = np.ctypeslib.as_array(C.cast(L.opencv_mat_get_data(fm), C.POINTER(C.c_uint8)), shape=(h * w * cn,)).reshape(h, w, cn).copy()
            frames.append((px, {"duration": L.webp_decoder_get_prev_frame_delay(d), "x_offset": L.webp_decoder_get_prev_frame_x_offset(d),
                                "y_offset": L.webp_decoder_get_prev_frame_y_offset(d), "dispose": L.webp_decoder_get_prev_frame_dispose(d),
                                "blend": L.webp_decoder_get_prev_frame_blend(d)}))
        else:
            frames.append(None)
        L.opencv_mat_release(fm)
        more = L.webp_decoder_has_more_frames(d)
        L.webp_decoder_advance_frame(d)
        if not more:
            break
    L.webp_decoder_release(d)
    L.opencv_mat_release(m)
    return info, frames, icc[: info["icc_len"]].tobytes()


def digest(res):
    if res is None:
        return "none"
    info, frames, icc = res
    h = hashlib.sha1()
    for f in frames:
        if f is None:
            h.update(b"failed")
        else:
            h.update(np.array(list(f[0].shape) + [f[1][k] for k in ("duration", "x_offset", "y_offset", "dispose", "blend")], dtype=np.int32).tobytes())
            h.update(f[0].tobytes())
    h.update(icc)
    return "%s:%s" % (",".join(str(info[k]) for k in ("width", "height", "has_alpha", "num_frames", "total_duration", "bgcolor", "loop_count", "icc_len")), h.hexdigest()[:16])


def mutations(seed, n):
    """Damaged containers: byte flips in the first 64 bytes and at chunk boundaries, truncations, chunk tag swaps."""
    rnd = random.Random(seed)
    fx = fixtures()
    small = {k: v for k, v in fx.items() if len(v) < 60000}
    out = {}
    names = sorted(small)
    for i in range(n):
        name = names[i % len(names)]
        b = bytearray(small[name])
        kind = rnd.randrange(4)
        if kind == 0:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(min(len(b), 96))] = rnd.randrange(256)
        elif kind == 1:
            b = b[: rnd.randrange(12, len(b))]
        elif kind == 2:
            p = rnd.randrange(len(b))
            b[p] ^= 1 << rnd.randrange(8)
        else:
            pos = 12
            chunks = []
            while pos + 8 <= len(b):
                chunks.append(pos)
                pos += 8 + ((struct.unpack_from("<I", b, pos + 4)[0] + 1) & ~1)
            if chunks:
                c = rnd.choice(chunks)
                b[c : c + 4] = rnd.choice([b"VP8X", b"ANIM", b"ANMF", b"ALPH", b"VP8 ", b"VP8L", b"ICCP", b"JUNK"])
        out["%s#%d" % (name, i)] = bytes(b)
    return out


def ref_digest(oracle, data):
    info = oracle.ref_webp_info(data)
    if info is None:
        return "none"
    return digest((info, oracle.ref_webp_frames(data), oracle.ref_webp_icc(data)))


def test_decoder_matches_the_reference_library_live(W, oracle):
    """Container walk + frame bitstream assembly + the system libwebp's VP8 / VP8L / ALPH decode against libwebpmux + libwebp 1.5.0 of the
    reference: canvas, alpha flag, frame count, durations, offsets, dispose / blend, ICC and every decoded pixel."""
    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built (oracle/_ref/librefwebp.so)")
    cases = dict(fixtures())
    cases.update(mutations(fresh_seed(7), 400))
    bad = [n for n, d in cases.items() if digest(product_decode(W, d)) != ref_digest(oracle, d)]
    assert not bad, bad[:10]


def _frame_subchunk_cases():
    """The first ANMF frame of party-discord.webp (ALPH + "VP8 ") with its sub-chunk area rebuilt: what libwebp 1.5.0's MuxImageParse makes
    of every ordering (src/mux/muxread.c: a frame is "partial" from its header until its image chunk; an unknown chunk while partial, a
    known non-image chunk anywhere, a second ALPH or image chunk and a frame that ends partial all fail WebPMuxCreate)."""
    base = fixtures()["party-discord.webp"]

    def chunks(pos, end):
        out = []
        while pos + 8 <= end:
            sz = struct.unpack_from("<I", base, pos + 4)[0]
            out.append((pos, bytes(base[pos : pos + 4]), sz))
            pos += 8 + ((sz + 1) & ~1)
        return out

    p0, _, s0 = [c for c in chunks(12, len(base)) if c[1] == b"ANMF"][0]

    def ck(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    subs = [ck(t, base[p + 8 : p + 8 + s]) for p, t, s in chunks(p0 + 8 + 16, p0 + 8 + s0)]
    A, I = [x for x in subs if x[:4] == b"ALPH"][0], [x for x in subs if x[:4] == b"VP8 "][0]
    U = ck(b"JUNK", b"abcd")

    def rebuild(sub_bytes):
        payload = base[p0 + 8 : p0 + 8 + 16] + sub_bytes
        body = base[12:p0] + b"ANMF" + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"") + base[p0 + 8 + ((s0 + 1) & ~1) :]
        return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WEBP" + body

    # (sub-chunks, accepted by the reference's WebPMuxCreate)
    table = {"orig": (A + I, True), "unknown_first": (U + A + I, False), "unknown_between": (A + U + I, False), "unknown_after": (A + I + U, True),
             "no_alph": (I, True), "no_alph_unknown_first": (U + I, False), "second_image_after": (A + I + I, False), "alph_after": (A + I + A, False),
             "alph_after_noalph": (I + A, False), "two_alph": (A + A + I, False), "iccp_first": (ck(b"ICCP", b"abcd") + A + I, False),
             "iccp_after": (A + I + ck(b"ICCP", b"abcd"), False), "exif_after": (A + I + ck(b"EXIF", b"abcd"), False),
             "xmp_after": (A + I + ck(b"XMP ", b"abcd"), False), "anim_after": (A + I + ck(b"ANIM", b"abcdef"), False),
             "anmf_after": (A + I + ck(b"ANMF", b"0123456789abcdef"), False), "vp8x_after": (A + I + ck(b"VP8X", b"0123456789"), False),
             "unknown_after_then_image": (A + I + U + I, False), "image_then_unknown_then_alph": (I + U + A, False),
             "two_unknown_after": (A + I + U + U, True), "trailing_7_bytes": (A + I + b"JUNKxyz", False), "alph_only": (A, False), "empty": (b"", False)}
    return {n: (rebuild(sb), ok) for n, (sb, ok) in table.items()}


def test_frame_subchunk_orderings_accepted_and_refused_like_the_reference_mux(W, oracle):
    """Found by the live differential test on fresh seeds in round 6 (a flipped byte in a frame's ALPH tag: the reference refuses the file,
    the product served it): the sub-chunk rules of a frame, every ordering by hand. The verdicts in the table were read off the reference's
    libwebpmux (and are compared with it live where oracle/_ref is built)."""
    cases = _frame_subchunk_cases()
    # "First chunk should be VP8, VP8L or VP8X" (WebPMuxCreateInternal): anything else in front of an otherwise good file
    for name, data in fixtures().items():
        if len(data) < 200000:
            for first in (b"JUNK", b"ALPH", b"ICCP", b"ANIM"):
                body = first + struct.pack("<I", 6) + b"abcdef" + data[12:]
                cases["%s behind a %s chunk" % (name, first.decode())] = (b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WEBP" + body, False)
    for name, (data, ok) in cases.items():
        mine = product_decode(W, data)
        assert (mine is not None) == ok, name
        if oracle.ref_webp() is not None:
            assert (oracle.ref_webp_info(data) is not None) == ok, (name, "the table is the reference's")
            assert digest(mine) == ref_digest(oracle, data), name


def test_frames_that_do_not_fit_the_canvas_sized_decode_buffer_fail_like_the_reference(W, oracle):
    """The reference decodes every frame into ONE buffer of canvas width x height x 4 bytes (webp.cpp:129-131, 339-350): with a VP8X canvas
    smaller than its frames (a flipped byte in the canvas width: fresh-seed finding of round 6) WebPDecodeBGRAInto refuses the frame --
    webp_decoder_decode returns false for it, the container itself stays acceptable."""
    for name, pos, val in (("party-discord.webp", 24, 0x12), ("animated-webp-supported.webp", 24, 0x27)):
        d = bytearray(fixtures()[name])
        d[pos] = val
        info, frames, _ = product_decode(W, bytes(d))
        assert info["num_frames"] == len(frames) and len(frames) > 1
        cw, ch = info["width"], info["height"]
        whole = product_decode(W, fixtures()[name])[1]
        for k, (f, g) in enumerate(zip(frames, whole)):
            h, w, cn = g[0].shape
            fits = w * cn * h <= cw * ch * 4
            assert (f is not None) == fits, (name, k)
            if fits:
                assert np.array_equal(f[0], g[0])
        assert any(f is None for f in frames), name
        if oracle.ref_webp() is not None:
            assert digest((info, frames, b"")) == ref_digest(oracle, bytes(d)), name


def test_decoder_matches_recorded_reference_answers(W):
    gold = json.load(open(GOLD))
    cases = dict(fixtures())
    cases.update(mutations(7, 400))
    assert set(gold["decode"]) == set(cases)
    bad = [n for n, d in cases.items() if digest(product_decode(W, d)) != gold["decode"][n]]
    assert not bad, bad[:10]


# ------------------------------------------------------------------------------------------------ encoder half (host only)
def product_encode(L, frames, quality, delays=None, icc=b"", bgcolor=0xFFFFFFFF, loops=0, cap=32 << 20, extra_opts=()):
    """webp_encoder_create / write per frame / flush the way webpEncoder does (webp.go:174-256). frames: HxWx3/4 arrays."""
    out = np.zeros(cap, dtype=np.uint8)
    iccb = np.frombuffer(bytes(icc), dtype=np.uint8).copy() if icc else None
    e = L.webp_encoder_create(out.ctypes.data, cap, iccb.ctypes.data if icc else None, len(icc), bgcolor, loops)
    assert e
    opts = [64, int(quality)] + list(extra_opts)  # WebpQuality = cv::IMWRITE_WEBP_QUALITY
    arr = (C.c_int * len(opts))(*opts)
    for k, f in enumerate(frames):
        f = np.ascontiguousarray(f, dtype=np.uint8)
        h, w = f.shape[:2]
        cn = 1 if f.ndim == 2 else f.shape[2]
        m = L.opencv_mat_create_from_data(w, h, {1: 0, 3: 16, 4: 24}[cn], f.ctypes.data, f.size)
        r = L.webp_encoder_write(e, m, arr, len(opts), (delays[k] if delays else 0), 0, 0, 0, 0)
        L.opencv_mat_release(m)
        if not r:
            L.webp_encoder_release(e)
            return None
    n = L.webp_encoder_flush(e)
    L.webp_encoder_release(e)
    return out[:n].tobytes() if n else None


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _test_frames(n, w, h, cn, seed):
    """A small moving-square animation over a smooth background (sub-rectangles change from frame to frame)."""
    rnd = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), ((xx + yy) * 255 // max(1, w + h - 2))] + ([np.full((h, w), 255)] if cn == 4 else []), axis=2).astype(np.uint8)
    frames = []
    for k in range(n):
        f = base.copy()
        x0, y0 = (3 + 5 * k) % max(1, w - 9), (2 + 3 * k) % max(1, h - 7)
        f[y0:y0 + 7, x0:x0 + 9, :3] = rnd.integers(0, 256, (7, 9, 3), dtype=np.uint8)
        if cn == 4:
            f[y0:y0 + 3, x0:x0 + 4, 3] = rnd.integers(1, 256, (3, 4), dtype=np.uint8)  # partially transparent, never fully (lossless keeps RGB only where alpha > 0)
        frames.append(f)
    return frames


def test_still_writer_is_read_back_exactly_by_the_reference_library(W, oracle):
    """One frame -> still WebP (webp.cpp:707-751 + the ICCP mux, :556-575). Lossless (quality > 100): the reference's libwebp decodes the
    product's file to the very pixels that went in, the ICC profile is carried in an ICCP chunk exactly like the reference's own writer
    carries it. Lossy: same container shape (VP8X / ICCP / ALPH presence), and a faithful picture."""
    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    icc = oracle.ref_webp_icc(fixtures()["ferry_sunset.webp"])
    assert len(icc) == 536
    for cn in (3, 4):
        f = _test_frames(1, 61, 37, cn, 5)[0]
        for q, prof in ((101, icc), (101, b""), (80, icc), (80, b"")):
            ours = product_encode(W, [f], q, icc=prof)
            ref = oracle.ref_webp_encode_still(f, q, prof)
            assert ours and ref
            io, ir = oracle.ref_webp_info(ours), oracle.ref_webp_info(ref)
            assert io == dict(ir, icc_len=io["icc_len"]) and io["icc_len"] == len(prof) and oracle.ref_webp_icc(ours) == prof, (cn, q)
            assert (ours[12:16] == b"VP8X") == (ref[12:16] == b"VP8X"), (cn, q)
            back = oracle.ref_webp_frames(ours)[0][0]
            if q > 100:
                assert np.array_equal(back, f), (cn, q)
            else:  # another libwebp release on the product's side (1.2.2 vs the reference's 1.5.0): as faithful as the reference's own file
                ref_back = oracle.ref_webp_frames(ref)[0][0]
                assert back.shape == f.shape and _psnr(back, f) > _psnr(ref_back, f) - 1.5, (cn, q, _psnr(back, f), _psnr(ref_back, f))
    # grey input: cv::COLOR_GRAY2BGR before the import
    g = np.arange(40 * 30, dtype=np.uint8).reshape(30, 40)
    back = oracle.ref_webp_frames(product_encode(W, [g], 101))[0][0]
    assert np.array_equal(back, np.repeat(g[:, :, None], 3, axis=2))
    # nothing written -> flush reports failure (webp.cpp:503-508); an output buffer that is too small -> 0 as well
    assert product_encode(W, [], 80) is None
    assert product_encode(W, [_test_frames(1, 64, 64, 3, 1)[0]], 101, cap=64) is None


def test_animation_writer_plays_back_frame_for_frame_in_the_reference_library(W, oracle):
    """Two or more frames -> animated WebP (webp.cpp:631-706, 510-552). The reference assembles it with WebPAnimEncoder (kmin 3, kmax 4);
    the product codes every frame's changed rectangle on its own. Played back by libwebpdemux's WebPAnimDecoder of the reference, the
    lossless file shows exactly the frames that went in, for exactly their durations, with the loop count, background colour and ICC
    profile of the source."""
    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    icc = oracle.ref_webp_icc(fixtures()["ferry_sunset.webp"])
    for cn, n, w, h in ((4, 7, 48, 40), (3, 5, 33, 21), (4, 2, 9, 7)):
        frames = _test_frames(n, w, h, cn, 11)
        delays = [30 + 10 * k for k in range(n)]
        data = product_encode(W, frames, 101, delays=delays, icc=icc, bgcolor=0x11223344, loops=3)
        assert data
        info = oracle.ref_webp_info(data)
        assert (info["width"], info["height"], info["num_frames"], info["total_duration"], info["bgcolor"], info["loop_count"], info["icc_len"]) == (w, h, n, sum(delays), 0x11223344, 3, len(icc))
        assert oracle.ref_webp_icc(data) == icc
        canv, ts, loops, _ = oracle.ref_webp_play(data)
        assert len(canv) == n and loops == 3 and ts == list(np.cumsum(delays))
        for k in range(n):
            want = frames[k] if cn == 4 else np.concatenate([frames[k], np.full((h, w, 1), 255, np.uint8)], axis=2)
            assert np.array_equal(canv[k], want), (cn, k)
        # the product's own decoder reads its writer back too
        pi, pf, _ = product_decode(W, data)
        assert pi["num_frames"] == n and all(x is not None for x in pf)
    # an unchanged frame extends its predecessor instead of costing a chunk
    fr = _test_frames(3, 32, 24, 3, 2)
    data = product_encode(W, [fr[0], fr[0], fr[1]], 101, delays=[40, 50, 60])
    info = oracle.ref_webp_info(data)
    assert (info["num_frames"], info["total_duration"]) == (2, 150)
    canv, ts, _, _ = oracle.ref_webp_play(data)
    assert ts == [90, 150] and np.array_equal(canv[0][:, :, :3], fr[0]) and np.array_equal(canv[1][:, :, :3], fr[1])
    # lossy, with encoder options passed through the advanced API
    frames = _test_frames(4, 64, 48, 3, 9)
    data = product_encode(W, frames, 75, delays=[20] * 4, extra_opts=(1000, 2, 1005, 2))
    canv, ts, _, _ = oracle.ref_webp_play(data)
    assert len(canv) == 4 and all(_psnr(canv[k][:, :, :3], frames[k]) > 24 for k in range(4)), [_psnr(canv[k][:, :, :3], frames[k]) for k in range(4)]
    # a frame of another size cannot join the canvas (WebPAnimEncoderAdd fails in the reference)
    assert product_encode(W, [frames[0], frames[1][:20]], 80) is None


# ------------------------------------------------------------------------------------------------ ImageOps.Transform (device)
def _transform(data, dst_cap=64 << 20, **kw):
    import lilliput_amd as la

    d = la.Decoder(data)
    ops = la.ImageOps(2048)
    try:
        kw.setdefault("EncodeTimeout", 60 * 10**9)
        return ops.Transform(d, la.ImageOptions(**kw), dst_cap=dst_cap)
    finally:
        ops.Close()
        d.Close()


def _oracle_animated(oracle, frames, info, method, w, h):
    """ops.go:371-443 for a WebP source, restated on the CPU: every decoded (sub-)frame is blended ("over" in float, ops.go:566-582 ->
    opencv.cpp:556-667) or copied onto the canvas at its offset, the canvas is fitted / resized, then the frame's rectangle is cleared
    when it asks for disposal to the background (ops.go:552-563). Returns [(frame, duration_ms)]."""
    cn = 4 if info["has_alpha"] else 3
    canvas = np.zeros((info["height"], info["width"], cn), dtype=np.uint8)  # ClearToTransparent: zeros (black for three channels)
    out = []
    for px, m in frames:
        fh, fw = px.shape[:2]
        x, y = m["x_offset"], m["y_offset"]
        assert x + fw <= info["width"] and y + fh <= info["height"]
        if m["blend"] == 0:
            canvas[y:y + fh, x:x + fw] = oracle.blend_alpha(px, canvas[y:y + fh, x:x + fw])
        else:
            canvas[y:y + fh, x:x + fw] = px
        out.append((np.array(oracle.transform_static(canvas, 1, w, h, method, False), copy=True), m["duration"]))
        if m["dispose"] == 1:
            canvas[y:y + fh, x:x + fw] = 0
    return out


@pytest.mark.gpu
def test_config3_png_to_webp(W, oracle):
    """BASELINE configs[2]: testdata/ferry_sunset.png -> 512 x 512 WebP. The no-upscale rule makes it 297 x 297; the frame
    handed to the encoder equals the oracle's bit for bit, the file is what the webp.hpp writer makes of that frame (ICC profile of the PNG
    carried in an ICCP chunk), and the reference's libwebp reads it back -- exactly, when lossless is asked for."""
    import lilliput_amd as la

    data = open(os.path.join(ROOT, "tests", "golden", "inputs_png", "ferry_sunset.png"), "rb").read()
    icc = oracle.ref_png_icc(data) if oracle.ref_meta() is not None else None
    px = oracle.ref_png_decode(data) if oracle.ref_png() is not None else None
    d = la.Decoder(data)
    my_icc = d.ICC()
    d.Close()
    if icc is not None:
        assert my_icc == icc and len(icc) > 128
    for q in (85, 101):
        pre = la.parse_raw_frames(_transform(data, FileType=".bgra-frames", Width=512, Height=512, ResizeMethod=la.ImageOpsFit))[0][0]
        assert pre.shape == (297, 297, 3)
        if px is not None:
            assert np.array_equal(pre, oracle.transform_static(px, 1, 512, 512, oracle.FIT, False))
        out = _transform(data, FileType=".webp", Width=512, Height=512, ResizeMethod=la.ImageOpsFit, EncodeOptions={la.WebpQuality: q})
        assert out == product_encode(W, [pre], q, icc=my_icc), q
        if oracle.ref_webp() is not None:
            info = oracle.ref_webp_info(out)
            assert (info["width"], info["height"], info["num_frames"], info["has_alpha"], info["icc_len"]) == (297, 297, 1, 0, len(my_icc))
            assert oracle.ref_webp_icc(out) == my_icc
            back = oracle.ref_webp_frames(out)[0][0]
            if q > 100:
                assert np.array_equal(back, pre)
            else:
                ref_back = oracle.ref_webp_frames(oracle.ref_webp_encode_still(pre, q, my_icc))[0][0]
                assert _psnr(back, pre) > _psnr(ref_back, pre) - 1.5


@pytest.mark.gpu
def test_webp_sources_through_the_animated_loop(W, oracle):
    """Animated (and still) WebP sources through ImageOps.Transform with the raw frame sink: decode (host) -> blend / copy at the frame's
    offset onto the HBM-resident canvas -> Fit / Resize -> dispose, frame by frame, against the CPU restatement fed by the REFERENCE's
    libwebp frames. Bit-exact where the resize is a copy or an integer box, +-1 LSB where its taps are fractional."""
    import lilliput_amd as la

    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    fx = fixtures()
    M = {la.ImageOpsFit: oracle.FIT, la.ImageOpsResize: oracle.RESIZE, la.ImageOpsNoResize: oracle.NO_RESIZE}
    for name, method, w, h, tol in (("complex_dispose_and_blend.webp", la.ImageOpsNoResize, 0, 0, 0), ("party-discord.webp", la.ImageOpsNoResize, 0, 0, 0),
                                    ("animated-webp-supported.webp", la.ImageOpsFit, 200, 200, 0), ("animated-webp-supported.webp", la.ImageOpsResize, 77, 33, 1),
                                    ("big_buck_bunny_720_5s.webp", la.ImageOpsFit, 128, 128, 1), ("ferry_sunset.webp", la.ImageOpsFit, 99, 99, 0),
                                    ("firefox-gray-alpha.webp", la.ImageOpsNoResize, 0, 0, 0)):
        info, frames = oracle.ref_webp_info(fx[name]), oracle.ref_webp_frames(fx[name])
        d = la.Decoder(fx[name])
        hd = d.Header()
        assert d.Description() == "WEBP" and (hd["width"], hd["height"], hd["num_frames"]) == (info["width"], info["height"], info["num_frames"])
        assert d.AnimationInfo() == (info["loop_count"], info["num_frames"], info["total_duration"], info["bgcolor"])
        d.Close()
        got = la.parse_raw_frames(_transform(fx[name], FileType=".bgra-frames", Width=w, Height=h, ResizeMethod=method, dst_cap=192 << 20))
        if info["num_frames"] == 1:  # not animated: decode -> Fit, no canvas (ops.go:450-452, 199-206)
            exp = [(oracle.transform_static(frames[0][0], 1, w, h, M[method], False), frames[0][1]["duration"])]
        else:
            exp = _oracle_animated(oracle, frames, info, M[method], w, h)
        assert len(got) == len(exp), name
        for k, ((g, ms), (e, dur)) in enumerate(zip(got, exp)):
            assert g.shape == e.shape and ms == dur, (name, k, g.shape, e.shape, ms, dur)
            dlt = np.abs(g.astype(int) - e.astype(int)).max()
            assert dlt <= tol, (name, method, k, int(dlt))


@pytest.mark.gpu
def test_config4_animated_sources_to_animated_webp(W, oracle):
    """BASELINE configs[3]: testdata/party-discord.gif and big_buck_bunny_720_5s.webp -> 128 x 128 animated WebP. The frames handed to the
    encoder are the raw-sink frames (checked against the oracle above and in test_gif.py); the file is what the webp.hpp writer makes of
    them; played back by the reference's libwebpdemux it shows those frames -- exactly in lossless mode -- for the source's durations,
    with the source's loop count and background colour."""
    import lilliput_amd as la

    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    gif = open(os.path.join(ROOT, "tests", "golden", "inputs_gif", "party-discord.gif"), "rb").read()
    for data, q, min_psnr in ((gif, 101, None), (gif, 80, 20.0), (fixtures()["big_buck_bunny_720_5s.webp"], 75, 27.0), (fixtures()["party-discord.webp"], 101, None)):
        pre = la.parse_raw_frames(_transform(data, FileType=".bgra-frames", Width=128, Height=128, ResizeMethod=la.ImageOpsFit))
        out = _transform(data, FileType=".webp", Width=128, Height=128, ResizeMethod=la.ImageOpsFit, EncodeOptions={la.WebpQuality: q})
        d = la.Decoder(data)
        loops, nfr, dur, bg = d.AnimationInfo()
        d.Close()
        assert len(pre) == nfr
        played = oracle.ref_webp_play(out)
        assert played is not None
        canv, ts, ploops, pbg = played
        info = oracle.ref_webp_info(out)
        assert (info["width"], info["height"]) == pre[0][0].shape[1::-1] and info["loop_count"] == loops and info["bgcolor"] == bg % 2**32 and info["total_duration"] == dur
        # frames equal to their predecessor were merged by the writer: walk the playback by end timestamps
        t, k = 0, 0
        for f, ms in pre:
            t += ms
            while ts[k] < t:
                k += 1
            want = f if f.shape[2] == 4 else np.concatenate([f, np.full(f.shape[:2] + (1,), 255, np.uint8)], axis=2)
            if min_psnr is None:
                vis = want[:, :, 3] > 0  # lossless keeps colour only where something is visible (WebPConfig.exact = 0, as in the reference)
                assert np.array_equal(canv[k][:, :, 3], want[:, :, 3]) and np.array_equal(canv[k][vis], want[vis]), (q, k)
            else:  # lossy: colour is only meaningful where something is visible
                vis = want[:, :, 3] > 0
                assert _psnr(canv[k][vis][:, :3], want[vis][:, :3]) > min_psnr, (q, k, _psnr(canv[k][vis][:, :3], want[vis][:, :3]))
        assert ts[-1] == dur
    # a WebP source cannot skip frames (webp.go:169-171), so a frame limit fails the Transform exactly as in the reference (ops.go:425-429)
    with pytest.raises(la.LilliputError) as e:
        _transform(fixtures()["party-discord.webp"], FileType=".webp", Width=16, Height=16, ResizeMethod=la.ImageOpsFit, MaxEncodeFrames=3)
    assert e.value.code == 9  # ErrSkipNotSupported
    # frame / duration limits and single-frame output go through the same encoder (ops.go:384-433)
    one = _transform(gif, FileType=".webp", Width=16, Height=16, ResizeMethod=la.ImageOpsFit, DisableAnimatedOutput=True, EncodeOptions={la.WebpQuality: 101})
    assert oracle.ref_webp_info(one)["num_frames"] == 1
    five = _transform(gif, FileType=".webp", Width=16, Height=16, ResizeMethod=la.ImageOpsFit, MaxEncodeFrames=5, EncodeOptions={la.WebpQuality: 101})
    assert oracle.ref_webp_info(five)["total_duration"] == 150 and len(oracle.ref_webp_play(five)[0]) <= 5


@pytest.mark.gpu
def test_animation_writer_file_size_against_the_reference_writer(W, oracle):
    """The product's animation writer places changed rectangles of whole frames (lp_abi_webp.cpp); the reference hands whole canvases to
    libwebp's WebPAnimEncoder (kmin 3 / kmax 4, webp.cpp:631-706), which also tries blended sub-frames and key frames. Same frames on
    playback (test above) -- here: what the difference costs in bytes, on the BASELINE configs[3] sources. The reference writer runs
    over the very frames the product's encoder was handed (the raw frame sink)."""
    import lilliput_amd as la

    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    gif = open(os.path.join(ROOT, "tests", "golden", "inputs_gif", "party-discord.gif"), "rb").read()
    ratios = {}
    for name, data, q in (("party-discord.gif lossless", gif, 101), ("party-discord.gif q80", gif, 80),
                          ("big_buck_bunny_720_5s.webp q75", fixtures()["big_buck_bunny_720_5s.webp"], 75),
                          ("party-discord.webp lossless", fixtures()["party-discord.webp"], 101)):
        pre = la.parse_raw_frames(_transform(data, FileType=".bgra-frames", Width=128, Height=128, ResizeMethod=la.ImageOpsFit))
        out = _transform(data, FileType=".webp", Width=128, Height=128, ResizeMethod=la.ImageOpsFit, EncodeOptions={la.WebpQuality: q})
        d = la.Decoder(data)
        loops, nfr, dur, bg = d.AnimationInfo()
        d.Close()
        frames = np.stack([f if f.shape[2] == 4 else np.concatenate([f, np.full(f.shape[:2] + (1,), 255, np.uint8)], axis=2) for f, _ in pre])
        ref = oracle.ref_webp_encode_anim(frames, [ms for _, ms in pre], q, loops, bg)
        assert ref is not None
        ratios[name] = (len(out), len(ref), round(len(out) / len(ref), 3))
    print("animated WebP, product bytes / reference bytes:", ratios)
    # lossless: rectangles against blended sub-frames cost little; lossy: the reference re-encodes only what changed too
    assert all(r[2] <= 1.15 for r in ratios.values()), ratios


@pytest.mark.gpu
def test_lossy_front_end_planes_equal_libwebps(W, oracle):
    """The lossy still writer codes Y'CbCr 4:2:0 planes; the reference lets libwebp derive them from the BGR frame (WebPEncodeBGR ->
    picture_csp_enc.c: 16-bit fixed-point luma, chroma from the 2 x 2 mean in linear light). The product derives them on the device
    (k_webp_yuv420) for frames the device holds: bit for bit the planes of the reference's libwebp, odd sizes and flat / banded /
    noisy content included; a frame with translucent pixels is left to libwebp's alpha-weighted import."""
    if oracle.ref_webp() is None:
        pytest.skip("reference libwebp driver not built")
    L = W
    L.lilliput_hip_webp_yuv420.restype = C.c_int
    L.lilliput_hip_webp_yuv420.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(8)

    def planes(px):
        h, w, cn = px.shape
        buf = np.ascontiguousarray(px)
        m = L.opencv_mat_create_from_data(w, h, 16 if cn == 3 else 24, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
        uvw, uvh = (w + 1) // 2, (h + 1) // 2
        y, u, v = np.zeros((h, w), np.uint8), np.zeros((uvh, uvw), np.uint8), np.zeros((uvh, uvw), np.uint8)
        r = L.lilliput_hip_webp_yuv420(m, y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
        L.opencv_mat_release(m)
        return r, y, u, v

    for h, w in ((1, 1), (2, 2), (5, 7), (2, 9), (33, 17), (64, 64), (101, 3), (297, 297), (480, 641)):
        for kind in range(4):
            px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            if kind == 1:
                px = (px // 32 * 32).astype(np.uint8)
            elif kind == 2:
                px[:] = rng.integers(0, 256, 3, dtype=np.uint8)
            elif kind == 3:  # opaque BGRA takes the same route
                px = np.concatenate([px, np.full((h, w, 1), 255, np.uint8)], axis=2)
            r, y, u, v = planes(px)
            ry, ru, rv, ralpha = oracle.ref_webp_yuv420(px)
            assert r == 0 and not ralpha
            assert np.array_equal(y, ry) and np.array_equal(u, ru) and np.array_equal(v, rv), (h, w, kind)
    px = rng.integers(0, 256, (9, 9, 4), dtype=np.uint8)
    assert planes(px)[0] == 1 and oracle.ref_webp_yuv420(px)[3]
