"""giflib_decoder_* ABI (lp_abi_gif.cpp): host container/LZW reader against the reference's giflib 5.2.2, and the device
compositing against the reference's render loop (oracle/ref_gif_driver.c), on the reference's own GIF fixtures, hand-built
edge cases and seeded mutations. Recorded answers: tests/golden/gif_golden.json (tests/golden/make_gif_golden.py)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest
from conftest import fresh_seed

import gif_cases

HERE = os.path.dirname(os.path.abspath(__file__))
CV_8UC4 = 24


class _Info(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("loop_count", "frame_count", "bg_red", "bg_green", "bg_blue", "bg_alpha", "duration_ms")]


@pytest.fixture(scope="module")
def G(hip_lib):
    L = hip_lib
    L.giflib_decoder_create.restype = C.c_void_p
    L.giflib_decoder_create.argtypes = [C.c_void_p]
    for n in ("get_width", "get_height", "get_num_frames", "get_frame_width", "get_frame_height", "get_prev_frame_delay", "get_prev_frame_disposal",
              "decode_frame_header", "skip_frame", "release"):
        getattr(L, "giflib_decoder_" + n).argtypes = [C.c_void_p]
    L.giflib_decoder_release.restype = None
    L.giflib_decoder_decode_frame.argtypes = [C.c_void_p, C.c_void_p]
    L.giflib_decoder_decode_frame.restype = C.c_bool
    L.giflib_decoder_get_animation_info.argtypes = [C.c_void_p]
    L.giflib_decoder_get_animation_info.restype = _Info
    L.lilliput_hip_gif_read_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]
    return L


class Dec:
    """gifDecoder of giflib.go:56-242 over the C ABI."""

    def __init__(self, L, data):
        self.L = L
        self.buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        self.mat = L.opencv_mat_create_from_data(len(data), 1, 0, self.buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)))
        self.h = L.giflib_decoder_create(self.mat) if self.mat else None

    def close(self):
        if self.h:
            self.L.giflib_decoder_release(self.h)
        if self.mat:
            self.L.opencv_mat_release(self.mat)

    def info(self):
        i = self.L.giflib_decoder_get_animation_info(self.h)
        return [i.loop_count, i.frame_count, i.bg_red, i.bg_green, i.bg_blue, i.bg_alpha, i.duration_ms]


def host_walk(L, data, skip=()):
    """Host half only (no GPU): per frame (meta[10], indices); final state 1 eof / 2 header error / 3 decode failed."""
    d = Dec(L, data)
    if not d.h:
        d.close()
        return None
    w, h = L.giflib_decoder_get_width(d.h), L.giflib_decoder_get_height(d.h)
    frames, st, k = [], 0, 0
    while True:
        if k in skip:
            st = L.giflib_decoder_skip_frame(d.h)
            k += 1
            if st:
                st = 1 if st == 1 else 2
                break
            continue
        st = L.giflib_decoder_decode_frame_header(d.h)
        k += 1
        if st:
            st = 1 if st == 1 else 2
            break
        fw, fh = L.giflib_decoder_get_frame_width(d.h), L.giflib_decoder_get_frame_height(d.h)
        idx = np.zeros(max(1, fw * fh), dtype=np.uint8)
        meta = (C.c_int * 10)()
        n = L.lilliput_hip_gif_read_frame(d.h, idx.ctypes.data, idx.size, meta, None)
        if n < 0:
            st = 3
            break
        frames.append((list(meta), idx[:n].copy()))
    info = d.info()
    d.close()
    return w, h, frames, st, info


def _host_digest(r):
    if r is None:
        return "none"
    h = hashlib.sha1()
    for meta, idx in r[2]:
        h.update(np.array(meta[:10], dtype=np.int32).tobytes())
        h.update(idx.tobytes())
    return "%dx%d:%d:%d:%s:%s" % (r[0], r[1], len(r[2]), r[3], ",".join(map(str, r[4])), h.hexdigest()[:16])


def _all_inputs():
    c = dict(gif_cases.fixtures())
    c.update(gif_cases.hand_cases())
    c.update(gif_cases.fuzz_cases(41, 500))
    return c


def test_host_reader_matches_recorded_giflib_answers(G):
    gold = json.load(open(os.path.join(HERE, "golden", "gif_golden.json")))["host"]
    bad = [k for k, v in _all_inputs().items() if _host_digest(host_walk(G, v)) != gold[k]]
    assert not bad, bad[:10]


def test_host_reader_matches_giflib_live(G, oracle):
    if oracle.ref_gif() is None:
        pytest.skip("oracle/_ref/librefgif.so not built (needs /root/reference)")
    cases = dict(gif_cases.fixtures())
    cases.update(gif_cases.hand_cases())
    cases.update(gif_cases.fuzz_cases(fresh_seed(77), 700))
    for name, data in cases.items():
        for skip in ((), (1,)):
            mine, ref = host_walk(G, data, skip), oracle.ref_gif_frames(data, skip=skip)
            assert (mine is None) == (ref is None), name
            if mine is None:
                continue
            assert mine[:2] == ref[:2] and mine[3] == ref[3] and len(mine[2]) == len(ref[2]), (name, skip, mine[3], ref[3])
            for (mm, mi), (_, rm, ri) in zip(mine[2], ref[2]):
                assert mm == rm[:10] and np.array_equal(mi, ri), (name, skip)
            assert mine[4] == oracle.ref_gif_info(data), name


def test_frame_whose_pixel_count_overflows_an_int_is_refused_like_the_reference(G):
    """giflib.cpp giflib_decoder_decode_frame: `desc.Width > INT_MAX / desc.Height` fails the frame. A 65535 x 65535 frame (4 294 836 225
    pixels) must not be allocated, must not wrap get_line's int length and must not unwind through the C ABI."""
    L = G
    data = gif_cases.gif(65535, 65535, [gif_cases.image(0, 0, 65535, 65535, [0, 1, 2, 3])])
    d = Dec(L, data)
    assert d.h
    assert L.giflib_decoder_decode_frame_header(d.h) == 0
    assert (L.giflib_decoder_get_frame_width(d.h), L.giflib_decoder_get_frame_height(d.h)) == (65535, 65535)
    idx = np.zeros(16, dtype=np.uint8)
    meta = (C.c_int * 10)()
    assert L.lilliput_hip_gif_read_frame(d.h, idx.ctypes.data, idx.size, meta, None) == -1
    d.close()


def test_animation_info_known_answers_of_the_reference_tests(G):
    """giflib_test.go:201-240: loop count, frame count, total duration."""
    fx = gif_cases.fixtures()
    for name, loops, frames, ms in (("party-discord.gif", 0, 16, 480), ("ferry_sunset.gif", 1, 1, 0), ("no-loop.gif", 1, 44, 4400),
                                    ("duplicate_number_of_loops.gif", 2, 2, 0), ("dispose_bgnd.gif", 0, 5, 5000)):
        d = Dec(G, fx[name])
        i = d.info()
        d.close()
        assert (i[0], i[1], i[6]) == (loops, frames, ms), name


def device_frames(L, data, lazy=False):
    """Full decode through the ABI the way gifDecoder.DecodeTo drives it: a fresh Mat header over the same pixel buffer per frame."""
    d = Dec(L, data)
    if not d.h:
        d.close()
        return None
    w, h = L.giflib_decoder_get_width(d.h), L.giflib_decoder_get_height(d.h)
    fb = np.zeros(w * h * 4, dtype=np.uint8)
    frames, st = [], 0
    while True:
        m = L.opencv_mat_create_from_data(w, h, CV_8UC4, fb.ctypes.data_as(C.c_void_p), C.c_size_t(fb.size))
        st = L.giflib_decoder_decode_frame_header(d.h)
        if st:
            L.opencv_mat_release(m)
            st = 1 if st == 1 else 2
            break
        ok = L.giflib_decoder_decode_frame(d.h, m)
        if ok and lazy:
            assert L.lilliput_hip_mat_sync_host(m) == 0
        L.opencv_mat_release(m)
        if not ok:
            st = 3
            break
        frames.append((fb.reshape(h, w, 4).copy(), L.giflib_decoder_get_prev_frame_delay(d.h), L.giflib_decoder_get_prev_frame_disposal(d.h)))
    d.close()
    return w, h, frames, st


def _canvas_digest(r):
    if r is None:
        return "none"
    h = hashlib.sha1()
    for f in r[2]:
        h.update(f[0].tobytes())
    return "%dx%d:%d:%d:%s" % (r[0], r[1], len(r[2]), r[3], h.hexdigest()[:16])


@pytest.mark.gpu
def test_frames_composited_on_device_match_recorded_reference(G):
    gold = json.load(open(os.path.join(HERE, "golden", "gif_golden.json")))["canvas"]
    cases = dict(gif_cases.fixtures())
    cases.update(gif_cases.hand_cases())
    cases.update(gif_cases.fuzz_cases(41, 120))
    bad = [k for k, v in cases.items() if _canvas_digest(device_frames(G, v)) != gold[k]]
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_frames_composited_on_device_match_reference_live(G, oracle):
    if oracle.ref_gif() is None:
        pytest.skip("oracle/_ref/librefgif.so not built")
    cases = dict(gif_cases.fixtures())
    cases.update(gif_cases.hand_cases())
    for name, data in cases.items():
        mine, ref = device_frames(G, data), oracle.ref_gif_frames(data)
        assert (mine is None) == (ref is None), name
        if mine is None:
            continue
        assert mine[:2] == ref[:2] and mine[3] == ref[3] and len(mine[2]) == len(ref[2]), name
        disp = {0: 0, 1: 0, 2: 1, 3: 2}
        for k, ((canvas, delay, dispose), (rc, rm, _)) in enumerate(zip(mine[2], ref[2])):
            assert np.array_equal(canvas, rc), (name, k)
            assert delay == rm[6] and dispose == disp.get(rm[5], 0), (name, k)


@pytest.mark.gpu
def test_canvas_stays_on_device_with_lazy_write_back(G):
    """Lazy host write-back: nothing reaches the pixel buffer until asked, and the frames are the same."""
    data = gif_cases.fixtures()["restore_previous.gif"]
    eager = device_frames(G, data)
    G.lilliput_hip_set_lazy_host(1)
    try:
        lazy = device_frames(G, data, lazy=True)
    finally:
        G.lilliput_hip_set_lazy_host(0)
    assert len(eager[2]) == len(lazy[2]) == 12
    for a, b in zip(eager[2], lazy[2]):
        assert np.array_equal(a[0], b[0])


# ------------------------------------------------------------------------------------------ ImageOps.Transform on GIF sources
def _transform(data, **kw):
    import lilliput_amd as la

    d = la.Decoder(data)
    ops = la.ImageOps(1024)
    try:
        kw.setdefault("EncodeTimeout", 30 * 10**9)
        return ops.Transform(d, la.ImageOptions(**kw), dst_cap=32 << 20)
    finally:
        ops.Close()
        d.Close()


@pytest.mark.gpu
def test_gif_to_jpeg_thumbnail_is_the_first_composited_frame(G, oracle):
    """GIF -> .jpeg: the OpenCV encoder returns content on the first frame, so the output is frame 0 through the animated
    composite path (ops.go:173-197), alpha dropped by the JPEG writer."""
    import lilliput_amd as la

    for name, w, h in (("party-discord.gif", 16, 16), ("restore_previous.gif", 64, 48), ("ferry_sunset.gif", 200, 200)):
        data = gif_cases.fixtures()[name]
        out = _transform(data, FileType=".jpeg", Width=w, Height=h, ResizeMethod=la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
        canvas = oracle.ref_gif_frames(data, max_frames=1)[2][0][0]
        exp = oracle.transform_static(canvas, 1, w, h, oracle.FIT, False)
        assert out == oracle.jpeg_encode(exp, 85), name


@pytest.mark.gpu
def test_animated_loop_composites_resizes_and_disposes_every_frame(G, oracle):
    """The whole animated loop of ImageOps.Transform (ops.go:371-443) through the raw frame sink: every output frame is the
    reference canvas of that frame, fitted / resized / passed through, with the frame delay carried along."""
    import lilliput_amd as la

    fx = gif_cases.fixtures()
    for name in ("party-discord.gif", "restore_previous.gif", "dispose_bgnd.gif", "no_gce_first_frame.gif"):
        ref = oracle.ref_gif_frames(fx[name])
        for method, w, h in ((la.ImageOpsFit, 20, 12), (la.ImageOpsResize, 33, 17), (la.ImageOpsNoResize, 0, 0), (la.ImageOpsFit, 4096, 4096)):
            frames = la.parse_raw_frames(_transform(fx[name], FileType=".bgra-frames", Width=w, Height=h, ResizeMethod=method))
            assert len(frames) == len(ref[2]), (name, method)
            for k, ((got, ms), (canvas, meta, _)) in enumerate(zip(frames, ref[2])):
                exp = oracle.transform_static(canvas, 1, w, h, {la.ImageOpsFit: oracle.FIT, la.ImageOpsResize: oracle.RESIZE, la.ImageOpsNoResize: oracle.NO_RESIZE}[method], False)
                assert got.shape == exp.shape and np.array_equal(got, exp), (name, method, k)
                assert ms == meta[6] * 10, (name, k)


@pytest.mark.gpu
def test_animated_loop_limits(G, oracle):
    import lilliput_amd as la

    data = gif_cases.fixtures()["no-loop.gif"]           # 44 frames, 100 ms each
    d = la.Decoder(data)
    assert d.AnimationInfo()[:3] == (1, 44, 4400) and d.Description() == "GIF" and d.ICC() == b""
    assert d.Header()["num_frames"] == 44 and d.Header()["pixel_type"] == CV_8UC4
    d.Close()
    n = lambda **kw: len(la.parse_raw_frames(_transform(data, FileType=".bgra-frames", Width=32, Height=32, **kw)))
    assert n() == 44
    assert n(MaxEncodeFrames=5) == 5                      # skipToEnd + flush (ops.go:425-431)
    assert n(DisableAnimatedOutput=True) == 1             # ops.go:420-423
    assert n(MaxEncodeDuration=1050 * 10**6) == 10        # frames are dropped once their running duration passes the cap (ops.go:384-390)
    with pytest.raises(la.LilliputError) as e:
        _transform(data, FileType=".bgra-frames", Width=32, Height=32, EncodeTimeout=1)
    assert e.value.code == 7                              # ErrEncodeTimeout after the first frame
    # a static source through the same sink: one frame, delivered by the flush
    jpg = open(os.path.join(HERE, "golden", "inputs", "coast.jpg"), "rb").read()
    fr = la.parse_raw_frames(_transform(jpg, FileType=".bgra-frames", Width=32, Height=24, ResizeMethod=la.ImageOpsFit))
    assert len(fr) == 1 and np.array_equal(fr[0][0], oracle.transform_static(oracle.jpeg_decode(jpg), 1, 32, 24, oracle.FIT, False))


@pytest.mark.gpu
def test_batch_serves_gif_items_next_to_jpegs(G, oracle, fixture_bytes):
    """lilliput_hip_batch_transform: GIF items take the one-image path (first frame -> JPEG) while the JPEG parts run."""
    import lilliput_amd as la

    fx = gif_cases.fixtures()
    names = ["party-discord.gif", "restore_previous.gif", "ferry_sunset.gif"]
    sources = [fixture_bytes["coast.jpg"], fx[names[0]], fixture_bytes["field.jpg"], fx[names[1]], fx[names[2]], b"GIF89a" + b"\0" * 4]
    b = la.Batch(0)
    res = b.transform(sources, 48, 48, quality=80)
    b.close()
    assert [r.status for r in res] == [0, 0, 0, 0, 0, 1]
    for k, n in ((1, names[0]), (3, names[1]), (4, names[2])):
        canvas = oracle.ref_gif_frames(fx[n], max_frames=1)[2][0][0]
        exp = oracle.transform_static(canvas, 1, 48, 48, oracle.FIT, False)
        assert res[k].data == oracle.jpeg_encode(exp, 80), n
        assert (res[k].width, res[k].height) == exp.shape[1::-1]
    assert res[0].data == oracle.transform_jpeg_thumbnail(fixture_bytes["coast.jpg"], 48, 48, 80)


# ------------------------------------------------------------------------------------------ GIF -> GIF
@pytest.mark.gpu
def test_gif_to_gif_is_byte_identical_to_the_reference_writer(G, oracle):
    """ImageOps.Transform with FileType ".gif": decoder canvases -> composite -> Fit -> palette mapping on the device -> LZW on
    the host. The bytes equal what the reference's libgif writes when driven like giflib.cpp:762-1306."""
    import lilliput_amd as la

    if oracle.ref_gif() is None:
        pytest.skip("oracle/_ref/librefgif.so not built")
    fx = gif_cases.fixtures()
    hand = gif_cases.hand_cases()
    sources = {n: fx[n] for n in ("party-discord.gif", "restore_previous.gif", "dispose_bgnd.gif", "no_gce_first_frame.gif", "duplicate_number_of_loops.gif", "no-loop.gif")}
    sources.update({n: hand[n] for n in ("partial_frames", "local_maps", "interlaced", "transparent_first", "comment_and_app", "big_dictionary", "long_runs", "two_gce")})
    for name, data in sources.items():
        hdr = oracle.ref_gif_frames(data, max_frames=1)
        w, h = hdr[0], hdr[1]
        for method, tw, th in ((la.ImageOpsFit, max(1, w // 2), max(1, h // 2)), (la.ImageOpsNoResize, 0, 0), (la.ImageOpsResize, w + 3, max(1, h - 1))):
            om = {la.ImageOpsFit: oracle.FIT, la.ImageOpsResize: oracle.RESIZE, la.ImageOpsNoResize: oracle.NO_RESIZE}[method]
            exp = oracle.ref_gif_transcode(data, lambda c: oracle.transform_static(c, 1, tw, th, om, False))
            got = _transform(data, FileType=".gif", Width=tw, Height=th, ResizeMethod=method)
            assert exp is not None and got == exp, (name, method, len(got), len(exp))


@pytest.mark.gpu
def test_gif_output_needs_a_gif_source_and_respects_frame_limits(G, oracle, fixture_bytes):
    import lilliput_amd as la

    with pytest.raises(la.LilliputError) as e:
        _transform(fixture_bytes["coast.jpg"], FileType=".gif", Width=16, Height=16)
    assert e.value.code == 10
    data = gif_cases.fixtures()["no-loop.gif"]
    got = _transform(data, FileType=".gif", Width=40, Height=40, MaxEncodeFrames=3)
    if oracle.ref_gif() is not None:
        assert got == oracle.ref_gif_transcode(data, lambda c: oracle.transform_static(c, 1, 40, 40, oracle.FIT, False), max_frames=3)
    back = oracle.ref_gif_frames(got) if oracle.ref_gif() is not None else None
    assert back is None or (len(back[2]) == 3 and back[:2] == (40, 40))


@pytest.mark.gpu
def test_reference_known_answer_no_gce_first_frame(G):
    """giflib_test.go:20-66 (discord/lilliput#267): a first frame without a Graphic Control Extension declares no transparent
    colour, so palette index 0 (opaque yellow, the left half of frame 0) must survive GIF -> GIF: pixel (4,4) of the first output
    frame is (255,255,0,255)."""
    import lilliput_amd as la

    data = gif_cases.fixtures()["no_gce_first_frame.gif"]
    d = la.Decoder(data)
    h = d.Header()
    d.Close()
    out = _transform(data, FileType=".gif", Width=h["width"], Height=h["height"], ResizeMethod=la.ImageOpsNoResize)
    frames = device_frames(G, out)
    assert frames is not None and len(frames[2]) >= 1
    b, g, r, a = (int(v) for v in frames[2][0][0][4, 4])
    assert (r, g, b, a) == (255, 255, 0, 255)
