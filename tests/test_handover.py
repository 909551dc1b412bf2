"""The pixel hand-over item (include/lilliput_hip.h lilliput_hip_pixels_header): frames decoded outside the library -- the reference
sends AVIF to libavif and MP4 / MOV / WEBM to libavcodec (lilliput.go:136-164), serial codecs that stay on the host -- enter the
device path at BGR(A), through the same orientation / Fit / encode stages as a decoded JPEG."""
import struct

import numpy as np
import pytest


def _handover(px, orientation=1, stride=0, duration_ms=0, pad=0):
    h, w = px.shape[:2]
    cn = 1 if px.ndim == 2 else px.shape[2]
    rowb = w * cn
    st = stride or rowb
    body = b"".join(px[y].tobytes() + b"\xAA" * (st - rowb) for y in range(h))
    if st != rowb:
        body = body[:len(body) - (st - rowb)]   # the last row carries no padding
    return b"LPPIXELS" + struct.pack("<6I", w, h, cn, stride, orientation, duration_ms) + body + b"\0" * pad


def _transform(blob, ext, w, h, method, **kw):
    import lilliput_amd as la
    d = la.Decoder(blob)
    ops = la.ImageOps(8192)
    try:
        return ops.Transform(d, la.ImageOptions(ext, w, h, method, EncodeTimeout=10**10, **kw)), d.Description()
    finally:
        ops.Close()
        d.Close()


def test_header_validation(hip_lib):
    import lilliput_amd as la
    px = np.zeros((4, 5, 3), np.uint8)
    good = _handover(px)
    d = la.Decoder(good)
    assert d.Description() == "PIXELS"
    hd = d.Header()
    assert (hd["width"], hd["height"], hd["num_frames"], hd["orientation"]) == (5, 4, 1, 1)
    d.Close()
    bad = [good[:-1],                                                       # one byte short
           b"LPPIXELS" + struct.pack("<6I", 5, 4, 2, 0, 1, 0) + bytes(40),  # two channels
           b"LPPIXELS" + struct.pack("<6I", 5, 4, 3, 14, 1, 0) + bytes(60), # stride below a row
           b"LPPIXELS" + struct.pack("<6I", 5, 4, 3, 0, 9, 0) + bytes(60),  # orientation out of range
           b"LPPIXELS" + struct.pack("<6I", 0, 4, 3, 0, 1, 0) + bytes(60),
           b"LPPIXELS" + struct.pack("<6I", 70000, 1, 1, 0, 1, 0) + bytes(70000)]
    for b in bad:
        with pytest.raises(la.LilliputError):
            la.Decoder(b)


@pytest.mark.gpu
def test_handed_over_pixels_take_the_jpeg_path(hip_lib, fixture_bytes):
    """The frame our own decoder produces for a JPEG, handed back as pixels, must give the byte-identical thumbnail: everything after
    decode is the same device code."""
    import lilliput_amd as la
    src = fixture_bytes["coast.jpg"]
    raw, _ = _transform(src, ".bgra-frames", 0, 0, la.ImageOpsNoResize)
    (px, _dur), = la.parse_raw_frames(raw)
    h, w, cn = px.shape
    want, _ = _transform(src, ".jpeg", 96, 64, la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
    got, desc = _transform(_handover(px), ".jpeg", 96, 64, la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
    assert desc == "PIXELS" and got == want
    # ... and against the ORACLE's frame for the hand-over route (not only the product's other route): the pre-encode frame within
    # +-1 LSB of transform_static on the handed-over pixels (a fractional scale: float taps), the output its byte-exact encoding
    from oracle import oracle as O
    O.lib()
    for o in (1, 3, 6, 8):
        blob = _handover(px, orientation=o)
        ref = O.transform_any_frame(blob, 96, 64)
        raw2, _ = _transform(blob, ".bgra-frames", 96, 64, la.ImageOpsFit)
        (frame, _d), = la.parse_raw_frames(raw2)
        assert frame.shape == ref.shape and np.abs(frame.astype(int) - ref.astype(int)).max() <= 1, o
        out2, _ = _transform(blob, ".jpeg", 96, 64, la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
        assert out2 == O.jpeg_encode(frame, 85), o
    exact = _handover(np.ascontiguousarray(px[:64, :96]))                   # 96 x 64 -> 48 x 32: an integer scale, bit-exact
    out3, _ = _transform(exact, ".jpeg", 48, 32, la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
    assert out3 == O.transform_any_to_jpeg(exact, 48, 32, 85)
    got, _ = _transform(_handover(px, stride=w * cn + 13, pad=7), ".jpeg", 96, 64, la.ImageOpsFit, EncodeOptions={la.JpegQuality: 85})
    assert got == want                                                      # padded rows, trailing bytes
    # orientation 6 (rotate 90 degrees clockwise to display) == the pre-rotated frame with orientation 1
    rot = np.ascontiguousarray(np.rot90(px, -1))
    a, _ = _transform(_handover(px, orientation=6), ".png", 40, 60, la.ImageOpsFit, NormalizeOrientation=True)
    b, _ = _transform(_handover(rot), ".png", 40, 60, la.ImageOpsFit, NormalizeOrientation=True)
    assert a == b
    # BGRA and gray frames: through the PNG writer and back
    rng = np.random.default_rng(2)
    for shape in ((33, 47, 4), (20, 31)):
        q = rng.integers(0, 256, shape, dtype=np.uint8)
        out, _ = _transform(_handover(q), ".bgra-frames", 0, 0, la.ImageOpsNoResize)
        (back, _d), = la.parse_raw_frames(out)
        assert np.array_equal(back.reshape(q.shape), q)


@pytest.mark.gpu
def test_handover_items_in_a_batch(hip_lib, fixture_bytes):
    import lilliput_amd as la
    src = fixture_bytes["coast.jpg"]
    raw, _ = _transform(src, ".bgra-frames", 0, 0, la.ImageOpsNoResize)
    (px, _dur), = la.parse_raw_frames(raw)
    item = _handover(px)
    b = la.Batch(0)
    res = b.transform([src, item, item[:40], src], 96, 64, quality=85)
    b.close()
    assert [r.status for r in res] == [0, 0, 1, 0]
    assert res[1].data == res[0].data == res[3].data
    from oracle import oracle as O
    O.lib()
    exact = _handover(np.ascontiguousarray(px[:64, :96]))                   # integer scale: the batch's bytes are the oracle's
    b = la.Batch(0)
    res = b.transform([exact, src, exact], 48, 32, quality=85)
    b.close()
    assert res[0].status == 0 and res[0].data == res[2].data == O.transform_any_to_jpeg(exact, 48, 32, 85)
