"""BASELINE configs[4] in miniature as a parity test: a 64-item mixed-format stream (JPEG 70 / PNG 15 / WebP 10 / handed-over decoded
frames 5 %, sides 512 - 2048, squares and 4:3) through ONE batch call and through the two-slot node entry point; EVERY item is held
against the reference CPU path (oracle.transform_any_to_jpeg: the reference's own libjpeg-turbo / libpng / libwebp decode ->
OpenCV-semantics INTER_AREA -> libjpeg-turbo-arithmetic encode; SURVEY.md 8d, ops.go:352-479)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mix():
    from lilliput_amd import synth

    pools = synth.firehose_pool(6, 512, 2048, seed=11)
    return synth.firehose_items(pools, 64, seed=12)


def test_mix_generator_follows_the_survey_law():
    from lilliput_amd import synth

    assert [k for k, _ in synth.FIREHOSE_MIX] == ["jpeg", "png", "webp", "pixels"] and abs(sum(p for _, p in synth.FIREHOSE_MIX) - 1.0) < 1e-9
    pools = {k: [b"x"] for k, _ in synth.FIREHOSE_MIX}
    kinds = [k for k, _ in synth.firehose_items(pools, 4000, seed=3)]
    share = {k: kinds.count(k) / 4000 for k, _ in synth.FIREHOSE_MIX}
    assert abs(share["jpeg"] - 0.70) < 0.03 and abs(share["png"] - 0.15) < 0.03 and abs(share["webp"] - 0.10) < 0.03 and abs(share["pixels"] - 0.05) < 0.02


def test_oracle_serves_every_format_of_the_mix(oracle):
    """(no GPU) the checker itself: every format yields a 256 x 256 (or Fit-shaped) JPEG; the JPEG route equals transform_jpeg_thumbnail."""
    from lilliput_amd import synth

    pools = synth.firehose_pool(1, 512, 640, seed=5, workers=1)
    for kind, (d,) in pools.items():
        out = oracle.transform_any_to_jpeg(d, 96, 96, 85)
        if out is None:
            pytest.skip("oracle/_ref is not built here (%s)" % kind)
        assert out[:2] == b"\xff\xd8" and oracle.jpeg_info(out)["width"] <= 96
        if kind == "jpeg":
            assert out == oracle.transform_jpeg_thumbnail(d, 96, 96, 85)


@pytest.mark.gpu
def test_every_item_of_a_mixed_stream_matches_the_reference_path(hip_lib, oracle, mix):
    import lilliput_amd as la

    sys.path.insert(0, ROOT)
    import bench

    datas = [d for _, d in mix]
    kinds = [k for k, _ in mix]
    assert len(set(kinds)) >= 3
    ops = la.ImageOps(8192)
    b = la.Batch(0)
    n = la.Node([0, 0])
    try:
        for who, res in (("batch", b.transform(datas, 256, 256, quality=85, dst_cap=512 << 10)), ("node", n.transform(datas, 256, 256, quality=85, dst_cap=512 << 10))):
            bad = []
            for i, (k, d) in enumerate(mix):
                if res[i].status != 0:
                    bad.append((who, i, k, "status %d" % res[i].status))
                    continue
                v = bench.firehose_check(la, oracle, ops, d, res[i].data, 256, 85)
                assert v is not None, "oracle/_ref must be built on the GPU box"
                if not v:
                    bad.append((who, i, k, len(d)))
            assert not bad, bad[:10]
        # a different target (fractional scales everywhere, Fit to a non-square box)
        res = b.transform(datas[:24], 200, 120, quality=70, dst_cap=512 << 10)
        for i, (k, d) in enumerate(mix[:24]):
            assert res[i].status == 0, (i, k)
            # the reference path's bytes, or a pre-encode frame within +-1 LSB of the oracle's that the output encodes byte-exactly (the rule of
            # the bench's gate; round 3 compared decoded thumbnails to within 8 here)
            assert bench.firehose_check(la, oracle, ops, d, res[i].data, 200, 70, height=120), (i, k)
    finally:
        ops.Close()
        b.close()
        n.close()


def test_avif_feeder_frames_are_the_reference_decoders(oracle):
    """(no GPU) Real AVIF files in the mix: the bench's host feeder (Pillow's bundled libavif -> hand-over item) must hand the library
    the pixels the reference's own libavif + dav1d decode (oracle/_ref/librefavif.so, avif.cpp:164-321), or the gate would compare two
    different images."""
    import struct

    from lilliput_amd import synth

    if not synth.avif_supported() or oracle.ref_avif() is None:
        pytest.skip("Pillow without AVIF, or oracle/_ref/librefavif.so not built")
    for seed, side in ((4001, 512), (4002, 776), (4003, 1032)):
        d = synth.firehose_source("avif", seed, side)
        assert oracle.is_avif(d)
        px, orientation = oracle.ref_avif_decode(d)
        h = synth.avif_to_handover(d)
        w_, h_, cn, stride, ori, _ = struct.unpack("<6I", h[8:32].tobytes())
        assert (h_, w_, cn) == px.shape and ori == orientation == 1 and stride == 0
        assert np.array_equal(h[32:].reshape(px.shape), px), (seed, side)
        assert oracle.transform_any_to_jpeg(d, 128, 128, 85) == oracle.transform_any_to_jpeg(h.tobytes(), 128, 128, 85)
    out = oracle.cpu_path_run([synth.firehose_source("avif", 4004, 640)], 96, 96, 85, threads=1, jobs=1)
    assert out["ok"] == 1 and out["outputs"][0] == oracle.transform_any_to_jpeg(synth.firehose_source("avif", 4004, 640), 96, 96, 85)


@pytest.mark.gpu
def test_avif_items_through_the_feeder_match_the_reference_path(hip_lib, oracle):
    """AVIF file -> host feeder -> hand-over item -> device (orientation, Fit, encode): the reference CPU path's answer, which starts from
    the file (reference libavif + dav1d decode -> INTER_AREA restatement -> libjpeg-turbo encode)."""
    import lilliput_amd as la
    from lilliput_amd import synth

    sys.path.insert(0, ROOT)
    import bench

    if not synth.avif_supported() or oracle.ref_avif() is None:
        pytest.skip("Pillow without AVIF, or oracle/_ref/librefavif.so not built")
    files = [synth.firehose_source("avif", 4100 + i, side) for i, side in enumerate((512, 640, 904, 1400))]
    frames = [synth.avif_to_handover(d) for d in files]
    ops = la.ImageOps(8192)
    b = la.Batch(0)
    try:
        for w, h in ((256, 256), (200, 120)):
            res = b.transform(frames, w, h, quality=85, dst_cap=512 << 10)
            for d, f, r in zip(files, frames, res):
                assert r.status == 0
                assert bench.firehose_check(la, oracle, ops, f.tobytes(), r.data, w, 85, height=h, ref_data=d)
    finally:
        ops.Close()
        b.close()
