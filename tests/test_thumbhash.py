"""The reference's own ThumbHash known answers (/root/reference/thumbhash_test.go:63-81), reproduced END TO END through the product:
NewDecoder -> ImageOps.Transform(FileType ".thumbhash", NoResize, NormalizeOrientation) -- device decode (JPEG / PNG; WebP payload on the host), device
orientation, samples gathered on the device, hash in the reference's float order."""
import base64
import os

import pytest

import png_cases
from test_png import THUMBHASH as PNG_HASHES


@pytest.mark.gpu
def test_reference_thumbhash_known_answers_through_the_product(golden, fixture_bytes):
    import lilliput_amd as la

    want = {n: g["thumbhash_b64"] for n, g in golden.items() if g["thumbhash_b64"]}
    assert len(want) == 9
    sources = {n: fixture_bytes[n] for n in want}
    want.update(PNG_HASHES)
    sources.update({n: png_cases.fixtures()[n] for n in PNG_HASHES})
    # thumbhash_test.go:78: the one WebP fixture (grey + alpha, lossless) -- container walk + libwebp decode + BGRA on the device
    want["firefox-gray-alpha.webp"] = "4AeKBQA7oFl7lqhmaDBp92yJJ1h2iHB2Rw=="
    sources["firefox-gray-alpha.webp"] = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs_webp", "firefox-gray-alpha.webp"), "rb").read()
    assert len(want) == 15  # every known answer of thumbhash_test.go:63-81
    ops = la.ImageOps(4096)
    for name, data in sources.items():
        d = la.Decoder(data)
        h = d.Header()
        out = ops.Transform(d, la.ImageOptions(".thumbhash", h["width"], h["height"], la.ImageOpsNoResize, True, EncodeTimeout=10**10))
        d.Close()
        assert base64.b64encode(out).decode() == want[name], name
    ops.Close()


@pytest.mark.gpu
def test_thumbhash_of_resized_and_animated_sources_matches_the_oracle(oracle, fixture_bytes):
    import gif_cases
    import lilliput_amd as la

    ops = la.ImageOps(2048)
    # a resized JPEG: the hash is taken from the resized frame
    data = fixture_bytes["large-sunrise.jpg"]
    d = la.Decoder(data)
    out = ops.Transform(d, la.ImageOptions(".thumbhash", 300, 120, la.ImageOpsFit, False, EncodeTimeout=10**10))
    d.Close()
    info = oracle.jpeg_info(data)
    assert out == oracle.thumbhash(oracle.transform_static(oracle.jpeg_decode(data), info["orientation"], 300, 120, oracle.FIT, False))
    # a GIF: first composited frame (BGRA, with transparency)
    gif = gif_cases.fixtures()["party-discord.gif"]
    d = la.Decoder(gif)
    out = ops.Transform(d, la.ImageOptions(".thumbhash", 0, 0, la.ImageOpsNoResize, False, EncodeTimeout=10**10))
    d.Close()
    if oracle.ref_gif() is not None:
        assert out == oracle.thumbhash(oracle.ref_gif_frames(gif, max_frames=1)[2][0][0])
    ops.Close()
