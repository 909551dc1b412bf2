"""Fractional INTER_AREA straight from 4:2:0 planes (lp_area_core.h / k_area_420): the batch path's route for JPEGs whose
Fit / Resize is not an integer scale (a 4000 x 4000 source to 256 x 256, say). The per-pixel walk is host + device code, so its
order of operations is pinned on the CPU against the oracle (decode -> cv::ExifTransform -> crop -> cv::resize INTER_AREA,
opencv.go:294-374 + resize.cpp ResizeArea_Invoker); the GPU tests then run the kernel itself through the batch ABI."""
import ctypes as C
import io

import numpy as np
import pytest


def _jpeg(rgb, w, h, subsampling=2, quality=90):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb[:h, :w])).save(b, "JPEG", quality=quality, subsampling=subsampling)
    return b.getvalue()


def _with_exif_orientation(jpeg, o):
    tiff = b"II*\x00\x08\x00\x00\x00" + b"\x01\x00" + b"\x12\x01\x03\x00\x01\x00\x00\x00" + bytes([o, 0, 0, 0]) + b"\x00\x00\x00\x00"
    payload = b"Exif\x00\x00" + tiff
    return jpeg[:2] + b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload + jpeg[2:]


# (w, h, box w, box h): fractional scales with 6 / 10 / 18 / 34 / 66-tap instantiations, odd sizes (replicated chroma edges, an
# odd last luma row / column), very narrow and very flat sources, scales just above 1 and the widest one the kernel takes
CASES = ((250, 243, 100, 100), (256, 256, 100, 77), (301, 200, 64, 64), (512, 512, 33, 33), (512, 400, 7, 9), (97, 131, 50, 50),
         (333, 222, 100, 100), (512, 512, 300, 300), (6, 200, 5, 50), (511, 509, 17, 16), (5, 5, 4, 4), (512, 9, 500, 2), (509, 512, 8, 500))


def _crop_plan(oracle, w, h, o, tw, th, method):
    """Output size and crop rectangle (oriented coordinates) as oracle.transform_static / ops.go:449-479 derive them, normalize off."""
    fw, fh = (h, w) if o >= 5 else (w, h)
    if method == oracle.FIT:
        nw, nh = oracle.calculate_expected_size(w, h, tw, th)  # header dimensions
        left, top, wpc, hpc = oracle.fit_crop_rect(fw, fh, nw, nh)
        return nw, nh, left, top, wpc, hpc
    return max(tw, 1), max(th, 1), 0, 0, fw, fh


def test_area_walk_on_the_host_matches_the_oracle(hip_lib, oracle):
    from lilliput_amd import synth

    u8p = C.POINTER(C.c_uint8)
    fn = hip_lib.lilliput_hip_area420_host
    fn.argtypes = [u8p, u8p, u8p, C.c_uint32, C.c_uint32] + [C.c_int] * 10 + [u8p]
    fn.restype = C.c_int
    rgb = synth.synth_rgb(11, 512)
    rng = np.random.default_rng(5)
    cases = list(CASES)
    for _ in range(8):
        w, h = int(rng.integers(5, 512)), int(rng.integers(2, 512))
        cases.append((w, h, int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1))))
    ran = 0
    for n, (w, h, tw, th) in enumerate(cases):
        for ss in ((2, 1, 0) if n % 3 == 0 else (2,) if n % 3 == 1 else (1, 0)):  # PIL's subsampling: 2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4
            data = _jpeg(rgb, w, h, ss)
            planes = [np.ascontiguousarray(oracle.jpeg_decode_plane(data, c)) for c in range(3)]
            px = oracle.jpeg_decode(data)
            for o in range(1, 9):
                for method in (oracle.FIT, oracle.RESIZE):
                    exp = oracle.transform_static(px, o, tw, th, method, False)
                    nw, nh, left, top, wpc, hpc = _crop_plan(oracle, w, h, o, tw, th, method)
                    out = np.zeros((nh, nw, 3), np.uint8)
                    rc = fn(planes[0].ctypes.data_as(u8p), planes[1].ctypes.data_as(u8p), planes[2].ctypes.data_as(u8p), planes[0].shape[1],
                            planes[1].shape[1], w, h, ss, o, left, top, wpc, hpc, nw, nh, out.ctypes.data_as(u8p))
                    if rc == 1:  # integer scale, an up-scaling axis, or more than 66 (34 when the axes swap) taps: not these kernels'
                        continue
                    assert rc == 0
                    ran += 1
                    assert exp.shape == out.shape and np.array_equal(exp, out), (w, h, tw, th, ss, o, method)
    assert ran > 400


@pytest.mark.gpu
def test_fractional_scales_all_orientations_bit_exact(batch, oracle):
    """Orientations 1-4 take k_area_420, 5-8 k_area_420t (4:2:0, 4:2:2 and 4:4:4 each have their instantiations); tap counts past the
    kernels' take the frame route: all must give the oracle's bytes."""
    from lilliput_amd import synth

    rgb = synth.synth_rgb(11, 512)
    for (w, h, tw, th) in CASES:
        for ss in ((2,) if (w * 7 + h) % 3 else (2, 1, 0)):
            data = _jpeg(rgb, w, h, ss)
            for o in range(1, 9):
                d = _with_exif_orientation(data, o)
                for method in (oracle.FIT, oracle.RESIZE):
                    r = batch.transform([d], tw, th, method=method, normalize=False, quality=85)[0]
                    assert r.status == 0
                    frame = oracle.transform_static(oracle.jpeg_decode(d), o, tw, th, method, False)
                    assert (r.width, r.height) == (frame.shape[1], frame.shape[0]), (w, h, tw, th, o, method)
                    assert r.data == oracle.jpeg_encode(frame, 85), (w, h, tw, th, ss, o, method)


@pytest.mark.gpu
def test_fractional_scales_in_one_mixed_batch(batch, oracle):
    """One call with images of every route (integer scale, fractional row-wise, fractional axis-swapping; 4:2:0, 4:2:2, 4:4:4) and several tap counts:
    the kernels share launches through their op lists."""
    from lilliput_amd import synth

    rgb = synth.synth_rgb(23, 512)
    srcs = []
    for i, (w, h) in enumerate(((500, 500), (512, 512), (400, 300), (333, 222), (512, 256), (250, 243), (97, 131), (300, 500))):
        d = _jpeg(rgb, w, h, (2, 2, 1, 0)[i % 4])
        srcs.append(_with_exif_orientation(d, 1 + (i * 3) % 8))
    res = batch.transform(srcs, 64, 64, quality=85)
    for d, r in zip(srcs, res):
        assert r.status == 0
        o = oracle.jpeg_info(d)["orientation"]
        frame = oracle.transform_static(oracle.jpeg_decode(d), o, 64, 64, oracle.FIT, False)
        assert r.data == oracle.jpeg_encode(frame, 85)
