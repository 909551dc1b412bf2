"""Damaged, cut and padded JPEG streams for tests/test_damaged.py (CPU and GPU halves draw the same cases).

Kinds (inside the entropy-coded data of the last scan):
  0 a few random byte substitutions        1 cut short, closed with EOI            2 a 32-byte run of 00 / FF / D0 / 7F
  3 a byte pair that looks like a marker below 0xC0 (FF 01 .. FF BF)                4 a restart marker with another number / an RSTn put in
  5 a restart marker removed                6 a real marker (DHT, EOI, SOS, APP0, COM, SOF0, DQT) in the data
  7 cut short, nothing behind (the buffer simply ends)                              8 bytes inserted before / removed before a restart marker
  9 one flipped bit
"""
import io

import numpy as np

KINDS = 10


def bases():
    """Small baseline files: 4:2:0, 4:4:4, restart interval = one MCU row, 4:2:2 with optimised tables, restart intervals of 3 and 1
    MCUs, grey -- each a list entry (name, bytes)."""
    from PIL import Image

    from lilliput_amd import synth

    rgb = synth.synth_rgb(31, 512)[:192, :256]
    out = []
    for name, kw in (("420", {"subsampling": 2}), ("444", {"subsampling": 0}), ("420_rows", {"subsampling": 2, "restart_marker_rows": 1}),
                     ("422_opt", {"subsampling": 1, "optimize": True}), ("420_dri3", {"subsampling": 2, "restart_marker_blocks": 3}),
                     ("444_dri1", {"subsampling": 0, "restart_marker_blocks": 1})):
        b = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(rgb)).save(b, "JPEG", quality=88, **kw)
        out.append((name, b.getvalue()))
    g = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb[:, :, 0])).save(g, "JPEG", quality=70)
    out.append(("grey", g.getvalue()))
    return out


def progressive_bases():
    from PIL import Image

    from lilliput_amd import synth

    rgb = synth.synth_rgb(33, 512)[:96, :128]
    out = []
    for name, kw in (("prog_420", {"subsampling": 2}), ("prog_444_rows", {"subsampling": 0, "restart_marker_rows": 1}), ("prog_420_dri2", {"subsampling": 2, "restart_marker_blocks": 2})):
        b = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(rgb)).save(b, "JPEG", quality=85, progressive=True, **kw)
        out.append((name, b.getvalue()))
    return out


def damage(base, rng, kind):
    d = bytearray(base)
    sos = base.rfind(b"\xff\xda")
    lo, hi = sos + 14, len(d) - 2
    rsts = [i for i in range(lo, hi) if d[i] == 0xFF and 0xD0 <= d[i + 1] <= 0xD7]
    if kind == 0:
        for p in rng.integers(lo, hi, rng.integers(1, 6)):
            d[p] = int(rng.integers(0, 256))
    elif kind == 1:
        d = d[: int(rng.integers(lo + 6, hi))] + b"\xff\xd9"
    elif kind == 2:
        p = int(rng.integers(lo, hi - 40))
        d[p : p + 32] = bytes([int(rng.choice([0, 0xFF, 0xD0, 0x7F]))]) * 32
    elif kind == 3:
        p = int(rng.integers(lo, hi - 4))
        d[p] = 0xFF
        d[p + 1] = int(rng.integers(1, 0xC0))
    elif kind == 4:
        if rsts and rng.integers(0, 2):
            d[int(rng.choice(rsts)) + 1] = 0xD0 + int(rng.integers(0, 8))
        else:
            p = int(rng.integers(lo, hi - 4))
            d[p] = 0xFF
            d[p + 1] = 0xD0 + int(rng.integers(0, 8))
    elif kind == 5:
        if rsts:
            i = int(rng.choice(rsts))
            del d[i : i + 2]
        else:
            d[int(rng.integers(lo, hi))] ^= 0x10
    elif kind == 6:
        p = int(rng.integers(lo, hi - 4))
        d[p] = 0xFF
        d[p + 1] = int(rng.choice([0xC4, 0xD9, 0xDA, 0xE0, 0xFE, 0xC0, 0xDB]))
    elif kind == 7:
        d = d[: int(rng.integers(lo + 6, hi))]
    elif kind == 8:
        if rsts:
            i = int(rng.choice(rsts))
            if rng.integers(0, 2):
                d[i:i] = bytes(int(x) for x in rng.integers(0, 255, rng.integers(1, 9)))
            else:
                del d[max(lo, i - int(rng.integers(1, 9))) : i]
        else:
            p = int(rng.integers(lo, hi - 10))
            del d[p : p + int(rng.integers(1, 9))]
    else:
        d[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
    return bytes(d)


def tails(base, upto=12):
    """The end of the buffer: EOI missing with k filler bytes behind the data, the last k bytes missing, the same closed with EOI.
    Whether cv::JpegDecoder still returns the image depends on where libjpeg's read-ahead falls (lilliput_amd/csrc/lp_jbits.h)."""
    out = []
    for k in range(upto):
        out.append(("pad%d" % k, base[:-2] + b"\x55" * k))
        out.append(("cut%d" % k, base[: -2 - k]))
        out.append(("cut%d_eoi" % k, base[: -2 - k] + b"\xff\xd9"))
    return out


def cases(seed, per_base=60, progressive=False):
    """(tag, bytes) for every base: `per_base` damaged variants (kinds in turn) + the tail cases."""
    rng = np.random.default_rng(seed)
    out = []
    for name, base in (progressive_bases() if progressive else bases()):
        for it in range(per_base):
            out.append(("%s/k%d/%d" % (name, it % KINDS, it), damage(base, rng, it % KINDS)))
        for tag, data in tails(base):
            out.append(("%s/%s" % (name, tag), data))
    return out
