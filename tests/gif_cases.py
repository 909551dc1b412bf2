"""Deterministic GIF inputs: the reference's own fixtures (tests/golden/inputs_gif, from /root/reference/testdata), hand-built
edge cases and seeded mutations. Shared by tests/test_gif.py and tests/golden/make_gif_golden.py."""
import os
import random
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "inputs_gif")


def fixtures():
    return {n: open(os.path.join(FIX, n), "rb").read() for n in sorted(os.listdir(FIX))}


def lzw_encode(indices, min_code):
    """A plain GIF LZW encoder (variable width, clear at 4096) -> data sub-blocks."""
    clear, eof = 1 << min_code, (1 << min_code) + 1
    table = {bytes([i]): i for i in range(clear)}
    nxt, width = eof + 1, min_code + 1
    out, acc, nbits = bytearray(), 0, 0

    def emit(code):
        nonlocal acc, nbits
        acc |= code << nbits
        nbits += width
        while nbits >= 8:
            out.append(acc & 255)
            acc >>= 8
            nbits -= 8

    emit(clear)
    w = b""
    for px in indices:
        wk = w + bytes([px])
        if wk in table:
            w = wk
            continue
        emit(table[w])
        if nxt < 4096:
            table[wk] = nxt
            nxt += 1
            if nxt > (1 << width) and width < 12:
                width += 1
        else:
            emit(clear)
            table = {bytes([i]): i for i in range(clear)}
            nxt, width = eof + 1, min_code + 1
        w = bytes([px])
    if w:
        emit(table[w])
    emit(eof)
    if nbits:
        out.append(acc & 255)
    blocks = bytearray()
    for i in range(0, len(out), 255):
        chunk = out[i : i + 255]
        blocks += bytes([len(chunk)]) + chunk
    return bytes(blocks) + b"\0"


def gce(disposal=0, delay=0, transparent=None, user=0):
    packed = (disposal << 2) | (user << 1) | (0 if transparent is None else 1)
    return b"\x21\xf9\x04" + bytes([packed]) + struct.pack("<H", delay) + bytes([transparent or 0]) + b"\0"


def image(left, top, w, h, indices, min_code=3, interlace=False, local=None):
    packed = (0x40 if interlace else 0) | ((0x80 | (len(local) // 3).bit_length() - 2) if local else 0)
    if interlace:
        rows = [indices[r * w : (r + 1) * w] for r in range(h)]
        order = [r for o, j in ((0, 8), (4, 8), (2, 4), (1, 2)) for r in range(o, h, j)]
        indices = b"".join(bytes(rows[r]) for r in order)
    return b"\x2c" + struct.pack("<HHHH", left, top, w, h) + bytes([packed]) + (local or b"") + bytes([min_code]) + lzw_encode(bytes(indices), min_code)


def gif(sw, sh, records, palette=None, bg=0, version=b"89a"):
    pal = palette if palette is not None else bytes(v for i in range(8) for v in ((i * 36) & 255, (i * 73) & 255, (255 - i * 30) & 255))
    packed = (0x80 | ((len(pal) // 3).bit_length() - 2)) if pal else 0
    return b"GIF" + version + struct.pack("<HH", sw, sh) + bytes([packed, bg, 0]) + pal + b"".join(records) + b"\x3b"


def hand_cases():
    rnd = random.Random(3)
    px = lambda n, hi=8: bytes(rnd.randrange(hi) for _ in range(n))
    netscape = b"\x21\xff\x0bNETSCAPE2.0\x03\x01\x05\x00\x00"
    c = {
        "static": gif(16, 12, [image(0, 0, 16, 12, px(192))]),
        "interlaced": gif(16, 13, [image(0, 0, 16, 13, px(208), interlace=True)]),
        "transparent_first": gif(16, 12, [gce(0, 5, transparent=2), image(0, 0, 16, 12, px(192))]),
        "partial_frames": gif(20, 20, [netscape, gce(1, 3), image(0, 0, 20, 20, px(400)), gce(2, 4, transparent=1), image(5, 6, 8, 9, px(72)), gce(3, 1),
                                     image(2, 2, 10, 4, px(40)), gce(0, 0, transparent=0), image(12, 1, 8, 19, px(152))]),
        "frame_off_canvas": gif(10, 10, [gce(2, 1), image(6, 7, 8, 8, px(64)), gce(3, 1), image(9, 9, 4, 4, px(16)), gce(1, 1), image(0, 0, 10, 10, px(100))]),
        "frame_outside": gif(10, 10, [image(0, 0, 10, 10, px(100)), gce(2, 2), image(12, 3, 4, 4, px(16)), image(1, 1, 3, 3, px(9))]),
        "local_maps": gif(12, 12, [image(0, 0, 12, 12, px(144, 4), min_code=2, local=bytes(range(12))), gce(2, 9, transparent=3),
                                  image(3, 3, 6, 6, px(36, 16), min_code=4, local=bytes(range(48)))], palette=b""),
        "no_color_map": gif(8, 8, [image(0, 0, 8, 8, px(64))], palette=b""),
        "index_out_of_range": gif(8, 8, [image(0, 0, 8, 8, px(64, 8), min_code=3)], palette=bytes(range(12))),
        "bg_out_of_range": gif(8, 8, [gce(2, 1, transparent=1), image(2, 2, 4, 4, px(16)), image(0, 0, 2, 2, px(4))], bg=200),
        "dispose_prev_first": gif(10, 10, [gce(3, 1), image(2, 2, 5, 5, px(25)), gce(3, 1), image(4, 4, 5, 5, px(25)), image(0, 0, 3, 3, px(9))]),
        "gce_short": gif(8, 8, [b"\x21\xf9\x03\x05\x01\x00\x00", image(0, 0, 8, 8, px(64)), image(1, 1, 2, 2, px(4))]),
        "two_gce": gif(8, 8, [gce(2, 7, transparent=3), gce(1, 9), image(1, 1, 6, 6, px(36))]),
        "comment_and_app": gif(8, 8, [b"\x21\xfe\x05hello\x00", netscape, b"\x21\xff\x0bNETSCAPE2.0\x03\x01\x09\x00\x00", gce(0, 2), image(0, 0, 8, 8, px(64)), gce(0, 0), image(0, 0, 8, 8, px(64))]),
        "zero_size_frame": gif(8, 8, [image(0, 0, 0, 4, b"")]),
        "no_terminator": gif(8, 8, [gce(1, 2), image(0, 0, 8, 8, px(64))])[:-1],
        "junk_record": gif(8, 8, [image(0, 0, 8, 8, px(64)), b"\x00"]),
        "version_87a": gif(8, 8, [image(0, 0, 8, 8, px(64))], version=b"87a"),
        "version_junk": gif(8, 8, [image(0, 0, 8, 8, px(64))], version=b"xyz"),
        "not_gif": b"GIG89a" + gif(8, 8, [image(0, 0, 8, 8, px(64))])[6:],
        "zero_width_screen": gif(0, 8, [image(0, 0, 8, 8, px(64))]),
        "code_size_9": gif(8, 8, [b"\x2c" + struct.pack("<HHHH", 0, 0, 8, 8) + b"\0\x09" + lzw_encode(px(64), 8)]),
        "code_size_1": gif(8, 8, [image(0, 0, 8, 8, px(64, 2), min_code=1)]),
        "big_dictionary": gif(64, 64, [image(0, 0, 64, 64, bytes(rnd.randrange(256) for _ in range(4096)), min_code=8)], palette=bytes(rnd.randrange(256) for _ in range(768))),
        "long_runs": gif(64, 64, [image(0, 0, 64, 64, bytes([1] * 4096))]),
        "short_data": gif(8, 8, [b"\x2c" + struct.pack("<HHHH", 0, 0, 8, 8) + b"\0\x03" + lzw_encode(px(30), 3), image(0, 0, 2, 2, px(4))]),
        "extra_data": gif(8, 8, [b"\x2c" + struct.pack("<HHHH", 0, 0, 4, 4) + b"\0\x03" + lzw_encode(px(64), 3), gce(2, 1), image(0, 0, 2, 2, px(4))]),
    }
    return c


def fuzz_cases(seed, n):
    rnd = random.Random(seed)
    base = list(hand_cases().values())
    fx = fixtures()
    base += [fx[k] for k in ("dispose_bgnd.gif", "no_gce_first_frame.gif", "party-discord.gif", "duplicate_number_of_loops.gif")]
    out = {}
    for it in range(n):
        data = bytearray(rnd.choice(base))
        mode = rnd.randrange(4)
        if mode == 0:
            for _ in range(rnd.randrange(1, 4)):
                data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 1:
            data[rnd.randrange(len(data))] ^= 1 << rnd.randrange(8)
        elif mode == 2:
            data = data[: rnd.randrange(len(data) + 1)]
        else:
            q = rnd.randrange(6, len(data))
            data = data[:q] + data[q + rnd.randrange(1, 3) :]
        out["fz%d_%d" % (seed, it)] = bytes(data)
    return out
