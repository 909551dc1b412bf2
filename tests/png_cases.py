"""Deterministic PNG inputs: the reference's fixtures (tests/golden/inputs_png, from /root/reference/data and testdata), generated
files covering every colour type / bit depth / filter / Adam7 / tRNS / PLTE combination, and seeded mutations.
Shared by tests/test_png.py and tests/golden/make_png_golden.py."""
import os
import random
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "inputs_png")
SIG = b"\x89PNG\r\n\x1a\n"
CH = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
A7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def fixtures():
    return {n: open(os.path.join(FIX, n), "rb").read() for n in sorted(os.listdir(FIX))}


def chunk(t, d, crc=None):
    c = zlib.crc32(t + d) if crc is None else crc
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", c & 0xFFFFFFFF)


def _filter_row(ft, cur, prev, bpp):
    out = bytearray(len(cur))
    for i, v in enumerate(cur):
        a = cur[i - bpp] if i >= bpp else 0
        b = prev[i] if prev else 0
        c = prev[i - bpp] if prev and i >= bpp else 0
        if ft == 0: p = 0
        elif ft == 1: p = a
        elif ft == 2: p = b
        elif ft == 3: p = (a + b) >> 1
        else:
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
        out[i] = (v - p) & 255
    return bytes(out)


def _pack_rows(samples, w, h, depth, nch):
    """samples[y][x] = tuple of nch sample values -> packed row bytes."""
    rows = []
    for y in range(h):
        if depth == 16:
            rows.append(b"".join(struct.pack(">H", s) for px in samples[y] for s in px))
        elif depth == 8:
            rows.append(bytes(s for px in samples[y] for s in px))
        else:
            bits, acc, n = bytearray(), 0, 0
            for px in samples[y]:
                acc = (acc << depth) | px[0]; n += depth
                if n == 8: bits.append(acc); acc = n = 0
            if n: bits.append(acc << (8 - n))
            rows.append(bytes(bits))
    return rows


def make_png(w, h, ct, depth, rnd, interlace=False, filters=None, extra=(), palette_n=None, trns=None, level=6, idat_split=0, smooth=True):
    nch = CH[ct]
    maxv = (1 << depth) - 1
    if ct == 3:
        pn = palette_n or min(256, 1 << depth)
        samples = [[(rnd.randrange(min(pn + (2 if palette_n else 0), 1 << depth)),) for _ in range(w)] for _ in range(h)]
    elif smooth:
        samples = [[tuple(min(maxv, max(0, ((x * 7 + y * 3 + c * 40) * maxv // 255 + rnd.randrange(-3, 4)) & maxv if depth >= 8 else rnd.randrange(maxv + 1))) for c in range(nch)) for x in range(w)] for y in range(h)]
    else:
        samples = [[tuple(rnd.randrange(maxv + 1) for _ in range(nch)) for _ in range(w)] for _ in range(h)]
    bits = depth * nch
    bpp = max(1, bits // 8)
    raw = bytearray()
    passes = A7 if interlace else ((0, 0, 1, 1),)
    for (x0, y0, dx, dy) in passes:
        sub = [row[x0::dx] for row in samples[y0::dy]]
        if not sub or not sub[0]:
            continue
        rows = _pack_rows(sub, len(sub[0]), len(sub), depth, nch)
        prev = None
        for r, cur in enumerate(rows):
            ft = filters[r % len(filters)] if filters else rnd.randrange(5)
            raw += bytes([ft]) + _filter_row(ft, cur, prev, bpp)
            prev = cur
    z = zlib.compress(bytes(raw), level)
    chunks = [chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ct, 0, 0, 1 if interlace else 0))]
    if ct == 3:
        chunks.append(chunk(b"PLTE", bytes(rnd.randrange(256) for _ in range(3 * pn))))
    if trns is not None:
        chunks.append(chunk(b"tRNS", trns))
    chunks += list(extra)
    if idat_split:
        step = max(1, len(z) // idat_split)
        chunks += [chunk(b"IDAT", z[i : i + step]) for i in range(0, len(z), step)]
    else:
        chunks.append(chunk(b"IDAT", z))
    chunks.append(chunk(b"IEND", b""))
    return SIG + b"".join(chunks), samples


def generated():
    rnd = random.Random(9)
    out = {}
    for ct, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for il in (False, True):
                w, h = rnd.randrange(1, 40), rnd.randrange(1, 40)
                out["ct%d_d%d_%s_%dx%d" % (ct, depth, "a7" if il else "np", w, h)] = make_png(w, h, ct, depth, rnd, interlace=il)[0]
    for ft in range(5):
        out["filter%d_rgba" % ft] = make_png(70, 67, 6, 8, rnd, filters=[ft])[0]
        out["filter%d_rgb16" % ft] = make_png(33, 70, 2, 16, rnd, filters=[ft])[0]
        out["filter%d_gray1" % ft] = make_png(77, 66, 0, 1, rnd, filters=[ft])[0]
    out["tall_130_rows"] = make_png(9, 130, 2, 8, rnd)[0]
    out["wide"] = make_png(300, 3, 6, 8, rnd)[0]
    out["one_pixel"] = make_png(1, 1, 2, 8, rnd)[0]
    out["a7_tiny_2x2"] = make_png(2, 2, 6, 8, rnd, interlace=True)[0]
    out["a7_1x5"] = make_png(1, 5, 0, 8, rnd, interlace=True)[0]
    out["idat_split"] = make_png(40, 40, 2, 8, rnd, idat_split=7)[0]
    out["stored"] = make_png(20, 20, 6, 8, rnd, level=0)[0]
    # tRNS variants
    png, smp = make_png(24, 24, 2, 8, rnd, smooth=False)
    key = smp[3][4]
    out["rgb8_key"] = make_png(24, 24, 2, 8, random.Random(1), trns=struct.pack(">HHH", *key), smooth=False)[0]
    out["rgb8_key_hi_bits"] = make_png(24, 24, 2, 8, random.Random(1), trns=struct.pack(">HHH", key[0] | 0x300, key[1], key[2]), smooth=False)[0]
    out["rgb16_key"] = make_png(12, 12, 2, 16, random.Random(2), trns=struct.pack(">HHH", 7, 8, 9), smooth=False)[0]
    out["gray8_key"] = make_png(12, 12, 0, 8, random.Random(3), trns=struct.pack(">H", 5))[0]
    out["pal_trns"] = make_png(30, 30, 3, 8, random.Random(4), trns=bytes(range(0, 200, 10)))[0]
    out["pal_small_palette_big_indices"] = make_png(30, 30, 3, 8, random.Random(5), palette_n=5)[0]
    out["pal4_trns"] = make_png(31, 9, 3, 4, random.Random(6), trns=bytes([0, 128]))[0]
    out["pal_trns_too_long"] = make_png(8, 8, 3, 2, random.Random(7), trns=bytes(9))[0]
    out["rgba_with_trns"] = make_png(8, 8, 6, 8, random.Random(8), trns=bytes(6))[0]
    out["rgb_bad_trns_len"] = make_png(8, 8, 2, 8, random.Random(8), trns=bytes(5))[0]
    # damage
    good = make_png(16, 16, 2, 8, random.Random(10))[0]
    idat = good.find(b"IDAT")
    out["bad_filter_byte"] = None  # built below
    raw = bytearray()
    r2 = random.Random(11)
    rows = [bytes(r2.randrange(256) for _ in range(48)) for _ in range(16)]
    for r, row in enumerate(rows):
        raw += bytes([7 if r == 9 else 0]) + row
    out["bad_filter_byte"] = SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", 16, 16, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")
    clean = b"".join(b"\0" + row for row in rows)
    zraw = zlib.compress(clean)
    ih = chunk(b"IHDR", struct.pack(">IIBBBBB", 16, 16, 8, 2, 0, 0, 0))
    out["short_data"] = SIG + ih + chunk(b"IDAT", zlib.compress(clean[:300])) + chunk(b"IEND", b"")
    out["extra_data"] = SIG + ih + chunk(b"IDAT", zlib.compress(clean + b"\0" * 99)) + chunk(b"IEND", b"")
    out["bad_adler"] = SIG + ih + chunk(b"IDAT", zraw[:-4] + b"\1\2\3\4") + chunk(b"IEND", b"")
    out["no_adler"] = SIG + ih + chunk(b"IDAT", zraw[:-4]) + chunk(b"IEND", b"")
    out["truncated_zlib"] = SIG + ih + chunk(b"IDAT", zraw[: len(zraw) // 2]) + chunk(b"IEND", b"")
    out["idat_crc"] = SIG + ih + chunk(b"IDAT", zraw, crc=5) + chunk(b"IEND", b"")
    out["no_iend"] = SIG + ih + chunk(b"IDAT", zraw)
    out["iend_crc"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"IEND", b"", crc=1)
    out["late_idat"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", b"zz") + chunk(b"IEND", b"")
    out["late_idat_crc"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"IDAT", b"zz", crc=1) + chunk(b"IEND", b"")
    out["critical_after"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"ABCD", b"") + chunk(b"IEND", b"")
    out["anc_crc_after"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"tEXt", b"k\0v", crc=3) + chunk(b"IEND", b"")
    out["ihdr_after"] = SIG + ih + chunk(b"IDAT", zraw) + ih + chunk(b"IEND", b"")
    out["plte_after"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"PLTE", b"\1\2\3") + chunk(b"IEND", b"")
    out["iend_with_data"] = SIG + ih + chunk(b"IDAT", zraw) + chunk(b"IEND", b"xx")
    out["zero_idat_first"] = SIG + ih + chunk(b"IDAT", b"") + chunk(b"IDAT", zraw) + chunk(b"IEND", b"")
    out["small_window"] = SIG + ih + chunk(b"IDAT", bytes([0x08, 0x1D]) + zraw[2:]) + chunk(b"IEND", b"")
    return out


def fuzz(seed, n):
    rnd = random.Random(seed)
    base = [v for v in generated().values() if len(v) < 6000]
    out = {}
    for it in range(n):
        data = bytearray(rnd.choice(base))
        mode = rnd.randrange(4)
        if mode == 0:
            for _ in range(rnd.randrange(1, 3)):
                data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 1:
            data[rnd.randrange(8, len(data))] ^= 1 << rnd.randrange(8)
        elif mode == 2:
            data = data[: rnd.randrange(len(data) + 1)]
        else:  # flip a byte and repair the chunk CRC so that the damage reaches the decoder
            i = 8
            spans = []
            while i + 12 <= len(data):
                L = struct.unpack(">I", data[i : i + 4])[0]
                if i + 12 + L > len(data): break
                spans.append((i, L)); i += 12 + L
            if spans:
                i, L = rnd.choice(spans)
                if L:
                    data[i + 8 + rnd.randrange(L)] ^= 1 << rnd.randrange(8)
                    data[i + 8 + L : i + 12 + L] = struct.pack(">I", zlib.crc32(bytes(data[i + 4 : i + 8 + L])) & 0xFFFFFFFF)
        out["fz%d_%d" % (seed, it)] = bytes(data)
    return out
