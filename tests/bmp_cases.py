"""BMP test inputs: every variant cv::BmpDecoder reads (1 / 4 / 8 bits with colour and grey palettes, 16-bit 5-5-5 and 5-6-5, 24, 32 with
and without bit fields, RLE8 / RLE4 with escapes, bottom-up and top-down, OS/2 and V4 / V5 headers) and damaged files. Deterministic."""
import random
import struct


def _palette(n, grey, rnd):
    if grey:
        return [(i * 255 // max(1, n - 1),) * 3 for i in range(n)]
    return [(rnd.randrange(256), rnd.randrange(256), rnd.randrange(256)) for _ in range(n)]


def _rle(rows, bpp, rnd, deltas=False):
    """rows: lists of palette indices, in FILE order (first row = first in the file). BI_RLE8 / BI_RLE4 with encoded runs, absolute runs,
    end-of-line, optional delta escapes over runs of index 0 ... and end-of-bitmap."""
    out = bytearray()
    for row in rows:
        x = 0
        while x < len(row):
            run = 1
            while x + run < len(row) and run < 255 and row[x + run] == row[x]:
                run += 1
            if bpp == 4 and run < 3 and x + run < len(row):  # alternate two indices as one encoded run
                a, b = row[x], row[x + 1]
                alt = 2
                while x + alt < len(row) and alt < 254 and row[x + alt] == (a if alt % 2 == 0 else b):
                    alt += 1
                if alt >= 4:
                    out += bytes([alt, (a << 4) | b])
                    x += alt
                    continue
            if run >= 3 or x + run >= len(row):
                out += bytes([run, row[x] if bpp == 8 else (row[x] << 4) | row[x]])
                x += run
                continue
            n = min(len(row) - x, rnd.randrange(3, 40))
            n = max(3, n) if len(row) - x >= 3 else 0
            if n == 0:  # one or two pixels left: encoded runs of one
                out += bytes([1, row[x] if bpp == 8 else (row[x] << 4)])
                x += 1
                continue
            out += bytes([0, n])
            if bpp == 8:
                out += bytes(row[x:x + n])
                if n & 1:
                    out.append(0)
            else:
                packed = bytearray()
                for i in range(0, n, 2):
                    packed.append((row[x + i] << 4) | (row[x + i + 1] if i + 1 < n else 0))
                out += packed
                if len(packed) & 1:
                    out.append(0)
            x += n
        out += b"\x00\x00"
    out += b"\x00\x01"
    return bytes(out)


def make_bmp(w, h, bpp, seed=0, compression=0, topdown=False, header=40, grey=False, ncolors=None, masks=None, smooth=True, tail=b""):
    rnd = random.Random(seed * 7919 + w * 31 + h * 17 + bpp)
    pal = b""
    ncol = 0
    if bpp <= 8:
        ncol = ncolors if ncolors is not None else 1 << bpp
        entries = _palette(ncol, grey, rnd)
        pal = b"".join(bytes(e[:3]) + (b"" if header == 12 else b"\x00") for e in entries)
    # pixel rows, top to bottom
    rows = []
    for y in range(h):
        if bpp <= 8:
            lim = max(1, min(ncol, 1 << bpp))
            if smooth:
                row = [((x // 3 + y // 2 + rnd.randrange(2)) % lim) for x in range(w)]
            else:
                row = [rnd.randrange(lim) for _ in range(w)]
        elif bpp == 16:
            row = [rnd.randrange(1 << 16) for _ in range(w)]
        else:
            row = [tuple(rnd.randrange(256) for _ in range(bpp // 8)) for _ in range(w)]
        rows.append(row)
    file_rows = rows if topdown else rows[::-1]
    if compression in (1, 2):
        data = _rle(file_rows, bpp, rnd)
    else:
        data = bytearray()
        for row in file_rows:
            if bpp == 1:
                b = bytearray((w + 7) // 8)
                for x, v in enumerate(row):
                    b[x >> 3] |= (v & 1) << (7 - (x & 7))
            elif bpp == 4:
                b = bytearray((w + 1) // 2)
                for x, v in enumerate(row):
                    b[x >> 1] |= (v & 15) << (4 if x % 2 == 0 else 0)
            elif bpp == 8:
                b = bytearray(row)
            elif bpp == 16:
                b = bytearray(struct.pack("<%dH" % w, *row))
            else:
                b = bytearray(bytes(c for px in row for c in px))
            b += bytes(-len(b) % 4)
            data += b
        data = bytes(data)
    mask_bytes = b""
    if compression == 3 and header == 40:
        mask_bytes = struct.pack("<III", *(masks or ((0xF800, 0x07E0, 0x001F) if bpp == 16 else (0x00FF0000, 0x0000FF00, 0x000000FF))))
    if header == 12:
        dib = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        dib = struct.pack("<IiiHHIIiiII", header, w, -h if topdown else h, 1, bpp, compression, len(data), 2835, 2835, ncol if ncolors is not None else 0, 0)
        if header > 40:
            m = masks or ((0xF800, 0x07E0, 0x001F, 0) if bpp == 16 else (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000))
            extra = struct.pack("<IIII", *(tuple(m) + (0,) * (4 - len(m))))
            dib += extra[: header - 40] if header <= 56 else extra + bytes(header - 56)
    off = 14 + len(dib) + len(mask_bytes) + len(pal)
    return struct.pack("<2sIHHI", b"BM", off + len(data), 0, 0, off) + dib + mask_bytes + pal + data + tail


def generated():
    out = {}
    for bpp in (1, 4, 8):
        for grey in (False, True):
            for w, h in ((1, 1), (7, 5), (32, 9), (33, 16)):
                out["p%d_%s_%dx%d" % (bpp, "grey" if grey else "col", w, h)] = make_bmp(w, h, bpp, grey=grey)
        out["p%d_topdown" % bpp] = make_bmp(13, 6, bpp, topdown=True)
        out["p%d_os2" % bpp] = make_bmp(13, 6, bpp, header=12)
        out["p%d_fewcolors" % bpp] = make_bmp(13, 6, bpp, ncolors=2)
        out["p%d_noise" % bpp] = make_bmp(21, 11, bpp, smooth=False)
    for w, h in ((1, 1), (7, 5), (64, 3), (31, 17)):
        out["rgb24_%dx%d" % (w, h)] = make_bmp(w, h, 24)
        out["rgb32_%dx%d" % (w, h)] = make_bmp(w, h, 32)
        out["rgb16_%dx%d" % (w, h)] = make_bmp(w, h, 16)
    out["rgb24_topdown"] = make_bmp(9, 4, 24, topdown=True)
    out["rgb32_topdown"] = make_bmp(9, 4, 32, topdown=True)
    out["rgb16_565"] = make_bmp(9, 4, 16, compression=3)
    out["rgb16_555_fields"] = make_bmp(9, 4, 16, compression=3, masks=(0x7C00, 0x03E0, 0x001F))
    out["rgb32_fields"] = make_bmp(9, 4, 32, compression=3)
    out["rgb32_v4_fields"] = make_bmp(9, 4, 32, compression=3, header=108)
    out["rgb32_v5_fields"] = make_bmp(9, 4, 32, compression=3, header=124)
    out["rgb32_v4_odd_masks"] = make_bmp(9, 4, 32, compression=3, header=108, masks=(0x000000FF, 0x0000FF00, 0x00FF0000, 0xFF000000))
    out["rgb24_v5"] = make_bmp(9, 4, 24, header=124)
    out["rgb32_v3_52"] = make_bmp(9, 4, 32, compression=3, header=52)
    out["rgb32_v3_56"] = make_bmp(9, 4, 32, compression=3, header=56)
    for w, h in ((1, 1), (16, 4), (37, 9), (300, 5)):
        out["rle8_%dx%d" % (w, h)] = make_bmp(w, h, 8, compression=1)
        out["rle4_%dx%d" % (w, h)] = make_bmp(w, h, 4, compression=2)
    out["rle8_grey"] = make_bmp(37, 9, 8, compression=1, grey=True)
    out["rle4_grey"] = make_bmp(37, 9, 4, compression=2, grey=True)
    out["rle8_noise"] = make_bmp(40, 8, 8, compression=1, smooth=False)
    out["rle4_noise"] = make_bmp(40, 8, 4, compression=2, smooth=False)
    out["rle8_topdown"] = make_bmp(16, 4, 8, compression=1, topdown=True)
    out["rgb24_tail"] = make_bmp(9, 4, 24, tail=b"trailing bytes")
    return out


def fuzz(seed, n):
    """Damaged variants of the generated files: byte flips in the headers, truncations, bytes deleted."""
    rnd = random.Random(seed)
    base = generated()
    names = sorted(base)
    out = {}
    for i in range(n):
        name = names[rnd.randrange(len(names))]
        b = bytearray(base[name])
        kind = rnd.randrange(5)
        if kind == 0:
            b[rnd.randrange(min(len(b), 70))] = rnd.randrange(256)
        elif kind == 1:
            pos = rnd.randrange(min(len(b), 70))
            b[pos] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            del b[rnd.randrange(len(b)):]
        elif kind == 3 and len(b) > 80:
            pos = rnd.randrange(60, len(b))
            b[pos] = rnd.randrange(256)
        else:
            pos = rnd.randrange(len(b))
            del b[pos:pos + rnd.randrange(1, 5)]
        out["%s#%d" % (name, i)] = bytes(b)
    return out
