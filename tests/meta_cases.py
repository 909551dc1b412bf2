"""Deterministic inputs for the colour-metadata readers (opencv_decoder_get_jpeg_icc / get_png_icc / get_png_cicp):
hand-made container edge cases plus seeded mutations. Used by tests/test_meta.py and tests/golden/make_meta_golden.py."""
import os
import random
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def app2(seq, num, payload):
    body = b"ICC_PROFILE\0" + bytes([seq, num]) + payload
    return b"\xff\xe2" + struct.pack(">H", len(body) + 2) + body


def chunk(t, d, crc=None):
    c = zlib.crc32(t + d) if crc is None else crc
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", c & 0xFFFFFFFF)


def icc_profile(space=b"RGB ", cls=b"mntr", pcs=b"XYZ ", ver=0x02100000, intent=0, ntags=1, extra=b"", sig=b"acsp", tag_off=None, tag_sz=12, length=None):
    data_off = 132 + 12 * ntags
    tags = b"".join(b"desc" + struct.pack(">II", data_off if tag_off is None else tag_off, tag_sz) for _ in range(ntags))
    payload = b"\x11" * 12 + extra
    total = data_off + len(payload)
    h = (struct.pack(">I", total if length is None else length) + b"\0" * 4 + struct.pack(">I", ver) + cls + space + pcs + b"\0" * 12 + sig + b"\0" * 24 +
         struct.pack(">I", intent) + struct.pack(">III", 0xF6D6, 0x10000, 0xD32D) + b"\0" * 48)
    assert len(h) == 128
    return h + struct.pack(">I", ntags) + tags + payload


def png(chunks, ct=2, depth=8, w=4, h=4):
    ihdr = struct.pack(">IIBBBBB", w, h, depth, ct, 0, 0, 0)
    raw = b"".join(b"\0" + b"\x55" * (max(w, 1) * (3 if ct == 2 else 1)) for _ in range(max(h, 1)))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + b"".join(chunks) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def iccp(profile, name=b"icc", method=0, comp=None):
    return chunk(b"iCCP", name + b"\0" + bytes([method]) + (zlib.compress(profile) if comp is None else comp))


def hand_cases():
    rnd = random.Random(1)
    ex = bytes(rnd.randrange(256) for _ in range(200))
    good = icc_profile(extra=ex)
    C1 = chunk(b"cICP", bytes([9, 16, 0, 1]))
    C2 = chunk(b"cICP", bytes([1, 13, 0, 0]))
    IH = chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0))
    SIG = b"\x89PNG\r\n\x1a\n"
    P = lambda *c, **k: png(list(c), **k)
    cases = {
        "plain": P(iccp(good)), "gray_rgbprof": P(iccp(good), ct=0), "gray_grayprof": P(iccp(icc_profile(space=b"GRAY", extra=ex)), ct=0),
        "rgb_grayprof": P(iccp(icc_profile(space=b"GRAY", extra=ex))), "cmyk": P(iccp(icc_profile(space=b"CMYK", extra=ex))),
        "lab_pcs": P(iccp(icc_profile(pcs=b"Lab ", extra=ex))), "bad_pcs": P(iccp(icc_profile(pcs=b"Luv ", extra=ex))),
        "bad_sig": P(iccp(icc_profile(sig=b"acsq", extra=ex))), "ver4": P(iccp(icc_profile(ver=0x04200000, extra=ex))),
        "tags0": P(iccp(icc_profile(ntags=0, extra=ex))), "tags3": P(iccp(icc_profile(ntags=3, extra=ex))),
        "tag_outside": P(iccp(icc_profile(tag_off=100000, extra=ex))), "tag_sz_outside": P(iccp(icc_profile(tag_sz=100000, extra=ex))),
        "tag_unaligned": P(iccp(icc_profile(tag_off=145, extra=ex))), "tag_in_header": P(iccp(icc_profile(tag_off=4, extra=ex))),
        "len_odd": P(iccp(icc_profile(extra=ex + b"x"))), "len_odd_v4": P(iccp(icc_profile(extra=ex + b"x", ver=0x04200000))),
        "len_more": P(iccp(icc_profile(extra=ex, length=1000))), "len_less": P(iccp(icc_profile(extra=ex, length=200))),
        "len_131": P(iccp(icc_profile(extra=ex, length=131))), "method1": P(iccp(good, method=1)), "name_empty": P(iccp(good, name=b"")),
        "name_79": P(iccp(good, name=b"a" * 79)), "name_80": P(iccp(good, name=b"a" * 80)), "name_space": P(iccp(good, name=b" lead")),
        "name_hi": P(iccp(good, name=b"a\xa0b")), "trunc_z": P(iccp(good, comp=zlib.compress(good)[:-20])),
        "no_adler": P(iccp(good, comp=zlib.compress(good)[:-4])), "extra_z": P(iccp(good, comp=zlib.compress(good) + b"junkjunk")),
        "extra_out": P(iccp(good, comp=zlib.compress(good + b"\0" * 64))), "garbage_z": P(iccp(good, comp=b"\x00" * 120)),
        "short_chunk": P(iccp(icc_profile(), comp=zlib.compress(icc_profile(), 9))), "two_iccp": P(iccp(good), iccp(icc_profile(extra=ex[:100]))),
        "bad_then_good": P(iccp(icc_profile(sig=b"acsq", extra=ex)), iccp(good)), "srgb_then_iccp": P(chunk(b"sRGB", b"\0"), iccp(good)),
        "after_plte": P(chunk(b"PLTE", b"\1\2\3"), iccp(good)), "plte_after": P(iccp(good), chunk(b"PLTE", b"\1\2\3")),
        "crc_bad_iccp": P(chunk(b"iCCP", b"icc\0\0" + zlib.compress(good), crc=1)),
        "crc_bad_iccp_then_good": P(chunk(b"iCCP", b"icc\0\0" + zlib.compress(good), crc=1), iccp(icc_profile(extra=ex[:100]))),
        "crc_bad_other": P(chunk(b"gAMA", struct.pack(">I", 45455), crc=5), iccp(good)), "unknown_anc": P(chunk(b"abCd", b"xyz"), iccp(good)),
        "unknown_crit": P(chunk(b"ABCD", b"xyz"), iccp(good)), "bad_name_chunk": P(chunk(b"ab1d", b"xyz"), iccp(good)),
        "reserved_bit": P(chunk(b"cIcP", bytes([9, 16, 0, 1])), C1), "cicp_ok": P(C1, iccp(good)), "cicp_matrix": P(chunk(b"cICP", bytes([9, 16, 1, 1]))),
        "cicp_len5": P(chunk(b"cICP", bytes([9, 16, 0, 1, 0]))), "cicp_len5_then_good": P(chunk(b"cICP", bytes([9, 16, 0, 1, 0])), C2),
        "cicp_dup": P(C1, C2), "cicp_matrix_then_good": P(chunk(b"cICP", bytes([9, 16, 3, 1])), C2), "cicp_crc_then_good": P(chunk(b"cICP", bytes([9, 16, 0, 1]), crc=3), C2),
        "cicp_after_plte": P(chunk(b"PLTE", b"\1\2\3"), C1), "cicp_gray": P(C1, ct=0), "cicp_big": P(chunk(b"cICP", bytes([255, 255, 0, 255]))),
        "plte_dup": P(C1, chunk(b"PLTE", b"\1\2\3"), chunk(b"PLTE", b"\1\2\3")), "plte_bad_len_then_cicp": P(chunk(b"PLTE", b"\1\2\3\4"), C1),
        "plte_badcrc_then_cicp": P(chunk(b"PLTE", b"\1\2\3", crc=9), C1), "plte_empty": P(C1, chunk(b"PLTE", b"")),
        "plte_after_trns": P(chunk(b"tRNS", b"\0\1\0\2\0\3"), chunk(b"PLTE", b"\1\2\3"), C1), "plte_after_bad_bkgd": P(chunk(b"bKGD", b"\0"), chunk(b"PLTE", b"\1\2\3"), C1),
        "plte_gray": P(chunk(b"PLTE", b"\1\2\3"), C1, ct=0), "pal_noplte": P(C1, ct=3), "pal_ok": P(C1, chunk(b"PLTE", b"\1\2\3"), ct=3),
        "pal_badcrc": P(C1, chunk(b"PLTE", b"\1\2\3", crc=9), ct=3), "pal_empty": P(C1, chunk(b"PLTE", b""), ct=3), "pal_toomany": P(C1, chunk(b"PLTE", b"\1\2\3" * 257), ct=3),
        "iend_first": SIG + IH + C1 + chunk(b"IEND", b""), "no_ihdr": SIG + C1 + IH + chunk(b"IDAT", b"x"), "ihdr_dup": P(IH, C1),
        "ihdr_zero_w": P(C1, w=0), "ihdr_depth3": P(C1, depth=3), "ihdr_ct5": P(C1, ct=5), "ihdr_rgb_d4": P(C1, depth=4), "ihdr_big_w": P(C1, w=1000001),
        "ihdr_w_1m": P(C1, w=1000000), "big_anc": P(chunk(b"abCd", b"x" * 8000001), C1), "len_hi": SIG + IH + struct.pack(">I", 0x80000000) + b"abCd" + b"\0" * 40,
        "trunc_sig": SIG[:6], "trunc_hdr": P(C1)[:40], "trunc_before_idat": P(C1)[: 8 + 25 + 16], "trunc_in_idat_hdr": P(C1)[: 8 + 25 + 16 + 6],
        "trunc_at_idat_hdr": P(C1)[: 8 + 25 + 16 + 8], "idat_empty_first": SIG + IH + C1 + chunk(b"IDAT", b""),
        "many_anc": P(chunk(b"tRNS", b"\0\1\0\2\0\3"), chunk(b"bKGD", b"\0\1\0\2\0\3"), chunk(b"pHYs", b"\0" * 9), chunk(b"tEXt", b"k\0v"), chunk(b"cHRM", b"\0" * 32),
                      chunk(b"sBIT", b"\x09\x09\x09"), chunk(b"eXIf", b"MM\0*"), chunk(b"cLLI", b"\0" * 8), chunk(b"mDCV", b"\0" * 24), C1, iccp(good)),
    }
    for cm, fm, il in ((1, 0, 0), (0, 1, 0), (0, 0, 2), (0, 0, 1)):
        cases["ihdr_%d%d%d" % (cm, fm, il)] = SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, cm, fm, il)) + C1 + iccp(good) + chunk(b"IDAT", b"x")
    for cls in (b"abst", b"link", b"nmcl", b"scnr", b"prtr", b"spac", b"junk"):
        cases["cls_" + cls.decode()] = P(iccp(icc_profile(cls=cls, extra=ex)))
    for it in (3, 4, 0xFFFF, 0x10000):
        cases["intent_%x" % it] = P(iccp(icc_profile(intent=it, extra=ex)))
    out = []
    for k, v in cases.items():
        out.append(("png_icc", k, v))
        out.append(("png_cicp", k, v))
    return out


def fuzz_cases(seed, n):
    """Seeded chunk shuffles / byte flips / truncations of PNGs, and ICC chunk-set mutations of a JPEG fixture."""
    rnd = random.Random(seed)
    ex = bytes(rnd.randrange(256) for _ in range(200))
    good = icc_profile(extra=ex)
    pool = [chunk(b"cICP", bytes([9, 16, 0, 1])), chunk(b"cICP", bytes([1, 13, 0, 0])), chunk(b"cICP", bytes([9, 16, 5, 1])), iccp(good),
            iccp(icc_profile(space=b"GRAY", extra=ex)), iccp(icc_profile(extra=ex[:80], ver=0x04400000)), chunk(b"PLTE", b"\1\2\3" * 4), chunk(b"PLTE", b""),
            chunk(b"PLTE", b"\1\2\3", crc=1), chunk(b"PLTE", b"\1\2"), chunk(b"sRGB", b"\0"), chunk(b"gAMA", struct.pack(">I", 45455)), chunk(b"tRNS", b"\0\1\0\2\0\3"),
            chunk(b"abCd", b"hello"), chunk(b"iCCP", b"x\0\0" + zlib.compress(good), crc=2), chunk(b"bKGD", b"\0"), chunk(b"IDAT", b""), chunk(b"IEND", b"")]
    out = []
    for it in range(n):
        chunks = [rnd.choice(pool) for _ in range(rnd.randrange(0, 6))]
        ct = rnd.choice([0, 2, 2, 2, 3, 4, 6])
        depth = 8 if ct != 3 else rnd.choice([1, 2, 4, 8])
        data = bytearray(png(chunks, ct=ct, depth=depth))
        mode = rnd.randrange(4)
        if mode == 1:
            for _ in range(rnd.randrange(1, 4)):
                data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 2:
            data = data[: rnd.randrange(len(data) + 1)]
        elif mode == 3:
            data[rnd.randrange(8, len(data))] ^= 1 << rnd.randrange(8)
        out.append(("png_icc", "fz%d_%d" % (seed, it), bytes(data)))
        out.append(("png_cicp", "fz%d_%d" % (seed, it), bytes(data)))
    base = open(os.path.join(HERE, "golden", "inputs", "field.jpg"), "rb").read()
    for it in range(n):
        num = rnd.randrange(1, 5)
        parts = [bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 40))) for _ in range(num)]
        segs = [app2(i + 1, num, parts[i]) for i in range(num)]
        m = rnd.randrange(8)
        if m == 1:
            rnd.shuffle(segs)
        elif m == 2:
            segs.append(rnd.choice(segs))
        elif m == 3:
            segs.pop(rnd.randrange(len(segs)))
        elif m == 4:
            segs[rnd.randrange(len(segs))] = app2(rnd.randrange(0, 6), rnd.randrange(0, 6), b"zz")
        elif m == 5:
            segs.insert(rnd.randrange(len(segs) + 1), b"\xff\xe2\x00\x08ICC_PR")
        pos = 2 if rnd.random() < 0.5 else 20
        data = bytearray(base[:pos] + b"".join(segs) + base[pos:])
        hdr_end = data.find(b"\xff\xda") + 14
        mode = rnd.randrange(4)
        if mode == 1:
            for _ in range(rnd.randrange(1, 4)):
                data[rnd.randrange(hdr_end)] = rnd.randrange(256)
        elif mode == 2:
            data = data[: rnd.randrange(hdr_end + 4)]
        elif mode == 3:
            data[rnd.randrange(hdr_end)] ^= 1 << rnd.randrange(8)
        out.append(("jpeg_icc", "fz%d_%d" % (seed, it), bytes(data)))
    return out


def all_cases():
    fix = os.path.join(HERE, "golden", "inputs")
    out = [("jpeg_icc", n, open(os.path.join(fix, n), "rb").read()) for n in sorted(os.listdir(fix))]
    return out + hand_cases() + fuzz_cases(101, 600) + fuzz_cases(202, 600)
