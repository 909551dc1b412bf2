"""PBM / PGM / PPM files for tests/test_pxm.py and tests/golden/make_pxm_digests.py: every kind ("P1".."P6"), sample ranges on both
sides of 255, the header forms cv::PxMDecoder's number reader accepts (comments, runs of white space, a single separator before binary
samples), and seeded damage (truncations, flipped bytes, spliced text) -- written from the format's description, no reference file."""
import random


def make_pxm(kind, w, h, maxval=255, seed=1, sep=b"\n", comment=None, tail_sep=b"\n", ascii_width=0):
    """One file of `kind` (1..6). Binary kinds: packed rows (P4, MSB first, rows padded to bytes), one or two bytes per sample (big endian
    above 255). ASCII kinds: decimal samples separated by white space."""
    rnd = random.Random(seed)
    cn = 3 if kind in (3, 6) else 1
    head = b"P%d" % kind + sep
    if comment is not None:
        head += b"#" + comment + b"\n"
    head += b"%d" % w + sep + b"%d" % h
    if kind not in (1, 4):
        head += sep + b"%d" % maxval
    head += tail_sep
    if kind == 4:
        body = bytes(rnd.randrange(256) for _ in range(((w + 7) // 8) * h))
    elif kind in (5, 6):
        n = w * h * cn
        if maxval > 255:
            body = b"".join(rnd.randrange(maxval + 1).to_bytes(2, "big") for _ in range(n))
        else:
            body = bytes(rnd.randrange(maxval + 1) for _ in range(n))
    elif kind == 1:
        rows = []
        for _ in range(h):
            rows.append(b" ".join(b"%d" % rnd.randrange(2) for _ in range(w)))
        body = b"\n".join(rows) + b"\n"
    else:
        rows = []
        for _ in range(h):
            rows.append(b" ".join((b"%*d" % (ascii_width, rnd.randrange(maxval + 1))) for _ in range(w * cn)))
        body = b"\n".join(rows) + b"\n"
    return head + body


def generated():
    """(name, file) of every regular variant."""
    out = []
    seed = 100
    for kind in range(1, 7):
        for (w, h) in ((1, 1), (7, 3), (8, 8), (9, 5), (33, 17), (64, 48), (131, 67)):
            maxvals = (1,) if kind in (1, 4) else (255, 1, 15, 100, 254, 256, 1023, 4095, 65535)
            for mv in maxvals:
                seed += 1
                out.append(("k%d_%dx%d_m%d" % (kind, w, h, mv), make_pxm(kind, w, h, mv, seed)))
    for kind in range(1, 7):
        seed += 1
        out.append(("k%d_comment" % kind, make_pxm(kind, 21, 9, 255, seed, comment=b" made by a test")))
        out.append(("k%d_crlf" % kind, make_pxm(kind, 21, 9, 255, seed, sep=b"\r\n", tail_sep=b"\r\n")))
        out.append(("k%d_spaces" % kind, make_pxm(kind, 21, 9, 255, seed, sep=b"  \t ", tail_sep=b" ")))
        out.append(("k%d_tabs" % kind, make_pxm(kind, 21, 9, 255, seed, sep=b"\t", tail_sep=b"\t")))
        out.append(("k%d_zero_w" % kind, make_pxm(kind, 0, 9, 255, seed)))
        out.append(("k%d_zero_h" % kind, make_pxm(kind, 9, 0, 255, seed)))
        out.append(("k%d_maxval0" % kind, make_pxm(kind, 9, 4, 0, seed)))
        out.append(("k%d_maxval65536" % kind, make_pxm(kind, 9, 4, 65536, seed)))
        out.append(("k%d_wide" % kind, make_pxm(kind, 1500, 2, 255, seed)))
        out.append(("k%d_padded_numbers" % kind, make_pxm(kind, 12, 6, 255, seed, ascii_width=4)))
        out.append(("k%d_trailing" % kind, make_pxm(kind, 12, 6, 255, seed) + b"trailing bytes \x00\xff"))
    # samples beyond the announced range, negative signs, letters inside the samples, a comment between samples
    out.append(("p2_over_range", b"P2\n3 2\n15\n0 15 16 255 1000 7\n"))
    out.append(("p2_huge_number", b"P2\n3 2\n255\n0 15 99999999999999999999 255 1000 7\n"))
    out.append(("p2_negative", b"P2\n3 2\n255\n0 15 -4 255 100 7\n"))
    out.append(("p2_letter", b"P2\n3 2\n255\n0 15 x 255 100 7\n"))
    out.append(("p2_comment_inside", b"P2\n3 2\n255\n0 15 # a remark\n 4 255 100 7\n"))
    out.append(("p2_no_final_newline", b"P2\n3 2\n255\n0 15 4 255 100 7"))
    out.append(("p2_short", b"P2\n3 2\n255\n0 15 4 255 100\n"))
    out.append(("p3_short", b"P3\n2 2\n255\n1 2 3 4 5 6 7 8 9 10 11\n"))
    out.append(("p1_dense", b"P1\n4 2\n1 0 1 1\n0 0 1 0\n"))
    out.append(("p1_packed_digits", b"P1\n4 2\n1011\n0010\n"))
    out.append(("p1_twos", b"P1\n4 2\n1 2 0 3\n0 0 1 0\n"))
    out.append(("p5_header_only", b"P5\n4 4\n255\n"))
    out.append(("p5_no_separator", b"P5\n2 2\n255"))
    out.append(("p6_one_byte_short", make_pxm(6, 5, 5, 255, 3)[:-1]))
    out.append(("p4_one_byte_short", make_pxm(4, 13, 5, 1, 3)[:-1]))
    out.append(("p5_16_one_byte_short", make_pxm(5, 5, 5, 1000, 3)[:-1]))
    out.append(("p7", b"P7\n2 2\n255\n\x00\x01\x02\x03"))
    out.append(("p0", b"P0\n2 2\n255\n\x00\x01\x02\x03"))
    out.append(("no_space", b"P5x2 2\n255\n\x00\x01\x02\x03"))
    out.append(("two_bytes", b"P5"))
    out.append(("header_eof_in_comment", b"P5\n# never ends"))
    out.append(("header_missing_height", b"P5\n12\n"))
    out.append(("header_signs", b"P5\n+2 2\n255\n\x00\x01\x02\x03"))
    out.append(("header_neg", b"P5\n-2 2\n255\n\x00\x01\x02\x03"))
    out.append(("header_big", b"P5\n99999999 99999999\n255\n\x00"))
    out.append(("header_overflow", b"P5\n4294967298 2\n255\n\x00\x01\x02\x03"))
    out.append(("header_vt_ff", b"P5\x0b2\x0c2\n255\n\x00\x01\x02\x03"))
    return out


def fuzz(seed, n):
    """n damaged variants of the regular files: truncations, byte flips, spliced ASCII, digits rewritten in the header."""
    rnd = random.Random(seed)
    base = [v for k, v in generated() if len(v) < 6000 and not k.startswith(("header", "p7", "p0", "two", "no_space"))]
    out = []
    for i in range(n):
        d = bytearray(rnd.choice(base))
        kind = rnd.randrange(6)
        if kind == 0 and len(d) > 4:
            d = d[: rnd.randrange(3, len(d))]
        elif kind == 1:
            for _ in range(rnd.randrange(1, 6)):
                d[rnd.randrange(len(d))] = rnd.randrange(256)
        elif kind == 2:
            at = rnd.randrange(2, min(len(d), 24))
            d[at:at] = rnd.choice((b"#x\n", b" ", b"\n\n", b"9", b"0", b"-", b"\x00", b"#", b"\r", b"12345678901"))
        elif kind == 3:
            for _ in range(rnd.randrange(1, 4)):
                at = rnd.randrange(2, min(len(d), 20))
                d[at] = rnd.choice(b"0123456789 \n#\tx")
        elif kind == 4:
            at = rnd.randrange(len(d))
            d[at:at] = bytes(rnd.choice(b"0123456789 \n#-+ax\xff") for _ in range(rnd.randrange(1, 9)))
        else:
            at = rnd.randrange(len(d))
            del d[at: at + rnd.randrange(1, 12)]
        out.append(("fuzz%04d" % i, bytes(d)))
    return out
