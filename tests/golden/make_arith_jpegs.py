"""Writes tests/golden/inputs_arith/*.jpg -- arithmetic-coded JPEGs (SOF9 sequential, SOF10 progressive) from the reference's own
libjpeg-turbo compressor (oracle/_ref/libref.so, ref_jpeg_encode_ex with arith_code) -- and tests/golden/arith_golden.json: per file
"<h>x<w>x<c>:<sha1 of the pixels the reference's libjpeg decodes>" and, per component, a sha1 of the quantised coefficients it
decodes (jpeg_read_coefficients). The product's QM decoder (lilliput_amd/csrc/lp_arith_host.cpp) is held to these answers.
Run in the build container (needs /root/reference)."""
import ctypes as C, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

R = O.ref()
assert R is not None
R.ref_jpeg_encode_ex.restype = C.c_long
out_dir = os.path.join(ROOT, "tests", "golden", "inputs_arith")
rng = np.random.default_rng(77)
ARITH, COND, OWN_TABLES = 128, 256, 512


def photo(h, w, c, noise=9):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 95 * np.sin(x / 13.0 + k) + 35 * np.cos(y / 7.0 - k) for k in range(4)], -1) + rng.normal(0, noise, (h, w, 4))
    img = np.clip(img, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img[:, :, 0] if c == 1 else img[:, :, :c])


def enc(name, h, w, ncomp, mode, samp, q, dri=0, optimize=0, noise=9, force_baseline=1):
    px = photo(h, w, ncomp, noise)
    buf = np.zeros(h * w * 8 + 65536, np.uint8)
    n = R.ref_jpeg_encode_ex(px.ctypes.data_as(C.c_void_p), w, h, ncomp, mode, (C.c_int * 6)(*samp), q, force_baseline, dri, optimize | ARITH, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    assert n > 0, name
    open(os.path.join(out_dir, name + ".jpg"), "wb").write(buf[:n].tobytes())


S444, S422, S420, S440 = (1, 1, 1, 1, 1, 1), (2, 1, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1), (1, 2, 1, 1, 1, 1)
enc("arith_seq_420", 75, 101, 3, 0, S420, 85)
enc("arith_seq_444_dri3", 41, 67, 3, 0, S444, 92, dri=3)
enc("arith_seq_gray", 66, 50, 1, 0, S444, 70)
enc("arith_seq_422_cond", 58, 77, 3, 0, S422, 80, optimize=COND)
enc("arith_seq_440_own_tables_cond_dri1", 35, 90, 3, 0, S440, 60, dri=1, optimize=COND | OWN_TABLES)
enc("arith_seq_noisy_q100", 64, 64, 3, 0, S420, 100, noise=60)
enc("arith_seq_q1_16bit_tables", 48, 48, 3, 0, S420, 1, force_baseline=0)
enc("arith_seq_noninterleaved_420", 70, 85, 3, 0, S420, 85, optimize=16)
enc("arith_seq_tiny", 1, 1, 3, 0, S420, 90)
enc("arith_seq_narrow", 40, 3, 3, 0, S420, 90)
enc("arith_seq_cmyk", 52, 61, 4, 0, S444, 90)
enc("arith_prog_simple_420", 75, 101, 3, 0, S420, 85, optimize=2)
enc("arith_prog_simple_444_dri3_cond", 41, 67, 3, 0, S444, 92, dri=3, optimize=2 | COND)
enc("arith_prog_simple_gray", 66, 50, 1, 0, S444, 70, optimize=2)
enc("arith_prog_spectral_422", 58, 77, 3, 0, S422, 80, optimize=4)
enc("arith_prog_spectral_440_dri1", 35, 90, 3, 0, S440, 60, dri=1, optimize=4)
enc("arith_prog_deep_420", 90, 64, 3, 0, S420, 95, optimize=8)
enc("arith_prog_deep_gray_dri17_noisy", 120, 33, 1, 0, S444, 99, dri=17, optimize=8, noise=50)
enc("arith_prog_ycck", 58, 41, 4, 1, S440, 75, optimize=2)
enc("arith_seq_big_420", 512, 768, 3, 0, S420, 90, noise=25)
gold = {}
for f in sorted(os.listdir(out_dir)):
    d = open(os.path.join(out_dir, f), "rb").read()
    px = O.ref_jpeg_decode(d)
    nc = 1 if px.shape[2] == 1 else (4 if "cmyk" in f or "ycck" in f else 3)
    gold[f] = {"pixels": "%dx%dx%d:%s" % (px.shape[0], px.shape[1], px.shape[2], hashlib.sha1(px.tobytes()).hexdigest()[:16]),
               "coefs": [hashlib.sha1(np.ascontiguousarray(O.ref_jpeg_decode_coefs(d, c)).tobytes()).hexdigest()[:16] for c in range(nc)]}
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "arith_golden.json"), "w"), indent=0, sort_keys=True)
print(len(gold), "files,", sum(os.path.getsize(os.path.join(out_dir, f)) for f in gold), "bytes")
