"""Writes tests/golden/inputs_exotic/*.jpg with the reference's own libjpeg-turbo compressor (oracle/_ref/libref.so): samplings,
colour spaces and table widths Pillow cannot produce (4:4:0, 2x2 luma with custom chroma factors, RGB-in-JPEG with an Adobe
marker, YCbCr without JFIF, 16-bit quantisation tables / SOF1, restart intervals that are not a multiple of a row, progressive
files with libjpeg's default script, a spectral-selection-only script and a deep successive-approximation script, sequential
files in several scans and with three Huffman table pairs, four-component CMYK / YCCK files, sampling factors 3 and 4 and chroma sampled finer than luma), and
tests/golden/exotic_golden.json = "<h>x<w>x<c>:<sha1 of the pixels the reference's libjpeg decodes>" per file (four-component files: libjpeg's
CMYK rows through OpenCV's own icvCvt_CMYK2BGR_8u_C4C3R out of the reference's libopencv_imgcodecs.a, oracle/ref_cv_driver.cpp).
Run in the build container (needs /root/reference)."""
import ctypes as C, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

R = O.ref()
assert R is not None
R.ref_jpeg_encode_ex.restype = C.c_long
out_dir = os.path.join(ROOT, "tests", "golden", "inputs_exotic")
rng = np.random.default_rng(42)

def photo(h, w, c):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 95 * np.sin(x / 13.0 + k) + 35 * np.cos(y / 7.0 - k) for k in range(3)], -1) + rng.normal(0, 9, (h, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img[:, :, 0] if c == 1 else img)

def photo4(h, w):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 95 * np.sin(x / 13.0 + k) + 35 * np.cos(y / 7.0 - k) for k in range(4)], -1) + rng.normal(0, 9, (h, w, 4))
    return np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8))


def enc(name, h, w, ncomp, mode, samp, q, force_baseline=1, dri=0, optimize=0):
    px = photo4(h, w) if ncomp == 4 else photo(h, w, ncomp)
    buf = np.zeros(h * w * 4 + 65536, np.uint8)
    n = R.ref_jpeg_encode_ex(px.ctypes.data_as(C.c_void_p), w, h, ncomp, mode, (C.c_int * 6)(*samp), q, force_baseline, dri, optimize, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    assert n > 0, name
    open(os.path.join(out_dir, name + ".jpg"), "wb").write(buf[:n].tobytes())

S444, S422, S420, S440 = (1, 1, 1, 1, 1, 1), (2, 1, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1), (1, 2, 1, 1, 1, 1)
enc("s440", 61, 83, 3, 0, S440, 88)
enc("s440_dri5_opt", 100, 37, 3, 0, S440, 75, dri=5, optimize=1)
enc("s440_narrow", 40, 3, 3, 0, S440, 90)
enc("rgb_adobe_444", 45, 52, 3, 1, S444, 90)
enc("rgb_adobe_dri", 33, 70, 3, 1, S444, 60, dri=3)
enc("ycc_adobe_420", 70, 90, 3, 2, S420, 85)
enc("ycc_nomarker_422", 50, 66, 3, 3, S422, 85)
enc("q1_16bit_tables_420", 64, 64, 3, 0, S420, 1, force_baseline=0)
enc("q2_16bit_tables_gray", 57, 41, 1, 0, (1, 1, 1, 1, 1, 1), 2, force_baseline=0)
enc("q1_baseline_444", 30, 30, 3, 0, S444, 1)
enc("q100_420_opt", 48, 80, 3, 0, S420, 100, optimize=1)
enc("dri7_420", 90, 75, 3, 0, S420, 80, dri=7)
enc("dri1_444", 24, 40, 3, 0, S444, 92, dri=1)
enc("gray_dri2", 40, 56, 1, 0, (1, 1, 1, 1, 1, 1), 70, dri=2)
# progressive (SOF2): optimize bit 1 = jpeg_simple_progression, bit 2 = spectral selection only with one DC scan per component,
# bit 3 = successive approximation from Al = 3 with split AC bands (oracle/ref_driver.c ref_jpeg_encode_ex)
enc("prog_simple_420", 75, 101, 3, 0, S420, 85, optimize=2)
enc("prog_simple_444_opt_dri3", 41, 67, 3, 0, S444, 92, dri=3, optimize=3)
enc("prog_simple_gray", 66, 50, 1, 0, S444, 70, optimize=2)
enc("prog_spectral_422", 58, 77, 3, 0, S422, 80, optimize=4)
enc("prog_spectral_440_dri1", 35, 90, 3, 0, S440, 60, dri=1, optimize=5)
enc("prog_deep_420", 90, 64, 3, 0, S420, 95, optimize=8)
enc("prog_deep_gray_dri17", 120, 33, 1, 0, S444, 99, dri=17, optimize=9)
enc("prog_q1_16bit_tables", 48, 48, 3, 0, S420, 1, force_baseline=0, optimize=2)
enc("prog_narrow", 40, 3, 3, 0, S420, 90, optimize=2)
enc("prog_tiny", 1, 1, 3, 0, S420, 90, optimize=2)
# sequential files that go scan by scan: one scan per component (bit 4), Y+Cb then Cr (bits 4+6), a table pair per component
# (bit 5 with bit 0: table numbers 0..2, SOF1)
enc("seq_noninterleaved_420", 70, 85, 3, 0, S420, 85, optimize=16)
enc("seq_noninterleaved_444_dri4_opt", 33, 58, 3, 0, S444, 92, dri=4, optimize=17)
enc("seq_two_scans_422", 64, 49, 3, 0, S422, 75, optimize=16 + 64)
enc("seq_three_table_pairs_420", 55, 90, 3, 0, S420, 80, optimize=33)
enc("seq_three_table_pairs_noninterleaved_440", 47, 47, 3, 0, S440, 60, dri=2, optimize=16 + 33)
# four components (px = CMYK): mode 0 CMYK + Adobe marker, 1 YCCK + Adobe marker, 2 CMYK without a marker, 3 YCCK data without a marker
# (decoders take it for CMYK); K samples like the first component
enc("cmyk_adobe_444", 52, 61, 4, 0, S444, 90)
enc("cmyk_adobe_420k_dri3", 45, 77, 4, 0, S420, 80, dri=3)
enc("ycck_adobe_420", 66, 70, 4, 1, S420, 85)
enc("ycck_adobe_422_opt", 37, 95, 4, 1, S422, 92, optimize=1)
enc("ycck_adobe_440_progressive", 58, 41, 4, 1, S440, 75, optimize=2)
enc("cmyk_nomarker_444_progressive", 30, 44, 4, 2, S444, 88, optimize=2)
enc("ycck_nomarker_420_noninterleaved", 50, 50, 4, 3, S420, 70, optimize=16)
enc("cmyk_adobe_tiny", 1, 3, 4, 0, S420, 90)
# sampling factors beyond 2 and chroma sampled finer than luma (jdsample.c int_upsample = replication for every ratio but 2:1 / 1:2 / 2:2)
enc("samp_411", 40, 90, 3, 0, (4, 1, 1, 1, 1, 1), 85)
enc("samp_410_dri2", 70, 70, 3, 0, (4, 2, 1, 1, 1, 1), 80, dri=2)
enc("samp_chroma_mixed_22_21_11", 45, 63, 3, 0, (2, 2, 2, 1, 1, 1), 90)
enc("samp_luma_coarser_than_chroma_progressive", 50, 35, 3, 0, (1, 1, 2, 2, 2, 2), 75, optimize=2)
enc("samp_3x1", 33, 100, 3, 0, (3, 1, 1, 1, 1, 1), 88, optimize=1)
enc("samp_1x4", 100, 20, 3, 0, (1, 4, 1, 1, 1, 1), 70)
gold = {}
for f in sorted(os.listdir(out_dir)):
    d = open(os.path.join(out_dir, f), "rb").read()
    px = O.ref_jpeg_decode(d)
    gold[f] = "%dx%dx%d:%s" % (px.shape[0], px.shape[1], px.shape[2], hashlib.sha1(px.tobytes()).hexdigest()[:16])
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "exotic_golden.json"), "w"), indent=0, sort_keys=True)
print(len(gold), "files,", sum(os.path.getsize(os.path.join(out_dir, f)) for f in gold), "bytes")
