"""Wall time of each opencv_* call of one Transform (decode -> crop -> resize -> encode), lazy host write-back on."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lilliput_amd as la
from lilliput_amd import synth

L = la.lib()
L.lilliput_hip_set_lazy_host(int(os.environ.get("LAZY", "1")))
AREA = C.c_int.in_dll(L, "CV_INTER_AREA").value
cases = {"large-sunrise 1300x1942": open(os.path.join(ROOT, "tests/golden/inputs/large-sunrise.jpg"), "rb").read(),
         "synthetic 4096x4096": synth.synth_jpeg(0, 4096)}
fb = [np.zeros(4096 * 4096 * 4, np.uint8), np.zeros(4096 * 4096 * 4, np.uint8)]
out = np.zeros(1 << 20, np.uint8)
for name, data in cases.items():
    src = np.frombuffer(data, np.uint8).copy()
    acc = {}
    N = 12
    for it in range(N):
        t = [time.perf_counter()]
        def mark(k):
            t.append(time.perf_counter())
            if it >= 2:
                acc[k] = acc.get(k, 0.0) + t[-1] - t[-2]
        em = L.opencv_mat_create_from_data(len(data), 1, 0, src.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)))
        dec = L.opencv_decoder_create(em)
        assert L.opencv_decoder_read_header(dec)
        w, h = L.opencv_decoder_get_width(dec), L.opencv_decoder_get_height(dec)
        mark("create+read_header")
        m = L.opencv_mat_create_from_data(w, h, 16, fb[0].ctypes.data_as(C.c_void_p), C.c_size_t(fb[0].size))
        assert L.opencv_decoder_read_data(dec, m)
        mark("read_data")
        side = min(w, h)
        v = L.opencv_mat_crop(m, (w - side) // 2, (h - side) // 2, side, side)
        d = L.opencv_mat_create_from_data(256, 256, 16, fb[1].ctypes.data_as(C.c_void_p), C.c_size_t(fb[1].size))
        L.opencv_mat_resize(v, d, 256, 256, AREA)
        mark("crop+resize")
        om = L.opencv_mat_create_empty_from_data(out.size, out.ctypes.data_as(C.c_void_p))
        enc = L.opencv_encoder_create(b".jpeg", om)
        opts = (C.c_int * 2)(1, 85)
        assert L.opencv_encoder_write(enc, d, opts, 2)
        n = L.opencv_mat_get_height(om)
        mark("encode")
        for x in (enc,):
            L.opencv_encoder_release(x)
        for x in (om, d, v, m):
            L.opencv_mat_release(x)
        L.opencv_decoder_release(dec)
        L.opencv_mat_release(em)
        mark("release")
    print(name, "->", n, "bytes;", ", ".join("%s %.2f ms" % (k, v / (N - 2) * 1e3) for k, v in acc.items()), "; total %.2f ms" % (sum(acc.values()) / (N - 2) * 1e3))
