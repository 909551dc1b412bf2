"""Development check of the wave-per-scan progressive decoder (lp_kernels_prog.hip): coefficients against the host threads' on synthetic
progressive files (first mismatch reported by component / block / coefficient), then batch throughput host vs device.
Usage: python scripts/r06_prog_dev.py [check|bench] [side ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la
from lilliput_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "check"
sides = [int(a) for a in sys.argv[2:]] or [64, 256, 1024]
L = la.binding.lib()
b = la.Batch(0)


def host_coefs(d, c):
    a = np.frombuffer(d, np.uint8)
    out = np.zeros(1 << 24, np.int16)
    bw, bh = C.c_int(), C.c_int()
    rc = L.lilliput_hip_progressive_coefs_host(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size), C.c_int(c), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                                               C.byref(bw), C.byref(bh), C.c_int(4))
    assert rc == 0, rc
    return out[: bw.value * bh.value * 64].reshape(bh.value, bw.value, 64)


if what == "check":
    bad = 0
    for side in sides:
        for seed, kw in ((1, {}), (2, {"restart_rows": 1}), (3, {"subsampling": 0}), (4, {"subsampling": 1, "quality": 50})):
            q = kw.pop("quality", 90)
            d = synth.synth_jpeg(seed, side, q, progressive=True, **kw)
            for mode in (1,):
                L.lilliput_hip_set_progressive_entropy(mode)
                for c in range(3):
                    got = b.decode_jpeg_coefs(d, c)
                    exp = host_coefs(d, c)
                    if not np.array_equal(got, exp):
                        bad += 1
                        w = np.argwhere(got != exp)
                        print("MISMATCH side %d seed %d mode %d comp %d: %d coefficients differ, first (by, bx, nat) = %s got %d want %d; rows touched %s" % (
                            side, seed, mode, c, len(w), w[0], got[tuple(w[0])], exp[tuple(w[0])], sorted(set(w[:, 0]))[:8]), flush=True)
                        zz = np.array(sorted(set(w[:, 2]))[:16])
                        print("   natural indices differing:", zz)
            print("side %d seed %d checked" % (side, seed), flush=True)
    L.lilliput_hip_set_progressive_entropy(-1)
    st = (C.c_uint64 * 3)()
    L.lilliput_hip_progressive_stats(st)
    print("device images %d, gave up %d, device scans %d" % tuple(st))
    print("mismatching components:", bad)
    sys.exit(1 if bad else 0)

for side in sides:
    files = [synth.synth_jpeg(i, side, 90, progressive=True) for i in range(8)]
    for n in ([int(x) for x in os.environ["BATCHES"].split(",")] if os.environ.get("BATCHES") else (16, 64, 256)):
        if side >= 4096 and n > 64 and not os.environ.get("BATCHES"):
            continue
        srcs = [files[i % 8] for i in range(n)]
        row = []
        for mode in (0, 1):
            L.lilliput_hip_set_progressive_entropy(mode)
            b.transform(srcs, 256, 256)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                r = b.transform(srcs, 256, 256)
            dt = (time.perf_counter() - t0) / reps
            assert all(x.status == 0 for x in r)
            row.append("%s %8.2f ms %8.1f img/s" % (("host", "device")[mode], dt * 1e3, n / dt))
            if mode == 0:
                ref = [x.data for x in r]
            else:
                assert [x.data for x in r] == ref, "device outputs differ from the host route's"
        print("side %5d  batch %3d  %s | %s" % (side, n, row[0], row[1]), flush=True)
L.lilliput_hip_set_progressive_entropy(-1)
st = (C.c_uint64 * 3)()
L.lilliput_hip_progressive_stats(st)
print("device images %d, gave up %d, device scans %d" % tuple(st))
