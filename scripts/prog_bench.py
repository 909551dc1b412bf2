"""Progressive-source throughput: batches of identical-size progressive JPEGs through Batch.transform (upload + decode + resample +
encode), next to the same pixels saved as baseline. Usage: python scripts/prog_bench.py [side ...]"""
import io
import sys
import time

import numpy as np
from PIL import Image

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la
from lilliput_amd import synth

sides = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]
b = la.Batch(0)
for side in sides:
    rgb = synth.synth_rgb(1, side)
    files = {}
    for prog in (False, True):
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=90, subsampling=2, progressive=prog)
        files[prog] = buf.getvalue()
    for n in (1, 16, 64):
        if side >= 4096 and n > 16:
            continue
        row = []
        for prog in (False, True):
            srcs = [files[prog]] * n
            b.transform(srcs, 256, 256)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                r = b.transform(srcs, 256, 256)
            dt = (time.perf_counter() - t0) / reps
            assert all(x.status == 0 for x in r)
            row.append("%s %8.2f ms %8.1f img/s" % ("progressive" if prog else "baseline   ", dt * 1e3, n / dt))
        print("side %5d  batch %3d  %s | %s  (%d / %d KB)" % (side, n, row[0], row[1], len(files[False]) >> 10, len(files[True]) >> 10), flush=True)
