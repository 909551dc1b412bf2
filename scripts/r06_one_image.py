"""Latency of ONE image through the batched entry point and through Part A's call sequence (one caller). Usage: python scripts/r06_one_image.py [side ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la
from lilliput_amd import synth

sides = [int(a) for a in sys.argv[1:]] or [512, 4096]
b = la.Batch(0)
for side in sides:
    d = synth.synth_jpeg(3, side, 90)
    for _ in range(20):
        b.transform([d], 256, 256)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        r = b.transform([d], 256, 256)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("side %d Batch.transform of one image: p50 %.3f ms p10 %.3f p90 %.3f" % (side, ts[100] * 1e3, ts[20] * 1e3, ts[180] * 1e3), flush=True)
    if os.environ.get("LILLIPUT_HIP_TRACE"):
        continue
    for part in ("A", "C"):
        la.service_sim([d], 1, 50, 256, 256, 85, part=part)
        r = la.service_sim([d], 1, 400, 256, 256, 85, part=part)
        lat = np.sort(r["latency_ms"])
        print("side %d Part %s, one caller: %.1f img/s, p50 %.3f ms p90 %.3f ms (ok %d of %d)" % (side, part, r["ok"] / r["seconds"], lat[len(lat) // 2], lat[int(len(lat) * 0.9)], r["ok"], r["jobs"]), flush=True)
