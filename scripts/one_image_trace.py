"""One image through the batch entry point, repeated: wall time per call and the library's own stage breakdown (LILLIPUT_HIP_TRACE=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lilliput_amd as la
from lilliput_amd import synth

for size in (int(a) for a in (sys.argv[1:] or ["512", "4096"])):
    data = synth.synth_jpeg(0, size)
    b = la.Batch(0)
    ts = []
    for it in range(30):
        t = time.perf_counter()
        r = b.transform([data], 256, 256, quality=85)
        ts.append(time.perf_counter() - t)
        assert r[0].status == 0
    ts = sorted(ts[5:])
    print("size %d: one-image batch call p50 %.3f ms, min %.3f ms" % (size, ts[len(ts) // 2] * 1e3, ts[0] * 1e3), flush=True)
    b.close()
