"""One batch of progressive sources through Batch.transform (for rocprofv3 traces). Usage: python scripts/r06_prog_one.py side n mode [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la
from lilliput_amd import synth

side, n, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L = la.binding.lib()
L.lilliput_hip_set_progressive_entropy(mode)
files = [synth.synth_jpeg(i, side, 90, progressive=True) for i in range(min(n, 8))]
srcs = [files[i % len(files)] for i in range(n)]
b = la.Batch(0)
b.transform(srcs, 256, 256)
for _ in range(reps):
    t0 = time.perf_counter()
    r = b.transform(srcs, 256, 256)
    dt = time.perf_counter() - t0
    assert all(x.status == 0 for x in r)
    print("side %d n %d mode %d: %.2f ms %.1f img/s" % (side, n, mode, dt * 1e3, n / dt), flush=True)
