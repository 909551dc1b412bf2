import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lilliput_amd as la
from lilliput_amd import synth
side = int(sys.argv[1])
d = synth.synth_jpeg(3, side, 90)
b = la.Batch(0)
for _ in range(30):
    b.transform([d], 256, 256)
time.sleep(0.05)
t0 = time.perf_counter(); b.transform([d], 256, 256); print("last call %.3f ms" % ((time.perf_counter() - t0) * 1e3))
