"""One caller through Part A (ops.go's call sequence, a lone chain served on the caller's thread): a few calls for a kernel trace. Usage: python scripts/r06_one_parta.py side [jobs]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lilliput_amd as la
from lilliput_amd import synth
side = int(sys.argv[1]); jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = synth.synth_jpeg(3, side, 90)
r = la.service_sim([d], 1, jobs, 256, 256, 85, part="A")
lat = np.sort(r["latency_ms"])
print("side %d Part A one caller: %.1f img/s p50 %.3f ms (ok %d of %d)" % (side, r["ok"] / r["seconds"], lat[len(lat) // 2], r["ok"], r["jobs"]))
