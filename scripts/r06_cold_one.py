"""One caller, many distinct sources that are not in the cache (what bench.py --workload abi --threads 1 measures): the latency distribution. Usage: python scripts/r06_cold_one.py [side] [distinct] [part]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lilliput_amd as la
from lilliput_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 128
part = sys.argv[3] if len(sys.argv) > 3 else "A"
srcs = [np.frombuffer(synth.synth_jpeg(i, side, 90), dtype=np.uint8) for i in range(nd)]
la.service_sim(srcs, 1, 64, 256, 256, 85, part=part)
r = la.service_sim(srcs, 1, 1024, 256, 256, 85, part=part)
lat = np.asarray(r["latency_ms"])
print("side %d, %d distinct, Part %s: %.1f img/s | mean %.3f ms p10 %.3f p50 %.3f p75 %.3f p90 %.3f p99 %.3f max %.3f" % (side, nd, part, r["ok"] / r["seconds"], lat.mean(), *np.percentile(lat, [10, 50, 75, 90, 99]), lat.max()))
print("in order, first 40:", " ".join("%.2f" % x for x in lat[:40]))
