"""Per-dispatch counters of kernels matching a prefix from a rocprofv3 --pmc CSV. Usage: python scripts/r06_pmc_list.py <dir> <prefix> [last_n_dispatches]"""
import csv, glob, os, sys
from collections import OrderedDict, defaultdict
d, pref = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = OrderedDict()
for r in csv.DictReader(open(f)):
    if not r["Kernel_Name"].startswith(pref):
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {"grid": r.get("Grid_Size", "?")})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = list(disp)
if last:
    keys = keys[-last:]
for k in keys:
    v = disp[k]
    print("dispatch %6d grid %s: %s" % (k, v.pop("grid"), "  ".join("%s=%.0f" % kv for kv in sorted(v.items()))))
