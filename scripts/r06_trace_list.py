"""Lists the launches of kernels matching a prefix from a rocprofv3 kernel trace CSV, in start order: name, duration (us), grid.
Usage: python scripts/r06_trace_list.py <dir> <prefix> [last_n]"""
import csv, glob, os, sys
d, pref = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(pref)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if last:
    rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    print("%-28s start %10.1f us  dur %10.1f us  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][:28], (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
