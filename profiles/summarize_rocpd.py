"""Dump the per-kernel statistics (rocprofv3 --kernel-trace --stats, rocpd sqlite output) to a markdown table.
usage: python profiles/summarize_rocpd.py gpurun_out/<dir>/<name>_results.db > profiles/<round>_<what>.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("| kernel | calls | total us | avg us | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, tot, avg, pct in rows:
    print("| `%s` | %d | %.1f | %.1f | %.2f |" % (name.split("(")[0].replace("void ", ""), calls, tot, avg, pct))
