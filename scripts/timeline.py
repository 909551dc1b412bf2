"""Timeline of the LAST end-to-end step in a rocprofv3 --kernel-trace --memory-copy-trace run (scripts/trace_e2e.sh):
per 5 ms bin, how busy the H2D copy engine was and how many kernels were running (sum of kernel durations / bin width)."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "")) for r in csv.DictReader(open(kt))]
M = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], 0) for r in csv.DictReader(open(mt))]
big = sorted(m for m in M if m[1] - m[0] > 1e6 and "HOST_TO_DEVICE" in m[2])  # the chunk uploads (the trace carries no size column)
# steps = clusters of big H2D copies separated by > 20 ms
steps, cur = [], [big[0]]
for m in big[1:]:
    if m[0] - cur[-1][1] > 20e6:
        steps.append(cur); cur = []
    cur.append(m)
steps.append(cur)
last = steps[-1]
t0 = last[0][0]
kk = [k for k in K if k[0] >= t0 - 1e6]
t1 = max(k[1] for k in kk)
print("last step: %.1f ms from the first chunk upload to the last kernel; %d chunk uploads, sum of their durations %.1f ms, last one ends at %.1f ms" % (
    (t1 - t0) / 1e6, len(last), sum(m[1] - m[0] for m in last) / 1e6, (max(m[1] for m in last) - t0) / 1e6))
B = 5e6
nb = int((t1 - t0) / B) + 1
h2d, kern = [0.0] * nb, [0.0] * nb
byk = defaultdict(lambda: [0.0] * nb)
def spread(arr, a, b):
    a, b = max(a, t0), min(b, t1)
    i = int((a - t0) / B)
    while a < b and i < nb:
        e = min(b, t0 + (i + 1) * B)
        arr[i] += e - a
        a = e; i += 1
for m in last: spread(h2d, m[0], m[1])
for k in kk:
    spread(kern, k[0], k[1]); spread(byk[k[2].split("<")[0][:14]], k[0], k[1])
names = sorted(byk, key=lambda n: -sum(byk[n]))[:7]
print("bin(ms)  H2D-busy  kernels-in-flight  " + "  ".join(n.ljust(14) for n in names))
for i in range(nb):
    print("%5.0f    %5.2f     %5.2f              " % (i * B / 1e6, h2d[i] / B, kern[i] / B) + "  ".join(("%5.2f" % (byk[n][i] / B)).ljust(14) for n in names))
