"""Turn rocprofv3 CSV output (gpurun_out/<tag>/...) into the committed summaries.
  python profiles/summarize_csv.py stats <dir>            -> markdown table of <dir>/**/*_kernel_stats.csv
  python profiles/summarize_csv.py pmc <fetch_dir> <write_dir> <images_per_launch> -> markdown table + profiles/r01_pmc_traffic.json"""
import csv, glob, json, os, sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    assert hits, (d, suffix)
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


if sys.argv[1] == "stats":
    rows = list(csv.DictReader(open(find(sys.argv[2], "kernel_stats.csv"))))
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows:
        print("| `%s` | %d | %.1f | %.1f | %.2f |" % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
else:
    def per_kernel(d, counter):
        acc, cnt = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if r["Counter_Name"] != counter:
                continue
            acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
            cnt[short(r["Kernel_Name"])] += 1
        return {k: acc[k] / cnt[k] for k in acc}
    fetch, write = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
    n = int(sys.argv[4])
    out = {}
    print("| kernel | FETCH_SIZE KiB/launch | x2 (gfx950 correction) MB/image | WRITE_SIZE KiB/launch | MB/image |")
    print("|---|---:|---:|---:|---:|")
    for k in sorted(fetch, key=lambda k: -fetch[k] - write.get(k, 0)):
        if not k.startswith("k_"):
            continue
        f, w = fetch[k], write.get(k, 0.0)
        out[k.split("<")[0]] = {"fetch_kib_per_launch": f, "write_kib_per_launch": w, "images_per_launch": n, "hbm_bytes_per_image": (2 * f + w) * 1024 / n}
        print("| `%s` | %.0f | %.2f | %.0f | %.2f |" % (k, f, 2 * f * 1024 / n / 1e6, w, w * 1024 / n / 1e6))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r01_pmc_traffic.json"), "w"), indent=1)
