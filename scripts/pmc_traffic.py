"""HBM traffic per kernel and image from the PMC passes of scripts/pmc_sq.sh, stamped with a hash of the kernel sources.
  python scripts/pmc_traffic.py gpurun_out/<tag>_sq <images per launch> profiles/rNN_pmc_traffic.json
hbm_bytes_per_image = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / images per launch -- FETCH_SIZE and WRITE_SIZE count KiB, and on gfx950
FETCH_SIZE counts 64-byte requests as 32 (MI355X_MICROARCH.md, HBM / rocprofv3 section). bench.py reads the newest such file and
withholds roofline.traffic when the stamp no longer matches the sources it runs."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(d, counter):
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert hits, d
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(hits[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip().split("<DevMem")[0] # k_huff_write<DevMem<8, 2, 1> > -> k_huff_write
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


def main():
    d, n, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    fetch, write = per_kernel(os.path.join(d, "fetch"), "FETCH_SIZE"), per_kernel(os.path.join(d, "write"), "WRITE_SIZE")
    import bench
    out = {"kernel_source_sha16": bench.kernel_source_sha16(),
           "measured_with": "scripts/pmc_sq.sh %s (FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes, kernel-trace only); hbm_bytes_per_image = "
                            "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / images per launch: the gfx950 correction of MI355X_MICROARCH.md" % os.path.basename(d)}
    print("| kernel | FETCH_SIZE x 2 MB / image | WRITE_SIZE MB / image | HBM MB / image |\n|---|---:|---:|---:|")
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
        if not k.startswith("k_"):
            continue
        f, w = fetch[k] * 1024 / n, write.get(k, 0.0) * 1024 / n
        out[k] = {"hbm_bytes_per_image": 2 * f + w, "fetch_size_bytes_per_image_uncorrected": f, "write_size_bytes_per_image": w, "images_per_launch": n}
        print("| `%s` | %.2f | %.2f | %.2f |" % (k, 2 * f / 1e6, w / 1e6, (2 * f + w) / 1e6))
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
