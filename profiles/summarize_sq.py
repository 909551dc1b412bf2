"""rocprofv3 passes of scripts/pmc_sq.sh (gpurun_out/<tag>/) -> the per-kernel table committed as profiles/rNN_*_sq_counters.md.
  python profiles/summarize_sq.py gpurun_out/<tag> <images per launch> [json-out]
Per kernel (averages per launch, exclusive = one stream, dispatches serialised by the counter collection):
  duration from the counter-free kernel trace; VALU / SALU / LDS instructions per wave; VALU issue utilisation =
  SQ_INSTS_VALU x 2 cycles (wave64 on a SIMD-32, scripts/microbench.hip) / (SIMDs x duration x clock); wave-cycle split
  (SQ_WAIT_ANY, SQ_WAIT_INST_ANY, the rest = issuing); LDS bank-conflict share; mean waves per SIMD = SQ_WAVE_CYCLES x 4 /
  (SQ_BUSY-derived duration); HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md)."""
import csv, glob, json, os, sys
from collections import defaultdict

d, nimg = sys.argv[1], int(sys.argv[2])
SIMDS, CLK = 1024, 2.4e9


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n.split("<")[0] if not n.startswith("k_resample") and not n.startswith("k_idct") else n


def counters(p):
    vals = defaultdict(lambda: defaultdict(list))
    hits = glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        return {}
    for r in csv.DictReader(open(hits[0])):
        vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # median over the launches of the run (the first launch of a process carries one-off work)
    return {k: {c: sorted(v)[len(v) // 2] for c, v in vals[k].items()} for k in vals}


dur = {}
hits = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(hits[0])):
    dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
C = {}
for p in ("sq1", "sq2", "sq3", "fetch", "write"):
    for k, v in counters(p).items():
        C.setdefault(k, {}).update(v)
rows, out = [], {}
for k, (us, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    c = C.get(k, {})
    if not k.startswith("k_") or not c:
        continue
    waves = c.get("SQ_WAVES", 0) or 1
    wc = c.get("SQ_WAVE_CYCLES", 0) * 4 or 1  # quad-cycles -> cycles
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e-6) if c.get("GRBM_GUI_ACTIVE") else CLK  # the counter is summed over the 8 XCDs
    valu_util = c.get("SQ_INSTS_VALU", 0) * 2 / (SIMDS * us * 1e-6 * CLK)
    hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    fetch_raw, write_raw = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
    o = {"us_per_launch": us, "us_per_image": us / nimg, "waves": waves, "valu_per_wave": c.get("SQ_INSTS_VALU", 0) / waves,
         "salu_per_wave": c.get("SQ_INSTS_SALU", 0) / waves, "lds_per_wave": c.get("SQ_INSTS_LDS", 0) / waves,
         "valu_issue_util": valu_util, "wait_any": c.get("SQ_WAIT_ANY", 0) * 4 / wc, "wait_inst": c.get("SQ_WAIT_INST_ANY", 0) * 4 / wc,
         "lds_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0) or c.get("SQ_ACTIVE_INST_LDS", 1)),
         "waves_per_simd": wc / (SIMDS * us * 1e-6 * CLK), "hbm_bytes_per_image": hbm / nimg, "fetch_size_bytes_per_image_uncorrected": fetch_raw / nimg,
         "write_size_bytes_per_image": write_raw / nimg, "gui_clock_ghz": clk / 1e9}
    out[k] = o
    rows.append("| `%s` | %.1f | %.2f | %d | %.0f | %.0f | %.0f | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.1f | %.2f |" % (
        k, us, us / nimg, waves, o["valu_per_wave"], o["salu_per_wave"], o["lds_per_wave"], 100 * valu_util, 100 * o["wait_any"], 100 * o["wait_inst"],
        100 * o["lds_conflict_share"], o["waves_per_simd"], hbm / nimg / 1e6))
print("| kernel | us / launch | us / image | waves | VALU / wave | SALU / wave | LDS / wave | VALU issue util | wave cycles in s_waitcnt | issue-stalled | LDS conflict share | mean waves / SIMD | HBM MB / image |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
print("\n".join(rows))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
