"""Instruction mix of the main loop (largest backward-branch region) of one kernel in a gfx950 listing, split into the two issue-cost
classes scripts/r06_valu_rates.hip measured (fast: add / sub / logic / mov / f32 add-mul-fma ~1.0x; slow: everything else ~1.5x).
Usage: python scripts/r06_isa_mix.py <listing.s> <kernel symbol prefix>"""
import collections, re, sys
L = open(sys.argv[1]).read().splitlines()
start = [i for i, l in enumerate(L) if l.startswith(sys.argv[2]) and ":" in l.split()[0]][0]
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
body = L[start:end]
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
loop = body[best[0]:best[1] + 1]
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32",
        "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_lshrrev_b32", "v_ashrrev_i32"}
cnt = collections.Counter()
for l in loop:
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].startswith("//"):
        continue
    cnt[re.sub(r"_e32$|_e64$", "", t[0])] += 1
valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
fast = sum(v for k, v in cnt.items() if k in FAST)
print("main loop: %d instructions: %d VALU (%d fast-class, %d slow-class), %d SALU, %d vector memory" % (
    sum(cnt.values()), valu, fast, valu - fast, sum(v for k, v in cnt.items() if k.startswith("s_")), sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_")))))
print("VALU cost in v_add_u32 units (slow = 1.5): %.0f = %.2f x the instruction count" % (fast + 1.5 * (valu - fast), (fast + 1.5 * (valu - fast)) / max(1, valu)))
for k, v in cnt.most_common(30):
    print("  %-30s %4d  %s" % (k, v, "fast" if k in FAST else "slow" if k.startswith("v_") else ""))
