"""Checks the gfx950 listing of lp_kernels_prog.hip: the hand-managed registers of its in-flight loads (v100 ... v104, see PW_RING_* in the
source) must appear ONLY in the hand-written instructions -- a global_load into them, or the v_mov that takes them behind an s_waitcnt --
and the kernel must use no scratch. Exit code 1 otherwise. Usage: python scripts/r06_check_prog_isa.py [hipcc]"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "lilliput_amd", "csrc", "lp_kernels_prog.hip")
hipcc = sys.argv[1] if len(sys.argv) > 1 else "/opt/rocm/bin/hipcc"
out = os.path.join(tempfile.mkdtemp(), "pw.s")
subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
bad, loads, takes = [], 0, 0
for i, l in enumerate(lines):
    t = l.strip()
    if not re.search(r"\bv10[0-4]\b", t) or t.startswith(";") or t.startswith("."):
        continue
    if re.match(r"global_load_(sshort|dword) v10[0-4], v\[\d+:\d+\], off$", t):
        loads += 1
    elif re.match(r"v_mov_b32 v\d+, v10[0-4]$", t):
        takes += 1
        prev = next(x.strip() for x in reversed(lines[:i]) if x.strip() and not x.strip().startswith(";"))
        if not prev.startswith("s_waitcnt vmcnt("):
            bad.append((i + 1, "take without a wait in front: " + prev))
    else:
        bad.append((i + 1, t))
scratch = [l for l in lines if "private_segment_fixed_size" in l and not l.strip().endswith(" 0")]
print("hand-issued loads %d, takes %d, foreign uses %d, scratch %s" % (loads, takes, len(bad), "yes" if scratch else "none"))
for b in bad:
    print("  line %d: %s" % b)
sys.exit(1 if bad or scratch or not loads else 0)
