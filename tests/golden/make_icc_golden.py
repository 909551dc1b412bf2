"""Records the colorimetry of the reference's five canned ICC profiles (/root/reference/icc_profiles/*.icc, the blobs behind
color_info.cpp:43-68 cicp_get_icc_profile and lilliput.go:18-22 SRGBICCProfile) as numbers: header fields, white point, chromatic
adaptation matrix, colorants, parametric curve. tests/test_color.py holds this library's generated profiles to them.
Run in the build container (needs /root/reference):  python tests/golden/make_icc_golden.py
"""
import json
import os
import struct

REF = "/root/reference/icc_profiles"
NAMES = {"srgb": 1, "displayp3": 12, "rec2020": 9, "rec601_ntsc": 6, "rec601_pal": 5}  # profile -> a cICP primaries value that selects it


def fixed(b, off, n):
    return [struct.unpack(">i", b[off + 4 * k:off + 4 * k + 4])[0] / 65536.0 for k in range(n)]


def read_profile(b):
    out = {"size": len(b), "version": b[8:12].hex(), "class": b[12:16].decode(), "space": b[16:20].decode(), "pcs": b[20:24].decode(),
           "illuminant": fixed(b, 68, 3), "tags": {}}
    n = struct.unpack(">I", b[128:132])[0]
    for i in range(n):
        sig, off, size = struct.unpack(">4sII", b[132 + 12 * i:144 + 12 * i])
        typ = b[off:off + 4]
        if typ == b"XYZ ":
            val = fixed(b, off + 8, 3)
        elif typ == b"sf32":
            val = fixed(b, off + 8, 9)
        elif typ == b"para":
            ft = struct.unpack(">H", b[off + 8:off + 10])[0]
            val = {"type": ft, "params": fixed(b, off + 12, {0: 1, 1: 3, 2: 4, 3: 5, 4: 7}[ft])}
        else:
            continue
        out["tags"][sig.decode()] = val
    return out


if __name__ == "__main__":
    gold = {}
    for name, prim in NAMES.items():
        with open(os.path.join(REF, name + "_profile.icc"), "rb") as f:
            gold[name] = dict(read_profile(f.read()), primaries=prim)
    with open(os.path.join(os.path.dirname(__file__), "icc_golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("wrote", len(gold), "profiles")
