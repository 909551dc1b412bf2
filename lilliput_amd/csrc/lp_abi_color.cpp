// lp_abi_color.cpp -- the colour-signalling ABI of /root/reference/color_info.hpp: HDR detection (ICC 'cicp' tag, PNG cICP code
// points), the ICC header check, ICC profiles for a cICP primaries value, and the HDR -> SDR tone map (device: LpEngine::tonemap).
//
// Two things are built differently from the reference:
//  * the reference parses ICC blobs with Little-CMS (cmsOpenProfileFromMem + cmsReadTag(cicp)); reading one 12-byte tag is a header
//    and tag-table walk, done here with the same acceptance rules lcms applies on that route;
//  * the reference embeds five third-party .icc files as byte arrays (icc_profiles/*.h). This library derives the same kind of
//    profile (ICC v4.2 display class, matrix/TRC, D50 PCS, Bradford 'chad') from the standards' chromaticities and transfer
//    parameters at first use, so the colorimetry is the reference's while the bytes are this library's own
//    (tests/test_color.py compares tag values against the reference's profiles).
#include <math.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "lp_abi.h"
#include "lp_abi_guard.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline void put16(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 8); v.push_back(x); }
inline void put_sig(std::vector<uint8_t>& v, const char* s) { v.insert(v.end(), s, s + 4); }
inline void put_fix(std::vector<uint8_t>& v, double x) { put32(v, (uint32_t)(int32_t)lrint(x * 65536.0)); } // s15Fixed16Number

struct Mat3 { double m[3][3]; };
Mat3 mul(const Mat3& a, const Mat3& b)
{
    Mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
Mat3 inverse(const Mat3& a)
{
    const double (*m)[3] = a.m;
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    Mat3 r;
    r.m[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
    r.m[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; r.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    r.m[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; r.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; r.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    return r;
}

struct Space {
    const char* name;
    double xy[3][2];            // red, green, blue chromaticities
    double wx, wy;              // white point
    double g, a, b, c, d;       // parametric curve type 3: Y = (aX + b)^g for X >= d, cX below
};
// ITU-R BT.709 / IEC 61966-2-1 (sRGB), SMPTE EG 432-1 (Display P3), ITU-R BT.2020, SMPTE 170M (BT.601 525), ITU-R BT.470 BG (BT.601 625)
const double kSrgbCurve[5] = {2.4, 1 / 1.055, 0.055 / 1.055, 1 / 12.92, 0.04045};
const double k709Curve[5] = {1 / 0.45, 1 / 1.099, 0.099 / 1.099, 1 / 4.5, 0.081};
const double k2020Curve[5] = {1 / 0.45, 1 / 1.0993, 0.0993 / 1.0993, 1 / 4.5, 0.081}; // BT.2020's 12-bit alpha
const Space kSpaces[5] = {
    {"sRGB", {{0.64, 0.33}, {0.30, 0.60}, {0.15, 0.06}}, 0.3127, 0.3290, kSrgbCurve[0], kSrgbCurve[1], kSrgbCurve[2], kSrgbCurve[3], kSrgbCurve[4]},
    {"Display P3", {{0.680, 0.320}, {0.265, 0.690}, {0.150, 0.060}}, 0.3127, 0.3290, kSrgbCurve[0], kSrgbCurve[1], kSrgbCurve[2], kSrgbCurve[3], kSrgbCurve[4]},
    {"Rec. 2020", {{0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}}, 0.3127, 0.3290, k2020Curve[0], k2020Curve[1], k2020Curve[2], k2020Curve[3], k2020Curve[4]},
    {"Rec. 601 NTSC", {{0.630, 0.340}, {0.310, 0.595}, {0.155, 0.070}}, 0.3127, 0.3290, k709Curve[0], k709Curve[1], k709Curve[2], k709Curve[3], k709Curve[4]},
    {"Rec. 601 PAL", {{0.64, 0.33}, {0.29, 0.60}, {0.15, 0.06}}, 0.3127, 0.3290, k709Curve[0], k709Curve[1], k709Curve[2], k709Curve[3], k709Curve[4]},
};

void mluc(std::vector<uint8_t>& v, const char* text)
{
    const size_t n = strlen(text);
    put_sig(v, "mluc"); put32(v, 0); put32(v, 1); put32(v, 12);
    put_sig(v, "enUS"); put32(v, (uint32_t)n * 2); put32(v, 28);
    for (size_t i = 0; i < n; i++) put16(v, (uint8_t)text[i]);
    while (v.size() & 3) v.push_back(0);
}

std::vector<uint8_t> build_profile(const Space& sp)
{
    // RGB -> XYZ of the native white, then Bradford adaptation to the PCS illuminant (D50 as ICC.1 encodes it)
    Mat3 prim;
    for (int k = 0; k < 3; k++) { prim.m[0][k] = sp.xy[k][0] / sp.xy[k][1]; prim.m[1][k] = 1.0; prim.m[2][k] = (1 - sp.xy[k][0] - sp.xy[k][1]) / sp.xy[k][1]; }
    const double W[3] = {sp.wx / sp.wy, 1.0, (1 - sp.wx - sp.wy) / sp.wy};
    const Mat3 pinv = inverse(prim);
    double s[3];
    for (int k = 0; k < 3; k++) s[k] = pinv.m[k][0] * W[0] + pinv.m[k][1] * W[1] + pinv.m[k][2] * W[2];
    Mat3 rgb2xyz;
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) rgb2xyz.m[i][k] = prim.m[i][k] * s[k];
    const Mat3 brad = {{{0.8951, 0.2664, -0.1614}, {-0.7502, 1.7135, 0.0367}, {0.0389, -0.0685, 1.0296}}};
    const double D50[3] = {0.9642, 1.0, 0.8249};
    double cs[3], cd[3];
    for (int i = 0; i < 3; i++) { cs[i] = brad.m[i][0] * W[0] + brad.m[i][1] * W[1] + brad.m[i][2] * W[2]; cd[i] = brad.m[i][0] * D50[0] + brad.m[i][1] * D50[1] + brad.m[i][2] * D50[2]; }
    Mat3 scale = {{{cd[0] / cs[0], 0, 0}, {0, cd[1] / cs[1], 0}, {0, 0, cd[2] / cs[2]}}};
    const Mat3 chad = mul(inverse(brad), mul(scale, brad));
    const Mat3 col = mul(chad, rgb2xyz);

    // tag data
    struct Tag { const char* sig; uint32_t off, size; };
    std::vector<Tag> tags;
    std::vector<uint8_t> body;
    const uint32_t n_tags = 10, base = 128 + 4 + 12 * n_tags;
    auto begin = [&](const char* sig) { tags.push_back({sig, base + (uint32_t)body.size(), 0}); };
    auto end = [&]() { tags.back().size = base + (uint32_t)body.size() - tags.back().off; while (body.size() & 3) body.push_back(0); };
    begin("desc"); mluc(body, sp.name); end();
    begin("cprt"); mluc(body, "No copyright, use freely"); end();
    begin("wtpt"); put_sig(body, "XYZ "); put32(body, 0); for (int i = 0; i < 3; i++) put_fix(body, D50[i]); end();
    begin("chad"); put_sig(body, "sf32"); put32(body, 0); for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) put_fix(body, chad.m[i][k]); end();
    const char* colorant[3] = {"rXYZ", "gXYZ", "bXYZ"};
    for (int k = 0; k < 3; k++) { begin(colorant[k]); put_sig(body, "XYZ "); put32(body, 0); for (int i = 0; i < 3; i++) put_fix(body, col.m[i][k]); end(); }
    begin("rTRC"); put_sig(body, "para"); put32(body, 0); put16(body, 3); put16(body, 0);
    put_fix(body, sp.g); put_fix(body, sp.a); put_fix(body, sp.b); put_fix(body, sp.c); put_fix(body, sp.d); end();
    tags.push_back({"gTRC", tags.back().off, tags.back().size}); // the three channels share one curve
    tags.push_back({"bTRC", tags.back().off, tags.back().size});

    std::vector<uint8_t> p;
    put32(p, base + (uint32_t)body.size());         // profile size
    put32(p, 0);                                    // preferred CMM
    put32(p, 0x04200000);                           // version 4.2
    put_sig(p, "mntr"); put_sig(p, "RGB "); put_sig(p, "XYZ ");
    put16(p, 2025); put16(p, 1); put16(p, 1); put16(p, 0); put16(p, 0); put16(p, 0);
    put_sig(p, "acsp");
    put32(p, 0); put32(p, 0); put32(p, 0); put32(p, 0); // platform, flags, manufacturer, model
    put32(p, 0); put32(p, 0);                       // attributes
    put32(p, 0);                                    // rendering intent: perceptual
    for (int i = 0; i < 3; i++) put_fix(p, D50[i]); // PCS illuminant
    put32(p, 0);                                    // creator
    p.resize(128, 0);                               // profile ID (not computed) + reserved
    put32(p, n_tags);
    for (const Tag& t : tags) { put_sig(p, t.sig); put32(p, t.off); put32(p, t.size); }
    p.insert(p.end(), body.begin(), body.end());
    return p;
}

const std::vector<uint8_t>& profile(int idx)
{
    static std::vector<uint8_t> cache[5];
    static std::once_flag once;
    std::call_once(once, [] { for (int i = 0; i < 5; i++) cache[i] = build_profile(kSpaces[i]); });
    return cache[idx];
}

} // namespace

extern "C" {

bool cicp_is_hdr_transfer(uint8_t transfer) { return transfer == 16 || transfer == 18; } // color_info.cpp:38-41: PQ, HLG

bool icc_header_is_sane(const uint8_t* icc, size_t icc_len) // color_info.cpp:70-79
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!icc || icc_len < 128) return false;
    return (size_t)be32(icc) == icc_len;
}
LP_ABI_CATCH("icc_header_is_sane", return false)

// color_info.cpp:17-36: the ICC profile's 'cicp' tag names PQ or HLG. The acceptance rules are those of lcms on this route:
// 'acsp' magic, at most 100 tags, a tag is ignored when it does not fit inside the profile, the element is type 'cicp' and
// exactly 12 bytes (Type_VideoSignal_Read).
bool is_hdr_transfer_function(const uint8_t* icc, size_t len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!icc || len == 0 || len > 1024 * 1024 || len < 132) return false;
    if (memcmp(icc + 36, "acsp", 4) != 0) return false;
    size_t limit = be32(icc);
    if (limit >= len) limit = len;
    const uint32_t n = be32(icc + 128);
    if (n > 100 || 132 + (size_t)n * 12 > len) return false;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* e = icc + 132 + (size_t)i * 12;
        const uint64_t off = be32(e + 4), size = be32(e + 8);
        if (size == 0 || off == 0 || off + size > limit) continue;
        if (memcmp(e, "cicp", 4) != 0) continue;
        if (size != 12 || memcmp(icc + off, "cicp", 4) != 0) return false;
        const uint8_t transfer = icc[off + 9];
        return transfer == 16 || transfer == 18;
    }
    return false;
}
LP_ABI_CATCH("is_hdr_transfer_function", return false)

const uint8_t* cicp_get_icc_profile(uint8_t primaries, size_t* profile_size) // color_info.cpp:43-68
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    int idx;
    switch (primaries) {
    case 11: case 12: idx = 1; break; // SMPTE RP 431-2, EG 432-1: P3 primaries, D65
    case 9: idx = 2; break;
    case 5: idx = 4; break;           // BT.470 BG
    case 6: idx = 3; break;           // SMPTE 170M
    default: idx = 0; break;
    }
    const std::vector<uint8_t>& p = profile(idx);
    if (profile_size) *profile_size = p.size();
    return p.data();
}
LP_ABI_CATCH("cicp_get_icc_profile", return nullptr)

const uint8_t* lilliput_hip_srgb_icc_profile(size_t* profile_size) // lilliput.go:18-22 SRGBICCProfile
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    const std::vector<uint8_t>& p = profile(0);
    if (profile_size) *profile_size = p.size();
    return p.data();
}
LP_ABI_CATCH("lilliput_hip_srgb_icc_profile", return nullptr)

// color_info.cpp:112-204. src: width * height * 3 samples of src_depth bits; dst: width * height * 3 bytes. Host pointers.
void tonemap_rgb_to_sdr(const uint16_t* src, uint8_t* dst, int width, int height, int src_depth, uint8_t transfer, uint8_t primaries)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!src || !dst || width <= 0 || height <= 0 || src_depth < 1 || src_depth > 16) return;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) { fprintf(stderr, "lilliput_hip: tone map failed (no device)\n"); return; }
    if (eng->tonemap_host(src, dst, width, height, src_depth, transfer, primaries)) fprintf(stderr, "lilliput_hip: tone map failed: %s\n", eng->last_error().c_str());
}
LP_ABI_CATCH("tonemap_rgb_to_sdr", return)

// color_info.cpp:206-236. Host pointer, tightly packed; alpha untouched.
void tonemap_rgb_8u_inplace(uint8_t* pixels, int width, int height, int channels, uint8_t transfer, uint8_t primaries)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!pixels || width <= 0 || height <= 0 || (channels != 3 && channels != 4)) return;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) { fprintf(stderr, "lilliput_hip: tone map failed (no device)\n"); return; }
    if (eng->tonemap_host8(pixels, width, height, channels, transfer, primaries)) fprintf(stderr, "lilliput_hip: tone map failed: %s\n", eng->last_error().c_str());
}
LP_ABI_CATCH("tonemap_rgb_8u_inplace", return)

// Framebuffer.TonemapToSDR (opencv.go:794-812) for a Mat whose pixels live on the device: no host round trip.
int lilliput_hip_mat_tonemap(opencv_mat mat, uint8_t transfer, uint8_t primaries)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!m || m->rows <= 0 || m->cols <= 0) return 0;
    const int type_cn = ((m->type >> 3) & 511) + 1, depth = m->type & 7;
    if (depth != 0 || (type_cn != 3 && type_cn != 4)) return 0; // TonemapToSDR returns without touching other layouts
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) return -1;
    if (eng->tonemap(lp_mat_frame(m), transfer, primaries)) { lp_set_error(eng->last_error()); return -1; }
    m->dev_valid = true;
    return lp_mat_to_host(m, eng) ? 0 : -1;
}
LP_ABI_CATCH("lilliput_hip_mat_tonemap", return 0)

} // extern "C"
