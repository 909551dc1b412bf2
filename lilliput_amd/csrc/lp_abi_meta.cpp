// lp_abi_meta.cpp -- the colour-metadata readers of the opencv.hpp C ABI (/root/reference/opencv.hpp:118-132):
// host-side byte walks, no device work. The reference implements them by handing the buffer to libjpeg-turbo /
// libpng (/root/reference/opencv.cpp:252-296, 314-395) and by splicing a chunk into a finished PNG (:413-463); the
// Go callers are openCVDecoder.ICC / CICP (/root/reference/opencv.go:697-767) and ImageOps.applyOutputCICP
// (/root/reference/ops.go:306-333). Here the container walks are written out directly, following the published
// behaviour of the two libraries' header readers (libjpeg-turbo 3.1.0 jdmarker.c/jdinput.c/jdicc.c, libpng 1.6.47
// pngread.c/pngrutil.c/png.c); tests/test_meta.py compares them with the reference's own prebuilt libraries on
// well-formed and deliberately broken streams.
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "../../include/lilliput_hip.h"
#include "lp_png.h"
#include "lp_abi_guard.h"

namespace {
// ------------------------------------------------------------------------------------------------ JPEG
// The memory source of libjpeg never fails: past the end of the buffer it hands out a fake EOI (FF D9) again and again
// (jdatasrc.c fill_mem_input_buffer), so a truncated segment is "completed" with that pattern and the walk goes on.
struct Src {
    const uint8_t* p;
    size_t n, i;
    uint8_t get() { const uint8_t v = i < n ? p[i] : (((i - n) & 1) ? 0xD9 : 0xFF); i++; return v; }
    uint32_t get2() { const uint32_t hi = get(); return (hi << 8) | get(); }
};

// What jpeg_read_header(require_image = TRUE) has to see before it reports JPEG_HEADER_OK: SOI, a frame header the
// library accepts, well-formed table segments, then SOS (jdmarker.c read_markers, jdinput.c initial_setup). Any libjpeg
// error, or reaching (real or fake) EOI first, makes the reference return 0. APP2 payloads seen on the way are collected.
struct App2 { const uint8_t* d; uint32_t len; };

bool jpeg_header_ok(const uint8_t* s, size_t n, std::vector<App2>& app2)
{
    Src r{s, n, 0};
    if (r.get() != 0xFF || r.get() != 0xD8) return false;   // first_marker: JERR_NO_SOI
    bool have_sof = false, lossless = false;
    int ncomp = 0, prec = 0, width = 0, height = 0;
    int comp_id[256], comp_samp[256];
    for (;;) {
        // next_marker: skip garbage up to an 0xFF, skip fill 0xFFs; a zero after 0xFF is stuffed data -> keep looking
        uint8_t m;
        for (;;) {
            uint8_t c = r.get();
            while (c != 0xFF) c = r.get();
            do c = r.get(); while (c == 0xFF);
            if (c != 0) { m = c; break; }
        }
        if (m == 0xD8) return false;                         // JERR_SOI_DUPLICATE
        if (m == 0xD9) return false;                         // EOI before SOS: JERR_NO_IMAGE
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; // TEM / RSTn: parameterless
        const bool sof_ok = m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC3 || m == 0xC9 || m == 0xCA || m == 0xCB;
        const bool known = sof_ok || m == 0xC4 || m == 0xCC || m == 0xDA || m == 0xDB || m == 0xDC || m == 0xDD || (m >= 0xE0 && m <= 0xEF) || m == 0xFE;
        if (!known) return false;                            // JERR_SOF_UNSUPPORTED (C5-C7, C8, CD-CF) / JERR_UNKNOWN_MARKER
        long length = (long)r.get2();
        if (sof_ok) { // get_sof
            prec = r.get(); height = (int)r.get2(); width = (int)r.get2(); ncomp = r.get();
            length -= 8;
            if (have_sof) return false;                                // JERR_SOF_DUPLICATE
            if (height <= 0 || width <= 0 || ncomp <= 0) return false; // JERR_EMPTY_IMAGE
            if (length != (long)ncomp * 3) return false;               // JERR_BAD_LENGTH
            for (int c = 0; c < ncomp; c++) {
                comp_id[c] = r.get();
                comp_samp[c] = r.get();
                (void)r.get();
            }
            lossless = m == 0xC3 || m == 0xCB;
            have_sof = true;
        } else if (m == 0xC4) { // get_dht
            length -= 2;
            while (length > 16) {
                const int idx = r.get();
                long cnt = 0;
                for (int k = 1; k <= 16; k++) cnt += r.get();
                length -= 17;
                if (cnt > 256 || cnt > length) return false;           // JERR_BAD_HUFF_TABLE
                for (long k = 0; k < cnt; k++) (void)r.get();
                length -= cnt;
                if ((idx & 0x10 ? idx - 0x10 : idx) >= 4) return false; // JERR_DHT_INDEX
            }
            if (length != 0) return false;                             // JERR_BAD_LENGTH
        } else if (m == 0xDB) { // get_dqt: always reads 64 entries per table
            length -= 2;
            while (length > 0) {
                const int b = r.get();
                if ((b & 15) >= 4) return false;                       // JERR_DQT_INDEX
                for (int k = 0; k < ((b >> 4) ? 128 : 64); k++) (void)r.get();
                length -= 65;
                if (b >> 4) length -= 64;
            }
            if (length != 0) return false;
        } else if (m == 0xDD) { // get_dri
            if (length != 4) return false;
            (void)r.get2();
        } else if (m == 0xCC) { // get_dac
            length -= 2;
            while (length > 0) {
                const int idx = r.get(), val = r.get();
                length -= 2;
                if (idx >= 32) return false;                           // JERR_DAC_INDEX
                if (idx < 16 && (val & 15) > (val >> 4)) return false; // JERR_DAC_VALUE
            }
            if (length != 0) return false;
        } else if (m == 0xDA) { // get_sos, then jdinput.c initial_setup on the first scan
            if (!have_sof) return false;                               // JERR_SOS_NO_SOF
            const int ns = r.get();
            if (length != (long)ns * 2 + 6 || ns < 1 || ns > 4) return false; // JERR_BAD_LENGTH
            int cur[4] = {-1, -1, -1, -1}; // cur_comp_info[], indexed by scan position while filled, by frame index while probed
            for (int k = 0; k < ns; k++) {
                const int cc = r.get();
                (void)r.get();
                int f = 0;
                for (; f < ncomp && f < 4; f++)
                    if (comp_id[f] == cc && cur[f] < 0) break;
                if (f == ncomp || f == 4) return false;                // JERR_BAD_COMPONENT_ID
                cur[k] = f;
                for (int q = 0; q < k; q++)
                    if (cur[q] == f) return false;                     // "this CSi should differ from the previous CSi"
            }
            if (height > 65500 || width > 65500) return false;         // JPEG_MAX_DIMENSION
            if (lossless ? (prec < 2 || prec > 16) : (prec != 8 && prec != 12)) return false; // JERR_BAD_PRECISION
            if (ncomp > 10) return false;                              // MAX_COMPONENTS
            for (int c = 0; c < ncomp; c++) {
                const int hs = comp_samp[c] >> 4, vs = comp_samp[c] & 15;
                if (hs < 1 || hs > 4 || vs < 1 || vs > 4) return false; // JERR_BAD_SAMPLING
            }
            return true;
        } else { // APPn / COM / DNL: length-prefixed, skipped or saved
            if (length < 2) continue;                                  // skip_variable / save_marker shrug off a bogus length word
            const size_t at = r.i;
            const uint32_t dl = (uint32_t)length - 2;
            if (m == 0xE2 && at + dl <= n) app2.push_back(App2{s + at, dl}); // one that runs off the end can never be followed by SOS
            r.i += dl;
        }
    }
}

// jdicc.c jpeg_read_icc_profile: the chunks carry (sequence number, count); all must agree, be present once, non-empty in total
int jpeg_icc_assemble(const std::vector<App2>& app2, uint8_t* dst, size_t cap)
{
    static const char kTag[12] = {'I', 'C', 'C', '_', 'P', 'R', 'O', 'F', 'I', 'L', 'E', 0};
    int num = 0;
    bool present[256] = {false};
    uint32_t dlen[256] = {0};
    const uint8_t* dptr[256] = {nullptr};
    for (const App2& a : app2) {
        if (a.len < 14 || memcmp(a.d, kTag, 12) != 0) continue;
        if (num == 0) num = a.d[13];
        else if (num != a.d[13]) return 0;
        const int seq = a.d[12];
        if (seq <= 0 || seq > num) return 0;
        if (present[seq]) return 0;
        present[seq] = true;
        dlen[seq] = a.len - 14;
        dptr[seq] = a.d + 14;
    }
    if (num == 0) return 0;
    size_t total = 0;
    for (int s = 1; s <= num; s++) {
        if (!present[s]) return 0;
        total += dlen[s];
    }
    if (total == 0 || total > cap) return 0;
    size_t o = 0;
    for (int s = 1; s <= num; s++) { memcpy(dst + o, dptr[s], dlen[s]); o += dlen[s]; }
    return (int)total;
}

} // namespace

static const uint8_t kPngSig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static inline bool is_type(const uint8_t* t, const char* name) { return memcmp(t, name, 4) == 0; }

extern "C" {

int opencv_decoder_get_jpeg_icc(void* src, size_t src_len, void* dest, size_t dest_len) // opencv.cpp:252-296
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!src || !dest) return 0;
    std::vector<App2> app2;
    if (!jpeg_header_ok(static_cast<const uint8_t*>(src), src_len, app2)) return 0;
    return jpeg_icc_assemble(app2, static_cast<uint8_t*>(dest), dest_len);
}
LP_ABI_CATCH("opencv_decoder_get_jpeg_icc", return 0)

int opencv_decoder_get_png_icc(void* src, size_t src_len, void* dest, size_t dest_len) // opencv.cpp:314-344
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!src || !dest) return 0;
    LpPngInfo info;
    if (!lp_png_read_info(static_cast<const uint8_t*>(src), src_len, info)) return 0;
    if (info.icc.empty() || info.icc.size() > dest_len) return 0;
    memcpy(dest, info.icc.data(), info.icc.size());
    return (int)info.icc.size();
}
LP_ABI_CATCH("opencv_decoder_get_png_icc", return 0)

int opencv_decoder_get_png_cicp(void* src, size_t src_len, uint8_t* primaries, uint8_t* transfer, uint8_t* matrix, uint8_t* full_range) // opencv.cpp:357-395
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!src) return 0;
    LpPngInfo info;
    if (!lp_png_read_info(static_cast<const uint8_t*>(src), src_len, info) || !info.have_cicp) return 0;
    *primaries = info.cicp[0];
    *transfer = info.cicp[1];
    *matrix = info.cicp[2];
    *full_range = info.cicp[3];
    return 1;
}
LP_ABI_CATCH("opencv_decoder_get_png_cicp", return 0)

// Splice a 16-byte cICP chunk in right after IHDR of a finished PNG, in place; returns the new length, or the old one
// when the buffer is not a PNG or has no room (opencv.cpp:413-463).
size_t opencv_png_insert_cicp(void* png, size_t png_len, size_t png_cap, uint8_t primaries, uint8_t transfer, uint8_t matrix, uint8_t full_range)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    uint8_t* b = static_cast<uint8_t*>(png);
    const size_t add = 16;
    if (!b || png_len < 8 + 12 || png_len + add > png_cap) return png_len;
    if (memcmp(b, kPngSig, 8) != 0 || !is_type(b + 12, "IHDR")) return png_len;
    const size_t at = 8 + 12 + (size_t)be32(b + 8);
    if (at > png_len) return png_len;
    uint8_t c[16] = {0, 0, 0, 4, 'c', 'I', 'C', 'P', primaries, transfer, matrix, full_range, 0, 0, 0, 0};
    const uint32_t crc = (uint32_t)crc32(0, c + 4, 8);
    c[12] = (uint8_t)(crc >> 24); c[13] = (uint8_t)(crc >> 16); c[14] = (uint8_t)(crc >> 8); c[15] = (uint8_t)crc;
    memmove(b + at + add, b + at, png_len - at);
    memcpy(b + at, c, add);
    return png_len + add;
}
LP_ABI_CATCH("opencv_png_insert_cicp", return 0)

} // extern "C"
