// lp_png.cpp -- see lp_png.h.
#include "lp_png.h"
#include "lp_inflate.h"
#include <atomic>

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>

namespace {
const uint8_t kPngSig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline bool is_type(const uint8_t* t, const char* name) { return memcmp(t, name, 4) == 0; }

// ICC profile checks libpng applies before it keeps an iCCP profile (png.c png_icc_check_length/_header/_tag_table)
bool icc_profile_acceptable(const std::vector<uint8_t>& p, int color_type)
{
    if (p.size() < 132) return false;
    const uint32_t plen = be32(p.data());
    if (plen != p.size()) return false;
    if ((plen & 3) && p[8] > 3) return false;                                // "invalid length": from ICC v4 on the size is a multiple of four
    const uint32_t tags = be32(p.data() + 128);
    if (tags > 357913930u || 132 + (uint64_t)tags * 12 > plen) return false; // "tag count too large"
    if (be32(p.data() + 64) >= 0xffff) return false;                         // rendering intent out of range: "invalid rendering intent"
    if (be32(p.data() + 36) != 0x61637370u) return false;                    // 'acsp'
    static const uint8_t d50[12] = {0x00, 0x00, 0xf6, 0xd6, 0x00, 0x01, 0x00, 0x00, 0x00, 0x00, 0xd3, 0x2d};
    (void)d50; // a PCS illuminant other than D50 only draws a warning
    const uint32_t space = be32(p.data() + 16);
    if (space == 0x52474220u) { if (!(color_type & 2)) return false; }       // 'RGB ' needs a colour PNG
    else if (space == 0x47524159u) { if (color_type & 2) return false; }     // 'GRAY' needs a grey PNG
    else return false;                                                       // "invalid ICC profile color space"
    const uint32_t cls = be32(p.data() + 12);
    if (cls == 0x61627374u /* abst */ || cls == 0x6c696e6bu /* link */) return false; // not a display/input/output profile: rejected
    if (cls == 0x6e6d636cu /* nmcl */) { /* only a warning */ }
    const uint32_t pcs = be32(p.data() + 20);
    if (pcs != 0x58595a20u && pcs != 0x4c616220u) return false;              // 'XYZ ' / 'Lab '
    for (uint32_t t = 0; t < tags; t++) {
        const uint8_t* e = p.data() + 132 + (size_t)t * 12;
        const uint32_t off = be32(e + 4), sz = be32(e + 8);
        if (off > plen || sz > plen - off) return false;                     // "ICC profile tag outside profile"
    }
    return true;
}

// png_read_info up to the first IDAT. false = libpng would have raised an error (the reference then reports nothing).
} // namespace

bool lp_png_read_info(const uint8_t* s, size_t n, LpPngInfo& out)
{
    if (n < 8 || memcmp(s, kPngSig, 8) != 0) return false;
    size_t i = 8;
    bool have_ihdr = false, have_plte = false, have_iccp = false, seen_cicp = false, after_plte_slot = false;
    int color_type = 0;
    for (;;) {
        if (n - i < 8) return false;                                   // read past the end
        const uint32_t len = be32(s + i);
        const uint8_t* type = s + i + 4;
        if (len > 0x7fffffffu) return false;                           // "PNG unsigned integer out of range"
        for (int k = 0; k < 4; k++) {
            const uint8_t c = type[k];
            if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return false; // "bad header (invalid type)"
        }
        if (type[2] & 0x20) return false;                              // reserved bit (third letter lower case): same error
        const bool is_idat = is_type(type, "IDAT");
        if (is_idat) {
            if (!have_ihdr) return false;                              // "Missing IHDR before IDAT"
            if (color_type == 3 && !have_plte) return false;           // "Missing PLTE before IDAT"
            out.idat_off = i;
            return true;
        }
        if (n - i - 8 < (size_t)len + 4) return false;                 // truncated chunk
        const uint8_t* d = s + i + 8;
        const bool crc_ok = be32(d + len) == (uint32_t)crc32(crc32(0, type, 4), d, len);
        const bool ancillary = (type[0] & 0x20) != 0;
        i += 12 + (size_t)len;
        if (!crc_ok && !ancillary && !is_type(type, "PLTE")) return false; // CRC error in a critical chunk
        if (is_type(type, "IHDR")) {
            if (have_ihdr) return false;                               // "out of place"
            if (len != 13) return false;                               // "invalid"
            const uint32_t w = be32(d), h = be32(d + 4);
            const int depth = d[8], ct = d[9];
            bool ok = w != 0 && h != 0 && w <= 0x7fffffffu && h <= 0x7fffffffu && w <= 1000000u && h <= 1000000u;
            ok = ok && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16);
            ok = ok && (ct == 0 || ct == 2 || ct == 3 || ct == 4 || ct == 6);
            ok = ok && !((ct == 3 && depth > 8) || ((ct == 2 || ct == 4 || ct == 6) && depth < 8));
            ok = ok && d[10] == 0 && d[11] == 0 && d[12] <= 1;
            if (!ok) return false;                                     // "Invalid IHDR data"
            have_ihdr = true;
            color_type = ct;
            out.width = w; out.height = h; out.depth = depth; out.color_type = ct; out.interlace = d[12];
            continue;
        }
        if (!have_ihdr) return false;                                  // every handler: "missing IHDR"
        if (is_type(type, "IEND")) return false;                       // before any IDAT: "out of place"
        if (is_type(type, "PLTE")) {
            // critical only for palette images; in the other colour types a broken or misplaced PLTE is shrugged off
            if (color_type == 3) {
                if (!crc_ok || have_plte || after_plte_slot || len == 0 || len > 768 || len % 3) return false;
                have_plte = true;
                int np = (int)(len / 3);
                if (np > (1 << out.depth)) np = 1 << out.depth;            // entries beyond what the bit depth can address are dropped
                out.num_palette = np;
                memcpy(out.palette, d, (size_t)np * 3);
                continue;
            }
            if (have_plte || after_plte_slot) continue;                // "duplicate" / "out of place" (after tRNS or bKGD)
            if (!(color_type & 2)) continue;                           // "ignored in grayscale PNG"
            if (len > 768 || len % 3) continue;                        // "invalid"
            if (len == 0) return false;                                // png_set_PLTE: "Invalid palette"
            have_plte = true;                                          // counts as seen even with a CRC error
            continue;
        }
        if (!ancillary) return false;                                  // "unhandled critical chunk"
        if (!crc_ok && !is_type(type, "iCCP")) continue;               // ancillary chunk with a CRC error: dropped with a warning (the iCCP
                                                                       // reader only warns and keeps the profile it has already inflated)
        if (is_type(type, "cICP")) {
            if (have_plte || seen_cicp || len != 4) continue;          // out of place / duplicate / invalid: benign
            seen_cicp = true;                                          // from here on a further cICP is a duplicate ...
            if (d[2] != 0) continue;                                   // ... even when this one is unusable: RGB data needs identity matrix coefficients
            memcpy(out.cicp, d, 4);
            out.have_cicp = true;
        } else if (is_type(type, "iCCP")) {
            if (have_plte || have_iccp) continue;                      // out of place / duplicate (only an accepted profile counts)
            if (len < 81 + 11) continue;                               // "too short": libpng reads 81 bytes for the keyword and wants a minimal zlib stream after them
            uint32_t k = 0;
            while (k < 80 && d[k]) k++;
            if (k == 0 || k > 79) continue;                            // "bad keyword"
            if (d[k + 1] != 0) continue;                               // "bad compression method"
            // Inflate exactly as many bytes as the profile header announces; libpng only asks that they all arrive (a
            // missing checksum or further output is "extra compressed data", a warning).
            std::vector<uint8_t> prof(132);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit(&zs) != Z_OK) continue;
            zs.next_in = const_cast<uint8_t*>(d + k + 2);
            zs.avail_in = len - (k + 2);
            zs.next_out = prof.data();
            zs.avail_out = 132;
            (void)inflate(&zs, Z_NO_FLUSH);
            bool good = zs.avail_out == 0;
            if (good) {
                const uint32_t plen = be32(prof.data());
                good = plen >= 132 && plen <= 8000000u;                // "too short" / user_chunk_malloc_max
                if (good) {
                    prof.resize(plen);
                    zs.next_out = prof.data() + 132;
                    zs.avail_out = plen - 132;
                    if (zs.avail_out) (void)inflate(&zs, Z_FINISH);
                    good = zs.avail_out == 0;
                }
            }
            inflateEnd(&zs);
            if (good && icc_profile_acceptable(prof, color_type)) { out.icc.swap(prof); have_iccp = true; }
        }
        else if (is_type(type, "tRNS")) { // png_handle_tRNS; an accepted tRNS or bKGD also closes the slot in which PLTE may appear
            if (out.num_trans) continue;                               // "duplicate"
            if (color_type == 0 && len == 2) { out.trans_key[0] = (uint16_t)((d[0] << 8) | d[1]); out.num_trans = 1; }
            else if (color_type == 2 && len == 6) {
                for (int c = 0; c < 3; c++) out.trans_key[c] = (uint16_t)((d[2 * c] << 8) | d[2 * c + 1]);
                out.num_trans = 1;
            } else if (color_type == 3 && have_plte && len >= 1 && len <= 256 && (int)len <= out.num_palette) {
                memcpy(out.trans_alpha, d, len);
                out.num_trans = (int)len;
            } else continue;                                           // "invalid" / "out of place" / "invalid with alpha channel"
            after_plte_slot = true;
        } else if (is_type(type, "bKGD")) {
            if (color_type == 3 ? (have_plte && len == 1) : (color_type & 2) ? len == 6 : len == 2) after_plte_slot = true;
        }
        // every other ancillary chunk is irrelevant to these readers
    }
}

// ------------------------------------------------------------------------------------------------ image data
namespace {
const uint32_t kA7x0[7] = {0, 4, 0, 2, 0, 1, 0}, kA7y0[7] = {0, 0, 4, 0, 2, 0, 1}, kA7dx[7] = {8, 8, 4, 4, 2, 2, 1}, kA7dy[7] = {8, 8, 8, 4, 4, 2, 2};
inline size_t packed_row_bytes(uint32_t w, int bits_per_pixel) { return ((size_t)w * (size_t)bits_per_pixel + 7) / 8; }

bool chunk_name_ok(const uint8_t* type)
{
    for (int k = 0; k < 4; k++) {
        const uint8_t c = type[k];
        if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return false;
    }
    return !(type[2] & 0x20);
}
}

void lp_png_pass_geometry(const LpPngInfo& info, int pass, uint32_t* pw, uint32_t* ph, uint32_t* x0, uint32_t* y0, uint32_t* dx, uint32_t* dy)
{
    if (!info.interlace) { *pw = info.width; *ph = pass == 0 ? info.height : 0; *x0 = *y0 = 0; *dx = *dy = 1; return; }
    *x0 = kA7x0[pass]; *y0 = kA7y0[pass]; *dx = kA7dx[pass]; *dy = kA7dy[pass];
    *pw = info.width > *x0 ? (info.width - *x0 + *dx - 1) / *dx : 0;
    *ph = info.height > *y0 ? (info.height - *y0 + *dy - 1) / *dy : 0;
}

size_t lp_png_filtered_size(const LpPngInfo& info)
{
    const int bpp = info.depth * lp_png_channels_in_file(info.color_type);
    size_t total = 0;
    for (int p = 0; p < (info.interlace ? 7 : 1); p++) {
        uint32_t pw, ph, x0, y0, dx, dy;
        lp_png_pass_geometry(info, p, &pw, &ph, &x0, &y0, &dx, &dy);
        if (pw && ph) total += (size_t)ph * (1 + packed_row_bytes(pw, bpp)); // empty passes carry no data at all
    }
    return total;
}

namespace {
// How libpng pulls IDAT data (pngrutil.c png_read_IDAT_data): at most 8192 bytes of one chunk at a time, the chunk's CRC
// verified when the reader moves on to the next chunk, which has to be another IDAT. The granularity matters: whether a
// damaged stream end is met while a row is being produced (fatal) or only while the remainder is swallowed (a warning)
// depends on where the buffer boundaries fall.
struct IdatFeed {
    const uint8_t* s;
    size_t n, i;          // cursor: next unread byte of the file
    uint32_t left = 0;    // bytes of the current IDAT not handed to zlib yet
    uint32_t crc = 0;
    bool finish_crc()     // png_crc_finish(0) on a critical chunk
    {
        if (n - i < 4) return false;
        const bool ok = be32(s + i) == crc;
        i += 4;
        return ok;
    }
    bool open_chunk(bool first)
    {
        if (n - i < 8) return false;
        const uint32_t len = be32(s + i);
        const uint8_t* type = s + i + 4;
        if (len > 0x7fffffffu || !chunk_name_ok(type)) return false;
        if (!is_type(type, "IDAT")) return false; // "Not enough image data" (never taken for the first one)
        (void)first;
        crc = (uint32_t)crc32(0, type, 4);
        i += 8;
        left = len;
        return true;
    }
    bool fill(z_stream& zs)
    {
        while (left == 0) {
            if (!finish_crc()) return false;
            if (!open_chunk(false)) return false;
        }
        const uint32_t take = left < 8192u ? left : 8192u; // PNG_IDAT_READ_SIZE
        if (n - i < take) return false;
        zs.next_in = const_cast<uint8_t*>(s + i);
        zs.avail_in = take;
        crc = (uint32_t)crc32(crc, s + i, take);
        i += take;
        left -= take;
        return true;
    }
    bool skip_rest() // png_crc_finish(idat_size): the unread tail of the last IDAT, then its CRC
    {
        if (n - i < left) return false;
        crc = (uint32_t)crc32(crc, s + i, left);
        i += left;
        left = 0;
        return finish_crc();
    }
};
}

// png_read_end: the chunks after the image data, up to IEND (from offset i, the first byte behind the last IDAT's CRC)
static bool png_tail_ok(const uint8_t* s, size_t n, size_t i, const LpPngInfo& info)
{
    for (;;) {
        if (n - i < 8) return false;           // no IEND: read past the end
        const uint32_t len = be32(s + i);
        const uint8_t* type = s + i + 4;
        if (len > 0x7fffffffu || !chunk_name_ok(type)) return false;
        if (n - i - 8 < (size_t)len + 4) return false;
        const uint8_t* d = s + i + 8;
        const bool crc_ok = be32(d + len) == (uint32_t)crc32(crc32(0, type, 4), d, len);
        const bool ancillary = (type[0] & 0x20) != 0;
        i += 12 + (size_t)len;
        if (is_type(type, "IEND")) return true;                        // a CRC error or a payload here only draws a warning
        if (is_type(type, "IDAT")) { if (!crc_ok) return false; continue; } // "Too many IDATs found" is benign, a CRC error is not
        if (is_type(type, "IHDR")) return false;                       // "out of place"
        if (is_type(type, "PLTE")) { if (!crc_ok && info.color_type == 3) return false; continue; } // after IDAT: "out of place", benign
        if (!ancillary) return false;                                  // "unhandled critical chunk" (or its CRC error)
    }
}

// The common case in one sweep: every IDAT chunk whole, the whole image as the output buffer -- zlib then stays in its fast loop
// instead of being entered once per row with 8 KiB of input (libpng's pattern, reproduced below for the exact verdict on anything
// unusual; per-row calls cost ~1.5x: the window copy on every return and the call overhead). Returns 1 when the stream is a perfectly
// ordinary one -- consecutive IDAT chunks with good CRCs, a deflate stream that ends exactly with the last row and the last IDAT byte,
// filter bytes 0..4, a well-formed tail -- and `filtered` holds it; 0 = anything else: the caller takes libpng's walk, whose accept /
// reject rules (warnings against errors, where damage is tolerated) are then authoritative.
// png_read_filter_row's check of every row's filter byte
static bool png_filter_bytes_ok(const LpPngInfo& info, const LpBytes& filtered)
{
    const int bits = info.depth * lp_png_channels_in_file(info.color_type);
    size_t o = 0;
    for (int p = 0; p < (info.interlace ? 7 : 1); p++) {
        uint32_t pw, ph, x0, y0, dx, dy;
        lp_png_pass_geometry(info, p, &pw, &ph, &x0, &y0, &dx, &dy);
        if (!pw || !ph) continue;
        const size_t row = 1 + packed_row_bytes(pw, bits);
        for (uint32_t r = 0; r < ph; r++, o += row)
            if (filtered[o] > 4) return false;
    }
    return true;
}

static std::atomic<int> g_own_inflater{getenv("LILLIPUT_HIP_PNG_ZLIB") ? 0 : 1}; // A/B and tests: zlib for every file
int lp_png_set_inflater(int own) { return g_own_inflater.exchange(own ? 1 : 0); }

// ... and the same sweep through the library's own inflater (lp_inflate.cpp), which only ever says
// "ordinary, here it is" or "ask zlib". LILLIPUT_HIP_PNG_ZLIB=1 skips it (A/B, tests). Returns 1 = done, 0 = the inflater declined
// (zlib's one-sweep may still take the stream), -1 = declined for a reason the zlib sweep would find too (a chunk that does not fit the
// file, an IDAT CRC, a filter byte above 4, a broken tail): the caller goes straight to the row-wise walk instead of inflating a
// damaged file three times.
static int png_idat_own(const uint8_t* s, size_t n, const LpPngInfo& info, LpBytes& filtered)
{
    static thread_local std::vector<uint8_t> z; // the IDAT payloads, joined
    size_t total = 0, i = info.idat_off;
    for (;;) { // first walk: sizes
        if (n - i < 12) return -1;
        const uint32_t len = be32(s + i);
        if (len > 0x7fffffffu || n - i - 8 < (size_t)len + 4) return -1;
        if (!is_type(s + i + 4, "IDAT")) break;
        total += len;
        i += 12 + (size_t)len;
    }
    if (total < 6) return 0;
    if (z.size() < total + LP_INFLATE_PAD) z.resize(total + LP_INFLATE_PAD);
    size_t o = 0;
    for (i = info.idat_off;;) {
        const uint32_t len = be32(s + i);
        const uint8_t* type = s + i + 4;
        if (!is_type(type, "IDAT")) break;
        const uint8_t* d = s + i + 8;
        if (be32(d + len) != lp_crc32((uint32_t)crc32(0, type, 4), d, len)) return -1;
        memcpy(z.data() + o, d, len);
        o += len;
        i += 12 + (size_t)len;
    }
    memset(z.data() + total, 0, LP_INFLATE_PAD);
    if (lp_inflate_exact(z.data(), total, filtered.data(), filtered.size()) != 1) return 0;
    if (!png_filter_bytes_ok(info, filtered)) return -1;
    return png_tail_ok(s, n, i, info) ? 1 : -1;
}

static int png_idat_fast(const uint8_t* s, size_t n, const LpPngInfo& info, LpBytes& filtered)
{
    const size_t need = filtered.size();
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 0) != Z_OK) return 0;
    struct ZEnd { z_stream* z; ~ZEnd() { inflateEnd(z); } } zend{&zs};
    zs.next_out = filtered.data();
    zs.avail_out = (uInt)std::min<size_t>(need, 0xffffffffu);
    if (need > 0xffffffffu) return 0;
    size_t i = info.idat_off;
    bool ended = false;
    for (;;) {
        if (n - i < 12) return 0;
        const uint32_t len = be32(s + i);
        const uint8_t* type = s + i + 4;
        if (len > 0x7fffffffu || n - i - 8 < (size_t)len + 4) return 0;
        if (!is_type(type, "IDAT")) break;
        if (ended) return 0; // data behind the end of the stream: libpng's rules decide
        const uint8_t* d = s + i + 8;
        if (be32(d + len) != lp_crc32((uint32_t)crc32(0, type, 4), d, len)) return 0;
        zs.next_in = const_cast<uint8_t*>(d);
        zs.avail_in = len;
        if (len) {
            const int ret = inflate(&zs, Z_NO_FLUSH);
            if (ret == Z_STREAM_END) { if (zs.avail_in) return 0; ended = true; }
            else if (ret != Z_OK || zs.avail_in) return 0; // an error, or output that does not fit the image
        }
        i += 12 + (size_t)len;
    }
    if (!ended || zs.avail_out != 0) return 0;
    if (!png_filter_bytes_ok(info, filtered)) return 0;
    return png_tail_ok(s, n, i, info) ? 1 : 0;
}

bool lp_png_read_idat(const uint8_t* s, size_t n, const LpPngInfo& info, LpBytes& filtered)
{
    {
        const size_t need0 = lp_png_filtered_size(info);
        filtered.resize(need0); // LpBytes: nothing is cleared -- inflate writes every byte of an ordinary stream
        static const bool no_fast = getenv("LILLIPUT_HIP_PNG_ROWWISE") != nullptr; // A/B: libpng's row-by-row pattern for every file
        int own = 0;
        if (!no_fast && g_own_inflater.load(std::memory_order_relaxed) && need0 && (own = png_idat_own(s, n, info, filtered)) == 1) return true;
        if (!no_fast && need0 && own == 0 && png_idat_fast(s, n, info, filtered) == 1) return true;
    }
    IdatFeed in{s, n, info.idat_off};
    if (!in.open_chunk(true)) return false;
    const size_t need = lp_png_filtered_size(info);
    filtered.assign(need, 0);
    const int bits = info.depth * lp_png_channels_in_file(info.color_type);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 0) != Z_OK) return false; // window size from the stream header, as libpng asks for by default
    struct ZEnd { z_stream* z; ~ZEnd() { inflateEnd(z); } } zend{&zs};
    bool ended = false;
    size_t o = 0;
    for (int p = 0; p < (info.interlace ? 7 : 1); p++) {
        uint32_t pw, ph, x0, y0, dx, dy;
        lp_png_pass_geometry(info, p, &pw, &ph, &x0, &y0, &dx, &dy);
        if (!pw || !ph) continue;
        const size_t row = 1 + packed_row_bytes(pw, bits);
        for (uint32_t r = 0; r < ph; r++, o += row) {
            if (ended) return false;                         // rows are still owed after the stream said it was over
            zs.next_out = filtered.data() + o;
            zs.avail_out = (uInt)row;
            do {
                if (zs.avail_in == 0 && !in.fill(zs)) return false;
                const int ret = inflate(&zs, Z_NO_FLUSH);
                if (ret == Z_STREAM_END) { ended = true; break; }
                if (ret != Z_OK) return false;               // zlib's message becomes a png_chunk_error while a row is being read
            } while (zs.avail_out > 0);
            if (zs.avail_out > 0) return false;              // "Not enough image data"
            if (filtered[o] > 4) return false;               // png_read_filter_row: "bad adaptive filter value"
        }
    }
    if (!ended) { // png_read_finish_IDAT: swallow what is left of the stream; damage found now only draws a warning ...
        size_t produced = 0;
        uint8_t tmp[1024];
        do {
            if (zs.avail_in == 0 && !in.fill(zs)) return false; // ... except running out of IDAT chunks: "Not enough image data"
            zs.next_out = tmp;
            zs.avail_out = sizeof(tmp);
            const int ret = inflate(&zs, Z_NO_FLUSH);
            produced += sizeof(tmp) - zs.avail_out;
            if (ret != Z_OK) break;                          // the end of the stream, or a benign error
        } while (produced > 0);
    }
    if (!in.skip_rest()) return false;
    return png_tail_ok(s, n, in.i, info);
}
