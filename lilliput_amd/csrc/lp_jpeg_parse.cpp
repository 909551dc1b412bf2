// lp_jpeg_parse.cpp -- host-side marker parsing for the JPEG decode path.
//
// Stands where cv::JpegDecoder::readHeader -> jpeg_read_header sits in the reference
// (/root/reference/opencv.cpp:126-140 opencv_decoder_read_header): SOI, APPn (JFIF / Adobe / EXIF
// orientation, as surfaced by opencv_decoder_get_orientation, opencv.cpp:160-164), DQT, SOF0/1, DHT,
// DRI, SOS. Only what the device kernels need is kept; the entropy-coded segment itself is never
// walked byte-by-byte on the host unless the file does not end in EOI.
#include "lp_jpeg_parse.h"

#include <string.h>

#include <memory>

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

static int exif_orientation(const uint8_t* p, size_t n)
{
    if (n < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 0;
    const uint8_t* t = p + 6;
    size_t tn = n - 6;
    bool le;
    if (t[0] == 'I' && t[1] == 'I') le = true;
    else if (t[0] == 'M' && t[1] == 'M') le = false;
    else return 0;
    auto r16 = [&](size_t o) -> unsigned { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
    auto r32 = [&](size_t o) -> uint32_t {
        return le ? ((uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) | ((uint32_t)t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24))
                  : (((uint32_t)t[o] << 24) | ((uint32_t)t[o + 1] << 16) | ((uint32_t)t[o + 2] << 8) | (uint32_t)t[o + 3]);
    };
    if (r16(2) != 42) return 0;
    size_t ifd = r32(4);
    if (ifd + 2 > tn) return 0;
    unsigned cnt = r16(ifd);
    for (unsigned i = 0; i < cnt; i++) {
        size_t e = ifd + 2 + 12 * (size_t)i;
        if (e + 12 > tn) return 0;
        if (r16(e) == 0x0112) return (int)r16(e + 8);
    }
    return 0;
}

void lp_build_huff_slot(LpHuffSet* hs, int slot, const uint8_t bits[17], const uint8_t* vals)
{
    // every prefix starts out as "no code here" (canonical search answers libjpeg's length-17 rule for it)
    for (int i = 0; i < LP_LUT_SIZE; i++) hs->lut[slot][i] = LP_E_NO_SLICE;
    if (slot == 0) hs->lut2_used = 0; // slots are built in order 0..3 and share the second-level pool
    // canonical code assignment (T.81 Annex C). Short codes fill their span of the first level; a long code gets its prefix's
    // slice of the second level (allocated on first use) and fills its span of the LP_LUT2_BITS bits after the prefix.
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        int valptr = k, mincode = code;
        for (int i = 0; i < bits[l]; i++, k++, code++) {
            if (l <= LP_LUT_BITS) {
                int first = code << (LP_LUT_BITS - l), n = 1 << (LP_LUT_BITS - l);
                for (int j = 0; j < n && first + j < LP_LUT_SIZE; j++) hs->lut[slot][first + j] = lp_lut_entry(slot, l, vals[k]);
            } else {
                const uint32_t left = (uint32_t)code << (16 - l);           // the code, left aligned in 16 bits
                if (left >= 0x10000u) continue;                              // over-subscribed table (rejected elsewhere); never index out of range
                const uint32_t prefix = left >> (16 - LP_LUT_BITS);
                uint16_t& e1 = hs->lut[slot][prefix];
                if (LP_E_BITS(e1) != 0) continue;                               // a shorter code owns the prefix: not a prefix code, leave it to libjpeg's order
                if (LP_E_SLICE(e1) == 0xffu) {
                    if (hs->lut2_used >= LP_LUT2_SUBS) continue;             // pool exhausted: canonical search serves this prefix
                    e1 = (uint16_t)(hs->lut2_used << 5);
                    memset(hs->lut2 + ((size_t)hs->lut2_used << LP_LUT2_BITS), 0, sizeof(uint16_t) << LP_LUT2_BITS);
                    hs->lut2_used++;
                }
                const uint32_t sub = LP_E_SLICE(e1), first = left & ((1u << LP_LUT2_BITS) - 1u), n = 1u << (16 - l);
                for (uint32_t j = 0; j < n && first + j < (1u << LP_LUT2_BITS); j++) hs->lut2[(sub << LP_LUT2_BITS) | (first + j)] = lp_lut_entry(slot, l, vals[k]);
            }
        }
        hs->maxcode[slot][l] = bits[l] ? code - 1 : -1;
        hs->valoff[slot][l] = valptr - mincode;
        code <<= 1;
    }
    hs->maxcode[slot][0] = -1;
    hs->maxcode[slot][17] = 0x7fffffff;
    hs->valoff[slot][0] = 0;
    memset(hs->vals[slot], 0, 256);
    memcpy(hs->vals[slot], vals, (size_t)(k > 256 ? 256 : k));
}

// The counting passes' multi-symbol entries (LpHuffSet::lutm, lp_types.h). From the window's bit i on, symbols are taken while the
// whole CODE of the next one lies inside the LP_LUT_BITS window (its extra bits may reach beyond: the counting passes skip them
// unread) and the group stays a run the lane logic can apply in one step: one block (an EOB closes the group; so does a group whose
// advance would reach 64 with certainty), at most 31 bits (LP_E_BITS is five bits wide; the bit reader's ring geometry assumes as much).
void lp_build_huff_multi(LpHuffSet* hs, const int ac_of_dc[2])
{
    for (int slot = 0; slot < 4; slot++) {
        const int ac = slot >= 2 ? slot : ac_of_dc[slot];
        for (uint32_t i = 0; i < LP_LUT_SIZE; i++) {
            const uint16_t e1 = hs->lut[slot][i];
            hs->lutm[slot][i] = e1;
            if (LP_E_BITS(e1) == 0 || ac < 2 || ac > 3) continue; // the prefix of a long code / no code; a DC slot without one AC slot behind it
            if (e1 & 0x8000u) continue;                             // the first symbol ends the block
            uint32_t nbits = LP_E_BITS(e1), adv = (LP_E_RUNX(e1) & 15u) + 1u, nsym = 1;
            bool eob = false;
            while (nbits < LP_LUT_BITS) {
                const uint32_t known = LP_LUT_BITS - nbits;     // window bits left for the next code
                const uint16_t e = hs->lut[ac][(i << nbits) & (LP_LUT_SIZE - 1u)];
                const uint32_t n = LP_E_BITS(e), sz = LP_E_SIZE(e);
                if (n == 0 || n - sz > known) break;            // a long code, or a code that reaches past the window: not decided by these bits
                if (nbits + n > 31u) break;
                if (e & 0x8000u) { eob = true; nbits += n; nsym++; break; }
                const uint32_t a = (LP_E_RUNX(e) & 15u) + 1u;   // ZRL: run 15, size 0 -> 16 coefficients
                if (adv + a > 63u) break;                       // (a group of 64 or more can never be applied: z + advance <= 64 with z >= 0 ... keep the field six bits)
                adv += a; nbits += n; nsym++;
            }
            if (nsym < 2) continue;
            hs->lutm[slot][i] = (uint16_t)(nbits | ((eob ? adv : adv - 1u) << 9) | (eob ? 0x8000u : 0u));
        }
    }
}

// T.81 Annex K.3 typical Huffman tables (jstdhuff.c): what the encoder writes, and what libjpeg-turbo falls back to for a
// table id 0/1 that no DHT defined (jdhuff.c jinit_huff_decoder -> std_huff_tables, "Motion JPEG frames typically do not
// include the Huffman tables"). Order: DC luma, AC luma, DC chroma, AC chroma.
const uint8_t lp_std_huff_bits[4][17] = {{0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
                                     {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d},
                                     {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
                                     {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
const uint8_t lp_std_huff_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t lp_std_huff_ac_luma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t lp_std_huff_ac_chroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};


// jdhuff.c jpeg_make_d_derived_tbl (run by libjpeg for the tables a scan uses): the code space must not overflow
// (the all-ones code of any length is reserved) and DC symbols are categories 0..15 -- else JERR_BAD_HUFF_TABLE.
static bool huff_table_valid(const uint8_t bits[17], const uint8_t* vals, bool is_dc)
{
    long code = 0;
    int tot = 0;
    for (int l = 1; l <= 16; l++) {
        code += bits[l];
        tot += bits[l];
        if (code >= (1L << l)) return false;
        code <<= 1;
    }
    if (is_dc)
        for (int q = 0; q < tot; q++)
            if (vals[q] > 15) return false;
    return true;
}

int lp_jpeg_parse(const uint8_t* d, size_t n, LpJpegHeader* out)
{
    const int rc = lp_jpeg_parse_opts(d, n, out, false);
    // entropy-coded data that runs to the end of the buffer with no marker behind it: whether cv::JpegDecoder still returns the image
    // depends on where libjpeg's read-ahead falls (lp_jbits.h) -- the serial route, which models it, decides
    if (rc == LP_PARSE_OK && !out->scan_path && out->open_end) return lp_jpeg_parse_opts(d, n, out, true);
    return rc;
}

int lp_jpeg_parse_opts(const uint8_t* d, size_t n, LpJpegHeader* out, bool force_scans)
{
    *out = LpJpegHeader();
    memset(&out->j, 0, sizeof(out->j));
    memset(&out->huff, 0, sizeof(out->huff));
    LpJpeg& j = out->j;
    j.orientation = 1;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return LP_PARSE_NOT_JPEG;
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    bool latched[4] = {false, false, false, false}; // progressive: quantisation table fixed at the component's first scan
    uint16_t latched_qt[4][64];
    uint8_t hbits[2][4][17];
    uint8_t hvals[2][4][256];
    bool h_ok[2][4] = {{false, false, false, false}, {false, false, false, false}};
    int cid[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0}, td[4] = {0, 0, 0, 0}, ta[4] = {0, 0, 0, 0};
    bool have_sof = false, saw_jfif = false, saw_adobe = false, sof_unsupported = false, sof_bad_sampling = false, progressive = false;
    struct RawScan { unsigned ns; int comp[4], td[4], ta[4]; unsigned Ss, Se, Ah, Al, dri; bool sequential; uint8_t bits[8][17], vals[8][256]; size_t ecs_off, ecs_len; LpArithScan ar; };
    bool arith = false;     // SOF9 / SOF10: QM-coded scans (jdarith.c), always taken scan by scan
    uint8_t dac_L[16], dac_U[16], dac_K[16]; // DAC conditioning as jpeg_create_decompress / start of every datastream leaves it: L = 0, U = 1, Kx = 5
    for (int t = 0; t < 16; t++) { dac_L[t] = 0; dac_U[t] = 1; dac_K[t] = 5; }
    bool seq_scans = false; // a sequential file that goes scan by scan (see LpProgScan::sequential); decided at its first SOS
    std::vector<RawScan> raw_scans;
    unsigned sof_nc = 0;
    int scan_comp[4] = {-1, -1, -1, -1};
    int adobe_tf = 0;
    size_t i = 2;
    size_t ecs = 0;
    bool ran_off_end = false;
    // The walk accepts and rejects what jdmarker.c read_markers does (the reference decodes through it): garbage between
    // segments is skipped, an unknown marker code, a repeated SOI/SOF, or a table segment whose length does not add up is
    // an error; running off the end is "no image".
    for (;;) {
        // next_marker
        unsigned m;
        for (;;) {
            while (i < n && d[i] != 0xFF) i++;
            while (i < n && d[i] == 0xFF) i++;
            if (i >= n) { if (!raw_scans.empty()) { m = 0xD9; ran_off_end = true; break; } return LP_PARSE_TRUNCATED; }
            m = d[i++];
            if (m != 0) break; // FF 00: stuffed data, keep looking
        }
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9 && !raw_scans.empty()) { out->saw_eoi = ran_off_end ? 0 : 1; break; } // end of a progressive file (or of the buffer: no EOI)
        if (m == 0xD8 || m == 0xD9) return LP_PARSE_NOT_JPEG; // JERR_SOI_DUPLICATE / EOI before any scan
        const bool is_sof = m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC;
        const bool known = (m >= 0xC0 && m <= 0xCF && m != 0xC8) || m == 0xDA || m == 0xDB || m == 0xDC || m == 0xDD || (m >= 0xE0 && m <= 0xEF) || m == 0xFE;
        if (!known) return LP_PARSE_NOT_JPEG;                  // JERR_UNKNOWN_MARKER
        if (m == 0xC5 || m == 0xC6 || m == 0xC7 || m == 0xCD || m == 0xCE || m == 0xCF) return LP_PARSE_NOT_JPEG; // JERR_SOF_UNSUPPORTED
        if (i + 2 > n) return LP_PARSE_TRUNCATED;
        size_t L = be16(d + i);
        if (L < 2) {
            // skip_variable / get_interesting_appn shrug off a bogus length word; the table and frame readers do not
            if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE || m == 0xDC) { i += 2; continue; }
            return LP_PARSE_NOT_JPEG;                          // JERR_BAD_LENGTH
        }
        if (i + L > n) return LP_PARSE_TRUNCATED;
        const uint8_t* p = d + i + 2;
        size_t pl = L - 2;
        const size_t seg_end = i + L;
        if (m == 0xDB) { // get_dqt
            size_t k = 0;
            while (k < pl) {
                unsigned pq = p[k] >> 4, t = p[k] & 15;
                k++;
                if (t > 3 || k + (pq ? 128 : 64) > pl) return LP_PARSE_NOT_JPEG; // JERR_DQT_INDEX / JERR_BAD_LENGTH
                for (int z = 0; z < 64; z++) {
                    unsigned v;
                    if (pq) { v = be16(p + k); k += 2; } else v = p[k++];
                    qt[t][kZigzag[z]] = (uint16_t)v;
                }
                qt_ok[t] = true;
            }
        } else if (m == 0xC4) { // get_dht
            size_t k = 0;
            while (pl - k > 16) {
                unsigned tc = p[k] >> 4, th = p[k] & 15;
                k++;
                unsigned tot = 0;
                uint8_t bits[17];
                bits[0] = 0;
                for (int b = 1; b <= 16; b++) { bits[b] = p[k++]; tot += bits[b]; }
                if (tot > 256 || tot > pl - k) return LP_PARSE_NOT_JPEG;  // JERR_BAD_HUFF_TABLE
                if (tc > 1 || th > 3) return LP_PARSE_NOT_JPEG;           // JERR_DHT_INDEX
                memcpy(hbits[tc][th], bits, 17);
                memset(hvals[tc][th], 0, 256);
                memcpy(hvals[tc][th], p + k, tot);
                k += tot;
                h_ok[tc][th] = true;
            }
            if (k != pl) return LP_PARSE_NOT_JPEG;                        // JERR_BAD_LENGTH
        } else if (is_sof) { // get_sof
            if (have_sof) return LP_PARSE_NOT_JPEG;                       // JERR_SOF_DUPLICATE
            if (pl < 6) return LP_PARSE_NOT_JPEG;
            const unsigned prec = p[0], nc = p[5];
            j.height = be16(p + 1);
            j.width = be16(p + 3);
            if (j.height == 0 || j.width == 0 || nc == 0) return LP_PARSE_NOT_JPEG; // JERR_EMPTY_IMAGE
            if (pl != 6 + 3u * nc) return LP_PARSE_NOT_JPEG;             // JERR_BAD_LENGTH
            progressive = m == 0xC2 || m == 0xCA;
            arith = m == 0xC9 || m == 0xCA;
            if (m != 0xC0 && m != 0xC1 && m != 0xC2 && !arith) { sof_unsupported = true; } // lossless (SOF3, SOF11): judged at SOS
            if (prec != 8 || (nc != 1 && nc != 3 && nc != 4)) sof_unsupported = true; // 12-bit, two-component
            sof_nc = nc;
            for (unsigned c = 0; c < nc; c++) {
                const unsigned hs = p[7 + 3 * c] >> 4, vs = p[7 + 3 * c] & 15;
                if (hs < 1 || hs > 4 || vs < 1 || vs > 4) sof_bad_sampling = true; // JERR_BAD_SAMPLING, raised at the first SOS
                if (c < 4) {
                    cid[c] = p[6 + 3 * c];
                    j.hs[c] = (uint8_t)hs;
                    j.vs[c] = (uint8_t)vs;
                    tq[c] = p[8 + 3 * c];
                }
            }
            j.ncomp = (uint8_t)(nc <= 4 ? nc : 4);
            have_sof = true;
        } else if (m == 0xDD) { // get_dri
            if (L != 4) return LP_PARSE_NOT_JPEG;
            j.dri = be16(p);
        } else if (m == 0xCC) { // get_dac
            if (pl & 1) return LP_PARSE_NOT_JPEG;
            for (size_t k = 0; k < pl; k += 2) {
                if (p[k] >= 32) return LP_PARSE_NOT_JPEG;                                   // JERR_DAC_INDEX
                if (p[k] < 16 && (p[k + 1] & 15) > (p[k + 1] >> 4)) return LP_PARSE_NOT_JPEG; // JERR_DAC_VALUE
                if (p[k] < 16) { dac_L[p[k]] = p[k + 1] & 15; dac_U[p[k]] = p[k + 1] >> 4; }
                else dac_K[p[k] - 16] = p[k + 1];
            }
        } else if (m == 0xE0) {
            if (pl >= 14 && memcmp(p, "JFIF\0", 5) == 0) saw_jfif = true;
        } else if (m == 0xE1) {
            int o = exif_orientation(p, pl);
            if (o >= 1 && o <= 8 && j.orientation == 1) j.orientation = (uint8_t)o;
        } else if (m == 0xEE) {
            if (pl >= 12 && memcmp(p, "Adobe", 5) == 0) { saw_adobe = true; adobe_tf = p[11]; }
        } else if (m == 0xDA) { // get_sos, then the checks of jdinput.c initial_setup / jdhuff.c on the first scan
            if (!have_sof || pl < 1) return LP_PARSE_NOT_JPEG;            // JERR_SOS_NO_SOF
            unsigned ns = p[0];
            if (L != ns * 2 + 6 || ns < 1 || ns > 4) return LP_PARSE_NOT_JPEG; // JERR_BAD_LENGTH
            if (j.height > 65500 || j.width > 65500 || sof_bad_sampling) return LP_PARSE_NOT_JPEG;
            if (sof_unsupported) return LP_PARSE_UNSUPPORTED;
            int cur[4] = {-1, -1, -1, -1};
            for (unsigned s = 0; s < ns; s++) {
                int cs = p[1 + 2 * s], t = p[2 + 2 * s], c;
                // libjpeg-turbo's matching rule: the first frame component with this id whose slot is still free
                for (c = 0; c < j.ncomp; c++) if (cid[c] == cs && cur[c] < 0) break;
                if (c == j.ncomp) return LP_PARSE_NOT_JPEG;               // JERR_BAD_COMPONENT_ID
                cur[s] = c;
                for (unsigned q = 0; q < s; q++) if (cur[q] == c) return LP_PARSE_NOT_JPEG;
                // jpeg_make_d_derived_tbl rejects an index above 3 -- of the tables a scan actually builds: both in a sequential
                // scan, only the DC or the AC one in a progressive scan (checked below)
                if (!progressive && !arith && ((t >> 4) > 3 || (t & 15) > 3)) return LP_PARSE_NOT_JPEG; // JERR_NO_HUFF_TABLE
                if (s < 4) { scan_comp[s] = c; td[c] = t >> 4; ta[c] = t & 15; }
            }
            if (raw_scans.empty() && !ecs) out->one_pass = !progressive && ns == (unsigned)j.ncomp; // jdinput.c initial_setup: has_multiple_scans
            if (!progressive && raw_scans.empty()) {
                // A sequential file the baseline kernels do not take as it is -- fewer components in the first scan than in the frame
                // (more scans follow), components in another order, table numbers 2 / 3 -- is decoded scan by scan like a progressive one.
                seq_scans = ns != j.ncomp || j.ncomp == 4; // four components (CMYK / YCCK): up to ten blocks per MCU, always scan by scan
                // sampling factors the baseline kernels do not take (anything but luma 1x1 / 2x1 / 1x2 / 2x2 over 1x1 chroma): same walk
                if (j.ncomp == 3 && (j.hs[1] != 1 || j.vs[1] != 1 || j.hs[2] != 1 || j.vs[2] != 1 || j.hs[0] > 2 || j.vs[0] > 2)) seq_scans = true;
                for (unsigned s = 0; s < ns; s++) seq_scans = seq_scans || cur[s] != (int)s || (p[2 + 2 * s] >> 4) > 1 || (p[2 + 2 * s] & 15) > 1;
                seq_scans = seq_scans || force_scans || arith;
            }
            if (ns > 1) { // jdinput.c per_scan_setup: an interleaved MCU holds at most D_MAX_BLOCKS_IN_MCU = 10 blocks
                unsigned blocks = 0;
                for (unsigned s = 0; s < ns; s++) blocks += j.hs[cur[s]] * j.vs[cur[s]];
                if (blocks > 10) return LP_PARSE_NOT_JPEG;               // JERR_BAD_MCU_SIZE
            }
            if (progressive || seq_scans) { // jdphuff.c start_pass_phuff_decoder / jdhuff.c start_pass_huff_decoder: one of up to LP_MAX_SCANS scans
                RawScan rs;
                memset(&rs, 0, sizeof(rs));
                rs.ns = ns;
                rs.sequential = !progressive;
                for (unsigned s = 0; s < ns; s++) { rs.comp[s] = cur[s]; rs.td[s] = p[2 + 2 * s] >> 4; rs.ta[s] = p[2 + 2 * s] & 15; }
                rs.Ss = p[1 + 2 * ns]; rs.Se = p[2 + 2 * ns]; rs.Ah = p[3 + 2 * ns] >> 4; rs.Al = p[3 + 2 * ns] & 15;
                bool bad = false;
                if (rs.Ss == 0) { if (rs.Se != 0) bad = true; }
                else { if (rs.Se < rs.Ss || rs.Se > 63) bad = true; if (ns != 1) bad = true; }
                if (rs.Ah != 0 && rs.Ah - 1 != rs.Al) bad = true;
                if (rs.Al > 13) bad = true;
                if (bad && progressive) return LP_PARSE_NOT_JPEG;         // JERR_BAD_PROGRESSION (a sequential scan only warns: JWRN_NOT_SEQUENTIAL)
                for (unsigned s = 0; s < ns; s++) { // jdinput.c latch_quant_tables: a component keeps the table current at its first scan
                    const int c = cur[s];
                    if (latched[c]) continue;
                    if (tq[c] > 3 || !qt_ok[tq[c]]) return LP_PARSE_NOT_JPEG; // JERR_NO_QUANT_TABLE
                    memcpy(latched_qt[c], qt[tq[c]], sizeof(latched_qt[c]));
                    latched[c] = true;
                }
                rs.dri = j.dri;
                const bool dc_scan = rs.Ss == 0;
                if (arith) { // jdarith.c start_pass: any table number 0..15 is fine; the conditioning is what the DAC segments so far left
                    for (unsigned s = 0; s < ns; s++) {
                        rs.ar.dc_tbl[s] = (uint8_t)rs.td[s]; rs.ar.ac_tbl[s] = (uint8_t)rs.ta[s];
                        rs.ar.dc_L[s] = dac_L[rs.td[s] & 15]; rs.ar.dc_U[s] = dac_U[rs.td[s] & 15]; rs.ar.ac_K[s] = dac_K[rs.ta[s] & 15];
                    }
                } else if (rs.sequential) { // both tables of every component, Annex-K ones for undefined numbers 0 / 1 (jinit_huff_decoder -> std_huff_tables)
                    for (unsigned s = 0; s < ns; s++)
                        for (int cls = 0; cls < 2; cls++) {
                            const int id = cls ? rs.ta[s] : rs.td[s], slot = cls ? 4 + (int)s : (int)s;
                            if (h_ok[cls][id]) { memcpy(rs.bits[slot], hbits[cls][id], 17); memcpy(rs.vals[slot], hvals[cls][id], 256); }
                            else if (id < 2) {
                                memcpy(rs.bits[slot], lp_std_huff_bits[2 * id + cls], 17);
                                memset(rs.vals[slot], 0, 256);
                                if (cls) memcpy(rs.vals[slot], id ? lp_std_huff_ac_chroma : lp_std_huff_ac_luma, 162);
                                else memcpy(rs.vals[slot], lp_std_huff_dc_vals, 12);
                            } else
                                return LP_PARSE_NOT_JPEG;                 // JERR_NO_HUFF_TABLE
                            if (!huff_table_valid(rs.bits[slot], rs.vals[slot], cls == 0)) return LP_PARSE_NOT_JPEG;
                        }
                } else if (!(dc_scan && rs.Ah != 0))                      // a DC refinement scan reads raw bits only
                    for (unsigned s = 0; s < ns; s++) {
                        const int cls = dc_scan ? 0 : 1, id = dc_scan ? rs.td[s] : rs.ta[s];
                        if (id > 3) return LP_PARSE_NOT_JPEG;             // JERR_NO_HUFF_TABLE
                        // no Annex-K fallback here: std_huff_tables() runs in jinit_huff_decoder only, a progressive file defines what it uses
                        if (!h_ok[cls][id]) return LP_PARSE_NOT_JPEG;     // JERR_NO_HUFF_TABLE
                        const uint8_t* tb = hbits[cls][id];
                        const uint8_t* tv = hvals[cls][id];
                        memcpy(rs.bits[s], tb, 17);
                        memcpy(rs.vals[s], tv, 256);
                        if (!huff_table_valid(rs.bits[s], rs.vals[s], cls == 0)) return LP_PARSE_NOT_JPEG;
                    }
                rs.ecs_off = seg_end;
                size_t q = seg_end;
                for (; q + 1 < n; q++) {
                    // (memchr: the entropy-coded bytes are nine tenths of a progressive file, and a byte loop over them was 0.3 ms per
                    // 1024 x 1024 file -- 20 ms of a 64-image chunk's ingest)
                    const void* ff = memchr(d + q, 0xFF, n - 1 - q);
                    if (!ff) { q = n; break; }
                    q = (size_t)((const uint8_t*)ff - d);
                    const unsigned c = d[q + 1];
                    if (c == 0 || c == 0xFF || (c >= 0xD0 && c <= 0xD7)) continue;
                    // Scans with a restart interval: a byte pair that only looks like a marker (code below 0xC0: nothing libjpeg knows) does
                    // not end the data -- the decoder runs dry there, and jpeg_resync_to_restart skips the pair at the next interval
                    // boundary and reads on (lp_jbits.h). (A scan that ENDS with such a pair pending is refused by the scan decoders when
                    // libjpeg reads the file to its end, as read_markers refuses it: JERR_UNKNOWN_MARKER.)
                    if (rs.dri && c < 0xC0) continue;
                    break;
                }
                if (q + 1 >= n) q = n;                                    // ran off the end: the scan takes what is there
                rs.ecs_len = q - seg_end;
                if (raw_scans.size() >= LP_MAX_SCANS) return LP_PARSE_UNSUPPORTED;
                raw_scans.push_back(rs);
                if (!ecs) ecs = seg_end;
                i = q;
                if (q >= n || (q + 1 < n && d[q + 1] == 0xD9)) { out->saw_eoi = q < n; break; }
                if (rs.sequential && raw_scans.size() == 1 && ns == j.ncomp) break; // a one-scan file: libjpeg reads nothing past the scan
                continue;
            }
            ecs = seg_end; // one interleaved scan, components in frame order, table numbers 0 / 1: the baseline kernels' case
            break;
        }
        i = seg_end;
    }
    if (!have_sof || !ecs || j.width == 0 || j.height == 0) return LP_PARSE_NOT_JPEG;
    (void)sof_nc;
    // std_huff_tables(): ids 0 and 1 get the Annex-K luminance / chrominance tables when no DHT defined them
    for (int t = 0; t < 2; t++) {
        if (!h_ok[0][t]) {
            memcpy(hbits[0][t], lp_std_huff_bits[2 * t], 17);
            memset(hvals[0][t], 0, 256);
            memcpy(hvals[0][t], lp_std_huff_dc_vals, 12);
            h_ok[0][t] = true;
        }
        if (!h_ok[1][t]) {
            memcpy(hbits[1][t], lp_std_huff_bits[2 * t + 1], 17);
            memset(hvals[1][t], 0, 256);
            memcpy(hvals[1][t], t ? lp_std_huff_ac_chroma : lp_std_huff_ac_luma, 162);
            h_ok[1][t] = true;
        }
    }
    if (j.ncomp == 1) { j.hs[0] = j.vs[0] = 1; }
    j.hmax = j.vmax = 1;
    for (int c = 0; c < j.ncomp; c++) {
        if (j.hs[c] < 1 || j.hs[c] > 4 || j.vs[c] < 1 || j.vs[c] > 4) return LP_PARSE_NOT_JPEG;
        if (j.hs[c] > j.hmax) j.hmax = j.hs[c];
        if (j.vs[c] > j.vmax) j.vmax = j.vs[c];
        if ((progressive || seq_scans) && !latched[c]) { memset(latched_qt[c], 0, sizeof(latched_qt[c])); continue; } // in no scan at all: stays zero (libjpeg: never dequantised)
        if (progressive || seq_scans) continue;                                                    // tables were checked scan by scan
        if (tq[c] > 3 || !qt_ok[tq[c]]) return LP_PARSE_NOT_JPEG;                                  // JERR_NO_QUANT_TABLE
        if (!h_ok[0][td[c]] || !h_ok[1][ta[c]]) return LP_PARSE_NOT_JPEG;                         // JERR_NO_HUFF_TABLE
        if (!huff_table_valid(hbits[0][td[c]], hvals[0][td[c]], true) || !huff_table_valid(hbits[1][ta[c]], hvals[1][ta[c]], false))
            return LP_PARSE_NOT_JPEG;
    }
    for (int c = 0; c < j.ncomp; c++) // jdsample.c jinit_upsampler refuses fractional ratios when decoding starts (JERR_FRACT_SAMPLE_NOTIMPL)
        if (j.hmax % j.hs[c] || j.vmax % j.vs[c]) return LP_PARSE_UNSUPPORTED;
    j.generic_sampling = (j.ncomp == 3 && (j.hs[1] != 1 || j.vs[1] != 1 || j.hs[2] != 1 || j.vs[2] != 1 || j.hs[0] > 2 || j.vs[0] > 2)) || j.ncomp == 4;
    j.mcus_x = (j.width + 8 * j.hmax - 1) / (8 * j.hmax);
    j.mcus_y = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
    unsigned bpm = 0;
    for (int c = 0; c < j.ncomp; c++) {
        for (int v = 0; v < j.vs[c]; v++)
            for (int h = 0; h < j.hs[c]; h++) {
                if (bpm >= LP_MAX_BPM && !(progressive || seq_scans)) return LP_PARSE_UNSUPPORTED;
                if (bpm < 8) { // the per-block tables only serve the baseline kernels (at most LP_MAX_BPM blocks)
                    j.blk_comp[bpm] = (uint8_t)c;
                    j.blk_h[bpm] = (uint8_t)h;
                    j.blk_v[bpm] = (uint8_t)v;
                }
                bpm++;
            }
        j.blk_first[c] = (uint8_t)(bpm - j.hs[c] * j.vs[c]);
        j.bw[c] = j.mcus_x * j.hs[c];
        j.bh[c] = j.mcus_y * j.vs[c];
        j.plane_stride[c] = j.bw[c] * 8;
        j.dc_tbl[c] = (uint8_t)td[c];
        j.ac_tbl[c] = (uint8_t)(2 + ta[c]);
        if (progressive || seq_scans) memcpy(j.qt[c], latched_qt[c], sizeof(j.qt[c]));
        else memcpy(j.qt[c], qt[tq[c]], sizeof(j.qt[c]));
    }
    j.bpm = (uint8_t)bpm;
    j.blkpack = 0;
    for (unsigned b = 0; b < bpm && b < 8; b++) {
        const unsigned c = j.blk_comp[b];
        j.blkpack |= (uint64_t)(c | ((unsigned)td[c] << 2) | ((unsigned)ta[c] << 3)) << (4 * b);
    }
    j.total_blocks = j.mcus_x * j.mcus_y * bpm;
    // libjpeg's colour space guess (jdapimin.c default_decompress_parms)
    if (j.ncomp == 1) j.colorspace = 1;
    else if (j.ncomp == 4) j.colorspace = saw_adobe ? (adobe_tf == 0 ? 4 : 5) : 4; // Adobe transform 2 (or an unknown one): YCCK; no marker: CMYK
    else if (saw_jfif) j.colorspace = 2;
    else if (saw_adobe) j.colorspace = adobe_tf == 0 ? 3 : 2;
    else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') j.colorspace = 3;
    else j.colorspace = 2;
    if (progressive || seq_scans) {
        out->scan_path = true;
        for (const RawScan& rs : raw_scans) {
            LpProgScanHost hs;
            memset(&hs.s, 0, sizeof(hs.s));
            memset(&hs.tables, 0, sizeof(hs.tables));
            hs.s.ns = rs.ns;
            for (unsigned s = 0; s < rs.ns; s++) hs.s.comp[s] = (uint8_t)rs.comp[s];
            hs.s.Ss = (uint8_t)rs.Ss; hs.s.Se = (uint8_t)rs.Se; hs.s.Ah = (uint8_t)rs.Ah; hs.s.Al = (uint8_t)rs.Al;
            hs.s.sequential = rs.sequential ? 1 : 0;
            hs.s.dri = rs.dri;
            if (rs.ns == 1) { // non-interleaved: the component's own blocks, not the MCU padding
                const int c = rs.comp[0];
                hs.s.mcux = (j.width * j.hs[c] + j.hmax * 8 - 1) / (j.hmax * 8);
                hs.s.mcuy = (j.height * j.vs[c] + j.vmax * 8 - 1) / (j.vmax * 8);
            } else { hs.s.mcux = j.mcus_x; hs.s.mcuy = j.mcus_y; }
            for (unsigned s = 0; s < rs.ns; s++) {
                const int c = rs.comp[s];
                uint32_t base = 0;
                for (int q = 0; q < c; q++) base += j.bw[q] * j.bh[q];
                hs.s.cblk[s] = base;
                hs.s.bw[s] = j.bw[c];
                hs.s.hs[s] = rs.ns == 1 ? 1 : j.hs[c];
                hs.s.vs[s] = rs.ns == 1 ? 1 : j.vs[c];
            }
            for (unsigned slot = 0; slot < 8 && !arith; slot++) { // canonical tables + an 8-bit first-level lookup (Huffman scans)
                const unsigned s = slot;
                if ((slot & 3u) >= rs.ns) continue;
                if (rs.sequential ? false : (slot >= 4 || (rs.Ss == 0 && rs.Ah != 0))) continue; // progressive: one table per component, none in a DC refinement
                int code = 0, k = 0;
                for (int l = 1; l <= 16; l++) {
                    const int mincode = code, valptr = k;
                    for (int q = 0; q < rs.bits[s][l]; q++, k++, code++)
                        if (l <= 8)
                            for (int f = 0; f < (1 << (8 - l)); f++) hs.tables.lut8[s][(code << (8 - l)) + f] = (uint16_t)((l << 8) | rs.vals[s][k]);
                    hs.tables.maxcode[s][l] = rs.bits[s][l] ? code - 1 : -1;
                    hs.tables.valoff[s][l] = valptr - mincode;
                    code <<= 1;
                }
                hs.tables.maxcode[s][0] = -1;
                hs.tables.maxcode[s][17] = 0x7fffffff;
                memcpy(hs.tables.vals[s], rs.vals[s], 256);
            }
            hs.ecs_off = rs.ecs_off;
            hs.ecs_len = rs.ecs_len;
            hs.arith = arith;
            hs.ar = rs.ar;
            out->scans.push_back(hs);
        }
        if (out->scans.empty()) return LP_PARSE_NOT_JPEG;
        if (progressive) {
            // jdphuff.c / jdarith.c start_pass: coef_bits[component][k] = Al of the last scan that carried coefficient k (-1: none did);
            // jdcoefct.c smoothing_ok (libjpeg-turbo 3.x: DC + the first nine AC coefficients): every component has its quantisation
            // table latched, none of those ten quantisers is zero, its DC is at least partly known -- and some AC precision is missing
            int cb[4][10];
            for (auto& row : cb) for (int& v : row) v = -1;
            for (const RawScan& rs : raw_scans)
                for (unsigned s = 0; s < rs.ns; s++)
                    for (int k = (int)rs.Ss; k <= (int)rs.Se && k < 10; k++) cb[rs.comp[s]][k] = (int)rs.Al;
            static const int first10[10] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24}; // natural positions of zigzag 0..9
            bool ok = true, useful = false;
            for (int c = 0; c < j.ncomp && c < 4 && ok; c++) {
                if (!latched[c] || cb[c][0] < 0) { ok = false; break; }
                for (int q = 0; q < 10; q++) ok = ok && latched_qt[c][first10[q]] != 0;
                for (int k = 1; k < 10; k++) useful = useful || cb[c][k] != 0;
            }
            out->ref_smooths = ok && useful;
            for (int c = 0; c < 4; c++) for (int k = 0; k < 10; k++) out->coef_bits[c][k] = (int8_t)cb[c][k];
        }
        out->arith = arith;
        out->decode_fails = !out->one_pass && !out->saw_eoi; // jpeg_start_decompress reads a multi-scan file to EOI; without one it suspends
        out->ecs_off = out->scans.front().ecs_off;
        out->ecs_len = out->scans.back().ecs_off + out->scans.back().ecs_len - out->ecs_off;
        return LP_PARSE_OK;
    }
    {   // slots are built in order 0..3 (the second-level pool is handed out in that order); an absent table id is an empty table
        static const uint8_t no_bits[17] = {0}, no_vals[1] = {0};
        int ac_of_dc[2] = {-1, -1}; // the AC slot behind each DC slot, when the scan's components agree on one
        bool clash[2] = {false, false};
        for (int c = 0; c < j.ncomp && c < LP_MAX_COMP; c++) {
            const int d = j.dc_tbl[c] & 1;
            if (ac_of_dc[d] >= 0 && ac_of_dc[d] != j.ac_tbl[c]) clash[d] = true;
            ac_of_dc[d] = j.ac_tbl[c];
        }
        for (int d = 0; d < 2; d++) if (clash[d]) ac_of_dc[d] = -1;
        // Most files of a stream carry the same tables (the Annex-K set of every encoder that does not optimise): the set built last on
        // this thread is kept with the DHT contents it was built from and copied when they come again (24 -> 9 us per header walk).
        struct Memo { bool valid = false; bool ok[2][2]; uint8_t bits[2][2][17]; uint8_t vals[2][2][256]; int ac_of_dc[2]; LpHuffSet set; };
        static thread_local std::unique_ptr<Memo> memo_owner(new Memo()); // freed at thread exit (short-lived cgo threads would otherwise leave 24 KB each behind)
        Memo* const memo = memo_owner.get();
        bool same = memo->valid && memo->ac_of_dc[0] == ac_of_dc[0] && memo->ac_of_dc[1] == ac_of_dc[1];
        for (int cls = 0; cls < 2 && same; cls++)
            for (int t = 0; t < 2 && same; t++)
                same = memo->ok[cls][t] == h_ok[cls][t] &&
                       (!h_ok[cls][t] || (memcmp(memo->bits[cls][t], hbits[cls][t], 17) == 0 && memcmp(memo->vals[cls][t], hvals[cls][t], 256) == 0));
        if (same) {
            memcpy(&out->huff, &memo->set, sizeof(LpHuffSet));
        } else {
            memset(&out->huff, 0, sizeof(LpHuffSet)); // (every byte defined: sets are compared with memcmp when a batch deduplicates them)
            for (int t = 0; t < 2; t++) lp_build_huff_slot(&out->huff, t, h_ok[0][t] ? hbits[0][t] : no_bits, h_ok[0][t] ? hvals[0][t] : no_vals);
            for (int t = 0; t < 2; t++) lp_build_huff_slot(&out->huff, 2 + t, h_ok[1][t] ? hbits[1][t] : no_bits, h_ok[1][t] ? hvals[1][t] : no_vals);
            lp_build_huff_multi(&out->huff, ac_of_dc);
            memo->valid = false;
            for (int cls = 0; cls < 2; cls++)
                for (int t = 0; t < 2; t++) {
                    memo->ok[cls][t] = h_ok[cls][t];
                    if (!h_ok[cls][t]) continue; // (an absent table's bits / vals were never written: nothing to copy, nothing compared above)
                    memcpy(memo->bits[cls][t], hbits[cls][t], 17);
                    memcpy(memo->vals[cls][t], hvals[cls][t], 256);
                }
            memo->ac_of_dc[0] = ac_of_dc[0]; memo->ac_of_dc[1] = ac_of_dc[1];
            memcpy(&memo->set, &out->huff, sizeof(LpHuffSet));
            memo->valid = true;
        }
    }
    // End of the scan: the common case is a file that ends in EOI; otherwise walk the ECS once.
    out->ecs_off = ecs;
    size_t end = n;
    if (n >= ecs + 2 && d[n - 2] == 0xFF && d[n - 1] == 0xD9) {
        end = n - 2;
        out->saw_eoi = 1;
    } else {
        out->open_end = true;
        for (size_t q = ecs; q + 1 < n; q++) {
            const void* ff = memchr(d + q, 0xFF, n - 1 - q);
            if (!ff) break;
            q = (size_t)((const uint8_t*)ff - d);
            unsigned c = d[q + 1];
            if (c == 0 || c == 0xFF || (c >= 0xD0 && c <= 0xD7)) continue;
            if (j.dri && c < 0xC0) continue; // not a marker libjpeg knows: with a restart interval the decoder reads past it (lp_jbits.h)
            end = q;
            out->saw_eoi = c == 0xD9;
            out->open_end = false;
            break;
        }
    }
    out->ecs_len = end - ecs;
    return LP_PARSE_OK;
}

// A look at the frame header only (segment lengths, no entropy-coded byte is touched): is this a progressive (SOF2) Huffman-coded JPEG?
// The batch front end sizes its chunks with it before the real header walk: the device decodes such files one wave per scan, at a
// latency that does not depend on how many of them a chunk holds.
bool lp_jpeg_sniff_progressive(const uint8_t* d, size_t n, uint64_t* coef_bytes)
{
    if (coef_bytes) *coef_bytes = 0;
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return false;
        const unsigned m = d[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xC2) {
            // the int16 coefficient bytes the frame header asks for (what lp_jpeg_parse's geometry gives: every component's MCU-padded blocks x 128)
            if (coef_bytes && i + 10 <= n) {
                const size_t h = ((size_t)d[i + 5] << 8) | d[i + 6], w = ((size_t)d[i + 7] << 8) | d[i + 8], nc = d[i + 9];
                if (nc >= 1 && nc <= 4 && i + 10 + 3 * nc <= n) {
                    size_t hmax = 1, vmax = 1, blocks_per_mcu = 0;
                    for (size_t c = 0; c < nc; c++) {
                        const size_t hs = d[i + 11 + 3 * c] >> 4, vs = d[i + 11 + 3 * c] & 15;
                        hmax = hs > hmax ? hs : hmax; vmax = vs > vmax ? vs : vmax;
                        blocks_per_mcu += hs * vs;
                    }
                    const size_t mx = (w + 8 * hmax - 1) / (8 * hmax), my = (h + 8 * vmax - 1) / (8 * vmax);
                    *coef_bytes = (uint64_t)mx * my * blocks_per_mcu * 128;
                }
            }
            return true;
        }
        if (m == 0xDA || m == 0xD9 || (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) return false;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
        i += 2 + (((size_t)d[i + 2] << 8) | d[i + 3]);
    }
    return false;
}
