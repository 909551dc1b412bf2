// lp_jpeg_parse.cpp -- host-side marker parsing for the JPEG decode path.
//
// Stands where cv::JpegDecoder::readHeader -> jpeg_read_header sits in the reference
// (/root/reference/opencv.cpp:126-140 opencv_decoder_read_header): SOI, APPn (JFIF / Adobe / EXIF
// orientation, as surfaced by opencv_decoder_get_orientation, opencv.cpp:160-164), DQT, SOF0/1, DHT,
// DRI, SOS. Only what the device kernels need is kept; the entropy-coded segment itself is never
// walked byte-by-byte on the host unless the file does not end in EOI.
#include "lp_jpeg_parse.h"

#include <string.h>

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

// LUT entry for a code of length l decoding to symbol v (see LpHuffSet)
static inline uint16_t lut_entry(int slot, int l, unsigned v)
{
    const bool ends_block = slot >= 2 && (v & 15u) == 0 && (v >> 4) != 15; // AC symbol of size 0 that is not ZRL: EOB (jdhuff.c decode_mcu)
    return (uint16_t)((ends_block ? 0x8000u : 0u) | ((unsigned)l << 8) | v);
}

void lp_build_huff_slot(LpHuffSet* hs, int slot, const uint8_t bits[17], const uint8_t* vals)
{
    memset(hs->lut[slot], 0, sizeof(hs->lut[slot]));
    // canonical code assignment (T.81 Annex C), first-level table, base of the long codes
    uint32_t base2 = 0x10000u;
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        int valptr = k, mincode = code;
        for (int i = 0; i < bits[l]; i++, k++, code++) {
            if (l <= LP_LUT_BITS) {
                int first = code << (LP_LUT_BITS - l), n = 1 << (LP_LUT_BITS - l);
                for (int j = 0; j < n && first + j < LP_LUT_SIZE; j++) hs->lut[slot][first + j] = lut_entry(slot, l, vals[k]);
            } else {
                uint32_t left = (uint32_t)code << (16 - l);
                if (left < base2) base2 = left;
            }
        }
        hs->maxcode[slot][l] = bits[l] ? code - 1 : -1;
        hs->valoff[slot][l] = valptr - mincode;
        code <<= 1;
    }
    hs->base2[slot] = base2;
    // second-level table: a slice of the shared pool covering [base2, base2 + n); slots are built in order 0..3 and take
    // what is left of the pool (Annex-K tables need 2 + 320 + 32 + 320 entries)
    uint32_t used = 0;
    for (int t = 0; t < slot; t++) used = hs->lut2_off[t] + hs->lut2_n[t] > used ? hs->lut2_off[t] + hs->lut2_n[t] : used;
    uint32_t need = 0x10000u - base2, n2 = need < LP_LUT2_POOL - used ? need : LP_LUT2_POOL - used;
    hs->lut2_off[slot] = used;
    hs->lut2_n[slot] = n2;
    memset(hs->lut2 + used, 0, n2 * sizeof(uint16_t));
    code = 0; k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l]; i++, k++, code++) {
            if (l <= LP_LUT_BITS) continue;
            uint32_t first = ((uint32_t)code << (16 - l)) - base2, n = 1u << (16 - l);
            for (uint32_t j = 0; j < n && first + j < n2; j++) hs->lut2[used + first + j] = lut_entry(slot, l, vals[k]);
        }
        code <<= 1;
    }
    hs->maxcode[slot][0] = -1;
    hs->maxcode[slot][17] = 0x7fffffff;
    hs->valoff[slot][0] = 0;
    memset(hs->vals[slot], 0, 256);
    memcpy(hs->vals[slot], vals, (size_t)(k > 256 ? 256 : k));
}

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
    memset(out, 0, sizeof(*out));
    LpJpeg& j = out->j;
    j.orientation = 1;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return LP_PARSE_NOT_JPEG;
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    uint8_t hbits[2][4][17];
    uint8_t hvals[2][4][256];
    bool h_ok[2][4] = {{false, false, false, false}, {false, false, false, false}};
    int cid[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    bool have_sof = false, saw_jfif = false, saw_adobe = false;
    int adobe_tf = 0;
    size_t i = 2;
    size_t ecs = 0;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return LP_PARSE_NOT_JPEG;
        unsigned m = d[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
        if (m == 0xD9) return LP_PARSE_NOT_JPEG;
        size_t L = be16(d + i + 2);
        if (L < 2 || i + 2 + L > n) return LP_PARSE_TRUNCATED;
        const uint8_t* p = d + i + 4;
        size_t pl = L - 2;
        if (m == 0xDB) {
            size_t k = 0;
            while (k < pl) {
                unsigned pq = p[k] >> 4, t = p[k] & 15;
                k++;
                if (t > 3 || k + (pq ? 128 : 64) > pl) return LP_PARSE_NOT_JPEG;
                for (int z = 0; z < 64; z++) {
                    unsigned v;
                    if (pq) { v = be16(p + k); k += 2; } else v = p[k++];
                    qt[t][kZigzag[z]] = (uint16_t)v;
                }
                qt_ok[t] = true;
            }
        } else if (m == 0xC4) {
            size_t k = 0;
            while (k < pl) {
                unsigned tc = p[k] >> 4, th = p[k] & 15;
                k++;
                if (tc > 1 || th > 3 || k + 16 > pl) return LP_PARSE_NOT_JPEG;
                unsigned tot = 0;
                hbits[tc][th][0] = 0;
                for (int b = 1; b <= 16; b++) { hbits[tc][th][b] = p[k++]; tot += hbits[tc][th][b]; }
                if (tot > 256 || k + tot > pl) return LP_PARSE_NOT_JPEG;
                memset(hvals[tc][th], 0, 256);
                memcpy(hvals[tc][th], p + k, tot);
                k += tot;
                h_ok[tc][th] = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (pl < 6) return LP_PARSE_NOT_JPEG;
            if (p[0] != 8) return LP_PARSE_UNSUPPORTED;
            j.height = be16(p + 1);
            j.width = be16(p + 3);
            j.ncomp = p[5];
            if (j.ncomp != 1 && j.ncomp != 3) return LP_PARSE_UNSUPPORTED;
            if (pl < 6 + 3u * j.ncomp) return LP_PARSE_NOT_JPEG;
            for (int c = 0; c < j.ncomp; c++) {
                cid[c] = p[6 + 3 * c];
                j.hs[c] = p[7 + 3 * c] >> 4;
                j.vs[c] = p[7 + 3 * c] & 15;
                tq[c] = p[8 + 3 * c] & 3;
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return LP_PARSE_UNSUPPORTED; // progressive, lossless, arithmetic
        } else if (m == 0xDD) {
            if (pl >= 2) j.dri = be16(p);
        } else if (m == 0xE0) {
            if (pl >= 5 && memcmp(p, "JFIF\0", 5) == 0) saw_jfif = true;
        } else if (m == 0xE1) {
            int o = exif_orientation(p, pl);
            if (o >= 1 && o <= 8 && j.orientation == 1) j.orientation = (uint8_t)o;
        } else if (m == 0xEE) {
            if (pl >= 12 && memcmp(p, "Adobe", 5) == 0) { saw_adobe = true; adobe_tf = p[11]; }
        } else if (m == 0xDA) {
            if (!have_sof || pl < 1) return LP_PARSE_NOT_JPEG;
            unsigned ns = p[0];
            if (ns != j.ncomp) return LP_PARSE_UNSUPPORTED; // non-interleaved / multi-scan
            if (pl < 1 + 2 * ns + 3) return LP_PARSE_NOT_JPEG;
            for (unsigned s = 0; s < ns; s++) {
                int cs = p[1 + 2 * s], t = p[2 + 2 * s], c;
                for (c = 0; c < j.ncomp; c++) if (cid[c] == cs) break;
                if (c == j.ncomp) return LP_PARSE_NOT_JPEG;
                if ((unsigned)c != s) return LP_PARSE_UNSUPPORTED;
                td[c] = (t >> 4) & 3;
                ta[c] = t & 3;
            }
            ecs = i + 2 + L;
            break;
        }
        i += 2 + L;
    }
    if (!have_sof || !ecs || j.width == 0 || j.height == 0) return LP_PARSE_NOT_JPEG;
    if (j.ncomp == 1) { j.hs[0] = j.vs[0] = 1; }
    j.hmax = j.vmax = 1;
    for (int c = 0; c < j.ncomp; c++) {
        if (j.hs[c] < 1 || j.hs[c] > 2 || j.vs[c] < 1 || j.vs[c] > 2) return LP_PARSE_UNSUPPORTED;
        if (j.hs[c] > j.hmax) j.hmax = j.hs[c];
        if (j.vs[c] > j.vmax) j.vmax = j.vs[c];
        if (!qt_ok[tq[c]] || !h_ok[0][td[c]] || !h_ok[1][ta[c]]) return LP_PARSE_NOT_JPEG;
        if (td[c] > 1 || ta[c] > 1) return LP_PARSE_UNSUPPORTED;
        if (!huff_table_valid(hbits[0][td[c]], hvals[0][td[c]], true) || !huff_table_valid(hbits[1][ta[c]], hvals[1][ta[c]], false))
            return LP_PARSE_NOT_JPEG;
    }
    if (j.ncomp == 3 && (j.hs[1] != 1 || j.vs[1] != 1 || j.hs[2] != 1 || j.vs[2] != 1)) return LP_PARSE_UNSUPPORTED;
    j.mcus_x = (j.width + 8 * j.hmax - 1) / (8 * j.hmax);
    j.mcus_y = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
    unsigned bpm = 0;
    for (int c = 0; c < j.ncomp; c++) {
        for (int v = 0; v < j.vs[c]; v++)
            for (int h = 0; h < j.hs[c]; h++) {
                if (bpm >= LP_MAX_BPM) return LP_PARSE_UNSUPPORTED;
                j.blk_comp[bpm] = (uint8_t)c;
                j.blk_h[bpm] = (uint8_t)h;
                j.blk_v[bpm] = (uint8_t)v;
                bpm++;
            }
        j.blk_first[c] = (uint8_t)(bpm - j.hs[c] * j.vs[c]);
        j.bw[c] = j.mcus_x * j.hs[c];
        j.bh[c] = j.mcus_y * j.vs[c];
        j.plane_stride[c] = j.bw[c] * 8;
        j.dc_tbl[c] = (uint8_t)td[c];
        j.ac_tbl[c] = (uint8_t)(2 + ta[c]);
        memcpy(j.qt[c], qt[tq[c]], sizeof(j.qt[c]));
    }
    j.bpm = (uint8_t)bpm;
    j.blkpack = 0;
    for (unsigned b = 0; b < bpm; b++) {
        const unsigned c = j.blk_comp[b];
        j.blkpack |= (uint64_t)(c | ((unsigned)td[c] << 2) | ((unsigned)ta[c] << 3)) << (4 * b);
    }
    j.total_blocks = j.mcus_x * j.mcus_y * bpm;
    // libjpeg's colour space guess (jdapimin.c default_decompress_parms)
    if (j.ncomp == 1) j.colorspace = 1;
    else if (saw_jfif) j.colorspace = 2;
    else if (saw_adobe) j.colorspace = adobe_tf == 0 ? 3 : 2;
    else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') j.colorspace = 3;
    else j.colorspace = 2;
    {   // slots are built in order 0..3 (the second-level pool is handed out in that order); an absent table id is an empty table
        static const uint8_t no_bits[17] = {0}, no_vals[1] = {0};
        for (int t = 0; t < 2; t++) lp_build_huff_slot(&out->huff, t, h_ok[0][t] ? hbits[0][t] : no_bits, h_ok[0][t] ? hvals[0][t] : no_vals);
        for (int t = 0; t < 2; t++) lp_build_huff_slot(&out->huff, 2 + t, h_ok[1][t] ? hbits[1][t] : no_bits, h_ok[1][t] ? hvals[1][t] : no_vals);
    }
    // End of the scan: the common case is a file that ends in EOI; otherwise walk the ECS once.
    out->ecs_off = ecs;
    size_t end = n;
    if (n >= ecs + 2 && d[n - 2] == 0xFF && d[n - 1] == 0xD9) {
        end = n - 2;
        out->saw_eoi = 1;
    } else {
        for (size_t q = ecs; q + 1 < n; q++) {
            if (d[q] != 0xFF) continue;
            unsigned c = d[q + 1];
            if (c == 0 || c == 0xFF || (c >= 0xD0 && c <= 0xD7)) continue;
            end = q;
            out->saw_eoi = c == 0xD9;
            break;
        }
    }
    out->ecs_len = end - ecs;
    return LP_PARSE_OK;
}
