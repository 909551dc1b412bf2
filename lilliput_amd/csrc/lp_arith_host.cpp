// lp_arith_host.cpp -- see lp_arith_host.h. Restates libjpeg-turbo 3.1.0 jdarith.c (source not in the reference tree; pinned against
// the reference's own libjpeg.a through oracle/_ref, tests/test_arith.py) after T.81 Annex D (the QM decoder) and F.1.4 / G.1.3.
#include "lp_arith_host.h"

#include "lp_jbits.h"

#include <string.h>

namespace {

// T.81 Table D.3: Qe, next index after an LPS, next index after an MPS, "exchange the sense of the MPS". Entry 113 is libjpeg's
// extra state with a fixed probability of 0.5 (the sign bits and the DC refinement bits).
struct Qe { uint16_t qe; uint8_t nlps, nmps, sw; };
const Qe kQe[114] = {
    {0x5A1D, 1, 1, 1},     {0x2586, 14, 2, 0},    {0x1114, 16, 3, 0},    {0x080B, 18, 4, 0},    {0x03D8, 20, 5, 0},    {0x01DA, 23, 6, 0},
    {0x00E5, 25, 7, 0},    {0x006F, 28, 8, 0},    {0x0036, 30, 9, 0},    {0x001A, 33, 10, 0},   {0x000D, 35, 11, 0},   {0x0006, 9, 12, 0},
    {0x0003, 10, 13, 0},   {0x0001, 12, 13, 0},   {0x5A7F, 15, 15, 1},   {0x3F25, 36, 16, 0},   {0x2CF2, 38, 17, 0},   {0x207C, 39, 18, 0},
    {0x17B9, 40, 19, 0},   {0x1182, 42, 20, 0},   {0x0CEF, 43, 21, 0},   {0x09A1, 45, 22, 0},   {0x072F, 46, 23, 0},   {0x055C, 48, 24, 0},
    {0x0406, 49, 25, 0},   {0x0303, 51, 26, 0},   {0x0240, 52, 27, 0},   {0x01B1, 54, 28, 0},   {0x0144, 56, 29, 0},   {0x00F5, 57, 30, 0},
    {0x00B7, 59, 31, 0},   {0x008A, 60, 32, 0},   {0x0068, 62, 33, 0},   {0x004E, 63, 34, 0},   {0x003B, 32, 35, 0},   {0x002C, 33, 9, 0},
    {0x5AE1, 37, 37, 1},   {0x484C, 64, 38, 0},   {0x3A0D, 65, 39, 0},   {0x2EF1, 67, 40, 0},   {0x261F, 68, 41, 0},   {0x1F33, 69, 42, 0},
    {0x19A8, 70, 43, 0},   {0x1518, 72, 44, 0},   {0x1177, 73, 45, 0},   {0x0E74, 74, 46, 0},   {0x0BFB, 75, 47, 0},   {0x09F8, 77, 48, 0},
    {0x0861, 78, 49, 0},   {0x0706, 79, 50, 0},   {0x05CD, 48, 51, 0},   {0x04DE, 50, 52, 0},   {0x040F, 50, 53, 0},   {0x0363, 51, 54, 0},
    {0x02D4, 52, 55, 0},   {0x025C, 53, 56, 0},   {0x01F8, 54, 57, 0},   {0x01A4, 55, 58, 0},   {0x0160, 56, 59, 0},   {0x0125, 57, 60, 0},
    {0x00F6, 58, 61, 0},   {0x00CB, 59, 62, 0},   {0x00AB, 61, 63, 0},   {0x008F, 61, 32, 0},   {0x5B12, 65, 65, 1},   {0x4D04, 80, 66, 0},
    {0x412C, 81, 67, 0},   {0x37D8, 82, 68, 0},   {0x2FE8, 83, 69, 0},   {0x293C, 84, 70, 0},   {0x2379, 86, 71, 0},   {0x1EDF, 87, 72, 0},
    {0x1AA9, 87, 73, 0},   {0x174E, 72, 74, 0},   {0x1424, 72, 75, 0},   {0x119C, 74, 76, 0},   {0x0F6B, 74, 77, 0},   {0x0D51, 75, 78, 0},
    {0x0BB6, 77, 79, 0},   {0x0A40, 77, 48, 0},   {0x5832, 80, 81, 1},   {0x4D1C, 88, 82, 0},   {0x438E, 89, 83, 0},   {0x3BDD, 90, 84, 0},
    {0x34EE, 91, 85, 0},   {0x2EAE, 92, 86, 0},   {0x299A, 93, 87, 0},   {0x2516, 86, 71, 0},   {0x5570, 88, 89, 1},   {0x4CA9, 95, 90, 0},
    {0x44D9, 96, 91, 0},   {0x3E22, 97, 92, 0},   {0x3824, 99, 93, 0},   {0x32B4, 99, 94, 0},   {0x2E17, 93, 86, 0},   {0x56A8, 95, 96, 1},
    {0x4F46, 101, 97, 0},  {0x47E5, 102, 98, 0},  {0x41CF, 103, 99, 0},  {0x3C3D, 104, 100, 0}, {0x375E, 99, 93, 0},   {0x5231, 105, 102, 0},
    {0x4C0F, 106, 103, 0}, {0x4639, 107, 104, 0}, {0x415E, 103, 99, 0},  {0x5627, 105, 106, 1}, {0x50E7, 108, 107, 0}, {0x4B85, 109, 103, 0},
    {0x5597, 110, 109, 0}, {0x504F, 111, 107, 0}, {0x5A10, 110, 111, 1}, {0x5522, 112, 109, 0}, {0x59EB, 112, 111, 1}, {0x5A1D, 113, 113, 0}};

struct Dec {
    LpJSrc src;         // the raw bytes and jdarith.c's cinfo->unread_marker: once a marker has been met the decoder is fed zero bytes
    int64_t c, a;
    int ct;             // -16 at the start of an interval; -1 = the decoder has given up (JWRN_ARITH_BAD_CODE)
    uint8_t dc_stats[16][64], ac_stats[16][256], fixed_bin[4];
    int32_t last_dc[4];
    int dc_ctx[4];

    int byte()          // jdarith.c get_byte over cv::JpegDecoder's source manager: an empty buffer is JERR_CANT_SUSPEND (the image fails)
    {
        if (src.p == src.end) { src.suspended = true; return -1; }
        return *src.p++;
    }
    // T.81 D.2: one binary decision with statistics bin *st (bit 7: sense of the MPS, bits 0..6: index into Table D.3)
    int decode(uint8_t* st)
    {
        while (a < 0x8000) { // renormalisation and byte-in, D.2.6
            if (--ct < 0) {
                int data;
                if (src.marker || src.suspended) data = 0;
                else {
                    data = byte();
                    if (data < 0) data = 0;
                    else if (data == 0xFF) {
                        do data = byte(); while (data == 0xFF); // fill bytes
                        if (data < 0) data = 0;
                        else if (data == 0) data = 0xFF;        // a stuffed zero
                        else { src.marker = data; data = 0; }   // a marker inside the segment is legal: zeros from here on
                    }
                }
                c = (c << 8) | data;
                if ((ct += 8) < 0)      // the first two bytes of an interval
                    if (++ct == 0) a = 0x8000;
            }
            a <<= 1;
        }
        int sv = *st;
        const Qe& q = kQe[sv & 0x7F];
        const int64_t qe = q.qe;
        const uint8_t nl = (uint8_t)(q.nlps | (q.sw << 7)), nm = q.nmps;
        int64_t temp = a - qe;
        a = temp;
        temp <<= ct;
        if (c >= temp) {
            c -= temp;
            if (a < qe) { a = qe; *st = (uint8_t)((sv & 0x80) ^ nm); }          // conditional LPS exchange
            else { a = qe; *st = (uint8_t)((sv & 0x80) ^ nl); sv ^= 0x80; }
        } else if (a < 0x8000) {
            if (a < qe) { *st = (uint8_t)((sv & 0x80) ^ nl); sv ^= 0x80; }      // conditional MPS exchange
            else *st = (uint8_t)((sv & 0x80) ^ nm);
        }
        return sv >> 7;
    }
};

} // namespace

int lp_arith_scan(const uint8_t* ecs, const uint8_t* file_end, const LpProgScan& sc, const LpArithScan& ar, int16_t* coef, bool whole_file)
{
    Dec* dp = new Dec();
    Dec& d = *dp;
    d.src.init(ecs, file_end);
    const uint32_t Ss = sc.Ss, Se = sc.Se, Ah = sc.Ah, Al = sc.Al;
    const bool seq = sc.sequential != 0;
    const bool dc_part = seq || (Ss == 0 && Ah == 0), ac_part = seq || Ss != 0;
    auto reset = [&] { // start_pass / process_restart: the statistics of the scan's tables, the predictors, the registers
        for (uint32_t s = 0; s < sc.ns; s++) {
            if (dc_part) { memset(d.dc_stats[ar.dc_tbl[s] & 15], 0, 64); d.last_dc[s] = 0; d.dc_ctx[s] = 0; }
            if (ac_part) memset(d.ac_stats[ar.ac_tbl[s] & 15], 0, 256);
        }
        d.c = 0; d.a = 0; d.ct = -16;
    };
    memset(d.fixed_bin, 0, sizeof(d.fixed_bin));
    d.fixed_bin[0] = 113;
    reset();
    int bad = 0;
    uint32_t rst_left = sc.dri, next_rst = 0;
    const int32_t p1 = 1 << Al, m1 = -(1 << Al);
    for (uint32_t my = 0; my < sc.mcuy; my++)
        for (uint32_t mx = 0; mx < sc.mcux; mx++) {
            if (sc.dri) { // decode_mcu_*: "Process restart marker if needed"
                if (rst_left == 0) {
                    // jdmarker.c read_restart_marker / jpeg_resync_to_restart (lp_jbits.h); out of bytes: JERR_CANT_SUSPEND
                    if (!d.src.restart_marker(next_rst)) { delete dp; return LP_SCAN_OUT_OF_DATA; }
                    next_rst = (next_rst + 1u) & 7u;
                    reset();
                    rst_left = sc.dri;
                }
                rst_left--;
            }
            if (d.src.suspended) { delete dp; return LP_SCAN_OUT_OF_DATA; } // the decoder asked for a byte the buffer does not hold
            if (d.ct == -1) continue; // "if error do nothing"
            if (seq || Ss == 0) {
                for (uint32_t s = 0; s < sc.ns && d.ct != -1; s++)
                    for (uint32_t v = 0; v < sc.vs[s] && d.ct != -1; v++)
                        for (uint32_t h = 0; h < sc.hs[s] && d.ct != -1; h++) {
                            int16_t* blk = coef + ((size_t)sc.cblk[s] + (size_t)(my * sc.vs[s] + v) * sc.bw[s] + mx * sc.hs[s] + h) * 64;
                            if (!seq && Ah != 0) { // decode_mcu_DC_refine: one fixed-probability bit per block
                                if (d.decode(d.fixed_bin)) blk[0] = (int16_t)(blk[0] | p1);
                                continue;
                            }
                            // decode_mcu_DC_first / the DC half of decode_mcu: F.1.4.4.1
                            const uint32_t tbl = ar.dc_tbl[s] & 15;
                            uint8_t* st = d.dc_stats[tbl] + d.dc_ctx[s];
                            if (d.decode(st) == 0) d.dc_ctx[s] = 0;
                            else {
                                const int sign = d.decode(st + 1);
                                st += 2; st += sign;
                                int m = d.decode(st);
                                if (m != 0) {
                                    st = d.dc_stats[tbl] + 20; // Table F.4: X1 = 20
                                    while (d.decode(st)) {
                                        if ((m <<= 1) == 0x8000) { d.ct = -1; bad = 1; break; }
                                        st += 1;
                                    }
                                    if (d.ct == -1) break;
                                }
                                // F.1.4.4.1.2: the conditioning category of the next difference
                                if (m < (int)((1L << ar.dc_L[s]) >> 1)) d.dc_ctx[s] = 0;
                                else if (m > (int)((1L << ar.dc_U[s]) >> 1)) d.dc_ctx[s] = 12 + sign * 4;
                                else d.dc_ctx[s] = 4 + sign * 4;
                                int v2 = m;
                                st += 14; // Figure F.24: the magnitude bit pattern
                                while (m >>= 1) if (d.decode(st)) v2 |= m;
                                v2 += 1;
                                if (sign) v2 = -v2;
                                d.last_dc[s] = (d.last_dc[s] + v2) & 0xffff;
                            }
                            blk[0] = (int16_t)(uint16_t)((uint32_t)d.last_dc[s] << (seq ? 0 : Al));
                            if (!seq) continue;
                            // the AC half of decode_mcu: Figure F.20
                            const uint32_t at = ar.ac_tbl[s] & 15;
                            for (uint32_t k = 1; k <= 63; k++) {
                                st = d.ac_stats[at] + 3 * (k - 1);
                                if (d.decode(st)) break; // EOB
                                while (d.decode(st + 1) == 0) {
                                    st += 3; k++;
                                    if (k > 63) { d.ct = -1; bad = 1; break; }
                                }
                                if (d.ct == -1) break;
                                const int sign = d.decode(d.fixed_bin);
                                st += 2;
                                int m = d.decode(st);
                                if (m != 0 && d.decode(st)) {
                                    m <<= 1;
                                    st = d.ac_stats[at] + (k <= ar.ac_K[s] ? 189 : 217);
                                    while (d.decode(st)) {
                                        if ((m <<= 1) == 0x8000) { d.ct = -1; bad = 1; break; }
                                        st += 1;
                                    }
                                    if (d.ct == -1) break;
                                }
                                int v2 = m;
                                st += 14;
                                while (m >>= 1) if (d.decode(st)) v2 |= m;
                                v2 += 1;
                                if (sign) v2 = -v2;
                                blk[k] = (int16_t)v2;
                            }
                        }
                continue;
            }
            // AC scans hold one component, one block per MCU
            int16_t* blk = coef + ((size_t)sc.cblk[0] + (size_t)my * sc.bw[0] + mx) * 64;
            const uint32_t at = ar.ac_tbl[0] & 15;
            if (Ah == 0) { // decode_mcu_AC_first
                for (uint32_t k = Ss; k <= Se; k++) {
                    uint8_t* st = d.ac_stats[at] + 3 * (k - 1);
                    if (d.decode(st)) break; // EOB
                    while (d.decode(st + 1) == 0) {
                        st += 3; k++;
                        if (k > Se) { d.ct = -1; bad = 1; break; }
                    }
                    if (d.ct == -1) break;
                    const int sign = d.decode(d.fixed_bin);
                    st += 2;
                    int m = d.decode(st);
                    if (m != 0 && d.decode(st)) {
                        m <<= 1;
                        st = d.ac_stats[at] + (k <= ar.ac_K[0] ? 189 : 217);
                        while (d.decode(st)) {
                            if ((m <<= 1) == 0x8000) { d.ct = -1; bad = 1; break; }
                            st += 1;
                        }
                        if (d.ct == -1) break;
                    }
                    int v2 = m;
                    st += 14;
                    while (m >>= 1) if (d.decode(st)) v2 |= m;
                    v2 += 1;
                    if (sign) v2 = -v2;
                    blk[k] = (int16_t)(uint16_t)((uint32_t)v2 << Al);
                }
            } else { // decode_mcu_AC_refine
                int kex = (int)Se;
                for (; kex >= (int)Ss; kex--) if (blk[kex]) break; // end of block of the previous stage (inside the band: what lies below it belongs to other scans, and cannot change the k > kex test)
                for (uint32_t k = Ss; k <= Se; k++) {
                    uint8_t* st = d.ac_stats[at] + 3 * (k - 1);
                    if ((int)k > kex && d.decode(st)) break; // EOB
                    for (;;) {
                        int16_t* co = blk + k;
                        if (*co) { // previously non-zero: a correction bit
                            if (d.decode(st + 2)) *co = (int16_t)(*co < 0 ? *co + m1 : *co + p1);
                            break;
                        }
                        if (d.decode(st + 1)) { // newly non-zero
                            *co = (int16_t)(d.decode(d.fixed_bin) ? m1 : p1);
                            break;
                        }
                        st += 3; k++;
                        if (k > Se) { d.ct = -1; bad = 1; break; }
                    }
                    if (d.ct == -1) break;
                }
            }
        }
    if (d.src.suspended) { delete dp; return LP_SCAN_OUT_OF_DATA; }
    // what jdmarker.c read_markers meets behind the scan (a file of several scans is read to its end before any pixel is returned)
    if (whole_file) {
        const int after = d.src.after_scan(sc.dri != 0);
        if (after) bad = after;
    }
    delete dp;
    return bad;
}
