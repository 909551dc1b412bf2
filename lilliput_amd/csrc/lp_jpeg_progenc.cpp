// lp_jpeg_progenc.cpp -- see lp_jpeg_progenc.h. Follows libjpeg-turbo 3.1.0 (source not in the reference tree; prebuilt
// deps/linux/amd64/lib/libjpeg.a): jcparam.c jpeg_simple_progression (the scan script), jcphuff.c encode_mcu_DC_first /
// AC_first / DC_refine / AC_refine + emit_eobrun (EOB runs, buffered correction bits), jchuff.c jpeg_gen_optimal_table
// (T.81 K.2 with libjpeg's tie-breaking and 16-bit length limiting), jcmarker.c (DHT only for the table a scan uses, selector
// nibbles of unused tables written as 0).
#include "lp_jpeg_progenc.h"

#include <string.h>

#include "lp_engine.h"
#include "lp_abi_guard.h"

namespace {
struct Scan { int ncomp; int comp[3]; int Ss, Se, Ah, Al; };

struct Table { // one Huffman table: statistics, then the optimal code
    long freq[257];
    uint8_t bits[17];
    uint8_t vals[256];
    int nvals;
    uint16_t code[256];
    uint8_t len[256];
};

int bit_length(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

// jchuff.c jpeg_gen_optimal_table
void gen_optimal_table(Table& t)
{
    int bits[33], codesize[257], others[257];
    long freq[257];
    memset(bits, 0, sizeof(bits));
    memset(codesize, 0, sizeof(codesize));
    for (int i = 0; i < 257; i++) others[i] = -1;
    memcpy(freq, t.freq, sizeof(freq));
    freq[256] = 1; // the pseudo symbol takes the all-ones code of the longest length
    for (;;) {
        // the two smallest non-zero frequencies; ties go to the larger symbol number
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; i++) if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; i--) // no code may be longer than 16 bits (T.81 K.2, figure K.3)
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) j--;
            bits[i] -= 2;
            bits[i - 1]++;
            bits[j + 1] += 2;
            bits[j]--;
        }
    int i = 16;
    while (bits[i] == 0) i--;
    bits[i]--; // the pseudo symbol
    t.bits[0] = 0;
    for (int l = 1; l <= 16; l++) t.bits[l] = (uint8_t)bits[l];
    int p = 0; // symbols by (original) code length, then by value
    for (int l = 1; l <= 32; l++)
        for (int s = 0; s <= 255; s++)
            if (codesize[s] == l) t.vals[p++] = (uint8_t)s;
    t.nvals = p;
    // jpeg_make_c_derived_tbl: canonical codes
    memset(t.len, 0, sizeof(t.len));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int q = 0; q < t.bits[l]; q++, k++) { t.code[t.vals[k]] = (uint16_t)code++; t.len[t.vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
}

struct Coder { // jcphuff.c phuff_entropy_encoder for one scan
    std::vector<uint8_t>* out;
    bool gather;
    Table* tbl;          // the scan's table (AC scans), or indexed by dc_tbl_no via dc[] (DC scans)
    uint64_t acc = 0;
    int nacc = 0;
    unsigned eobrun = 0;
    std::vector<uint8_t> corr; // buffered correction bits of the blocks inside the current EOB run (entropy->bit_buffer, BE)
    void emit_bits(unsigned v, int n)
    {
        if (gather || !n) return;
        acc = (acc << n) | (v & ((1u << n) - 1u));
        nacc += n;
        while (nacc >= 8) {
            const uint8_t c = (uint8_t)(acc >> (nacc - 8));
            out->push_back(c);
            if (c == 0xFF) out->push_back(0);
            nacc -= 8;
        }
    }
    void emit_symbol(Table& t, int s)
    {
        if (gather) t.freq[s]++;
        else emit_bits(t.code[s], t.len[s]);
    }
    void emit_buffered(const uint8_t* b, size_t n)
    {
        if (gather) return;
        for (size_t i = 0; i < n; i++) emit_bits(b[i], 1);
    }
    void emit_eobrun()
    {
        if (eobrun > 0) {
            const int nbits = bit_length(eobrun) - 1;
            emit_symbol(*tbl, nbits << 4);
            if (nbits) emit_bits(eobrun, nbits);
            eobrun = 0;
            emit_buffered(corr.data(), corr.size());
            corr.clear();
        }
    }
    void flush() { emit_bits(0x7F, 7); acc = 0; nacc = 0; } // fill the last byte with one bits
};
} // namespace

bool lp_jpeg_encode_progressive(int W, int H, int ncomp, int quality, const int16_t* coef, std::vector<uint8_t>& out)
{
    if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3)) return false;
    const int mcu = ncomp == 1 ? 8 : 16, mcus_x = (W + mcu - 1) / mcu, mcus_y = (H + mcu - 1) / mcu, bpm = ncomp == 1 ? 1 : 6;
    // blocks of a component: the padded grid (interleaved DC scans) and the component's own blocks (single-component scans)
    int gw[3], gh[3], rw[3], rh[3];
    for (int c = 0; c < ncomp; c++) {
        const int hs = (ncomp == 3 && c == 0) ? 2 : 1;
        gw[c] = mcus_x * hs; gh[c] = mcus_y * hs;
        rw[c] = ncomp == 1 ? (W + 7) / 8 : (W * hs + 15) / 16;
        rh[c] = ncomp == 1 ? (H + 7) / 8 : (H * hs + 15) / 16;
    }
    auto block = [&](int c, int by, int bx) -> const int16_t* {
        size_t idx;
        if (ncomp == 1) idx = (size_t)by * mcus_x + bx;
        else if (c == 0) idx = ((size_t)(by >> 1) * mcus_x + (bx >> 1)) * 6 + (by & 1) * 2 + (bx & 1);
        else idx = ((size_t)by * mcus_x + bx) * 6 + 3 + c;
        return coef + idx * 64;
    };
    // DC of a block of the padded grid. k_enc_fdct transforms edge-replicated pixels for the dummy blocks; libjpeg (jccoefct.c
    // compress_first_pass) gives a dummy block at the right edge the DC of the last real block of its row, a dummy block row at the
    // bottom the DC of the right-most block of the same MCU in the row above, and no AC at all -- AC scans never visit them.
    auto dc_of = [&](int c, int by, int bx) -> int {
        const int hs = (ncomp == 3 && c == 0) ? 2 : 1;
        while (by >= rh[c]) { by--; bx = bx / hs * hs + hs - 1; }
        if (bx >= rw[c]) bx = rw[c] - 1;
        return block(c, by, bx)[0];
    };
    (void)gw; (void)gh; (void)bpm;
    // ---- file header: what the baseline writer emits up to the frame header, with SOF2 in place of SOF0
    uint8_t hb[1024];
    uint16_t qt[2][64];
    const size_t hl = lp_build_jpeg_header(W, H, ncomp, quality, hb, qt);
    size_t sof = 0;
    for (size_t i = 2; i + 1 < hl; ) { // walk the segments up to SOF0
        if (hb[i] == 0xFF && hb[i + 1] == 0xC0) { sof = i; break; }
        i += 2 + ((size_t)hb[i + 2] << 8 | hb[i + 3]);
    }
    if (!sof) return false;
    out.clear();
    out.insert(out.end(), hb, hb + sof);
    const size_t sof_len = 2 + ((size_t)hb[sof + 2] << 8 | hb[sof + 3]);
    out.insert(out.end(), hb + sof, hb + sof + sof_len);
    out[out.size() - sof_len + 1] = 0xC2;
    // ---- jcparam.c jpeg_simple_progression
    std::vector<Scan> script;
    if (ncomp == 3) {
        script = {{3, {0, 1, 2}, 0, 0, 0, 1}, {1, {0}, 1, 5, 0, 2},  {1, {2}, 1, 63, 0, 1}, {1, {1}, 1, 63, 0, 1}, {1, {0}, 6, 63, 0, 2},
                  {1, {0}, 1, 63, 2, 1},      {3, {0, 1, 2}, 0, 0, 1, 0}, {1, {2}, 1, 63, 1, 0}, {1, {1}, 1, 63, 1, 0}, {1, {0}, 1, 63, 1, 0}};
    } else {
        script = {{1, {0}, 0, 0, 0, 1}, {1, {0}, 1, 5, 0, 2}, {1, {0}, 6, 63, 0, 2}, {1, {0}, 1, 63, 2, 1}, {1, {0}, 0, 0, 1, 0}, {1, {0}, 1, 63, 1, 0}};
    }
    for (const Scan& sc : script) {
        Table tabs[2]; // by table number: 0 luma, 1 chroma (DC scans may use both, an AC scan one)
        for (int pass = 0; pass < 2; pass++) { // gather statistics, then emit
            const bool gather = pass == 0;
            const bool dc_scan = sc.Ss == 0, needs_table = !(dc_scan && sc.Ah != 0);
            if (gather && !needs_table) continue; // a DC refinement scan is raw bits
            if (gather) { memset(tabs[0].freq, 0, sizeof(tabs[0].freq)); memset(tabs[1].freq, 0, sizeof(tabs[1].freq)); }
            else {
                // jcphuff.c finish_pass_gather_phuff + jcmarker.c write_scan_header: the tables this scan uses, each once
                bool did[2] = {false, false};
                for (int s = 0; s < sc.ncomp && needs_table; s++) {
                    const int tn = sc.comp[s] ? 1 : 0;
                    if (did[tn]) continue;
                    did[tn] = true;
                    gen_optimal_table(tabs[tn]);
                    out.push_back(0xFF); out.push_back(0xC4);
                    const int len = 2 + 1 + 16 + tabs[tn].nvals;
                    out.push_back((uint8_t)(len >> 8)); out.push_back((uint8_t)len);
                    out.push_back((uint8_t)(tn | (dc_scan ? 0 : 0x10)));
                    for (int l = 1; l <= 16; l++) out.push_back(tabs[tn].bits[l]);
                    out.insert(out.end(), tabs[tn].vals, tabs[tn].vals + tabs[tn].nvals);
                }
                out.push_back(0xFF); out.push_back(0xDA);
                const int len = 6 + 2 * sc.ncomp;
                out.push_back((uint8_t)(len >> 8)); out.push_back((uint8_t)len);
                out.push_back((uint8_t)sc.ncomp);
                for (int s = 0; s < sc.ncomp; s++) {
                    const int tn = sc.comp[s] ? 1 : 0;
                    out.push_back((uint8_t)(sc.comp[s] + 1));
                    // only the table a scan uses is named; the other nibble (and both in a DC refinement) is 0
                    out.push_back((uint8_t)(dc_scan ? (sc.Ah == 0 ? tn << 4 : 0) : tn));
                }
                out.push_back((uint8_t)sc.Ss); out.push_back((uint8_t)sc.Se); out.push_back((uint8_t)((sc.Ah << 4) | sc.Al));
            }
            Coder cd;
            cd.out = &out;
            cd.gather = gather;
            cd.tbl = &tabs[sc.comp[0] ? 1 : 0];
            const int Al = sc.Al;
            if (dc_scan) {
                int last_dc[3] = {0, 0, 0};
                const int my_n = sc.ncomp == 1 ? rh[sc.comp[0]] : mcus_y, mx_n = sc.ncomp == 1 ? rw[sc.comp[0]] : mcus_x;
                for (int my = 0; my < my_n; my++)
                    for (int mx = 0; mx < mx_n; mx++)
                        for (int s = 0; s < sc.ncomp; s++) {
                            const int c = sc.comp[s], nh = (sc.ncomp == 3 && c == 0) ? 2 : 1;
                            for (int v = 0; v < nh; v++)
                                for (int h = 0; h < nh; h++) {
                                    const int dc = dc_of(c, my * nh + v, mx * nh + h);
                                    if (sc.Ah == 0) { // encode_mcu_DC_first: the point transform is an arithmetic shift
                                        const int t2 = dc >> Al;
                                        int temp = t2 - last_dc[c], temp2 = temp;
                                        last_dc[c] = t2;
                                        if (temp < 0) { temp = -temp; temp2--; }
                                        const int nbits = bit_length((unsigned)temp);
                                        if (nbits > 11) return false; // JERR_BAD_DCT_COEF
                                        cd.emit_symbol(tabs[c ? 1 : 0], nbits);
                                        if (nbits) cd.emit_bits((unsigned)temp2, nbits);
                                    } else // encode_mcu_DC_refine
                                        cd.emit_bits((unsigned)(dc >> Al) & 1u, 1);
                                }
                        }
            } else {
                const int c = sc.comp[0];
                for (int by = 0; by < rh[c]; by++)
                    for (int bx = 0; bx < rw[c]; bx++) {
                        const int16_t* blk = block(c, by, bx); // zigzag order: element k is the k-th coefficient of the scan order
                        if (sc.Ah == 0) { // encode_mcu_AC_first
                            int r = 0;
                            for (int k = sc.Ss; k <= sc.Se; k++) {
                                int temp = blk[k], temp2;
                                if (temp == 0) { r++; continue; }
                                if (temp < 0) { temp = -temp; temp >>= Al; temp2 = ~temp; }
                                else { temp >>= Al; temp2 = temp; }
                                if (temp == 0) { r++; continue; } // non-zero only below the point transform
                                if (cd.eobrun > 0) cd.emit_eobrun();
                                while (r > 15) { cd.emit_symbol(*cd.tbl, 0xF0); r -= 16; }
                                const int nbits = bit_length((unsigned)temp);
                                if (nbits > 10) return false; // JERR_BAD_DCT_COEF
                                cd.emit_symbol(*cd.tbl, (r << 4) + nbits);
                                cd.emit_bits((unsigned)temp2, nbits);
                                r = 0;
                            }
                            if (r > 0) {
                                cd.eobrun++;
                                if (cd.eobrun == 0x7FFF) cd.emit_eobrun();
                            }
                        } else { // encode_mcu_AC_refine
                            int absv[64], eob = 0;
                            for (int k = sc.Ss; k <= sc.Se; k++) {
                                int temp = blk[k];
                                if (temp < 0) temp = -temp;
                                temp >>= Al;
                                absv[k] = temp;
                                if (temp == 1) eob = k; // the last coefficient that becomes non-zero in this scan
                            }
                            int r = 0;
                            std::vector<uint8_t> br; // correction bits of this block, appended behind cd.corr at the end
                            for (int k = sc.Ss; k <= sc.Se; k++) {
                                const int temp = absv[k];
                                if (temp == 0) { r++; continue; }
                                while (r > 15 && k <= eob) { // ZRLs, unless the EOB swallows them
                                    cd.emit_eobrun();
                                    cd.emit_symbol(*cd.tbl, 0xF0);
                                    r -= 16;
                                    cd.emit_buffered(br.data(), br.size());
                                    br.clear();
                                }
                                if (temp > 1) { br.push_back((uint8_t)(temp & 1)); continue; } // already non-zero: just its next bit
                                cd.emit_eobrun();
                                cd.emit_symbol(*cd.tbl, (r << 4) + 1);
                                cd.emit_bits(blk[k] < 0 ? 0u : 1u, 1);
                                cd.emit_buffered(br.data(), br.size());
                                br.clear();
                                r = 0;
                            }
                            if (r > 0 || !br.empty()) {
                                cd.eobrun++;
                                cd.corr.insert(cd.corr.end(), br.begin(), br.end());
                                if (cd.eobrun == 0x7FFF || cd.corr.size() > 1000 - 64 + 1) cd.emit_eobrun();
                            }
                        }
                    }
                cd.emit_eobrun();
            }
            if (!gather) cd.flush();
        }
    }
    out.push_back(0xFF); out.push_back(0xD9);
    return true;
}

// Test access (no device work): progressive JPEG bytes from quantised coefficients in the encoder's own layout (MCU order, zigzag
// order per block). Returns the length, 0 on failure, or the negated length needed when cap is too small.
extern "C" long lilliput_hip_progressive_encode_coefs(int width, int height, int ncomp, int quality, const int16_t* coef, uint8_t* out, size_t cap)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    std::vector<uint8_t> v;
    if (!lp_jpeg_encode_progressive(width, height, ncomp, quality, coef, v)) return 0;
    if (v.size() > cap) return -(long)v.size();
    memcpy(out, v.data(), v.size());
    return (long)v.size();
}
LP_ABI_CATCH("lilliput_hip_progressive_encode_coefs", return -1)
