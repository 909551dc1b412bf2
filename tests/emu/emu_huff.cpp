// tests/emu/emu_huff.cpp -- DEVELOPMENT AID (tests only): runs the per-lane Huffman decoding logic of
// lilliput_amd/csrc/lp_huff_core.h serially on the CPU, lane by lane, in the same pass structure the HIP
// kernels use (speculate -> verify rounds -> scan -> write). It lets `pytest -m "not gpu"` check the
// self-synchronisation / checkpoint / ownership logic against the oracle without a GPU.
// It is NOT linked into liblilliput_hip.so and is never used as a fallback.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../lilliput_amd/csrc/lp_huff_core.h"
#include "../../lilliput_amd/csrc/lp_jpeg_parse.h"
#include "../../lilliput_amd/csrc/lp_unstuff_core.h"

template <int R, int T, int Q>
struct HostMemT {
    static constexpr int kRing = R, kEvery = T, kQuads = Q;
    const uint32_t* words; // linear big-endian words of the clean stream
    const LpHuffSet* hs;
    const uint32_t* rst;
    // Ring emulation: the device keeps R words per lane in LDS and tops them up every T steps; the emulation tracks the
    // same window and reports when the lane logic fetches outside of it.
    uint32_t fill = 0, lowest = 0;
    bool* window_violation;
    uint32_t fetch1(uint32_t w)
    {
        if (w >= fill || w + R < fill) *window_violation = true;
        return words[w];
    }
    uint32_t peek_np(uint32_t np)
    {
        const uint32_t w1 = (uint32_t)(3 - ((int32_t)np >> 5));
        const uint32_t hi = (np & 31u) ? fetch1(w1 - 1u) : 0u; // on a word boundary the shift is 0 and the upper word is not looked at (the device reads whatever the ring holds there)
        return (uint32_t)(((((uint64_t)hi) << 32) | fetch1(w1)) >> (np & 31u));
    }
    void reseek(uint32_t w) { fill = (w & ~3u) + R; }
    void topup(uint32_t p) { const uint32_t w = p >> 5; for (int i = 0; i < Q; i++) if (fill + 4u <= w + R) fill += 4; }
    bool any(bool p) const { return p; }
    bool any2(bool a, bool b) const { return a || b; }
    uint32_t lut(uint32_t t, uint32_t i) const { if (step_count) ++*step_count; return hs->lut[t][i]; }
    uint32_t lut2(uint32_t i) const { return hs->lut2[i]; }
    uint32_t lutc(uint32_t t, uint32_t i) const { if (step_count) ++*step_count; return hs->lut[t][i] | ((uint32_t)hs->lutm[t][i] << 16); }
    unsigned long long* step_count = nullptr;
    int32_t maxcode(uint32_t t, uint32_t l) const { return hs->maxcode[t][l]; }
    int32_t valoff(uint32_t t, uint32_t l) const { return hs->valoff[t][l]; }
    uint32_t val(uint32_t t, uint32_t i) const { return hs->vals[t][i & 255]; }
    uint32_t rst_bit(uint32_t k) const { return rst[k]; }
    void settle(uint32_t&) const {}
};
typedef HostMemT<8, 2, 1> HostMem;       // geometry of the SPEC / VERIFY kernels
typedef HostMemT<8, 2, 1> HostMemWrite;  // geometry of the WRITE kernel

struct HostSink { // one slot, flushed at the wave-uniform flush points like the device sink
    int16_t blk[64];
    int16_t* coef;      // decode-order blocks
    int16_t* pending = nullptr;
    HostSink() { memset(blk, 0, sizeof(blk)); }
    void put_dc(int32_t v, bool on) { if (on) blk[0] = (int16_t)v; }
    void put(uint32_t nat, int32_t v) { blk[nat & 63] = (int16_t)v; }
    uint32_t blk0 = 0;  // the lane's first block
    void end_block(uint32_t bc, bool on) { if (on) pending = coef + (size_t)(blk0 + (bc >> 5) - 1u) * 64; }
    bool stalled() const { return pending != nullptr; }
    void flush() { if (pending) { memcpy(pending, blk, 128); memset(blk, 0, sizeof(blk)); pending = nullptr; } }
    void finish() {}
};

struct HostCk {
    LpCkptPk* rec;
    void record(uint32_t k, const LpCkptPk& c) { rec[k] = c; }
    uint32_t pos(uint32_t k) const { return rec[k].p; }
    LpCkptPk load(uint32_t k) const { return rec[k]; }
};

// statistics of the last emu_decode_coefs call: table lookups (= decode steps) of the SPEC pass over all subsequences
static unsigned long long g_spec_steps = 0;
extern "C" unsigned long long emu_last_spec_steps() { return g_spec_steps; }

extern "C" int emu_decode_coefs(const uint8_t* data, size_t len, uint32_t S, uint32_t C, int comp, int16_t* out, size_t cap_elems,
                                int* bw, int* bh, int* rounds, int* nsub_out, int* spec_hits)
{
    LpJpegHeader h;
    int rc = lp_jpeg_parse(data, len, &h);
    if (rc) return -rc;
    if (h.scan_path) return -17; // not a file the subsequence-parallel kernels take (or one the parser sends through the serial route: open end)
    const LpJpeg& img = h.j;
    if (comp >= img.ncomp) return -10;
    // unstuff (mirrors k_unstuff_*): keep data bytes, one FF per FF..FF00 run, drop RSTn and record boundaries
    const uint8_t* raw = data + h.ecs_off;
    size_t rl = h.ecs_len;
    std::vector<uint8_t> clean;
    std::vector<uint32_t> rst;
    bool rst_numbers_ok = true;
    for (size_t q = 0; q < rl; q++) {
        uint8_t c = raw[q], prev = q ? raw[q - 1] : 0, next = q + 1 < rl ? raw[q + 1] : 0xD9;
        if (c == 0xFF) { if (next == 0) clean.push_back(0xFF); continue; }
        if (prev == 0xFF) {
            if (c == 0) continue;
            if (c >= 0xD0 && c <= 0xD7) { // k_unstuff_scatter: the k-th marker must be RST(k mod 8)
                if ((c & 7u) != (rst.size() & 7u)) rst_numbers_ok = false;
                rst.push_back((uint32_t)clean.size() * 8);
                continue;
            }
            return -11; // unexpected marker (state error bit 0)
        }
        clean.push_back(c);
    }
    // k_unstuff_scan: exactly the restart markers the MCU count asks for (state error bit 3)
    if (!rst_numbers_ok || rst.size() != (size_t)(img.dri ? (img.mcus_x * img.mcus_y + img.dri - 1u) / img.dri - 1u : 0u)) return -16;
    uint32_t total_bits = (uint32_t)clean.size() * 8;
    std::vector<uint32_t> words((clean.size() + 3) / 4 + 64, 0);
    for (size_t q = 0; q < clean.size(); q++) words[q >> 2] |= (uint32_t)clean[q] << (24 - 8 * (q & 3));
    uint32_t n_rst = (uint32_t)rst.size();
    rst.push_back(0);
    bool violation = false;
    LpImgCtx ic;
    ic.blkpack = (uint32_t)img.blkpack; ic.bpm = img.bpm; ic.n_rst = n_rst; ic.total_bits = total_bits; ic.total_blocks = img.total_blocks; ic.rst_blocks = img.dri * img.bpm; lp_ctx_tables(ic);
    // the engine's schedule (lp_engine.cpp run_decode)
    const LpCkSched cs = lp_make_sched(S, C ? C : 256); // the engine's schedule (lp_engine.cpp run_decode)
    const uint32_t K = cs.K;
    uint32_t nsub = (total_bits + S - 1) / S;
    *nsub_out = (int)nsub;
    std::vector<LpCkptPk> ck((size_t)nsub * K);
    std::vector<LpSubState> spec_ex(nsub), ex(nsub), entry_used(nsub);
    std::vector<LpSubSum> spec_tot(nsub), tot(nsub);
    for (uint32_t i = 0; i < nsub; i++) {
        LpSubState e{i * S, 0};
        HostMem m{words.data(), &h.huff, rst.data(), 0, 0, &violation};
        HostCk hc{&ck[(size_t)i * K]};
        uint32_t sub_end = i * S + S < total_bits ? i * S + S : total_bits;
        if (i == 0) g_spec_steps = 0;
        m.step_count = &g_spec_steps;
        lp_spec_pass(m, ic, sub_end, e, cs, hc, &spec_ex[i], &spec_tot[i]);
        ex[i] = spec_ex[i];
        tot[i] = spec_tot[i];
        entry_used[i] = LpSubState{0xffffffffu, 0xffffffffu};
    }
    int r = 0, hits = 0;
    for (;;) {
        int changed = 0;
        std::vector<LpSubState> snap(ex); // Jacobi sweep: every lane sees the previous round's exits (worst case on a GPU)
        for (uint32_t i = 1; i < nsub; i++) {
            LpSubState e = snap[i - 1];
            if (lp_state_eq(e, entry_used[i])) continue;
            HostMem m{words.data(), &h.huff, rst.data(), 0, 0, &violation};
            HostCk hc{&ck[(size_t)i * K]};
            uint32_t sub_end = i * S + S < total_bits ? i * S + S : total_bits;
            LpSubState ne = ex[i];
            LpSubSum nt;
            lp_verify_pass(m, ic, sub_end, e, K, hc, spec_ex[i], spec_tot[i], &ne, &nt);
            tot[i] = nt;
            if (r == 0 && lp_state_eq(ne, spec_ex[i])) hits++;
            if (!lp_state_eq(ne, ex[i])) { ex[i] = ne; changed++; }
            entry_used[i] = e;
        }
        r++;
        if (!changed) break;
        if (r > 1000) return -12;
    }
    *rounds = r;
    *spec_hits = hits;
    std::vector<LpSubSum> prefix(nsub);
    LpSubSum acc;
    lp_sum_zero(acc);
    for (uint32_t i = 0; i < nsub; i++) { prefix[i] = acc; acc = lp_sum_combine(acc, tot[i]); }
    if (acc.nblk < img.total_blocks) return -13;
    static const uint8_t zz[80] = LP_ZIGZAG_INIT;
    std::vector<int16_t> all((size_t)img.total_blocks * 64, 0x7fff);
    HostSink sink;
    sink.coef = all.data();
    uint32_t written = 0;
    bool irregular = false;
    for (uint32_t i = 0; i < nsub; i++) {
        LpSubState e = i ? ex[i - 1] : LpSubState{0, 0};
        HostMemWrite m{words.data(), &h.huff, rst.data(), 0, 0, &violation};
        sink.blk0 = prefix[i].nblk;
        bool irr = false;
        written += lp_write_pass(m, ic, e, ex[i].p, prefix[i], zz, sink, &irr);
        irregular = irregular || irr;
    }
    if (irregular) return -16; // k_huff_write: the restart intervals do not hold exactly their MCUs (state error bit 3)
    if (written != img.total_blocks) return -14;
    {   // DC differences -> absolute values (k_dc_scan on the device; same lane logic, one range)
        std::vector<int16_t> dcs(img.total_blocks);
        for (uint32_t q = 0; q < img.total_blocks; q++) dcs[q] = all[(size_t)q * 64];
        int32_t pred[LP_MAX_COMP] = {0, 0, 0};
        lp_dc_walk(dcs.data(), 0, img.mcus_x * img.mcus_y, img.bpm, img.dri, img.blk_comp, pred, true);
        for (uint32_t q = 0; q < img.total_blocks; q++) all[(size_t)q * 64] = dcs[q];
    }
    if (violation) return -15; // the lane logic read outside the ring window the device would hold
    *bw = (int)img.bw[comp]; *bh = (int)img.bh[comp];
    size_t ne = (size_t)img.bw[comp] * img.bh[comp] * 64;
    if (ne > cap_elems) return -3;
    const uint32_t hs = img.hs[comp], vs = img.vs[comp];
    for (uint32_t by = 0; by < img.bh[comp]; by++)
        for (uint32_t bx = 0; bx < img.bw[comp]; bx++) {
            size_t blk = ((size_t)(by / vs) * img.mcus_x + bx / hs) * img.bpm + img.blk_first[comp] + (by % vs) * hs + (bx % hs);
            memcpy(out + ((size_t)by * img.bw[comp] + bx) * 64, all.data() + blk * 64, 128);
        }
    return 0;
}

// ---- progressive (SOF2) scans: lp_prog_core.h lane logic, scan after scan in file order ----
#include "../../lilliput_amd/csrc/lp_prog_core.h"

struct HostProgMem {
    const uint32_t* words;
    size_t nwords;
    const uint32_t* rst;
    const LpProgHuff* ht;
    int16_t* coef; // the image's blocks
    int16_t* cur = nullptr;
    uint32_t word(uint32_t w) const { return w < nwords ? words[w] : 0u; }
    uint32_t rst_bit(uint32_t k) const { return rst[k]; }
    uint32_t lut8(uint32_t s, uint32_t i) const { return ht->lut8[s][i]; }
    int32_t maxcode(uint32_t s, uint32_t l) const { return ht->maxcode[s][l]; }
    int32_t valoff(uint32_t s, uint32_t l) const { return ht->valoff[s][l]; }
    uint32_t val(uint32_t s, uint32_t i) const { return ht->vals[s][i]; }
    void st(uint32_t blk, uint32_t e, int32_t v) { coef[(size_t)blk * 64 + e] = (int16_t)v; }
    int32_t ld(uint32_t blk, uint32_t e) const { return coef[(size_t)blk * 64 + e]; }
    uint64_t open(uint32_t blk)
    {
        cur = coef + (size_t)blk * 64;
        uint64_t nz = 0;
        for (int k = 0; k < 64; k++) nz |= (uint64_t)(cur[k] != 0) << k;
        return nz;
    }
    int32_t get(uint32_t e) const { return cur[e]; }
    void set(uint32_t e, int32_t v) { cur[e] = (int16_t)v; }
    void close(uint32_t) {}
};

extern "C" int emu_decode_coefs_progressive(const uint8_t* data, size_t len, int comp, int16_t* out, size_t cap_elems, int* bw, int* bh, int* nscans)
{
    LpJpegHeader h;
    int rc = lp_jpeg_parse_opts(data, len, &h, *nscans < 0); // *nscans < 0 on entry: force the scan-by-scan walk on a baseline file
    if (rc) return -rc;
    if (!h.scan_path) return -20;
    const LpJpeg& img = h.j;
    if (comp >= img.ncomp) return -10;
    size_t nblk = 0;
    for (int c = 0; c < img.ncomp; c++) nblk += (size_t)img.bw[c] * img.bh[c];
    std::vector<int16_t> coef(nblk * 64, 0);
    for (const LpProgScanHost& sh : h.scans) {
        const uint8_t* raw = data + sh.ecs_off;
        const size_t rl = sh.ecs_len;
        std::vector<uint8_t> clean;
        std::vector<uint32_t> rst;
        for (size_t q = 0; q < rl; q++) { // as k_unstuff_* (see emu_decode_coefs)
            uint8_t c = raw[q], prev = q ? raw[q - 1] : 0, next = q + 1 < rl ? raw[q + 1] : 0xD9;
            if (c == 0xFF) { if (next == 0) clean.push_back(0xFF); continue; }
            if (prev == 0xFF) {
                if (c == 0) continue;
                if (c >= 0xD0 && c <= 0xD7) { rst.push_back((uint32_t)clean.size() * 8); continue; }
                return -11;
            }
            clean.push_back(c);
        }
        std::vector<uint32_t> words((clean.size() + 3) / 4 + 4, 0);
        for (size_t q = 0; q < clean.size(); q++) words[q >> 2] |= (uint32_t)clean[q] << (24 - 8 * (q & 3));
        const uint32_t n_rst = (uint32_t)rst.size();
        rst.push_back(0);
        HostProgMem m{words.data(), words.size(), rst.data(), &sh.tables, coef.data()};
        lp_prog_scan(m, sh.s, (uint32_t)clean.size() * 8, n_rst);
    }
    *nscans = (int)h.scans.size();
    *bw = (int)img.bw[comp]; *bh = (int)img.bh[comp];
    const size_t ne = (size_t)img.bw[comp] * img.bh[comp] * 64;
    if (ne > cap_elems) return -3;
    size_t base = 0;
    for (int c = 0; c < comp; c++) base += (size_t)img.bw[c] * img.bh[c];
    static const uint8_t zz[80] = LP_ZIGZAG_INIT;
    for (size_t q = 0; q < (size_t)img.bw[comp] * img.bh[comp]; q++)
        for (int e = 0; e < 64; e++) out[q * 64 + zz[e]] = coef[(base + q) * 64 + e]; // stored in zigzag order
    return 0;
}

// The word-arithmetic byte classifier of the unstuff kernels (lp_unstuff_core.h) against the byte-by-byte definition of the classes
// (T.81 B.1.1.5 / jdhuff.c jpeg_fill_bit_buffer) on `iters` pseudo-random 16-byte groups drawn from marker-heavy alphabets, with
// every neighbour byte and segment-end position. Returns the number of disagreements.
extern "C" long emu_unstuff_classify_check(long iters, uint32_t seed)
{
    static const uint8_t alphabet[] = {0xFF, 0x00, 0xD0, 0xD7, 0xD8, 0xCF, 0xD9, 0x01, 0x80, 0xFE, 0x7F, 0xF8};
    uint32_t x = seed ? seed : 1u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    long bad = 0;
    for (long it = 0; it < iters; it++) {
        uint8_t b[16];
        const int mode = (int)(it & 3);
        for (int j = 0; j < 16; j++) b[j] = mode == 0 ? (uint8_t)rnd() : alphabet[rnd() % (mode == 1 ? 12u : mode == 2 ? 4u : 2u)];
        const uint32_t prev = alphabet[rnd() % 12u], next = alphabet[rnd() % 12u];
        const uint32_t pos0 = (rnd() % 4u) * 16u, raw_len = rnd() % 8u == 0 ? rnd() % 100u : pos0 + 1000u;
        uint32_t km = 0, rm = 0, err = 0;
        for (int j = 0; j < 16; j++) {
            const uint32_t c = b[j], pv = j == 0 ? prev : b[j - 1];
            uint32_t nx = j == 15 ? next : b[j + 1];
            const bool in = pos0 + j < raw_len;
            if (pos0 + j + 1 >= raw_len) nx = 0xD9; // nothing follows the last byte
            bool keep, rst = false;
            if (c == 0xFF) keep = nx == 0x00;
            else if (pv == 0xFF) { keep = false; rst = c >= 0xD0 && c <= 0xD7; if (in && c != 0 && !rst) err |= 1u; }
            else keep = true;
            if (in && keep) km |= 1u << j;
            if (in && rst) rm |= 1u << j;
        }
        uint32_t w[4], km2 = 0, rm2 = 0, err2 = 0;
        memcpy(w, b, 16);
        lp_unstuff_classify(w, prev, next, pos0, raw_len, km2, rm2, err2);
        if (km != km2 || rm != rm2 || err != err2) bad++;
        if (pos0 == 0) { // the HEAD form (a segment that starts `head` bytes into its first 16-byte group): as if those bytes did not exist
            const uint32_t head = 1u + rnd() % 15u, len2 = head + (rnd() % 4u == 0 ? rnd() % 40u : 1000u);
            uint32_t km3 = 0, rm3 = 0, err3 = 0;
            for (int j = (int)head; j < 16; j++) {
                const uint32_t c = b[j], pv = j == (int)head ? 0u : b[j - 1];
                uint32_t nx = j == 15 ? next : b[j + 1];
                const bool in = (uint32_t)j < len2;
                if ((uint32_t)j + 1 >= len2) nx = 0xD9;
                bool keep, rst = false;
                if (c == 0xFF) keep = nx == 0x00;
                else if (pv == 0xFF) { keep = false; rst = c >= 0xD0 && c <= 0xD7; if (in && c != 0 && !rst) err3 |= 1u; }
                else keep = true;
                if (in && keep) km3 |= 1u << j;
                if (in && rst) rm3 |= 1u << j;
            }
            uint32_t K3[4], R3[4], e3 = 0, km4 = 0, rm4 = 0;
            lp_unstuff_classify_masks<true, true>(w, 0u, next, 0u, len2, K3, R3, e3, head);
            for (int i = 0; i < 4; i++) { km4 |= lp_movemask4(K3[i]) << (4 * i); rm4 |= lp_movemask4(R3[i]) << (4 * i); }
            if (km3 != km4 || rm3 != rm4 || err3 != e3) bad++;
        }
        if (pos0 + 17u <= raw_len) { // the range-free form used for every chunk but a segment's last must agree where it applies
            uint32_t K1[4], R1[4], K2[4], R2[4], e1 = 0, e2 = 0;
            lp_unstuff_classify_masks<true>(w, prev, next, pos0, raw_len, K1, R1, e1);
            lp_unstuff_classify_masks<false>(w, prev, next, pos0, raw_len, K2, R2, e2);
            if (memcmp(K1, K2, 16) || memcmp(R1, R2, 16) || e1 != e2) bad++;
        }
    }
    return bad;
}

// The lookup tables the parser builds for a file: lut (one symbol per entry) and lutm (the counting passes' multi-symbol entries) --
// tests/test_host_logic.py checks lutm against an independent walk of lut.
extern "C" int emu_huff_tables(const uint8_t* data, size_t len, uint16_t* lut, uint16_t* lutm, int* lut_bits)
{
    LpJpegHeader h;
    int rc = lp_jpeg_parse(data, len, &h);
    if (rc) return -rc;
    if (h.scan_path) return -17;
    memcpy(lut, h.huff.lut, sizeof(h.huff.lut));
    memcpy(lutm, h.huff.lutm, sizeof(h.huff.lutm));
    *lut_bits = LP_LUT_BITS;
    return (int)h.j.ncomp;
}
