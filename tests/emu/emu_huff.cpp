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

struct HostMem {
    const uint32_t* words; // lane-interleaved layout (lp_clean_addr)
    const LpHuffSet* hs;
    const uint32_t* rst;
    uint32_t wps;
    uint32_t word(uint32_t w) const { return words[lp_clean_addr(w, wps)]; }
    bool any(bool p) const { return p; }
    uint32_t lut(uint32_t t, uint32_t i) const { return hs->lut[t][i]; }
    int32_t maxcode(uint32_t t, uint32_t l) const { return hs->maxcode[t][l]; }
    int32_t valoff(uint32_t t, uint32_t l) const { return hs->valoff[t][l]; }
    uint32_t val(uint32_t t, uint32_t i) const { return hs->vals[t][i & 255]; }
    uint32_t rst_bit(uint32_t k) const { return rst[k]; }
};

struct HostSink { // one slot, flushed at the wave-uniform flush points like the device sink
    int16_t blk[64];
    int16_t* coef[3];
    const LpJpeg* img;
    int16_t* pending = nullptr;
    HostSink() { memset(blk, 0, sizeof(blk)); }
    void put(uint32_t nat, int32_t v) { blk[nat & 63] = (int16_t)v; }
    void end_block(uint32_t c, uint32_t bx, uint32_t by) { pending = coef[c] + ((size_t)by * img->bw[c] + bx) * 64; }
    bool stalled() const { return pending != nullptr; }
    void flush() { if (pending) { memcpy(pending, blk, 128); memset(blk, 0, sizeof(blk)); pending = nullptr; } }
};

extern "C" int emu_decode_coefs(const uint8_t* data, size_t len, uint32_t S, uint32_t C, int comp, int16_t* out, size_t cap_elems,
                                int* bw, int* bh, int* rounds, int* nsub_out, int* spec_hits)
{
    LpJpegHeader h;
    int rc = lp_jpeg_parse(data, len, &h);
    if (rc) return -rc;
    const LpJpeg& img = h.j;
    if (comp >= img.ncomp) return -10;
    // unstuff (mirrors k_unstuff_*): keep data bytes, one FF per FF..FF00 run, drop RSTn and record boundaries
    const uint8_t* raw = data + h.ecs_off;
    size_t rl = h.ecs_len;
    std::vector<uint8_t> clean;
    std::vector<uint32_t> rst;
    for (size_t q = 0; q < rl; q++) {
        uint8_t c = raw[q], prev = q ? raw[q - 1] : 0, next = q + 1 < rl ? raw[q + 1] : 0xD9;
        if (c == 0xFF) { if (next == 0) clean.push_back(0xFF); continue; }
        if (prev == 0xFF) {
            if (c == 0) continue;
            if (c >= 0xD0 && c <= 0xD7) { rst.push_back((uint32_t)clean.size() * 8); continue; }
            return -11; // unexpected marker
        }
        clean.push_back(c);
    }
    uint32_t total_bits = (uint32_t)clean.size() * 8;
    const uint32_t wps = S / 32;
    const size_t nwords = ((clean.size() + 3) / 4 + 32 + (size_t)64 * wps - 1) / ((size_t)64 * wps) * ((size_t)64 * wps);
    std::vector<uint32_t> words(nwords, 0);
    for (size_t q = 0; q < clean.size(); q++) words[lp_clean_addr((uint32_t)(q >> 2), wps)] |= (uint32_t)clean[q] << (24 - 8 * (q & 3));
    rst.push_back(0);
    HostMem m{words.data(), &h.huff, rst.data(), wps};
    uint32_t n_rst = (uint32_t)rst.size() - 1;
    uint32_t K = S / C;
    uint32_t nsub = (total_bits + S - 1) / S;
    *nsub_out = (int)nsub;
    std::vector<LpCkpt> ck((size_t)nsub * K);
    std::vector<LpSubState> ex(nsub), entry_used(nsub);
    std::vector<LpSubSum> tot(nsub);
    for (uint32_t i = 0; i < nsub; i++) {
        LpSubState e{i * S, 0};
        ex[i].p = 0xffffffffu; ex[i].bz = 0;
        lp_count_pass(m, img, n_rst, total_bits, i, S, C, K, false, e, &ck[(size_t)i * K], &ex[i], &tot[i]);
        entry_used[i] = e;
    }
    int r = 0, hits = 0;
    for (;;) {
        int changed = 0;
        std::vector<LpSubState> snap(ex); // Jacobi sweep: every lane sees the previous round's exits (worst case on a GPU)
        for (uint32_t i = 1; i < nsub; i++) {
            LpSubState e = snap[i - 1];
            if (lp_state_eq(e, entry_used[i])) { if (r == 0) hits++; continue; }
            changed += lp_count_pass(m, img, n_rst, total_bits, i, S, C, K, true, e, &ck[(size_t)i * K], &ex[i], &tot[i]) ? 1 : 0;
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
    HostSink sink;
    sink.img = &img;
    std::vector<int16_t> cbuf[3];
    for (int c = 0; c < img.ncomp; c++) { cbuf[c].assign((size_t)img.bw[c] * img.bh[c] * 64, 0x7fff); sink.coef[c] = cbuf[c].data(); }
    uint32_t written = 0;
    for (uint32_t i = 0; i < nsub; i++) {
        LpSubState e = i ? ex[i - 1] : LpSubState{0, 0};
        written += lp_write_pass(m, img, n_rst, total_bits, e, ex[i].p, prefix[i], zz, sink);
    }
    if (written != img.total_blocks) return -14;
    *bw = (int)img.bw[comp]; *bh = (int)img.bh[comp];
    size_t ne = cbuf[comp].size();
    if (ne > cap_elems) return -3;
    memcpy(out, cbuf[comp].data(), ne * 2);
    return 0;
}
