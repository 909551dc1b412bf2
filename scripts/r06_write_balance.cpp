// scripts/r06_write_balance.cpp -- analysis aid (not part of the product): how evenly does the WRITE pass's work spread over the lanes of a wave?
// A serial walk of the clean stream with the product's own lane logic (lp_huff_core.h) counts the symbols and blocks that start in every
// S-bit subsequence; waves are 64 consecutive subsequences. Output: mean and maximum of the per-lane step estimate per wave.
// build: g++ -O2 -std=c++17 -o /tmp/wb/wb scripts/r06_write_balance.cpp lilliput_amd/csrc/lp_jpeg_parse.cpp ; run: /tmp/wb/wb file.jpg S
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../lilliput_amd/csrc/lp_huff_core.h"
#include "../lilliput_amd/csrc/lp_jpeg_parse.h"

struct Mem {
    static constexpr int kRing = 8, kEvery = 2, kQuads = 1;
    const uint32_t* words; const LpHuffSet* hs; const uint32_t* rst;
    uint32_t peek_np(uint32_t np) { const uint32_t w1 = (uint32_t)(3 - ((int32_t)np >> 5)); const uint32_t hi = (np & 31u) ? words[w1 - 1u] : 0u; return (uint32_t)(((((uint64_t)hi) << 32) | words[w1]) >> (np & 31u)); }
    void reseek(uint32_t) {} void topup(uint32_t) {}
    bool any(bool p) const { return p; } bool any2(bool a, bool b) const { return a || b; }
    uint32_t lut(uint32_t t, uint32_t i) const { return hs->lut[t][i]; } uint32_t lut2(uint32_t i) const { return hs->lut2[i]; }
    uint32_t lutc(uint32_t t, uint32_t i) const { return hs->lut[t][i] | ((uint32_t)hs->lutm[t][i] << 16); }
    int32_t maxcode(uint32_t t, uint32_t l) const { return hs->maxcode[t][l]; } int32_t valoff(uint32_t t, uint32_t l) const { return hs->valoff[t][l]; }
    uint32_t val(uint32_t t, uint32_t i) const { return hs->vals[t][i & 255]; } uint32_t rst_bit(uint32_t k) const { return rst[k]; } void settle(uint32_t&) const {}
};

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t len = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(len); if (fread(d.data(), 1, len, f) != len) return 1; fclose(f);
    const uint32_t S = argc > 2 ? atoi(argv[2]) : 16384;
    static LpJpegHeader h;
    if (lp_jpeg_parse(d.data(), len, &h) || h.scan_path) return 2;
    const LpJpeg& img = h.j;
    const uint8_t* raw = d.data() + h.ecs_off;
    std::vector<uint8_t> clean; std::vector<uint32_t> rst;
    for (size_t q = 0; q < h.ecs_len; q++) {
        uint8_t c = raw[q], prev = q ? raw[q - 1] : 0, next = q + 1 < h.ecs_len ? raw[q + 1] : 0xD9;
        if (c == 0xFF) { if (next == 0) clean.push_back(0xFF); continue; }
        if (prev == 0xFF) { if (c == 0) continue; if (c >= 0xD0 && c <= 0xD7) { rst.push_back((uint32_t)clean.size() * 8); continue; } return 3; }
        clean.push_back(c);
    }
    const uint32_t total_bits = (uint32_t)clean.size() * 8, n_rst = (uint32_t)rst.size();
    std::vector<uint32_t> words((clean.size() + 3) / 4 + 64, 0);
    for (size_t q = 0; q < clean.size(); q++) words[q >> 2] |= (uint32_t)clean[q] << (24 - 8 * (q & 3));
    rst.push_back(0);
    LpImgCtx ic; ic.blkpack = (uint32_t)img.blkpack; ic.bpm = img.bpm; ic.n_rst = n_rst; ic.total_bits = total_bits; ic.total_blocks = img.total_blocks; ic.rst_blocks = img.dri * img.bpm; lp_ctx_tables(ic);
    Mem m{words.data(), &h.huff, rst.data()};
    LpLane<Mem> L(m, ic);
    L.start(0, 0);
    const uint32_t nsub = (total_bits + S - 1) / S;
    std::vector<uint32_t> sym(nsub, 0), blk(nsub, 0), wide(nsub, 0);
    uint32_t owner = 0, blocks = 0; unsigned long long nsym = 0;
    while (blocks < img.total_blocks && L.pos() < total_bits) {
        uint32_t pk = L.peek();
        if (L.z == 0) { if (L.restart_check(pk)) pk = L.peek(); owner = std::min(nsub - 1, L.pos() / S); blk[owner]++; }
        auto s = L.step<true>(pk);
        sym[owner]++; nsym++;
        if (s.has_val && !s.is_dc && (s.val < -127 || s.val > 127)) wide[owner]++;
        if (s.block_done) blocks++;
    }
    // per-lane step estimate of the WRITE loop: symbols + the steps a finished block waits for the next flush (every 4th step; mean 1.5) + the skipped partial block (~half a block)
    double tot_mean = 0, tot_max = 0, tot_sorted = 0; uint32_t nw = 0;
    std::vector<double> cost(nsub);
    for (uint32_t i = 0; i < nsub; i++) cost[i] = sym[i] + 1.5 * blk[i];
    for (uint32_t w = 0; w * 64 < nsub; w++) {
        double mx = 0, sm = 0; uint32_t n = 0;
        for (uint32_t l = w * 64; l < std::min(nsub, w * 64 + 64); l++) { mx = std::max(mx, cost[l]); sm += cost[l]; n++; }
        tot_mean += sm / n; tot_max += mx; nw++;
    }
    std::vector<double> sorted(cost); std::sort(sorted.begin(), sorted.end());
    for (uint32_t w = 0; w * 64 < nsub; w++) { double mx = 0; for (uint32_t l = w * 64; l < std::min(nsub, w * 64 + 64); l++) mx = std::max(mx, sorted[l]); tot_sorted += mx; }
    printf("%s: %u x %u, %u blocks, %llu symbols (%.2f bits / symbol, %.1f / block), S %u -> %u subsequences, %u waves\n", argv[1], img.width, img.height, img.total_blocks, nsym,
           (double)total_bits / nsym, (double)nsym / img.total_blocks, S, nsub, nw);
    printf("  per-wave steps: mean-lane %.0f, max-lane %.0f (x %.2f); lanes sorted by cost: max-lane %.0f (x %.2f)\n", tot_mean / nw, tot_max / nw, tot_max / tot_mean, tot_sorted / nw, tot_sorted / tot_mean);
    double mn = 1e9, mx = 0; for (auto c : cost) { mn = std::min(mn, c); mx = std::max(mx, c); }
    printf("  lane cost min %.0f max %.0f; p10 %.0f p50 %.0f p90 %.0f\n", mn, mx, sorted[nsub / 10], sorted[nsub / 2], sorted[nsub * 9 / 10]);
    return 0;
}
