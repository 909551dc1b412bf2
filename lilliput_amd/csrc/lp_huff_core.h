// lp_huff_core.h -- per-lane baseline-JPEG Huffman decoding core (S1 in SURVEY.md 2a).
//
// Replaces libjpeg-turbo's serial jdhuff.c decode_mcu loop that the reference reaches through
// opencv_decoder_read_data (/root/reference/opencv.cpp:166-171) with a subsequence-parallel,
// self-synchronising decoder:
//   * the unstuffed entropy-coded stream is cut into subsequences of S bits, one per lane;
//   * COUNT pass, speculative: every lane decodes its own subsequence from a guessed state
//     (block 0 of an MCU, coefficient 0) and records its state + block/DC sums at K checkpoints;
//   * COUNT pass, verify: lane i restarts from lane i-1's exit state and decodes only until its
//     state coincides with a recorded checkpoint (JPEG streams self-synchronise after ~10 blocks);
//     repeated until no exit state changes (normally one round);
//   * an exclusive scan over the per-subsequence sums gives every lane its first block index and
//     its DC predictors;
//   * WRITE pass: every lane decodes the blocks that START inside its subsequence and emits their
//     64 coefficients (absolute DC), so each block is written by exactly one lane.
// Restart markers (DRI) are forced synchronisation points: at a block start fewer than 8 one-bits
// away from the next restart boundary the lane jumps to the boundary and resets its state.
//
// The code is host+device so that tests/emu can run the same lane logic on the CPU (development
// aid only -- the product never falls back to it).
#pragma once
#include "lp_types.h"

#if defined(__HIPCC__)
#define LP_HD __host__ __device__ __forceinline__
#else
#define LP_HD inline
#endif

// zigzag index -> natural (row-major) index; 16 guard entries like libjpeg's jpeg_natural_order
#define LP_ZIGZAG_INIT                                                                                    \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, \
     13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, \
     38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, \
     63, 63}

// Layout of the unstuffed stream in HBM: the stream is cut into subsequences of WPS = S/32 words; groups of 64
// consecutive subsequences are stored word-interleaved, so that when the 64 lanes of a wave each fetch "their next
// word" the accesses fall into a few contiguous 256-byte rows (coalesced HBM row loads) instead of 64 cache lines.
LP_HD uint32_t lp_clean_addr(uint32_t widx, uint32_t wps)
{
    uint32_t s = widx / wps, w = widx - s * wps;
    return ((s >> 6) * wps + w) * 64u + (s & 63u);
}

// Memory policy M must provide:
//   uint32_t word(uint32_t widx)            big-endian-corrected 32-bit word of the clean stream (applies lp_clean_addr)
//   bool any(bool)                          wave vote (host emulation: identity)
//   uint32_t lut(uint32_t tbl, uint32_t i)  first-level Huffman lookup
//   int32_t maxcode(tbl, l), valoff(tbl, l); uint32_t val(tbl, i)   canonical tables for long codes
//   uint32_t rst_bit(uint32_t k)            bit position of the k-th restart boundary
template <class M>
struct LpLane {
    const M& m;
    const LpJpeg& img;
    uint64_t buf;       // next bits, left aligned
    int avail;          // valid bits in buf
    uint32_t widx;      // next word to load
    uint32_t pending;   // == word(widx), fetched one iteration ahead by every lane of the wave at once
    uint32_t p;         // bit position of the next unread bit
    uint32_t b, z;      // block-in-MCU, zigzag index
    uint32_t next_rst;  // bit position of the next restart boundary (stream end when none left)
    uint32_t rst_k;     // index of that boundary
    uint32_t n_rst;
    uint32_t total_bits;

    LP_HD LpLane(const M& m_, const LpJpeg& img_, uint32_t n_rst_, uint32_t total_bits_)
        : m(m_), img(img_), buf(0), avail(0), widx(0), pending(0), p(0), b(0), z(0), next_rst(0), rst_k(0), n_rst(n_rst_),
          total_bits(total_bits_) {}

    LP_HD void seek(uint32_t pos)
    {
        p = pos;
        widx = pos >> 5;
        uint32_t off = pos & 31;
        uint64_t w0 = m.word(widx), w1 = m.word(widx + 1);
        buf = ((w0 << 32) | w1) << off;
        avail = 64 - (int)off;
        widx += 2;
        pending = m.word(widx);
    }
    // Wave-uniform fetch of the next word: issued by ALL lanes every iteration, consumed (maybe) one iteration later,
    // so the load latency overlaps a whole decode step and never sits behind a divergent branch.
    LP_HD void prefetch() { pending = m.word(widx); }
    LP_HD void start(uint32_t pos, uint32_t bz)
    {
        seek(pos);
        b = bz >> 8;
        z = bz & 255;
        // first restart boundary at or after pos (binary search; n_rst == 0 -> stream end)
        uint32_t lo = 0, hi = n_rst;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (m.rst_bit(mid) < pos) lo = mid + 1; else hi = mid;
        }
        rst_k = lo;
        next_rst = lo < n_rst ? m.rst_bit(lo) : total_bits;
    }
    LP_HD void refill()
    {
        if (avail <= 32) {
            buf |= (uint64_t)pending << (32 - avail);
            avail += 32;
            widx++;
        }
    }
    LP_HD void consume(uint32_t n)
    {
        buf <<= n;
        avail -= (int)n;
        p += n;
    }
    LP_HD uint32_t state_bz() const { return (b << 8) | z; }

    // At a block start: detect the end of a restart interval (or of the stream). Returns true when
    // the lane jumped to the boundary (DC predictors must be reset by the caller).
    LP_HD bool restart_check()
    {
        int32_t rem = (int32_t)(next_rst - p);
        if (rem >= 8) return false;
        bool jump = rem <= 0;
        if (!jump) jump = (uint32_t)(buf >> (64 - rem)) == ((1u << rem) - 1u);
        if (!jump) return false;
        uint32_t target = next_rst;
        if (rst_k < n_rst) {
            rst_k++;
            next_rst = rst_k < n_rst ? m.rst_bit(rst_k) : total_bits;
        } else {
            next_rst = 0x7fffffffu; // past the end of the stream: nothing left
        }
        seek(target);
        b = 0;
        z = 0;
        return true;
    }

    // Decode one Huffman symbol (+ its extra bits). On return:
    //   is_dc, k = zigzag index of the coefficient (valid when has_val), val, block_done.
    struct Sym { bool is_dc; bool has_val; bool block_done; uint32_t k; int32_t val; uint32_t comp; };
    LP_HD Sym step()
    {
        Sym r;
        r.is_dc = (z == 0);
        r.comp = img.blk_comp[b];
        uint32_t tbl = r.is_dc ? img.dc_tbl[r.comp] : img.ac_tbl[r.comp];
        uint32_t top = (uint32_t)(buf >> 48);
        uint32_t e = m.lut(tbl, top >> (16 - LP_LUT_BITS));
        uint32_t len = e >> 8, sym = e & 255;
        if (len == 0) { // code longer than LP_LUT_BITS: canonical search (T.81 F.2.2.3)
            len = 16;
            sym = 0;
            for (uint32_t l = LP_LUT_BITS + 1; l <= 16; l++) {
                int32_t code = (int32_t)(top >> (16 - l));
                if (code <= m.maxcode(tbl, l)) {
                    len = l;
                    sym = m.val(tbl, (uint32_t)(code + m.valoff(tbl, l)));
                    break;
                }
            }
        }
        consume(len);
        uint32_t s = sym & 15, run = sym >> 4;
        if (r.is_dc) run = 0;
        int32_t v = 0;
        if (s) {
            uint32_t x = (uint32_t)(buf >> (64 - s));
            consume(s);
            v = (x >> (s - 1)) ? (int32_t)x : (int32_t)x - (int32_t)((1u << s) - 1u);
        }
        r.val = v;
        if (r.is_dc) {
            r.has_val = true;
            r.k = 0;
            z = 1;
        } else if (s == 0) {
            r.has_val = false;
            r.k = 0;
            z = (run == 15) ? z + 16 : 64;
        } else {
            z += run;
            r.k = z;
            r.has_val = z < 64;
            z += 1;
        }
        r.block_done = z >= 64;
        if (r.block_done) {
            z = 0;
            b = (b + 1 == img.bpm) ? 0 : b + 1;
        }
        return r;
    }
};

LP_HD bool lp_state_eq(const LpSubState& a, const LpSubState& b) { return a.p == b.p && a.bz == b.bz; }

LP_HD void lp_sum_zero(LpSubSum& s)
{
    s.nblk = 0;
    s.nreset = 0;
    for (int c = 0; c < LP_MAX_COMP; c++) s.dc[c] = 0;
}

// a then b (b later in the stream)
LP_HD LpSubSum lp_sum_combine(const LpSubSum& a, const LpSubSum& b)
{
    LpSubSum r;
    r.nblk = a.nblk + b.nblk;
    r.nreset = a.nreset + b.nreset;
    for (int c = 0; c < LP_MAX_COMP; c++) r.dc[c] = b.nreset ? b.dc[c] : a.dc[c] + b.dc[c];
    return r;
}

// COUNT pass for subsequence `sub` of one image.
//   verify == false: speculative decode from (sub*S, b=0, z=0) (sub 0: the true start).
//   verify == true : decode from `entry` (= exit state of subsequence sub-1) until the state matches a
//                    recorded checkpoint, then splice the recorded remainder.
// ckpt[K], *exit_st, *total are this subsequence's records (read+written). Returns true when *exit_st changed.
template <class M>
LP_HD bool lp_count_pass(const M& m, const LpJpeg& img, uint32_t n_rst, uint32_t total_bits, uint32_t sub, uint32_t S,
                         uint32_t C, uint32_t K, bool verify, LpSubState entry, LpCkpt* ckpt, LpSubState* exit_st,
                         LpSubSum* total)
{
    LpLane<M> L(m, img, n_rst, total_bits);
    uint32_t sub_begin = sub * S;
    uint32_t sub_end = sub_begin + S;
    if (sub_end > total_bits) sub_end = total_bits;
    L.start(entry.p, entry.bz);
    LpSubSum sum;
    lp_sum_zero(sum);
    uint32_t k = 0;
    bool done = false, spliced = false;
    // Wave-uniform loop: every lane executes the same instruction stream; finished lanes are predicated off.
    while (m.any(!done)) {
        L.refill();
        L.prefetch();
        if (done) continue;
        if (L.z == 0 && L.restart_check()) {
            sum.nreset++;
            for (int c = 0; c < LP_MAX_COMP; c++) sum.dc[c] = 0;
        }
        LpSubState st;
        st.p = L.p;
        st.bz = L.state_bz();
        while (k < K && L.p >= sub_begin + k * C) {
            if (verify && lp_state_eq(ckpt[k].st, st)) {
                // synchronised with the recorded trajectory at checkpoint k: splice.
                LpSubSum rec = ckpt[k].sum, tot = *total;
                LpSubSum nt;
                nt.nblk = sum.nblk + (tot.nblk - rec.nblk);
                nt.nreset = sum.nreset + (tot.nreset - rec.nreset);
                bool no_reset_after = tot.nreset == rec.nreset;
                for (int c = 0; c < LP_MAX_COMP; c++) nt.dc[c] = no_reset_after ? sum.dc[c] + (tot.dc[c] - rec.dc[c]) : tot.dc[c];
                // re-base the recorded checkpoints k.. on the new prefix
                for (uint32_t j = k; j < K; j++) {
                    LpSubSum cj = ckpt[j].sum;
                    bool same_seg = cj.nreset == rec.nreset;
                    cj.nblk = sum.nblk + (cj.nblk - rec.nblk);
                    for (int c = 0; c < LP_MAX_COMP; c++) if (same_seg) cj.dc[c] = sum.dc[c] + (cj.dc[c] - rec.dc[c]);
                    cj.nreset = sum.nreset + (cj.nreset - rec.nreset);
                    ckpt[j].sum = cj;
                }
                *total = nt;
                done = true;
                spliced = true;
                break;
            }
            ckpt[k].st = st;
            ckpt[k].sum = sum;
            k++;
        }
        if (done) continue;
        if (L.p >= sub_end) { done = true; continue; }
        if (L.z == 0) sum.nblk++;
        typename LpLane<M>::Sym s = L.step();
        if (s.is_dc) sum.dc[s.comp] += s.val;
    }
    if (spliced) return false; // exit state unchanged
    LpSubState ne;
    ne.p = L.p;
    ne.bz = L.state_bz();
    bool changed = !lp_state_eq(ne, *exit_st);
    *exit_st = ne;
    *total = sum;
    return changed;
}

// WRITE pass for one subsequence. Sink S must provide:
//   void put(uint32_t natural_idx, int32_t v);          store one coefficient of the block being decoded
//   void end_block(uint32_t comp, uint32_t bx, uint32_t by);   the block is complete (queued for flushing)
//   bool stalled();                                      no free slot: the lane must wait for the next flush
//   void flush();                                        wave-uniform: write out every queued block
// Returns the number of blocks written.
#ifndef LP_FLUSH_EVERY
#define LP_FLUSH_EVERY 4
#endif
template <class M, class Sink>
LP_HD uint32_t lp_write_pass(const M& m, const LpJpeg& img, uint32_t n_rst, uint32_t total_bits, LpSubState entry,
                             uint32_t end_p, const LpSubSum& prefix, const uint8_t* zigzag, Sink& sink)
{
    LpLane<M> L(m, img, n_rst, total_bits);
    L.start(entry.p, entry.bz);
    uint32_t blk = prefix.nblk;
    int32_t pred[LP_MAX_COMP];
    for (int c = 0; c < LP_MAX_COMP; c++) pred[c] = prefix.dc[c];
    bool writing = false, done = false;
    uint32_t written = 0, iter = 0;
    // MCU coordinates of the current block, maintained incrementally
    uint32_t mcu = blk / img.bpm;
    uint32_t mx = mcu % img.mcus_x, my = mcu / img.mcus_x;
    while (m.any(!done)) {
        L.refill();
        L.prefetch();
        if ((++iter % LP_FLUSH_EVERY) == 0) sink.flush();
        if (done || sink.stalled()) continue;
        if (L.z == 0) {
            if (L.restart_check())
                for (int c = 0; c < LP_MAX_COMP; c++) pred[c] = 0;
            if (L.p >= end_p || blk >= img.total_blocks) { done = true; continue; }
            writing = true;
        } else if (L.p >= total_bits) { done = true; continue; } // truncated stream
        uint32_t bcur = L.b;
        typename LpLane<M>::Sym s = L.step();
        if (writing) {
            if (s.is_dc) {
                pred[s.comp] += s.val;
                sink.put(0, pred[s.comp]);
            } else if (s.has_val) {
                sink.put(zigzag[s.k], s.val);
            }
            if (s.block_done) {
                uint32_t c = img.blk_comp[bcur];
                sink.end_block(c, mx * img.hs[c] + img.blk_h[bcur], my * img.vs[c] + img.blk_v[bcur]);
                written++;
                blk++;
                if (L.b == 0) { // wrapped to the next MCU
                    mx++;
                    if (mx == img.mcus_x) { mx = 0; my++; }
                }
            }
        }
    }
    sink.flush();
    return written;
}
