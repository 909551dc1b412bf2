// lp_huff_core.h -- per-lane baseline-JPEG Huffman decoding core (S1 in SURVEY.md 2a).
//
// Replaces libjpeg-turbo's serial jdhuff.c decode_mcu loop that the reference reaches through
// opencv_decoder_read_data (/root/reference/opencv.cpp:166-171) with a subsequence-parallel,
// self-synchronising decoder:
//   * the unstuffed entropy-coded stream is cut into subsequences of S bits, one per lane;
//   * SPEC pass: every lane decodes its own subsequence from a guessed state (block 0 of an MCU,
//     coefficient 0) and records its state + block/DC sums at K checkpoints. Checkpoints are taken at
//     fixed ITERATIONS of the decode loop, so all 64 lanes of a wave record together (coalesced stores,
//     no divergent "some lane crossed a boundary" path in the hot loop);
//   * VERIFY pass: lane i restarts from lane i-1's exit state and decodes only until its state
//     coincides with a recorded checkpoint of subsequence i (JPEG streams self-synchronise after ~10
//     blocks), then splices the recorded remainder; repeated until no exit state changes;
//   * an exclusive scan over the per-subsequence sums gives every lane its first block index and
//     its DC predictors;
//   * WRITE pass: every lane decodes the blocks that START inside its subsequence and emits their
//     64 coefficients (absolute DC) in decode order, so each block is written by exactly one lane.
// Restart markers (DRI) are forced synchronisation points: at a block start fewer than 8 one-bits
// away from the next restart boundary the lane jumps to the boundary and resets its state.
//
// Hot-loop rules (measured on MI355X: the first version spent 64 % of its wave cycles in s_waitcnt):
//   * no global load per symbol: the bit reader pulls words from a per-lane ring that the memory
//     policy keeps topped up at wave-uniform iterations (device: LDS ring filled with 16-byte loads);
//   * no per-lane-indexed descriptor loads: block -> (component, DC table, AC table) comes from a packed
//     64-bit value held in scalar registers;
//   * events that are rare per lane but frequent per wave (checkpoints, long codes) must not run serial code:
//     checkpoints are wave-uniform, long codes take one second-level table lookup.
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

// Ring geometry of the bit reader. One decode step consumes at most 16 (code) + 15 (extra bits) = 31 bits, so
// LP_TOPUP_EVERY steps consume at most 8 words; after a top-up at most 3 ring words are free (16-byte granularity);
// the reader also looks one word ahead: 8 + 3 + 1 <= LP_RING_WORDS.
#define LP_RING_WORDS 16
#define LP_TOPUP_EVERY 8
#define LP_TOPUP_QUADS 2

// Per-image values every lane of a workgroup shares (scalar registers on the device).
struct LpImgCtx {
    uint64_t blkpack;       // LpJpeg::blkpack
    uint32_t bpm;
    uint32_t n_rst;         // restart boundaries found by the unstuff kernels
    uint32_t total_bits;    // length of the clean stream
    uint32_t total_blocks;
};

// Memory policy M must provide (per lane object, non-const):
//   uint32_t fetch(uint32_t widx)           big-endian-corrected word widx of the clean stream (must be inside the ring window)
//   void reseek(uint32_t widx)              the lane jumps: make [widx, widx + LP_RING_WORDS - 3) fetchable
//   void topup(uint32_t widx)               wave-uniform call every LP_TOPUP_EVERY steps: words below widx are dead, refill
//   bool any(bool)                          wave vote (host emulation: identity)
//   uint32_t lut(uint32_t tbl, uint32_t i), lut2(tbl, i), base2(tbl)
//   int32_t maxcode(tbl, l), valoff(tbl, l); uint32_t val(tbl, i)   canonical tables (third level, corrupt streams / huge tables)
//   uint32_t rst_bit(uint32_t k)            bit position of the k-th restart boundary
template <class M>
struct LpLane {
    M& m;
    const LpImgCtx& ic;
    uint64_t buf;       // next bits, left aligned
    int32_t avail;      // valid bits in buf
    uint32_t widx;      // next word to pull into buf
    uint32_t pending;   // == word(widx), fetched one step ahead
    uint32_t p;         // bit position of the next unread bit
    uint32_t b, z;      // block-in-MCU, zigzag index
    uint32_t next_rst;  // bit position of the next restart boundary (stream end when none left)
    uint32_t rst_k;     // index of that boundary

    LP_HD LpLane(M& m_, const LpImgCtx& ic_) : m(m_), ic(ic_), buf(0), avail(0), widx(0), pending(0), p(0), b(0), z(0), next_rst(0), rst_k(0) {}

    LP_HD void seek(uint32_t pos)
    {
        p = pos;
        widx = pos >> 5;
        const uint32_t off = pos & 31;
        m.reseek(widx);
        const uint64_t w0 = m.fetch(widx), w1 = m.fetch(widx + 1);
        buf = ((w0 << 32) | w1) << off;
        avail = 64 - (int32_t)off;
        widx += 2;
        pending = m.fetch(widx);
    }
    LP_HD void start(uint32_t pos, uint32_t bz)
    {
        seek(pos);
        b = bz >> 8;
        z = bz & 255;
        rst_k = 0;
        next_rst = ic.total_bits;
        if (ic.n_rst) { // first restart boundary at or after pos (binary search)
            uint32_t lo = 0, hi = ic.n_rst;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (m.rst_bit(mid) < pos) lo = mid + 1; else hi = mid;
            }
            rst_k = lo;
            next_rst = lo < ic.n_rst ? m.rst_bit(lo) : ic.total_bits;
        }
    }
    // Once per step, by every lane: move the look-ahead word into the bit buffer when there is room, look ahead again.
    LP_HD void refill()
    {
        if (avail <= 32) {
            buf |= (uint64_t)pending << (32 - avail);
            avail += 32;
            widx++;
        }
        pending = m.fetch(widx);
    }
    LP_HD uint32_t state_bz() const { return (b << 8) | z; }

    // At a block start: detect the end of a restart interval (or of the stream). Returns true when
    // the lane jumped to the boundary (DC predictors must be reset by the caller).
    LP_HD bool restart_check()
    {
        const int32_t rem = (int32_t)(next_rst - p);
        if (rem >= 8) return false;
        bool jump = rem <= 0;
        if (!jump) jump = (uint32_t)(buf >> (64 - rem)) == ((1u << rem) - 1u);
        if (!jump) return false;
        const uint32_t target = next_rst;
        if (rst_k < ic.n_rst) {
            rst_k++;
            next_rst = rst_k < ic.n_rst ? m.rst_bit(rst_k) : ic.total_bits;
        } else {
            next_rst = 0x7fffffffu; // past the end of the stream: nothing left
        }
        seek(target);
        b = 0;
        z = 0;
        return true;
    }

    // Second/third level of the code lookup (codes longer than LP_LUT_BITS).
    LP_HD uint32_t long_code(uint32_t tbl, uint32_t top)
    {
        const uint32_t idx = top - m.base2(tbl);
        uint32_t e = idx < LP_LUT2_SIZE ? m.lut2(tbl, idx) : 0u;
        if ((e >> 8) == 0) { // canonical search (T.81 F.2.2.3, jdhuff.c jpeg_huff_decode): corrupt prefix or a table with a very wide tail
            uint32_t len = 16, sym = 0;
            for (uint32_t l = LP_LUT_BITS + 1; l <= 16; l++) {
                const int32_t code = (int32_t)(top >> (16 - l));
                if (code <= m.maxcode(tbl, l)) {
                    len = l;
                    sym = m.val(tbl, (uint32_t)(code + m.valoff(tbl, l)));
                    break;
                }
            }
            e = (len << 8) | sym;
        }
        return e;
    }

    // Decode one Huffman symbol (+ its extra bits). On return:
    //   is_dc, k = zigzag index of the coefficient (valid when has_val), val, block_done, comp.
    struct Sym { bool is_dc; bool has_val; bool block_done; uint32_t k; int32_t val; uint32_t comp; };
    LP_HD Sym step()
    {
        Sym r;
        const uint32_t nib = (uint32_t)(ic.blkpack >> (b * 4u)) & 15u;
        r.is_dc = (z == 0);
        r.comp = nib & 3u;
        const uint32_t tbl = r.is_dc ? ((nib >> 2) & 1u) : 2u + (nib >> 3);
        const uint32_t top = (uint32_t)(buf >> 48);
        uint32_t e = m.lut(tbl, top >> (16 - LP_LUT_BITS));
        if ((e >> 8) == 0) e = long_code(tbl, top);
        const uint32_t len = e >> 8, sym = e & 255u;
        buf <<= len;
        const uint32_t s = sym & 15u;
        const uint32_t run = r.is_dc ? 0u : sym >> 4;
        const uint32_t x = (uint32_t)((buf >> 1) >> (63u - s)); // next s bits (0 when s == 0)
        buf <<= s;
        const uint32_t used = len + s;
        avail -= (int32_t)used;
        p += used;
        // HUFF_EXTEND: values whose first bit is 0 are negative
        r.val = x < ((1u << s) >> 1) ? (int32_t)x - (int32_t)((1u << s) - 1u) : (int32_t)x;
        uint32_t zn;
        if (r.is_dc) {
            r.has_val = true;
            r.k = 0;
            zn = 1;
        } else if (s == 0) {
            r.has_val = false;
            r.k = 0;
            zn = (run == 15) ? z + 16 : 64;
        } else {
            r.k = z + run;
            r.has_val = r.k < 64;
            zn = r.k + 1;
        }
        r.block_done = zn >= 64;
        if (r.block_done) {
            z = 0;
            b = (b + 1 == ic.bpm) ? 0 : b + 1;
        } else {
            z = zn;
        }
        return r;
    }
};

// a[c] += v without indexing a register array by a per-lane value (that would push the array to scratch memory)
LP_HD void lp_add3(int32_t a[LP_MAX_COMP], uint32_t c, int32_t v)
{
    a[0] += c == 0 ? v : 0;
    a[1] += c == 1 ? v : 0;
    a[2] += c == 2 ? v : 0;
}
LP_HD int32_t lp_get3(const int32_t a[LP_MAX_COMP], uint32_t c) { return c == 0 ? a[0] : c == 1 ? a[1] : a[2]; }

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

// What follows prefix `pre` inside `whole` (both measured from the same start): whole = combine(pre, tail).
LP_HD LpSubSum lp_sum_tail(const LpSubSum& whole, const LpSubSum& pre)
{
    LpSubSum t;
    t.nblk = whole.nblk - pre.nblk;
    t.nreset = whole.nreset - pre.nreset;
    for (int c = 0; c < LP_MAX_COMP; c++) t.dc[c] = t.nreset ? whole.dc[c] : whole.dc[c] - pre.dc[c];
    return t;
}

LP_HD int32_t lp_sx16(uint32_t v) { return (int32_t)(int16_t)(uint16_t)v; }

LP_HD LpSumPk lp_sum_pack(const LpSubSum& s)
{
    LpSumPk k;
    k.nblk = s.nblk;
    k.nreset = s.nreset;
    k.dc01 = ((uint32_t)s.dc[0] & 0xffffu) | ((uint32_t)s.dc[1] << 16);
    k.dc2 = (uint32_t)s.dc[2] & 0xffffu;
    return k;
}
LP_HD LpSubSum lp_sum_unpack(const LpSumPk& k)
{
    LpSubSum s;
    s.nblk = k.nblk;
    s.nreset = k.nreset;
    s.dc[0] = lp_sx16(k.dc01);
    s.dc[1] = lp_sx16(k.dc01 >> 16);
    s.dc[2] = lp_sx16(k.dc2);
    return s;
}
LP_HD LpCkptPk lp_ckpt_pack(const LpSubState& st, const LpSubSum& s)
{
    LpCkptPk k;
    k.p = st.p;
    k.bz_nreset = (st.bz & 0xffffu) | (s.nreset << 16);
    k.nblk_dc2 = (s.nblk & 0xffffu) | ((uint32_t)s.dc[2] << 16);
    k.dc01 = ((uint32_t)s.dc[0] & 0xffffu) | ((uint32_t)s.dc[1] << 16);
    return k;
}
LP_HD void lp_ckpt_unpack(const LpCkptPk& k, LpSubState& st, LpSubSum& s)
{
    st.p = k.p;
    st.bz = k.bz_nreset & 0xffffu;
    s.nreset = k.bz_nreset >> 16;
    s.nblk = k.nblk_dc2 & 0xffffu;
    s.dc[2] = lp_sx16(k.nblk_dc2 >> 16);
    s.dc[0] = lp_sx16(k.dc01);
    s.dc[1] = lp_sx16(k.dc01 >> 16);
}

LP_HD uint32_t lp_ck_iter(const LpCkSched& cs, uint32_t k) { return k < cs.nd ? (k + 1) * cs.td : cs.nd * cs.td + (k + 1 - cs.nd) * cs.ts; }

// SPEC pass for one subsequence: decode [entry.p, sub_end) from the guessed state.
// Ck must provide   void record(uint32_t k, const LpCkptPk&)   -- called by ALL lanes of the wave at the same iteration.
template <class M, class Ck>
LP_HD void lp_spec_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpSubState entry, const LpCkSched& cs, Ck& ck, LpSubState* exit_st,
                        LpSubSum* total)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    LpSubSum sum;
    lp_sum_zero(sum);
    uint32_t k = 0, iter = 0, next_ck = cs.K ? lp_ck_iter(cs, 0) : 0xffffffffu;
    bool done = false;
    // Wave-uniform loop: every lane executes the same instruction stream; finished lanes are predicated off.
    while (m.any(!done)) {
        L.refill();
        if ((iter & (LP_TOPUP_EVERY - 1)) == LP_TOPUP_EVERY - 1) m.topup(L.widx);
        if (!done && L.z == 0 && L.restart_check()) { // also catches the padded end of the stream
            sum.nreset++;
            for (int c = 0; c < LP_MAX_COMP; c++) sum.dc[c] = 0;
        }
        if (iter == next_ck) { // wave-uniform: iter, k and next_ck are the same in every lane
            LpSubState st;
            st.p = L.p;
            st.bz = L.state_bz();
            ck.record(k, lp_ckpt_pack(st, sum));
            k++;
            next_ck = k < cs.K ? lp_ck_iter(cs, k) : 0xffffffffu;
        }
        iter++;
        if (done) continue;
        if (L.p >= sub_end) { done = true; continue; }
        if (L.z == 0) sum.nblk++;
        const typename LpLane<M>::Sym s = L.step();
        if (s.is_dc) lp_add3(sum.dc, s.comp, s.val);
    }
    LpCkptPk none;
    none.p = 0xffffffffu; none.bz_nreset = 0; none.nblk_dc2 = 0; none.dc01 = 0;
    for (; k < cs.K; k++) ck.record(k, none);
    exit_st->p = L.p;
    exit_st->bz = L.state_bz();
    *total = sum;
}

// VERIFY pass for one subsequence: decode from `entry` (= current exit state of the previous subsequence) until the state
// matches a checkpoint recorded by the SPEC pass, then splice the recorded remainder (spec_exit / spec_total are the SPEC
// pass's immutable results). Without a match the lane runs to the end of the subsequence.
// Ck must provide   uint32_t pos(uint32_t k)   (cheap: position of checkpoint k, 0xffffffff if not recorded)
//                   LpCkptPk load(uint32_t k)  (the whole record; called once or twice per lane)
// Outputs the new exit state and total of the subsequence.
template <class M, class Ck>
LP_HD void lp_verify_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpSubState entry, uint32_t K, Ck& ck, const LpSubState& spec_exit,
                          const LpSubSum& spec_total, LpSubState* exit_st, LpSubSum* total)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    LpSubSum sum;
    lp_sum_zero(sum);
    uint32_t kk = 0, iter = 0;
    uint32_t cp = K ? ck.pos(0) : 0xffffffffu;
    bool done = false, spliced = false;
    while (m.any(!done)) {
        L.refill();
        if ((iter & (LP_TOPUP_EVERY - 1)) == LP_TOPUP_EVERY - 1) m.topup(L.widx);
        iter++;
        if (done) continue;
        if (L.z == 0 && L.restart_check()) {
            sum.nreset++;
            for (int c = 0; c < LP_MAX_COMP; c++) sum.dc[c] = 0;
        }
        while (cp < L.p) { // checkpoints are strictly ordered until the lane that recorded them finished
            kk++;
            cp = kk < K ? ck.pos(kk) : 0xffffffffu;
        }
        if (cp == L.p && kk < K) {
            LpSubState cst;
            LpSubSum csum;
            lp_ckpt_unpack(ck.load(kk), cst, csum);
            if (cst.bz == L.state_bz()) { // synchronised with the recorded trajectory at checkpoint kk
                *total = lp_sum_combine(sum, lp_sum_tail(spec_total, csum));
                *exit_st = spec_exit;
                done = true;
                spliced = true;
                continue;
            }
            kk++; // same position, different state: this checkpoint can never match
            cp = kk < K ? ck.pos(kk) : 0xffffffffu;
        }
        if (L.p >= sub_end) { done = true; continue; }
        if (L.z == 0) sum.nblk++;
        const typename LpLane<M>::Sym s = L.step();
        if (s.is_dc) lp_add3(sum.dc, s.comp, s.val);
    }
    if (!spliced) {
        exit_st->p = L.p;
        exit_st->bz = L.state_bz();
        *total = sum;
    }
}

// WRITE pass for one subsequence. Sink S must provide:
//   void put(uint32_t natural_idx, int32_t v);     store one coefficient of the block being decoded
//   void end_block(uint32_t blk);                  the block (decode-order index blk) is complete (queued for flushing)
//   bool stalled();                                no free slot: the lane must wait for the next flush
//   void flush();                                  wave-uniform: write out every queued block
// Returns the number of blocks written.
#ifndef LP_FLUSH_EVERY
#define LP_FLUSH_EVERY 2
#endif
template <class M, class Sink>
LP_HD uint32_t lp_write_pass(M& m, const LpImgCtx& ic, LpSubState entry, uint32_t end_p, const LpSubSum& prefix, const uint8_t* zigzag,
                             Sink& sink)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    uint32_t blk = prefix.nblk;
    int32_t pred[LP_MAX_COMP];
    for (int c = 0; c < LP_MAX_COMP; c++) pred[c] = prefix.dc[c];
    bool writing = false, done = false;
    uint32_t written = 0, iter = 0;
    while (m.any(!done)) {
        L.refill();
        if ((iter & (LP_TOPUP_EVERY - 1)) == LP_TOPUP_EVERY - 1) m.topup(L.widx);
        if ((iter % LP_FLUSH_EVERY) == LP_FLUSH_EVERY - 1) sink.flush();
        iter++;
        if (done || sink.stalled()) continue;
        if (L.z == 0) {
            if (L.restart_check())
                for (int c = 0; c < LP_MAX_COMP; c++) pred[c] = 0;
            if (L.p >= end_p || blk >= ic.total_blocks) { done = true; continue; }
            writing = true;
        } else if (L.p >= ic.total_bits) { done = true; continue; } // truncated stream
        const typename LpLane<M>::Sym s = L.step();
        if (writing) {
            if (s.is_dc) {
                lp_add3(pred, s.comp, s.val);
                sink.put(0, lp_get3(pred, s.comp));
            } else if (s.has_val) {
                sink.put(zigzag[s.k], s.val);
            }
            if (s.block_done) {
                sink.end_block(blk);
                written++;
                blk++;
            }
        }
    }
    sink.flush();
    return written;
}
