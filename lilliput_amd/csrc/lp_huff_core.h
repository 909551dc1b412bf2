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
// Inside a branch on a wave-uniform condition: keeps it a scalar branch. Without it the compiler folds the uniform test into the
// per-lane condition that follows and evaluates both in every iteration (the ring top-up test ran every step instead of every second).
#if defined(__HIP_DEVICE_COMPILE__)
#define LP_KEEP_UNIFORM_BRANCH() asm volatile("")
#else
#define LP_KEEP_UNIFORM_BRANCH() ((void)0)
#endif

// zigzag index -> natural (row-major) index; 16 guard entries like libjpeg's jpeg_natural_order
#define LP_ZIGZAG_INIT                                                                                    \
    {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, \
     13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, \
     38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, \
     63, 63}

// Ring geometry of the bit reader. A lane keeps the three stream words at its position in registers (w0, w1 = the 64 bits a
// peek shifts, w2 = the word after them, requested from the ring one step before it can be needed), so the ring read is off
// the symbol-to-symbol dependency chain. One decode step consumes at most 16 (code) + 15 (extra bits) = 31 bits, so kEvery = 8
// steps advance the position by at most 8 words; after a top-up at most 3 ring words are free (16-byte granularity) and a step
// may ask for word (p >> 5) + 2: 8 + 3 + 3 <= kRing = 16.
// The policy fixes the geometry: M::kRing words per lane, a top-up of at most M::kQuads 16-byte loads every M::kEvery steps.

// Per-image values every lane of a workgroup shares (scalar registers on the device).
struct LpImgCtx {
    uint32_t blkpack;       // low 32 bits of LpJpeg::blkpack (at most LP_MAX_BPM = 6 nibbles)
    uint32_t bpm;
    uint32_t n_rst;         // restart boundaries found by the unstuff kernels
    uint32_t total_bits;    // length of the clean stream
    uint32_t total_blocks;
};

// Memory policy M must provide (per lane object, non-const):
//   uint32_t fetch1(uint32_t w)             word w of the clean stream (big-endian corrected: bit 31 first); w within the ring window
//   void reseek(uint32_t w)                 the lane jumps: make [w, w + kRing - 3) fetchable
//   void topup(uint32_t p)                  wave-uniform call every kEvery steps with the lane's bit position: words below p >> 5 are dead, refill
//   bool any(bool)                          wave vote (host emulation: identity)
//   bool any_lt8(int32_t v)                 wave vote on v < 8 (see lp_near_boundary)
//   uint32_t lut(uint32_t tbl, uint32_t i), lut2(uint32_t i)   first-level entry of table tbl, entry i of the second-level pool
//   int32_t maxcode(tbl, l), valoff(tbl, l); uint32_t val(tbl, i)   canonical tables (third level, corrupt streams / huge tables)
//   uint32_t rst_bit(uint32_t k)            bit position of the k-th restart boundary
//   void settle(uint32_t& v)                v came from rst_bit() inside a rare branch: finish the load there (device), no-op on the host
//
// The lane keeps no refillable bit buffer: a peek is one 64-bit shift of (w0, w1) by p & 31. What bounds these kernels on
// MI355X is the latency of the serial chain symbol -> length -> position -> next symbol (PMC: the waves sit in s_waitcnt for
// half of their cycles with the VALU a third busy), so the chain holds exactly one LDS round trip -- the code-table lookup.
template <class M>
struct LpLane {
    M& m;
    const LpImgCtx& ic;
    uint32_t p;         // bit position of the next unread bit
    uint32_t z;         // zigzag index of the next coefficient (0 = a block starts here)
    uint32_t b4;        // 4 x block-in-MCU: the shift that brings the current block's nibble of ic.blkpack down (one add + compare per block
                        // end; the first version also kept the rotated nibble string: four more VALU instructions in every step)
    uint32_t next_rst;  // bit position of the next restart boundary (stream end when none left)
    uint32_t rst_k;     // index of that boundary
    uint32_t w0, w1, w2; // stream words (p >> 5), + 1, + 2

    LP_HD LpLane(M& m_, const LpImgCtx& ic_) : m(m_), ic(ic_), p(0), z(0), b4(0), next_rst(0), rst_k(0), w0(0), w1(0), w2(0) {}

    LP_HD uint32_t peek() const { return (uint32_t)(((((uint64_t)w0) << 32) | w1) << (p & 31u) >> 32); }
    LP_HD void load_window() // after a jump (m.reseek has been called)
    {
        const uint32_t w = p >> 5;
        w0 = m.fetch1(w);
        w1 = m.fetch1(w + 1u);
        w2 = m.fetch1(w + 2u);
    }
    // the position moves on by n <= 31 bits: slide the register window when it enters the next word, and ask for the word after
    // the window again (the same word as before when nothing slid; its value is first used one step later)
    LP_HD void advance(uint32_t n)
    {
        const uint32_t pn = p + n;
        const bool slid = ((p ^ pn) & 32u) != 0;
        w0 = slid ? w1 : w0;
        w1 = slid ? w2 : w1;
        w2 = m.fetch1((pn >> 5) + 2u);
        p = pn;
    }
    LP_HD void set_block(uint32_t nb) { b4 = 4u * nb; }
    LP_HD void start(uint32_t pos, uint32_t bz)
    {
        p = pos;
        m.reseek(pos >> 5);
        load_window();
        set_block(bz >> 8);
        z = bz & 255u;
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
            m.settle(next_rst);
        }
    }
    LP_HD uint32_t state_bz() const { return (b4 << 6) | z; } // (block-in-MCU << 8) | zigzag index

    // At a block start: detect the end of a restart interval (or of the stream). `pk` = peek(). Returns true when
    // the lane jumped to the boundary (DC predictors must be reset by the caller, and peek() must be redone).
    LP_HD bool restart_check(uint32_t pk)
    {
        int32_t rem = (int32_t)(next_rst - p);
        // A boundary BEHIND the lane: only a lane decoding from a wrong state can run over one in the middle of a block. It is
        // left behind (the lane keeps going and synchronises at a later boundary or by itself): jumping back would make the
        // lane visit the same positions twice, and a checkpoint of the first visit would splice the second visit's counts in
        // again (found by tests/test_gpu_sweep.py with one-MCU restart intervals).
        while (rem < 0) {
            if (rst_k < ic.n_rst) {
                rst_k++;
                next_rst = rst_k < ic.n_rst ? m.rst_bit(rst_k) : ic.total_bits;
            } else
                next_rst = 0x7fffffffu;
            m.settle(next_rst);
            rem = (int32_t)(next_rst - p);
        }
        if (rem >= 8) return false;
        bool jump = rem == 0;
        if (!jump) jump = (pk >> (32 - rem)) == ((1u << rem) - 1u);
        if (!jump) return false;
        const uint32_t target = next_rst;
        if (rst_k < ic.n_rst) {
            rst_k++;
            next_rst = rst_k < ic.n_rst ? m.rst_bit(rst_k) : ic.total_bits;
        } else {
            next_rst = 0x7fffffffu; // past the end of the stream: nothing left
        }
        m.settle(next_rst);
        p = target;
        m.reseek(target >> 5);
        load_window();
        set_block(0);
        z = 0;
        return true;
    }

    // Second level of the code lookup: e1 is the first-level entry of a prefix that belongs to codes longer than LP_LUT_BITS; its
    // low byte names a slice of the second-level pool, indexed by the LP_LUT2_BITS bits that follow the prefix (see LpHuffSet).
    // One dependent LDS read: a long code is rare per lane but shows up in most steps of a 64-lane wave.
    LP_HD uint32_t long_code(uint32_t tbl, uint32_t e1, uint32_t top)
    {
        const uint32_t sub = e1 & 0xffu;
        uint32_t e = sub != 0xffu ? m.lut2((sub << LP_LUT2_BITS) | (top & ((1u << LP_LUT2_BITS) - 1u))) : 0u;
        if ((e & 0x1f00u) == 0) { // canonical search (T.81 F.2.2.3, jdhuff.c jpeg_huff_decode): corrupt prefix, or a table whose long codes overflow the pool
            uint32_t len = 17, sym = 0; // no code matches: jpeg_huff_decode reads on to the sentinel length 17, warns and fakes a zero
            for (uint32_t l = LP_LUT_BITS + 1; l <= 16; l++) {
                const int32_t code = (int32_t)(top >> (16 - l));
                if (code <= m.maxcode(tbl, l)) {
                    len = l;
                    sym = m.val(tbl, (uint32_t)(code + m.valoff(tbl, l)));
                    break;
                }
            }
            e = (len << 8) | sym | ((tbl >= 2 && (sym & 15u) == 0 && (sym >> 4) != 15) ? 0x8000u : 0u);
        }
        return e;
    }

    // Decode one Huffman symbol (+ its extra bits) from pk = peek(). On return:
    //   is_dc, k = zigzag index of the coefficient (valid when has_val), val, block_done.
    // Written branch-free apart from the long-code lookup: every lane of the wave runs the same instructions.
    // NEED_VAL = false (the counting passes) skips the value reconstruction.
    struct Sym { bool is_dc; bool has_val; bool block_done; uint32_t k; int32_t val; };
    template <bool NEED_VAL>
    LP_HD Sym step(uint32_t pk)
    {
        Sym r;
        r.is_dc = (z == 0);
        // DC table id = bit 2 of the block's nibble, AC table id = bit 3 (+2); written as arithmetic so that it compiles to
        // selects, not to an exec-mask branch pair
        const uint32_t tbl = ((ic.blkpack >> (b4 + (r.is_dc ? 2u : 3u))) & 1u) + (r.is_dc ? 0u : 2u);
        uint32_t e = m.lut(tbl, pk >> (32 - LP_LUT_BITS));
        if ((e & 0x1f00u) == 0) e = long_code(tbl, e, pk >> 16);
        const uint32_t len = (e >> 8) & 31u;
        const uint32_t s = e & 15u;
        const uint32_t run = (e >> 4) & 15u;      // DC symbols are categories 0..15 (validated by the parser): run == 0
        r.val = 0;
        if (NEED_VAL) {
            const uint32_t t = pk << len;             // the extra bits, left aligned
            const uint32_t x = (t >> 1) >> (31u - s); // their value (0 when s == 0)
            // HUFF_EXTEND: a first extra bit of 0 means negative: val = x - (2^s - 1)
            const uint32_t neg = ~(uint32_t)((int32_t)t >> 31);
            r.val = (int32_t)(x - (neg & ((1u << s) - 1u)));
        }
        advance(len + s);
        // jdhuff.c decode_mcu: size 0 ends the block unless the run is 15 (ZRL); a coefficient whose index overruns 63 on
        // a corrupt stream still lands on jpeg_natural_order[64..79] = 63 (the zigzag table carries the same guard entries)
        const bool eob = (e & 0x8000u) != 0; // precomputed per table entry: an AC symbol of size 0 other than ZRL
        r.k = z + run;                            // DC: z == run == 0; at most 63 + 15
        r.has_val = r.is_dc || s != 0;
        const uint32_t zn = eob ? 64u : r.k + 1u; // ZRL (run 15, size 0) skips 16 coefficients: k + 1 == z + 16
        r.block_done = zn >= 64;
        z = r.block_done ? 0u : zn;
        // next block of the MCU; branch-free
        const uint32_t nb4 = b4 + 4u == 4u * ic.bpm ? 0u : b4 + 4u;
        b4 = r.block_done ? nb4 : b4;
        return r;
    }
};

// "Is a restart boundary (or the stream end) fewer than 8 bits ahead of a lane that stands at a block start and is still working?" as ONE
// signed value to compare with 8: the bits to the boundary, raised to at least 8 * z (so a lane inside a block never qualifies) or to 8
// when the lane is not working. The wave vote on `value < 8` is then a single v_cmp feeding the scalar branch; voting on the three
// conditions as a boolean cost eleven instructions per decode step (the compiler materialises the boolean in a VGPR and compares it again).
LP_HD int32_t lp_near_boundary(uint32_t next_rst, uint32_t p, uint32_t z, bool off)
{
    const int32_t rem = (int32_t)(next_rst - p), floor = off ? 8 : (int32_t)(z << 3);
    return rem > floor ? rem : floor;
}

LP_HD bool lp_state_eq(const LpSubState& a, const LpSubState& b) { return a.p == b.p && a.bz == b.bz; }

LP_HD void lp_sum_zero(LpSubSum& s)
{
    s.nblk = 0;
    s.nreset = 0;
}

// a then b (b later in the stream)
LP_HD LpSubSum lp_sum_combine(const LpSubSum& a, const LpSubSum& b)
{
    LpSubSum r;
    r.nblk = a.nblk + b.nblk;
    r.nreset = a.nreset + b.nreset;
    return r;
}

// What follows prefix `pre` inside `whole` (both measured from the same start): whole = combine(pre, tail).
LP_HD LpSubSum lp_sum_tail(const LpSubSum& whole, const LpSubSum& pre)
{
    LpSubSum t;
    t.nblk = whole.nblk - pre.nblk;
    t.nreset = whole.nreset - pre.nreset;
    return t;
}

LP_HD LpCkptPk lp_ckpt_pack(const LpSubState& st, const LpSubSum& s)
{
    LpCkptPk k;
    k.p = st.p;
    k.bz = st.bz;
    k.nblk = s.nblk;
    k.nreset = s.nreset;
    return k;
}
LP_HD void lp_ckpt_unpack(const LpCkptPk& k, LpSubState& st, LpSubSum& s)
{
    st.p = k.p;
    st.bz = k.bz;
    s.nblk = k.nblk;
    s.nreset = k.nreset;
}

LP_HD uint32_t lp_ck_iter(const LpCkSched& cs, uint32_t k) { return cs.it[k]; }
// it[k] from it[k - 1] (0 for k == 0): the rule of lp_make_sched as arithmetic, so that the decode loop needs no table load (a scalar
// load in the loop makes every iteration wait on lgkmcnt -- and with it on the lane's outstanding LDS reads)
LP_HD uint32_t lp_ck_next(uint32_t base, uint32_t k, uint32_t prev) { return prev + ((k < 4 || prev / 2 < base) ? base : prev / 2); }

// Schedule for subsequences of S bits; cbits tunes the first spacing (cbits / 32 iterations, 8 for the default 256).
inline LpCkSched lp_make_sched(uint32_t S, uint32_t cbits)
{
    LpCkSched cs;
    const uint32_t base = cbits / 32 > 2 ? cbits / 32 : 2, span = S / 4 > base ? S / 4 : base; // a lane runs about S/6.5 iterations
    uint32_t v = 0;
    cs.K = 0;
    cs.base = base;
    for (uint32_t k = 0; k < LP_MAX_CKPT; k++) {
        v = lp_ck_next(base, k, v);
        cs.it[k] = v;
        cs.K = k + 1;
        if (v >= span) break;
    }
    for (uint32_t k = cs.K; k < LP_MAX_CKPT; k++) cs.it[k] = 0xffffffffu;
    return cs;
}

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
    const uint32_t K = cs.K, ck_base = cs.base;
    uint32_t k = 0, iter = 0, next_ck = K ? lp_ck_next(ck_base, 0, 0) : 0xffffffffu;
    bool done = false;
    // Wave-uniform loop: every lane executes the same instruction stream; finished lanes are predicated off.
    // Single back edge, no `continue`: the register allocator then updates the lane state in place (the first version of
    // this loop carried ~30 v_mov copies per iteration across its exits).
    do {
        if ((iter & (M::kEvery - 1)) == M::kEvery - 1) { LP_KEEP_UNIFORM_BRANCH(); m.topup(L.p); }
        uint32_t pk = L.peek();
        if (m.any_lt8(lp_near_boundary(L.next_rst, L.p, L.z, done))) { // rare even per wave: a restart boundary or the stream end is near
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && L.z == 0 && L.restart_check(pk)) { // also catches the padded end of the stream
                sum.nreset++;
            }
            pk = L.peek();
        }
        if (iter == next_ck) { // wave-uniform: iter, k and next_ck are the same in every lane
            LpSubState st;
            st.p = L.p;
            st.bz = L.state_bz();
            ck.record(k, lp_ckpt_pack(st, sum));
            k++;
            next_ck = k < K ? lp_ck_next(ck_base, k, next_ck) : 0xffffffffu;
        }
        iter++;
        done = done || L.p >= sub_end;
        if (!done) {
            sum.nblk += L.z == 0 ? 1u : 0u;
            (void)L.template step<false>(pk);
        }
    } while (m.any(!done));
    LpCkptPk none;
    none.p = 0xffffffffu; none.bz = 0; none.nblk = 0; none.nreset = 0;
    for (; k < K; k++) ck.record(k, none);
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
    do { // same shape as the SPEC loop: one back edge, state updated in place
        if ((iter & (M::kEvery - 1)) == M::kEvery - 1) { LP_KEEP_UNIFORM_BRANCH(); m.topup(L.p); }
        uint32_t pk = L.peek();
        iter++;
        if (m.any_lt8(lp_near_boundary(L.next_rst, L.p, L.z, done))) {
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && L.z == 0 && L.restart_check(pk)) {
                sum.nreset++;
            }
            pk = L.peek();
        }
        if (!done) {
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
                } else {
                    kk++; // same position, different state: this checkpoint can never match
                    cp = kk < K ? ck.pos(kk) : 0xffffffffu;
                }
            }
        }
        done = done || L.p >= sub_end;
        if (!done) {
            sum.nblk += L.z == 0 ? 1u : 0u;
            (void)L.template step<false>(pk);
        }
    } while (m.any(!done));
    if (!spliced) {
        exit_st->p = L.p;
        exit_st->bz = L.state_bz();
        *total = sum;
    }
}

// WRITE pass for one subsequence. Sink S must provide:
//   void put_dc(int32_t v, bool on);               when `on`: the block's DC DIFFERENCE (made absolute later, see lp_dc_walk)
//   void put(uint32_t natural_idx, int32_t v);     store one AC coefficient of the block being decoded
//   void end_block(uint32_t blk, bool on);         when `on`: the block (decode-order index blk) is complete (queued for flushing)
//   bool stalled();                                no free slot: the lane must wait for the next flush
//   void flush();                                  wave-uniform: write out every queued block
//   void finish();                                 once, after the last flush
// Returns the number of blocks written.
#ifndef LP_FLUSH_EVERY
#define LP_FLUSH_EVERY 4   // 2 / 3 / 4 / 6 measured (8-word ring, top-up every 2 steps): 4 is the best trade of flush instructions against lanes waiting for the flush
#endif
template <class M, class Sink>
LP_HD uint32_t lp_write_pass(M& m, const LpImgCtx& ic, LpSubState entry, uint32_t end_p, const LpSubSum& prefix, const uint8_t* zigzag,
                             Sink& sink)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    uint32_t blk = prefix.nblk;
    bool writing = false, done = false;
    uint32_t written = 0, iter = 0;
    do {
        if ((iter & (M::kEvery - 1)) == M::kEvery - 1) { LP_KEEP_UNIFORM_BRANCH(); m.topup(L.p); }
        if ((iter % LP_FLUSH_EVERY) == LP_FLUSH_EVERY - 1) sink.flush();
        uint32_t pk = L.peek();
        iter++;
        const bool act = !done && !sink.stalled();
        if (m.any_lt8(lp_near_boundary(L.next_rst, L.p, L.z, !act))) {
            LP_KEEP_UNIFORM_BRANCH();
            if (act && L.z == 0) (void)L.restart_check(pk); // DC predictors restart in k_dc_scan, by MCU index
            pk = L.peek();
        }
        // a lane stops at the first block start at or after the end of its subsequence (or when the stream is truncated)
        const bool stop = L.z == 0 ? (L.p >= end_p || blk >= ic.total_blocks) : L.p >= ic.total_bits;
        done = done || (act && stop);
        if (act && !stop) {
            writing = writing || L.z == 0; // a lane that enters mid-block skips to the first block start
            const typename LpLane<M>::Sym s = L.template step<true>(pk);
            // few, flat predicated regions: every exec-mask branch costs scalar instructions in a loop that is issue bound
            sink.put_dc(s.val, writing && s.is_dc);
            if (writing && !s.is_dc && s.has_val) sink.put((uint32_t)zigzag[s.k], s.val);
            const bool bd = writing && s.block_done;
            sink.end_block(blk, bd);
            written += bd ? 1u : 0u;
            blk += bd ? 1u : 0u;
        }
    } while (m.any(!done));
    sink.flush();
    sink.finish();
    return written;
}

// DC differences -> absolute DC, in place, for MCUs [m0, m1) of one image given the predictors at m0 (jdhuff.c decode_mcu:
// last_dc_val per component, reset to 0 every restart interval -- process_restart runs every `dri` MCUs whatever the
// markers say). dc[] holds one value per block in decode order; comp_of[b] = component of block b of an MCU. Values wrap
// like libjpeg's store into a 16-bit JCOEF. Returns the predictors after m1 in pred[].
LP_HD void lp_dc_walk(int16_t* dc, uint32_t m0, uint32_t m1, uint32_t bpm, uint32_t dri, const uint8_t* comp_of, int32_t pred[LP_MAX_COMP], bool write)
{
    for (uint32_t m = m0; m < m1; m++) {
        if (dri && m % dri == 0) pred[0] = pred[1] = pred[2] = 0;
        for (uint32_t b = 0; b < bpm; b++) {
            const uint32_t c = comp_of[b];
            const int32_t v = pred[c] + dc[(size_t)m * bpm + b];
            pred[c] = v;
            if (write) dc[(size_t)m * bpm + b] = (int16_t)v;
        }
    }
}
