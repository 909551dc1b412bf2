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

#ifndef LP_MULTI
#define LP_MULTI 1   // SPEC / VERIFY decode several short symbols per step (LpHuffSet::lutm); 0 = one symbol per step (round 4's passes, A/B builds)
#endif
#if defined(__HIPCC__)
#define LP_HD __host__ __device__ __forceinline__
#else
#define LP_HD inline
#endif
// Inside a branch on a wave-uniform condition: keeps it a scalar branch. Without it the compiler folds the uniform test into the
// per-lane condition that follows and evaluates both in every iteration (the ring top-up test ran every step instead of every second).
// A/B switches of the decode steps (profiles/r06_write_stream.md)
#ifndef LP_X_LAYOUT
#define LP_X_LAYOUT 1   // rare paths out of line: the common step falls through instead of taking its branches (nothing for full launches,
                        // 3-4 % of a one-image call, whose lone waves pay for every taken branch)
#endif
#ifndef LP_X_FLUSH2
#define LP_X_FLUSH2 0   // the WRITE loop flushes every second step: a lane finishes at most one block between two flushes (a block is at least a DC
                        // symbol and an end-of-block), so no lane ever waits for a flush -- no stalled() test in the step, no idle steps
#endif
#if LP_X_LAYOUT
#define LP_RARE(x) __builtin_expect(!!(x), 0)
#else
#define LP_RARE(x) (x)
#endif
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

// Ring geometry of the bit reader. A peek reads the two stream words around the lane's position, words ceil(p / 32) - 1 and
// ceil(p / 32), from a per-lane ring. One decode step consumes at most 16 (code) + 15 (extra bits) = 31 bits, so kEvery = 8
// steps advance the position by at most 8 words; after a top-up at most 3 ring words are free (16-byte granularity) and a step
// may ask for word (p >> 5) + 1: 8 + 3 + 2 <= kRing = 16 (and 2 + 3 + 2 <= 8 for the ring of eight with a top-up every second step).
// The policy fixes the geometry: M::kRing words per lane, a top-up of at most M::kQuads 16-byte loads every M::kEvery steps.

// Per-image values every lane of a workgroup shares (scalar registers on the device).
struct LpImgCtx {
    uint32_t blkpack;       // low 32 bits of LpJpeg::blkpack (at most LP_MAX_BPM = 6 nibbles)
    uint32_t bpm;
    uint32_t n_rst;         // restart boundaries found by the unstuff kernels
    uint32_t total_bits;    // length of the clean stream
    uint32_t total_blocks;
    uint32_t rst_blocks;    // blocks per restart interval (dri x bpm), 0 = no restart interval
    uint32_t tb, nx;        // derived from blkpack / bpm by lp_ctx_tables(): 5-bit field i (at bit 5 i) belongs to block i of the MCU.
                            // tb: bits 0..1 = the block's DC table slot (0 / 1), bits 2..3 = its AC table slot (2 / 3);
                            // nx: 5 x the index of the block that follows it (0 after the last block of the MCU)
};
LP_HD void lp_ctx_tables(LpImgCtx& ic)
{
    ic.tb = ic.nx = 0;
    for (uint32_t i = 0; i < ic.bpm && i < LP_MAX_BPM; i++) {
        const uint32_t nib = (ic.blkpack >> (4u * i)) & 15u; // bit 2 = DC table id, bit 3 = AC table id
        ic.tb |= (((nib >> 2) & 1u) | ((2u + ((nib >> 3) & 1u)) << 2)) << (5u * i);
        ic.nx |= (5u * (i + 1u == ic.bpm ? 0u : i + 1u)) << (5u * i);
    }
}

// funnel shift: the low 32 bits of (hi:lo) >> (sh & 31) -- one v_alignbit_b32
LP_HD uint32_t lp_funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
#endif
}
// min(v, 1) as ONE instruction (the compiler turns the C form into a compare and a select, two instructions and a wait state)
LP_HD uint32_t lp_min1(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
    return r;
#else
    return v < 1u ? v : 1u;
#endif
}
// ... with a per-lane width (0..15 here; width 0 yields 0)
LP_HD uint32_t lp_bfe_w(uint32_t v, uint32_t off, uint32_t width)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return width ? (v >> (off & 31u)) & ((1u << width) - 1u) : 0u;
#endif
}
// bit field: (v >> (off & 31)) & ((1 << width) - 1) -- one v_bfe_u32; the hardware ignores the offset's upper bits like the host form does
LP_HD uint32_t lp_bfe(uint32_t v, uint32_t off, uint32_t width)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return (v >> (off & 31u)) & ((1u << width) - 1u);
#endif
}

// Memory policy M must provide (per lane object, non-const):
//   uint32_t peek_np(uint32_t np)           the 32 stream bits at the position np encodes (LpLane::np): with w1 = 3 - ((int32_t)np >> 5) -- the
//                                           word index ceil(p / 32) -- the funnel shift of words (w1 - 1, w1) by np. The device ring is laid out
//                                           so that both words are one two-address LDS read at a bit field of np
//   void reseek(uint32_t w)                 the lane jumps: make [w, w + kRing - 3) fetchable
//   void topup(uint32_t p)                  wave-uniform call every kEvery steps with the lane's bit position: words below p >> 5 are dead, refill
//   bool any(bool)                          wave vote (host emulation: identity)
//   bool any2(bool a, bool b)               any(a || b), as two votes or-ed in scalar registers (a vote on a compare costs nothing; a vote on
//                                           a combination of compares is materialised in a VGPR first: two more vector instructions)
//   uint32_t lut(uint32_t tbl, uint32_t i), lut2(uint32_t i)   first-level entry of table tbl, entry i of the second-level pool
//   uint32_t lutc(uint32_t tbl, uint32_t i)  lut[tbl][i] | lutm[tbl][i] << 16 (the counting passes' combined entry)
//   int32_t maxcode(tbl, l), valoff(tbl, l); uint32_t val(tbl, i)   canonical tables (third level, corrupt streams / huge tables)
//   uint32_t rst_bit(uint32_t k)            bit position of the k-th restart boundary
//   void settle(uint32_t& v)                v came from rst_bit() inside a rare branch: finish the load there (device), no-op on the host
//
// What bounds these kernels on MI355X is VALU issue (a wave64 instruction occupies its SIMD for four cycles; PMC and the ISA
// listing agree: SPEC ran 45 vector instructions per symbol in round 2), so the lane state is laid out for the fewest
// instructions per step, not for readability:
//   * the position is kept NEGATED (np = bias - p): a peek is then ONE funnel shift of two stream words by np -- taking the words
//     ceil(p / 32) - 1 and ceil(p / 32), the shift amount 32 - (p mod 32) taken mod 32 is right for every p, including p on a word
//     boundary (shift 0 returns the lower word);
//   * "is anything special about this step" (end of the subsequence, a restart boundary or the stream end fewer than 8 bits ahead)
//     is ONE signed compare of np with a per-lane limit; the wave branches to the slow path on the vote;
//   * block-in-MCU and the count of completed blocks share a register (bc): 5 x block in the low five bits -- the offset of the
//     block's field in ic.tb / ic.nx, which v_bfe_u32 reads without masking -- and the count above, incremented by the same add
//     that installs the next block;
//   * the table entry carries run + 64 x (ends the block), so z + runx + 1 > 63 is the block-done test for coefficient overrun,
//     ZRL and EOB alike.
template <class M>
struct LpLane {
    M& m;
    const LpImgCtx& ic;
    uint32_t np;        // LP_NP0 MINUS the bit position of the next unread bit (two's complement; LP_NP0 = 64, see pos())
    uint32_t z;         // zigzag index of the next coefficient (0 = a block starts here)
    uint32_t bc;        // bits 0..4: 5 x block-in-MCU; bits 5..31: blocks completed since start(), starting at -1 when start() was inside
                        // a block -- so the field is also "blocks STARTED since start()" minus (z != 0), and it is negative exactly while
                        // the lane has not seen a block start yet
    uint32_t next_rst;  // bit position of the next restart boundary (stream end when none left)
    uint32_t rst_k;     // index of that boundary
    uint32_t irregular; // the lane ran over a restart boundary in the middle of a block. In the counting passes that is a lane on a wrong
                        // state (nobody reads the flag there); in the WRITE pass, whose lanes start from verified states, it is a stream
                        // whose interval holds more data than its MCUs need -- libjpeg drops such data at the marker (lp_write_pass)

    LP_HD LpLane(M& m_, const LpImgCtx& ic_) : m(m_), ic(ic_), np(0), z(0), bc(0), next_rst(0), rst_k(0), irregular(0) {}

    // The bias of three words changes nothing for the funnel shift (it looks at np mod 32) and makes "the ring slot of word
    // ceil(p / 32)" a bit field of np itself: the policy stores word w at slot (3 - w) mod kRing, and (np >> 5) mod kRing is that slot.
    // The lane keeps no stream words in registers: a peek reads the two words around the position from the ring (one LDS
    // instruction) and funnel-shifts them. Round 2 kept a three-word register window with a look-ahead read, which took the LDS
    // round trip off the symbol-to-symbol chain and cost four vector instructions per step to slide; with the kernels bound by
    // instruction issue and four to eight waves per SIMD to hide the latency, the instructions were the worse deal.
    static constexpr uint32_t LP_NP0 = 96u;
    LP_HD uint32_t pos() const { return LP_NP0 - np; }
    LP_HD uint32_t peek() const { return m.peek_np(np); }
    LP_HD void load_window() // after a jump
    {
        const uint32_t wi = (pos() + 31u) >> 5; // ceil(p / 32)
        m.reseek(wi ? wi - 1u : 0u);
    }
    LP_HD void advance(uint32_t n) { np -= n; } // n <= 31 bits
    LP_HD void start(uint32_t pos_, uint32_t bz)
    {
        np = LP_NP0 - pos_;
        load_window();
        z = bz & 255u;
        bc = ((bz >> 8) & 31u) | (z ? 0xffffffe0u : 0u);
        rst_k = 0;
        next_rst = ic.total_bits;
        if (ic.n_rst) { // first restart boundary at or after pos (binary search)
            uint32_t lo = 0, hi = ic.n_rst;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (m.rst_bit(mid) < pos_) lo = mid + 1; else hi = mid;
            }
            rst_k = lo;
            next_rst = lo < ic.n_rst ? m.rst_bit(lo) : ic.total_bits;
            m.settle(next_rst);
        }
    }
    LP_HD uint32_t state_bz() const { return ((bc & 31u) << 8) | z; } // opaque to everything but start(): (5 x block-in-MCU << 8) | zigzag index
    LP_HD uint32_t started() const { return (bc >> 5) + (z ? 1u : 0u); } // blocks whose first symbol this lane has decoded since start() (mod 2^27)
    LP_HD bool seen_block_start() const { return (int32_t)bc >= 0; }

    // The limit the hot loops compare np with: the lane needs the slow path when p >= min(end, next_rst - 7), i.e. np <= -that.
    LP_HD int32_t limit(uint32_t end) const
    {
        const uint32_t r = next_rst - 7u; // next_rst >= 8 in any stream with a block in it; a wrapped value only sends the lane through the slow path
        const uint32_t lim = (int32_t)r < (int32_t)end ? r : end;
        return (int32_t)(LP_NP0 - lim);
    }

    // At a block start: detect the end of a restart interval (or of the stream). `pk` = peek(). Returns true when
    // the lane jumped to the boundary (DC predictors must be reset by the caller, and peek() must be redone).
    LP_HD bool restart_check(uint32_t pk)
    {
        int32_t rem = (int32_t)(next_rst - pos());
        // A boundary BEHIND the lane: only a lane decoding from a wrong state can run over one in the middle of a block. It is
        // left behind (the lane keeps going and synchronises at a later boundary or by itself): jumping back would make the
        // lane visit the same positions twice, and a checkpoint of the first visit would splice the second visit's counts in
        // again (found by tests/test_gpu_sweep.py with one-MCU restart intervals).
        while (rem < 0) {
            if (rst_k < ic.n_rst) {
                rst_k++;
                irregular = 1u;
                next_rst = rst_k < ic.n_rst ? m.rst_bit(rst_k) : ic.total_bits;
            } else
                next_rst = 0x7fffffffu;
            m.settle(next_rst);
            rem = (int32_t)(next_rst - pos());
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
        np = LP_NP0 - target;
        load_window();
        bc &= ~31u;
        z = 0;
        return true;
    }

    // Second level of the code lookup: e1 is the first-level entry of a prefix that belongs to codes longer than LP_LUT_BITS; it
    // names a slice of the second-level pool, indexed by the LP_LUT2_BITS bits that follow the prefix (see LpHuffSet).
    // One dependent LDS read: a long code is rare per lane but shows up in most steps of a 64-lane wave.
    LP_HD uint32_t long_code(uint32_t tbl, uint32_t e1, uint32_t top)
    {
        const uint32_t sub = LP_E_SLICE(e1);
        uint32_t e = sub != 0xffu ? m.lut2((sub << LP_LUT2_BITS) | (top & ((1u << LP_LUT2_BITS) - 1u))) : 0u;
        if (LP_RARE(LP_E_BITS(e) == 0)) { // canonical search (T.81 F.2.2.3, jdhuff.c jpeg_huff_decode): corrupt prefix, or a table whose long codes overflow the pool
            uint32_t len = 17, sym = 0; // no code matches: jpeg_huff_decode reads on to the sentinel length 17, warns and fakes a zero
            for (uint32_t l = LP_LUT_BITS + 1; l <= 16; l++) {
                const int32_t code = (int32_t)(top >> (16 - l));
                if (code <= m.maxcode(tbl, l)) {
                    len = l;
                    sym = m.val(tbl, (uint32_t)(code + m.valoff(tbl, l)));
                    break;
                }
            }
            e = lp_lut_entry((int)tbl, (int)len, sym);
        }
        return e;
    }

    // Decode one Huffman symbol (+ its extra bits) from pk = peek(). On return:
    //   is_dc, k = zigzag index of the coefficient (valid when has_val), val, block_done.
    // Written branch-free apart from the long-code lookup: every lane of the wave runs the same instructions.
    // NEED_VAL = false (the counting passes) skips the value reconstruction; what a pass does not read of Sym is never computed.
    struct Sym { bool is_dc; bool has_val; bool block_done; uint32_t k; int32_t val; };
    template <bool NEED_VAL>
    LP_HD Sym step(uint32_t pk)
    {
        Sym r;
        // DC table at a block start, the block's AC table inside: two bits of the block's field in ic.tb
        const uint32_t tbl = lp_bfe(ic.tb, bc + 2u * lp_min1(z), 2);
        uint32_t e;
        if (NEED_VAL || !LP_MULTI)
            e = m.lut(tbl, pk >> (32 - LP_LUT_BITS));
        else {
            // The counting passes take every symbol whose code lies inside the lookup window in ONE step (LpHuffSet::lutm: bits consumed,
            // zigzag advance, ends-the-block of the whole group in the upper half of the entry) -- unless the group would run past
            // coefficient 63 from this lane's z: then the one-symbol entry in the lower half, i.e. the step of the WRITE pass. The
            // bits above an entry's 16 stay in e: every field below is read with a bit-field extract.
            const uint32_t ec = m.lutc(tbl, pk >> (32 - LP_LUT_BITS));
            e = z + lp_bfe(ec, 25, 6) < 64u ? ec >> 16 : ec;
        }
        if (LP_E_BITS(e) == 0) e = long_code(tbl, e, pk >> 16);
        const uint32_t n = LP_E_BITS(e);          // code + extra bits
        const uint32_t s = LP_E_SIZE(e);
        const uint32_t runx = lp_bfe(e, 9, 7);    // LP_E_RUNX: run, + 64 when the symbol ends the block. DC symbols are categories 0..15 (validated by the parser): run == 0
        r.val = 0;
#ifdef LP_EXP_NOVAL
        if (false) { // timing experiment: what does the value reconstruction cost?
#else
        if (NEED_VAL) {
#endif
            const uint32_t x = lp_bfe_w(pk, 32u - n, s);  // the s extra bits that follow the code (0 when s == 0)
            // HUFF_EXTEND: a first extra bit of 0 means negative: val = x - (2^s - 1). nm = 1 - 2^s; the first bit is set iff 2x + nm > 0
            const uint32_t nm = (0xffffffffu << s) + 1u;
            r.val = (int32_t)(x + ((int32_t)(2u * x + nm) > 0 ? 0u : nm));
        }
        advance(n);
        // jdhuff.c decode_mcu: size 0 ends the block unless the run is 15 (ZRL, which skips 16 coefficients: k + 1 == z + 16); a
        // coefficient whose index overruns 63 on a corrupt stream still lands on jpeg_natural_order[64..79] = 63 (the zigzag table
        // carries the same guard entries)
        r.is_dc = z == 0;
        r.k = z + runx;                           // DC: z == run == 0; at most 63 + 15 when has_val
        r.has_val = r.is_dc || s != 0;
        const uint32_t zn = r.k + 1u;
        r.block_done = zn > 63u;
        z = r.block_done ? 0u : zn;
        // next block of the MCU, one more block completed; branch-free
        const uint32_t bn = ((bc | 31u) + 1u) | lp_bfe(ic.nx, bc, 5);
        bc = r.block_done ? bn : bc;
        return r;
    }
};

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
// Every checkpoint iteration is even (the counting loops run their steps in pairs): base is even, the geometric spacing rounded up.
LP_HD uint32_t lp_ck_next(uint32_t base, uint32_t k, uint32_t prev) { return prev + ((k < 4 || prev / 2 < base) ? base : ((prev / 2 + 1u) & ~1u)); }

// Schedule for subsequences of S bits; cbits tunes the first spacing (cbits / 32 iterations, 8 for the default 256).
inline LpCkSched lp_make_sched(uint32_t S, uint32_t cbits)
{
    LpCkSched cs;
    const uint32_t base = cbits / 32 > 2 ? (cbits / 32 + 1u) & ~1u : 2, span = S / 4 > base ? S / 4 : base; // a lane runs about S/6.5 iterations
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

// The lane's sums so far: blocks STARTED (first symbol decoded) and restart boundaries crossed since start().
template <class M>
LP_HD LpSubSum lp_lane_sum(const LpLane<M>& L, uint32_t nreset)
{
    LpSubSum s;
    s.nblk = L.started() & 0x07ffffffu;
    s.nreset = nreset;
    return s;
}

// The counting passes (SPEC, VERIFY) keep a finished lane RUNNING: masking it off costs four scalar instructions per step in a loop
// that is bound by the instructions it issues, vector and scalar alike (profiles/r03_c_final.md), while letting it decode on --
// garbage, past its subsequence -- costs nothing: its results were set aside when it finished (LpLaneExit), its limit can no longer
// be reached, its ring reads stay inside LDS and its stream loads inside the image's buffer descriptor.
struct LpLaneExit {
    uint32_t p, bz, nblk, nreset;
};
template <class M>
LP_HD LpLaneExit lp_lane_exit(const LpLane<M>& L, uint32_t nreset)
{
    LpLaneExit e;
    e.p = L.pos();
    e.bz = L.state_bz();
    e.nblk = L.started() & 0x07ffffffu;
    e.nreset = nreset;
    return e;
}

// SPEC pass for one subsequence: decode [entry.p, sub_end) from the guessed state.
// Ck must provide   void record(uint32_t k, const LpCkptPk&)   -- called by ALL lanes of the wave at the same iteration.
// Loop shape: the steps between two checkpoints run in pairs (one ring top-up per M::kEvery = 2 steps, no per-step test of the
// iteration count; every checkpoint iteration is even, see lp_ck_next); the checkpoint itself is recorded in the middle of the step
// that follows them -- after that step's restart check, before its decode, exactly where the one-test-per-step loop of round 2 had it.
template <class M, class Ck>
LP_HD void lp_spec_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpSubState entry, const LpCkSched& cs, Ck& ck, LpSubState* exit_st,
                        LpSubSum* total)
{
    static_assert(M::kEvery == 2, "the step pairs of the counting loops assume a top-up every second step");
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    uint32_t nreset = 0;
    const uint32_t K = cs.K, ck_base = cs.base;
    uint32_t k = 0, steps = 0, next_ck = K ? lp_ck_next(ck_base, 0, 0) : 0xffffffffu;
    bool done = false, live = true, near = false;
    int32_t nlim = L.limit(sub_end);
    LpLaneExit ex = lp_lane_exit(L, 0);
    uint32_t pk = 0;
    // first half of a step: the bits at the position and, rarely per wave, the restart check
    auto pre = [&] {
        pk = L.peek();
        near = m.any((int32_t)L.np <= nlim); // the subsequence ends here, or a restart boundary / the stream end is near
        if (LP_RARE(near)) {
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && L.z == 0 && L.restart_check(pk)) nreset++; // also catches the padded end of the stream
            pk = L.peek();
        }
    };
    // second half: a lane that has reached the end of its subsequence sets its results aside; then the decode, for every lane
    auto post = [&] {
        if (LP_RARE(near)) {
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && L.pos() >= sub_end) {
                ex = lp_lane_exit(L, nreset);
                done = true;
            }
            nlim = done ? (int32_t)0x80000000 : L.limit(sub_end);
            live = m.any(!done);
        }
        (void)L.template step<false>(pk);
    };
    while (live) {
        // the whole pairs before the next checkpoint (all of them, two at a time, once the checkpoints have run out)
        while (live && steps != next_ck) {
            m.topup(L.pos());
            pre(); post();
            pre(); post();
            steps += 2;
        }
        if (!live) break;
        m.topup(L.pos());
        pre();
        { // wave-uniform: every lane records checkpoint k now -- a finished lane its exit state again
            const LpLaneExit now = lp_lane_exit(L, nreset);
            LpCkptPk c;
            c.p = done ? ex.p : now.p; c.bz = done ? ex.bz : now.bz; c.nblk = done ? ex.nblk : now.nblk; c.nreset = done ? ex.nreset : now.nreset;
            ck.record(k, c);
            k++;
            next_ck = k < K ? lp_ck_next(ck_base, k, next_ck) : 0xffffffffu;
        }
        post();
        pre(); post();
        steps += 2;
    }
    LpCkptPk none;
    none.p = 0xffffffffu; none.bz = 0; none.nblk = 0; none.nreset = 0;
    for (; k < K; k++) ck.record(k, none);
    exit_st->p = ex.p;
    exit_st->bz = ex.bz;
    total->nblk = ex.nblk;
    total->nreset = ex.nreset;
}

// VERIFY pass for one subsequence: decode from `entry` (= current exit state of the previous subsequence) until the state
// matches a checkpoint recorded by the SPEC pass, then splice the recorded remainder (spec_exit / spec_total are the SPEC
// pass's immutable results). Without a match the lane runs to the end of the subsequence.
// Ck must provide   uint32_t pos(uint32_t k)   (cheap: position of checkpoint k, 0xffffffff if not recorded)
//                   LpCkptPk load(uint32_t k)  (the whole record; called once or twice per lane)
// Outputs the new exit state and total of the subsequence.
template <class Ck>
LP_HD int32_t lp_ck_npos(Ck& ck, uint32_t kk, uint32_t K) // checkpoint kk's position, negated like the lane's; "none" can never be reached
{
    const uint32_t cp = kk < K ? ck.pos(kk) : 0xffffffffu;
    return cp == 0xffffffffu ? (int32_t)0x80000000 : (int32_t)(96u - cp); // LpLane::LP_NP0
}
template <class M, class Ck>
LP_HD void lp_verify_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpSubState entry, uint32_t K, Ck& ck, const LpSubState& spec_exit,
                          const LpSubSum& spec_total, LpSubState* exit_st, LpSubSum* total)
{
    static_assert(M::kEvery == 2, "the step pairs of the counting loops assume a top-up every second step");
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    uint32_t nreset = 0, kk = 0;
    const int32_t never = (int32_t)0x80000000;
    int32_t ncp = lp_ck_npos(ck, 0, K);
    bool done = false, live = true;
    int32_t nlim = L.limit(sub_end);
    LpLaneExit ex = lp_lane_exit(L, 0);
    auto one_step = [&] { // same shape as a SPEC step, plus the look at the lane's next checkpoint
        uint32_t pk = L.peek();
        const bool near = m.any((int32_t)L.np <= nlim);
        if (LP_RARE(near)) {
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && L.z == 0 && L.restart_check(pk)) nreset++;
            pk = L.peek();
        }
        bool now_done = false;
        if (LP_RARE(m.any(ncp >= (int32_t)L.np))) { // some lane stands at or behind its next checkpoint (a finished lane's is out of reach)
            LP_KEEP_UNIFORM_BRANCH();
            if (!done) {
                while (ncp > (int32_t)L.np) { // checkpoints are strictly ordered until the lane that recorded them finished
                    kk++;
                    ncp = lp_ck_npos(ck, kk, K);
                }
                if (ncp == (int32_t)L.np && kk < K) {
                    LpSubState cst;
                    LpSubSum csum;
                    lp_ckpt_unpack(ck.load(kk), cst, csum);
                    if (cst.bz == L.state_bz()) { // synchronised with the recorded trajectory at checkpoint kk
                        const LpSubSum t = lp_sum_combine(lp_lane_sum(L, nreset), lp_sum_tail(spec_total, csum));
                        ex.p = spec_exit.p; ex.bz = spec_exit.bz; ex.nblk = t.nblk; ex.nreset = t.nreset;
                        now_done = true;
                    } else {
                        kk++; // same position, different state: this checkpoint can never match
                        ncp = lp_ck_npos(ck, kk, K);
                    }
                }
            }
        }
        if (LP_RARE(near || m.any(now_done))) {
            LP_KEEP_UNIFORM_BRANCH();
            if (!done && !now_done && L.pos() >= sub_end) {
                ex = lp_lane_exit(L, nreset);
                now_done = true;
            }
            done = done || now_done;
            if (done) ncp = never;
            nlim = done ? never : L.limit(sub_end);
            live = m.any(!done);
        }
        // (masked here, unlike in the SPEC pass: most lanes of a verify wave are finished most of the time -- a walk is a tenth of a
        // subsequence -- and sixty lanes decoding garbage cost the few that still work their LDS and load bandwidth: measured 7.6
        // against 7.1 us per image)
        if (!done) (void)L.template step<false>(pk);
    };
    while (live) {
        if (!done) m.topup(L.pos());
        one_step();
        one_step();
    }
    exit_st->p = ex.p;
    exit_st->bz = ex.bz;
    total->nblk = ex.nblk;
    total->nreset = ex.nreset;
}

// WRITE pass for one subsequence. Sink S must provide:
//   void put_dc(int32_t v, bool on);               when `on`: the block's DC DIFFERENCE (made absolute later, see lp_dc_walk)
//   void put(ZZ where, int32_t v);                 store one AC coefficient of the block being decoded; `where` = the caller's zigzag table
//                                                  entry for it (the natural index, or whatever address form the sink precomputed there)
//   void end_block(uint32_t bc, bool on);          when `on`: a block is complete (queued for flushing); bc = the lane's LpLane::bc after the
//                                                  step: the block's decode-order index is (first block of the lane) + (bc >> 5) - 1
//   bool stalled();                                no free slot: the lane must wait for the next flush
//   void flush();                                  wave-uniform: write out every queued block
//   void finish();                                 once, after the last flush
// Returns the number of blocks written.
#ifndef LP_FLUSH_EVERY
#define LP_FLUSH_EVERY 4   // 2 / 3 / 4 / 6 measured (8-word ring, top-up every 2 steps): 4 is the best trade of flush instructions against lanes waiting for the flush
#endif
// *irregular is set when the restart intervals of the stream do not hold exactly their MCUs (an interval that is short of blocks or
// holds more data than its blocks need, a boundary inside a block): what libjpeg makes of such a stream -- zero bits to the end of
// the interval, surplus data dropped at the marker -- is the serial decoder's business (lp_jbits.h); the caller sends the image there.
template <class M, class Sink, class ZZ>
LP_HD uint32_t lp_write_pass(M& m, const LpImgCtx& ic, LpSubState entry, uint32_t end_p, const LpSubSum& prefix, const ZZ* zigzag,
                             Sink& sink, bool* irregular)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    // A lane that enters mid-block skips to the first block start: L.bc counts that partial block from -1, so the block that
    // completes when the count field reads c (after the step) is block prefix.nblk + c - 1 of the image, and the lane is "writing"
    // exactly while the field is not negative before the step.
    const uint32_t left = ic.total_blocks > prefix.nblk ? ic.total_blocks - prefix.nblk : 0u; // blocks this lane may still write
    int32_t left_bc = (int32_t)(left << 5); // ... as a bound on L.bc; lowered to "the block at hand" once the lane is past its end (below)
    const uint32_t stream_end = end_p < ic.total_bits ? end_p : ic.total_bits;
    static_assert(M::kEvery == 2 && LP_FLUSH_EVERY == 4, "the step groups of the WRITE loop are written out for a top-up every second and a flush every fourth step");
    bool done = false;
    uint32_t written = 0;
    int32_t nlim = L.limit(stream_end);
    bool live = true;
    auto one_step = [&] {
        uint32_t pk = L.peek();
        const bool act = LP_X_FLUSH2 ? !done : !done && !sink.stalled();
        bool go = act;
        // slow path: the lane is at / past the end of its subsequence or of the stream, near a restart boundary, or out of blocks
        // (a stalled lane may cast a vote too: it costs a pass through here and changes nothing)
        if (LP_RARE(m.any2((int32_t)L.np <= nlim, (int32_t)L.bc >= left_bc))) {
            LP_KEEP_UNIFORM_BRANCH();
            if (act) {
                if (L.z == 0) { // DC predictors restart in k_dc_scan, by MCU index
                    const uint32_t k0 = L.rst_k;
                    // crossing into interval rst_k: exactly rst_k x (blocks per interval) blocks must lie behind the lane
                    if (L.restart_check(pk) && L.rst_k != k0 && prefix.nblk + (uint32_t)((int32_t)L.bc >> 5) != L.rst_k * ic.rst_blocks) L.irregular = 1u;
                }
                pk = L.peek();
                // a lane stops at the first block start at or after the end of its subsequence (or when the stream is truncated)
                const uint32_t p = L.pos();
                const bool stop = L.z == 0 ? (p >= end_p || (int32_t)L.bc >= left_bc) : p >= ic.total_bits;
                // the stream ends inside a block this lane is writing: libjpeg finishes the block (and its MCU) on zero bits -- the serial decoder's case
                if (stop && L.z != 0 && L.seen_block_start()) L.irregular = 1u;
                done = stop;
                go = !stop;
                // past the end of the subsequence inside a block: the lane finishes the block. Instead of coming through here at every
                // step until then, it is called back by the block count (and by the end of the stream)
                const bool finishing = !stop && L.z != 0 && p >= end_p;
                if (finishing) left_bc = (int32_t)((L.bc | 31u) + 1u);
                nlim = stop ? (int32_t)0x80000000 : L.limit(finishing ? ic.total_bits : stream_end);
                if (stop) left_bc = 0x7fffffff;
            }
            live = m.any(!done);
        }
        if (go) {
            const bool writing = L.seen_block_start();
            const typename LpLane<M>::Sym s = L.template step<true>(pk);
            // few, flat predicated regions: every exec-mask branch costs scalar instructions in a loop that is issue bound
            sink.put_dc(s.val, writing && s.is_dc);
            if (writing && !s.is_dc && s.has_val) sink.put(zigzag[s.k], s.val);
            const bool bd = writing && s.block_done;
            sink.end_block(L.bc, bd);
            written += bd ? 1u : 0u;
        }
    };
    // groups of four steps: a ring top-up before the second and the fourth, a flush before the fourth (the order the one-step loop
    // with its tests of the iteration count had); whether anyone is still working is looked at once per group
#if LP_X_FLUSH2
    while (live) {
        one_step();
        one_step();
        m.topup(L.pos());
        sink.flush();
    }
#else
    while (live) {
        one_step();
        m.topup(L.pos());
        one_step();
        one_step();
        m.topup(L.pos());
        sink.flush();
        one_step();
    }
#endif
    sink.flush();
    sink.finish();
    *irregular = L.irregular != 0u;
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
