// lp_tok_core.h -- the entropy decoder as ONE serial pass that emits fixed-size tokens, plus a data-parallel expansion of the tokens
// into coefficient blocks (S1 of SURVEY.md 2a; replaces libjpeg-turbo's jdhuff.c decode_mcu behind opencv_decoder_read_data,
// /root/reference/opencv.cpp:166-171).
//
// Round 2 decoded every symbol twice: a counting pass (SPEC, to learn where every subsequence really starts and how many blocks
// precede it) and a writing pass (WRITE, 2x the cost of SPEC: value reconstruction, zigzag lookup, LDS block slots, cooperative
// flushes, all inside the serial symbol loop). Here the speculative pass itself leaves what it decoded behind:
//   * every decode step emits exactly ONE 32-bit token -- (size, zigzag index after the symbol, the 16 stream bits that follow the
//     code) -- so lane i's token t lives at a fixed address T_s[i][t]: no offsets to compute, no slots, no flush; four tokens leave as
//     one 16-byte store per lane every fourth step;
//   * a lane's tokens are garbage until it has synchronised with the true symbol sequence and exact afterwards, so the verify pass
//     (which walks from the predecessor's true exit state until it meets one of the lane's checkpoints) writes the HEAD tokens of the
//     subsequence to T_v[i][0 .. h) and records where the speculative ones take over: T_s[i][c .. n);
//   * the expansion of tokens into 64-coefficient blocks needs no serial state at all: the token carries the zigzag position, the
//     block index is a count of block-end flags (a ballot + mbcnt per 64 tokens) on top of the subsequence's first block index from
//     the scan. It runs with every lane busy on coalesced loads -- the work the serial loop no longer does.
// The verify pass is resumable (LpVerState): the kernels run it in phases with a budget of steps each and compact the lanes that
// are still walking between phases, because a wave is as slow as its slowest lane and synchronisation distances have a long tail
// (measured on the 4096 x 4096 q90 workload: median 107 steps, p90 359, p99 793 of ~1500 - 3000 steps per subsequence).
#pragma once
#include "lp_huff_core.h"

// ---- token
//   bits  3..0   s   number of extra bits (the symbol's size category); 0 = no coefficient (EOB, ZRL) -- or a DC difference of 0
//   bits 10..4   zn  zigzag index AFTER the symbol: z + run + 1, or 64 for an end-of-block symbol. zn >= 64 ends the block; zn == 1 is
//                    the block's DC symbol (an AC symbol starts at z >= 1, so its zn is at least 2); the coefficient's zigzag index is
//                    zn - 1 (indices past 63 land on 63 like jpeg_natural_order's guard entries)
//   bits 31..16      the 16 stream bits after the code: the top s of them are the extra bits (HUFF_EXTEND happens in the expansion)
LP_HD uint32_t lp_tok_make(uint32_t pk, uint32_t len, uint32_t zn, uint32_t s) { return ((pk << len) & 0xffff0000u) | (zn << 4) | s; }
LP_HD uint32_t lp_tok_s(uint32_t t) { return t & 15u; }
LP_HD uint32_t lp_tok_zn(uint32_t t) { return (t >> 4) & 127u; }
LP_HD bool lp_tok_ends_block(uint32_t t) { return (t & 0x400u) != 0; } // zn >= 64
LP_HD int32_t lp_tok_value(uint32_t t) // jdhuff.c HUFF_EXTEND of the extra bits; 0 when s == 0
{
    const uint32_t s = t & 15u;
    const uint32_t x = (t >> 16) >> (16u - s);                 // s == 0: the whole field shifted out
    const uint32_t neg = (t & 0x80000000u) ? 0u : 0xffffffffu; // a first extra bit of 0 means negative
    return s ? (int32_t)(x - (neg & ((1u << s) - 1u))) : 0;
}
// Most tokens a subsequence of S bits can produce. Whatever state a lane is in, it walks the MCU's blocks in order, and a block costs
// at least its tables' shortest DC code + shortest AC code for two tokens (DC symbol, end of block): with min_mcu_bits = that sum over
// the MCU's blocks, S bits hold at most S * 2 * bpm / min_mcu_bits tokens (+ an MCU for the ends, + a window of the device sink).
// Annex-K tables: 2 bits per token; optimised chroma tables often carry 1-bit codes (data/large-sunrise.jpg: 20 bits per 12-token MCU).
LP_HD uint32_t lp_tok_cap(uint32_t S, uint32_t bpm, uint32_t min_mcu_bits)
{
    const uint32_t mb = min_mcu_bits ? min_mcu_bits : 1u;
    return ((uint32_t)(((uint64_t)S * 2u * bpm + mb - 1u) / mb) + 4u * bpm + 32u + 15u) & ~15u; // a multiple of 16: the device sink stores windows of 16 tokens
}
// shortest code of table slot t (0 = DC0, 1 = DC1, 2 = AC0, 3 = AC1)
inline uint32_t lp_huff_min_len(const LpHuffSet& hs, int t)
{
    for (uint32_t l = 1; l <= 16; l++)
        if (hs.maxcode[t][l] >= 0) return l;
    return 16;
}
inline uint32_t lp_min_mcu_bits(const LpHuffSet& hs, uint64_t blkpack, uint32_t bpm)
{
    uint32_t bits = 0;
    for (uint32_t b = 0; b < bpm; b++) {
        const uint32_t nib = (uint32_t)(blkpack >> (4 * b)) & 15u;
        bits += lp_huff_min_len(hs, (int)((nib >> 2) & 1u)) + lp_huff_min_len(hs, 2 + (int)((nib >> 3) & 1u));
    }
    return bits;
}

// What the verify pass leaves per subsequence for the expansion: T_v[0 .. head) then T_s[spec_from .. spec_n).
struct LpTokSpan {
    uint32_t head;          // tokens the verify pass wrote (0 for the first subsequence of an image: its speculative pass IS exact)
    uint32_t spec_from;     // first speculative token that is exact (= iteration of the matched checkpoint; spec_n when nothing matched)
};

// A verify walk that can be put down and picked up again.
struct LpVerState {
    uint32_t p, bz;         // decoder state before the next step
    uint32_t iter;          // steps done so far = tokens written to T_v
    uint32_t kk, ck_iter;   // next checkpoint of the subsequence to look at, and the speculative iteration it was taken at
    uint32_t nblk, nreset;  // sums so far
    uint32_t pad;
};

// One decode step that also builds the token. Same arithmetic as LpLane::step<false>.
template <class M>
LP_HD uint32_t lp_step_tok(LpLane<M>& L, uint32_t pk)
{
    M& m = L.m;
    const LpImgCtx& ic = L.ic;
    const bool is_dc = L.z == 0;
    const uint32_t tbl = ((ic.blkpack >> (L.b4 + (is_dc ? 2u : 3u))) & 1u) + (is_dc ? 0u : 2u);
    uint32_t e = m.lut(tbl, pk >> (32 - LP_LUT_BITS));
    if ((e & 0x1f00u) == 0) e = L.long_code(tbl, e, pk >> 16);
    const uint32_t len = (e >> 8) & 31u;
    const uint32_t s = e & 15u;
    const uint32_t run = (e >> 4) & 15u;
    const bool eob = (e & 0x8000u) != 0;
    const uint32_t zn = eob ? 64u : L.z + run + 1u;
    const uint32_t tok = lp_tok_make(pk, len, zn, s);
    L.advance(len + s);
    const bool block_done = zn >= 64u;
    L.z = block_done ? 0u : zn;
    const uint32_t nb4 = L.b4 + 4u == 4u * ic.bpm ? 0u : L.b4 + 4u;
    L.b4 = block_done ? nb4 : L.b4;
    return tok;
}

// SPEC pass that emits tokens. Tk must provide   void put(uint32_t u, uint32_t step, uint32_t tok, bool on)   -- called in every step by
// all lanes of the wave (u = position of the step inside its group of four, a compile-time constant after unrolling; step = the token's
// index, the same in every lane that is `on`; `on` = the lane decoded a symbol in this step) -- and   void finish(uint32_t step)   once
// after the loop. The device sink parks a window of 16 tokens per lane in LDS and the wave writes the windows out together, four lanes
// per 64-byte run (per-lane 16-byte stores 32 KB apart made the speculative pass 2.2x slower: 64 lines per store instruction).
// Ck as in lp_spec_pass. *ntok = tokens emitted (= decode steps of the lane).
template <class M, class Ck, class Tk>
LP_HD void lp_spec_tok_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpSubState entry, const LpCkSched& cs, Ck& ck, Tk& tk, LpSubState* exit_st,
                            LpSubSum* total, uint32_t* ntok)
{
    LpLane<M> L(m, ic);
    L.start(entry.p, entry.bz);
    LpSubSum sum;
    lp_sum_zero(sum);
    const uint32_t K = cs.K, ck_base = cs.base;
    uint32_t k = 0, iter = 0, next_ck = K ? lp_ck_next(ck_base, 0, 0) : 0xffffffffu, n = 0;
    bool done = false;
    do {
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if ((u & (M::kEvery - 1)) == M::kEvery - 1) m.topup(L.p); // kEvery is 2 or 4: a compile-time position inside the group
            uint32_t pk = L.peek();
            if (m.any_lt8(lp_near_boundary(L.next_rst, L.p, L.z, done))) {
                LP_KEEP_UNIFORM_BRANCH();
                if (!done && L.z == 0 && L.restart_check(pk)) sum.nreset++;
                pk = L.peek();
            }
            if (iter == next_ck) { // wave-uniform
                LpSubState st;
                st.p = L.p;
                st.bz = L.state_bz();
                ck.record(k, lp_ckpt_pack(st, sum));
                k++;
                next_ck = k < K ? lp_ck_next(ck_base, k, next_ck) : 0xffffffffu;
            }
            done = done || L.p >= sub_end;
            uint32_t tok = 0;
            if (!done) {
                sum.nblk += L.z == 0 ? 1u : 0u;
                tok = lp_step_tok(L, pk);
                n++;
            }
            tk.put(u, iter, tok, !done);
            iter++;
        }
    } while (m.any(!done));
    tk.finish(iter);
    LpCkptPk none;
    none.p = 0xffffffffu; none.bz = 0; none.nblk = 0; none.nreset = 0;
    for (; k < K; k++) ck.record(k, none);
    exit_st->p = L.p;
    exit_st->bz = L.state_bz();
    *total = sum;
    *ntok = n;
}

// VERIFY pass that emits the head tokens, in instalments: runs from `vs` until the subsequence synchronises with one of its
// checkpoints, ends, or vs.iter reaches `until_iter` (a multiple of 4). Returns true when the walk is over; then *exit_st / *total /
// *span are set. Ck as in lp_verify_pass plus   uint32_t pos(k). Tk as above (tokens go to T_v).
// active = false: a lane without a walk of its own that stays with the wave (the device sink writes the tokens out cooperatively).
template <class M, class Ck, class Tk>
LP_HD bool lp_verify_tok_pass(M& m, const LpImgCtx& ic, uint32_t sub_end, LpVerState& vs, uint32_t until_iter, uint32_t K, uint32_t ck_base, Ck& ck, Tk& tk,
                              const LpSubState& spec_exit, const LpSubSum& spec_total, uint32_t spec_n, LpSubState* exit_st, LpSubSum* total, LpTokSpan* span,
                              bool active = true)
{
    LpLane<M> L(m, ic);
    L.start(vs.p, vs.bz);
    LpSubSum sum;
    sum.nblk = vs.nblk; sum.nreset = vs.nreset;
    uint32_t kk = vs.kk, ck_iter = vs.ck_iter, iter = vs.iter;
    uint32_t step = m.uniform(vs.iter); // every walk of a wave is at the same step: 0 in the first instalment, the previous budget afterwards
    uint32_t cp = kk < K ? ck.pos(kk) : 0xffffffffu;
    bool done = !active, over = false, spliced = false;
    do {
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if ((u & (M::kEvery - 1)) == M::kEvery - 1) m.topup(L.p);
            uint32_t pk = L.peek();
            if (m.any_lt8(lp_near_boundary(L.next_rst, L.p, L.z, done))) {
                LP_KEEP_UNIFORM_BRANCH();
                if (!done && L.z == 0 && L.restart_check(pk)) sum.nreset++;
                pk = L.peek();
            }
            if (!done) {
                while (cp < L.p) { // checkpoints are strictly ordered until the lane that recorded them finished
                    kk++;
                    ck_iter = lp_ck_next(ck_base, kk, ck_iter);
                    cp = kk < K ? ck.pos(kk) : 0xffffffffu;
                }
                if (cp == L.p && kk < K) {
                    LpSubState cst;
                    LpSubSum csum;
                    lp_ckpt_unpack(ck.load(kk), cst, csum);
                    if (cst.bz == L.state_bz()) { // synchronised with the recorded trajectory at checkpoint kk: its tokens from ck_iter on are exact
                        *total = lp_sum_combine(sum, lp_sum_tail(spec_total, csum));
                        *exit_st = spec_exit;
                        span->head = iter;
                        span->spec_from = ck_iter;
                        done = over = spliced = true;
                    } else {
                        kk++;
                        ck_iter = lp_ck_next(ck_base, kk, ck_iter);
                        cp = kk < K ? ck.pos(kk) : 0xffffffffu;
                    }
                }
            }
            if (!done && L.p >= sub_end) { done = over = true; }
            uint32_t tok = 0;
            const bool on = !done;
            if (on) {
                sum.nblk += L.z == 0 ? 1u : 0u;
                tok = lp_step_tok(L, pk);
            }
            tk.put(u, step, tok, on);
            iter += on ? 1u : 0u;
            step++;
        }
        done = done || iter >= until_iter; // the budget (a multiple of 16) is checked between groups of four: a token window is never split over two instalments
    } while (m.any(!done));
    tk.finish(step);
    if (over && !spliced) {
        exit_st->p = L.p;
        exit_st->bz = L.state_bz();
        *total = sum;
        span->head = iter;
        span->spec_from = spec_n;
    }
    vs.p = L.p; vs.bz = L.state_bz(); vs.iter = iter; vs.kk = kk; vs.ck_iter = ck_iter; vs.nblk = sum.nblk; vs.nreset = sum.nreset;
    return over;
}

// ---- expansion, as the device does it per token (host: one token at a time; tests/emu)
// Sink must provide   void coef(uint32_t blk, uint32_t zigzag_k, int32_t v)   and   void dc(uint32_t blk, int32_t diff).
// Walks T_v[0 .. head) then T_s[spec_from .. spec_n) of one subsequence. `blk` = index of the block the first token belongs to
// (the block that is open at the subsequence's entry state, or the first one that starts in it). Returns the block index after the
// last token.
template <class Sink>
inline uint32_t lp_expand_tokens(const uint32_t* tv, const uint32_t* ts, const LpTokSpan& span, uint32_t spec_n, uint32_t blk, uint32_t total_blocks, Sink& sink)
{
    const uint32_t n = span.head + (spec_n > span.spec_from ? spec_n - span.spec_from : 0u);
    for (uint32_t j = 0; j < n; j++) {
        const uint32_t t = j < span.head ? tv[j] : ts[span.spec_from + (j - span.head)];
        const uint32_t zn = lp_tok_zn(t), s = lp_tok_s(t);
        if (blk < total_blocks) {
            if (zn == 1u) sink.dc(blk, lp_tok_value(t));
            else if (s) sink.coef(blk, zn - 1u, lp_tok_value(t));
        }
        blk += lp_tok_ends_block(t) ? 1u : 0u;
    }
    return blk;
}
