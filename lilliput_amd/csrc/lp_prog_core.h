// lp_prog_core.h -- one scan of a progressive (SOF2) JPEG, decoded by one lane.
//
// Replaces libjpeg-turbo's jdphuff.c decode_mcu_DC_first / decode_mcu_AC_first / decode_mcu_DC_refine / decode_mcu_AC_refine
// behind opencv_decoder_read_data (/root/reference/opencv.cpp:166-171): successive approximation (Ah/Al), spectral selection
// (Ss..Se), EOB runs and the correction bits of refinement scans, accumulated into the image's coefficient arena.
// The stream is the scan's unstuffed entropy-coded segment (k_unstuff_*): restart markers are already cut out and their bit
// positions recorded, bits past the end read as zero (libjpeg feeds zeros once it meets the next marker).
//
// Host+device like lp_huff_core.h, so that tests/emu runs the same logic on the CPU (development aid, never a fallback).
#pragma once
#include "lp_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LP_PHD __host__ __device__ __forceinline__
#else
#define LP_PHD inline
#endif

// Blocks are stored in ZIGZAG order (element k = the k-th coefficient of the scan order): a refinement scan walks a band of
// consecutive k, and the block's non-zero history is then one 64-bit mask whose bit k belongs to element k -- the walk over
// "the next r still-zero coefficients, correcting every non-zero one on the way" becomes a few bit operations instead of a
// visit to each of up to 63 coefficients. The IDCT undoes the order when it loads a block.
// natural (row-major) index -> zigzag index, for the consumers of the arena
#define LP_NAT2ZIGZAG_INIT                                                                                 \
    {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, \
     18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, \
     50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63}

// Memory policy P must provide:
//   uint32_t word(uint32_t w)                      word w of the clean stream, bit 31 first (any w the reader asks for below the capacity)
//   uint32_t rst_bit(uint32_t k)                   bit position of the k-th restart boundary
//   uint32_t lut8(uint32_t s, uint32_t i); int32_t maxcode(s, l), valoff(s, l); uint32_t val(s, i)    LpProgHuff of the scan
//   void st(uint32_t blk, uint32_t k, int32_t v)   store coefficient k (zigzag index) of a block (first scans: write only)
//   int32_t ld(uint32_t blk, uint32_t k)
//   uint64_t open(uint32_t blk)                    stage one block for an AC refinement; returns its non-zero mask (bit k = element k)
//   int32_t get(uint32_t k) / void set(uint32_t k, int32_t v) / void close(uint32_t blk)
template <class P>
struct LpProgBits {
    P& m;
    uint32_t p;         // next unread bit
    uint32_t total;     // end of the data the decoder may use: the next restart boundary or the end of the scan. libjpeg stops
                        // feeding real bytes at ANY marker and stuffs zero bits from there on (jdhuff.c jpeg_fill_bit_buffer)
    uint32_t cw, w0, w1;
    bool insufficient;  // jdhuff.c insufficient_data: a read went past `total` (JWRN_HIT_MARKER); until the next restart marker
                        // the remaining MCUs are left as they are
    uint32_t all_bits, n_rst, rst_k; // the whole stream, its restart boundaries, the next one to cross

    LP_PHD LpProgBits(P& m_, uint32_t total_bits, uint32_t n_rst_)
        : m(m_), p(0), total(n_rst_ ? m_.rst_bit(0) : total_bits), cw(0xfffffff0u), w0(0), w1(0), insufficient(false), all_bits(total_bits), n_rst(n_rst_), rst_k(0) {}
    // process_restart over an unstuffed stream (k_unstuff_*: the markers are cut out, their positions listed; the numbers of the RSTn
    // are not kept -- the device-lane build trusts them, the host reader lp_jbits.h does what libjpeg does with a wrong one)
    LP_PHD bool restart()
    {
        if (rst_k < n_rst) { // the marker is there: decoding resumes behind it ("reset out-of-data flag, unless ... up against end of data")
            seek(m.rst_bit(rst_k), rst_k + 1u < n_rst ? m.rst_bit(rst_k + 1u) : all_bits);
            insufficient = false;
        } else
            seek(all_bits, all_bits);
        rst_k++;
        return true;
    }
    LP_PHD void mcu_begin(uint32_t, bool) {}
    LP_PHD bool mcu_redo() { return false; }
    LP_PHD bool failed() const { return false; }
    LP_PHD uint32_t get_each(uint32_t n) { return get(n); } // n single-bit reads (libjpeg reads correction bits one by one; here the same bits)
    LP_PHD void seek(uint32_t pos, uint32_t new_total) // start of a restart interval: the cached words were cut at the old end
    {
        p = pos;
        total = new_total;
        cw = 0xfffffff0u;
    }
    LP_PHD uint32_t load(uint32_t w)
    {
        const uint32_t base = w << 5;
        if (base >= total) return 0u;
        uint32_t v = m.word(w);
        if (total - base < 32u) v &= ~(0xffffffffu >> (total - base));
        return v;
    }
    LP_PHD uint32_t peek() // the next 32 bits
    {
        const uint32_t w = p >> 5;
        if (w != cw) {
            if (w == cw + 1u) { w0 = w1; w1 = load(w + 1u); }
            else { w0 = load(w); w1 = load(w + 1u); }
            cw = w;
        }
        return (uint32_t)((((((uint64_t)w0) << 32) | w1) << (p & 31u)) >> 32);
    }
    LP_PHD uint32_t get(uint32_t n) // n <= 32
    {
        if (!n) return 0u;
        const uint32_t v = peek() >> (32u - n);
        p += n;
        insufficient = insufficient || p > total;
        return v;
    }
    LP_PHD uint32_t sym(uint32_t s) // jdhuff.c HUFF_DECODE / jpeg_huff_decode
    {
        const uint32_t pk = peek();
        uint32_t e = m.lut8(s, pk >> 24);
        uint32_t len = 17, v = 0; // no match (JWRN_HUFF_BAD_CODE): jpeg_huff_decode has walked on to the sentinel length 17 and fakes a zero
        if (e) { len = e >> 8; v = e & 255u; }
        else
            for (uint32_t l = 9; l <= 16; l++) {
                const int32_t code = (int32_t)(pk >> (32u - l));
                if (code <= m.maxcode(s, l)) { len = l; v = m.val(s, (uint32_t)(code + m.valoff(s, l)) & 255u); break; }
            }
        p += len;
        insufficient = insufficient || p > total;
        return v;
    }
};

LP_PHD int32_t lp_prog_extend(uint32_t v, uint32_t s) // HUFF_EXTEND
{
    return v < (1u << (s - 1u)) ? (int32_t)v - (int32_t)((1u << s) - 1u) : (int32_t)v;
}

LP_PHD uint32_t lp_popc64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll((unsigned long long)v);
#else
    return (uint32_t)__builtin_popcountll(v);
#endif
}
LP_PHD uint32_t lp_ctz64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((unsigned long long)v) - 1);
#else
    return (uint32_t)__builtin_ctzll(v);
#endif
}

// One correction bit for every coefficient of `bits` (ascending = scan order): a set bit moves a coefficient whose p1 bit is
// still clear away from zero by p1.
// A memory policy may bring its own versions of the two bit-mask walks of a refinement scan (P::kBulkCorrect: the host policy of
// lp_prog_host.cpp does them with PDEP where the CPU has it) -- same reads from the bit reader, same stores.
template <class P, class = void> struct LpProgHasBulk { static constexpr bool value = false; };
template <class P> struct LpProgHasBulk<P, decltype((void)P::kBulkCorrect)> { static constexpr bool value = true; };

// A store OUTSIDE the scan's band Ss..Se (only a damaged stream does that: a run that carries an AC-first scan past Se, or a refinement
// scan that runs out of still-zero coefficients and places its new one at Se + 1, jdphuff.c). libjpeg decodes the scans one after the
// other, so a later scan that owns that coefficient overwrites -- or refines -- the stray value; the dependency levels of
// lp_prog_levels let scans with disjoint bands run side by side and in any order, which is only the same thing while every scan stays
// inside its band. A policy that has P::stray() is told (the host threads then decode the image again in file order, the device
// decoders hand it to them); one without it (the CPU emulation of the tests: scans in file order anyway) is not.
template <class P, class = void> struct LpProgHasStray { static constexpr bool value = false; };
template <class P> struct LpProgHasStray<P, decltype((void)&P::stray)> { static constexpr bool value = true; };
template <class P>
LP_PHD void lp_prog_stray(P& m)
{
    if constexpr (LpProgHasStray<P>::value) m.stray();
}

template <class P, class B>
LP_PHD void lp_prog_correct(P& m, B& b, uint64_t bits, int32_t p1, int32_t m1)
{
    if constexpr (LpProgHasBulk<P>::value) {
        m.correct_bulk(b, bits, p1, m1);
        return;
    }
    while (bits) { // the correction bits of up to 32 coefficients come out of the stream in one read
        uint32_t n = lp_popc64(bits);
        if (n > 32u) n = 32u;
        uint32_t v = b.get_each(n) << (32u - n); // first coefficient's bit on top
        for (; n; n--, v <<= 1) {
            const uint32_t e = lp_ctz64(bits);
            bits &= bits - 1ull;
            if (v & 0x80000000u) {
                const int32_t co = m.get(e);
                if ((co & p1) == 0) m.set(e, co >= 0 ? co + p1 : co + m1);
            }
        }
    }
}

// Reader B: LpProgBits (an unstuffed stream with listed restart boundaries) or LpJBits (lp_jbits.h: the raw bytes, read the way libjpeg
// reads them under cv::JpegDecoder's source manager). Returns false when the reader ran out of bytes (LpJBits only): the image fails.
template <class P, class B>
LP_PHD bool lp_prog_scan_with(P& m, B& b, const LpProgScan& sc)
{
    const uint32_t Ss = sc.Ss, Se = sc.Se, Ah = sc.Ah, Al = sc.Al;
    const int32_t p1 = 1 << Al, m1 = -(1 << Al);
    int32_t pred[4] = {0, 0, 0, 0};
    uint32_t eobrun = 0, rst_left = sc.dri, bpm = 0;
    for (uint32_t s = 0; s < sc.ns; s++) bpm += (uint32_t)sc.hs[s] * sc.vs[s];
    for (uint32_t my = 0; my < sc.mcuy; my++)
        for (uint32_t mx = 0; mx < sc.mcux; mx++) {
            if (sc.dri && rst_left == 0) { // process_restart: the rest of the interval's bits are dropped, predictors and the EOB run start over
                if (!b.restart()) return false;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                eobrun = 0;
                rst_left = sc.dri;
            }
            // "If we've run out of data, don't modify the MCU" (every decode_mcu_* but the DC refinement, which cannot change anything
            // with zero bits anyway)
            if (b.insufficient && (sc.sequential || !(Ss == 0 && Ah != 0))) {
                if (sc.dri) rst_left--;
                continue;
            }
            if (sc.sequential) { // jdhuff.c decode_mcu: every block of the MCU whole, its DC difference then its AC run/size pairs
                const int32_t pred0[4] = {pred[0], pred[1], pred[2], pred[3]};
                b.mcu_begin(bpm, sc.dri == 0);
            mcu_again:
                for (uint32_t s = 0; s < sc.ns; s++)
                    for (uint32_t v = 0; v < sc.vs[s]; v++)
                        for (uint32_t h = 0; h < sc.hs[s]; h++) {
                            const uint32_t blk = sc.cblk[s] + (my * sc.vs[s] + v) * sc.bw[s] + mx * sc.hs[s] + h;
                            const uint32_t t = b.sym(s) & 15u;
                            if (t) pred[s] += lp_prog_extend(b.get(t), t);
                            m.st(blk, 0, pred[s]);
                            for (uint32_t k = 1; k < 64u; k++) {
                                const uint32_t rs = b.sym(4u + s), r = rs >> 4, sz = rs & 15u;
                                if (sz) {
                                    k += r;
                                    m.st(blk, k < 64u ? k : 63u, lp_prog_extend(b.get(sz), sz)); // jpeg_natural_order[64..79] = 63
                                } else if (r == 15u)
                                    k += 15u;
                                else
                                    break;
                            }
                        }
                if (b.mcu_redo()) { // decode_mcu_fast met a marker and gave the MCU up: decode_mcu_slow takes it from the same state
                    pred[0] = pred0[0]; pred[1] = pred0[1]; pred[2] = pred0[2]; pred[3] = pred0[3];
                    goto mcu_again;
                }
                if (b.failed()) return false;
                if (sc.dri) rst_left--;
                continue;
            }
            if (Ss == 0) { // DC scans may interleave components
                for (uint32_t s = 0; s < sc.ns; s++)
                    for (uint32_t v = 0; v < sc.vs[s]; v++)
                        for (uint32_t h = 0; h < sc.hs[s]; h++) {
                            const uint32_t blk = sc.cblk[s] + (my * sc.vs[s] + v) * sc.bw[s] + mx * sc.hs[s] + h;
                            if (Ah == 0) { // decode_mcu_DC_first
                                const uint32_t t = b.sym(s) & 15u;
                                if (t) pred[s] += lp_prog_extend(b.get(t), t);
                                m.st(blk, 0, (int32_t)((uint32_t)pred[s] << Al));
                            } else if (b.get(1)) // decode_mcu_DC_refine
                                m.st(blk, 0, m.ld(blk, 0) | p1);
                        }
            } else {
                const uint32_t blk = sc.cblk[0] + my * sc.bw[0] + mx;
                if (Ah == 0) { // decode_mcu_AC_first
                    if (eobrun > 0) eobrun--;
                    else
                        for (uint32_t k = Ss; k <= Se; k++) {
                            const uint32_t rs = b.sym(0), r = rs >> 4, t = rs & 15u;
                            if (t) {
                                k += r;
                                const int32_t val = lp_prog_extend(b.get(t), t);
                                // an index past 63 (corrupt stream) lands on jpeg_natural_order[64..79] = the last coefficient
                                if (k > Se) lp_prog_stray(m);
                                m.st(blk, k < 64u ? k : 63u, (int32_t)((uint32_t)val << Al));
                            } else if (r == 15u)
                                k += 15u;
                            else {
                                eobrun = (1u << r) + b.get(r) - 1u;
                                break;
                            }
                        }
                } else { // decode_mcu_AC_refine
                    uint64_t nz = m.open(blk);
                    const uint64_t band = Se >= 63u ? ~0ull : (1ull << (Se + 1u)) - 1ull; // elements 0..Se
                    uint32_t k = Ss;
                    if (eobrun == 0) {
                        while (k <= Se) {
                            const uint32_t rs = b.sym(0);
                            uint32_t r = rs >> 4;
                            int32_t t = (int32_t)(rs & 15u);
                            if (t) t = b.get(1) ? p1 : m1; // a new coefficient: its size is 1 whatever the symbol says (JWRN_HUFF_BAD_CODE otherwise)
                            else if (r != 15u) {
                                eobrun = (1u << r) + b.get(r);
                                break; // the rest of this band is handled as the first block of the run
                            }
                            // pass r still-zero coefficients and stop AT the next one (or behind the band), correcting the non-zero
                            // ones on the way: the (r + 1)-th zero at or after k
                            const uint64_t from_k = ~0ull << k;
                            uint64_t zeros = ~nz & from_k & band;
                            if constexpr (LpProgHasBulk<P>::value) zeros = m.drop_lowest(zeros, r);
                            else for (; r && zeros; r--) zeros &= zeros - 1ull;
                            const uint32_t stop = zeros ? lp_ctz64(zeros) : Se + 1u;
                            lp_prog_correct(m, b, nz & from_k & (stop >= 64u ? ~0ull : (1ull << stop) - 1ull), p1, m1);
                            k = stop;
                            if (t) { // index 64 (band ends at 63 and ran out of zeros) lands on the last coefficient, like jpeg_natural_order[64]
                                const uint32_t e = k < 64u ? k : 63u;
                                if (k > Se) lp_prog_stray(m);
                                m.set(e, t);
                                nz |= 1ull << e;
                            }
                            k++;
                        }
                    }
                    if (eobrun > 0) {
                        if (k <= Se) lp_prog_correct(m, b, nz & (~0ull << k) & band, p1, m1);
                        eobrun--;
                    }
                    m.close(blk);
                }
            }
            if (b.failed()) return false;
            if (sc.dri) rst_left--;
        }
    return true;
}

template <class P>
LP_PHD void lp_prog_scan(P& m, const LpProgScan& sc, uint32_t total_bits, uint32_t n_rst)
{
    LpProgBits<P> b(m, total_bits, n_rst);
    (void)lp_prog_scan_with(m, b, sc);
}
