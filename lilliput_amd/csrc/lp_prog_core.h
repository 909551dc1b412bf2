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

// zigzag index -> storage index inside a block (transposed natural order); entries past 63 are never used
#define LP_TZIGZAG_INIT                                                                                   \
    {0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,  13, 20, 27, 34, \
     41, 48, 56, 49, 42, 35, 28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, \
     52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63}

// Memory policy P must provide:
//   uint32_t word(uint32_t w)                      word w of the clean stream, bit 31 first (any w the reader asks for below the capacity)
//   uint32_t rst_bit(uint32_t k)                   bit position of the k-th restart boundary
//   uint32_t lut8(uint32_t s, uint32_t i); int32_t maxcode(s, l), valoff(s, l); uint32_t val(s, i)    LpProgHuff of the scan
//   uint32_t tz(uint32_t k)                        LP_TZIGZAG_INIT
//   void st(uint32_t blk, uint32_t e, int32_t v)   store one coefficient (first scans: write only)
//   int32_t ld(uint32_t blk, uint32_t e)
//   void open(uint32_t blk) / int32_t get(uint32_t e) / void set(uint32_t e, int32_t v) / void close(uint32_t blk)
//                                                  stage one block for an AC refinement (every coefficient of the band is read)
template <class P>
struct LpProgBits {
    P& m;
    uint32_t p;         // next unread bit
    uint32_t total;     // bits in the stream; everything behind reads as zero
    uint32_t cw, w0, w1;

    LP_PHD LpProgBits(P& m_, uint32_t total_) : m(m_), p(0), total(total_), cw(0xfffffff0u), w0(0), w1(0) {}
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
    LP_PHD uint32_t get(uint32_t n) // n <= 16
    {
        if (!n) return 0u;
        const uint32_t v = peek() >> (32u - n);
        p += n;
        return v;
    }
    LP_PHD uint32_t sym(uint32_t s) // jdhuff.c HUFF_DECODE / jpeg_huff_decode
    {
        const uint32_t pk = peek();
        uint32_t e = m.lut8(s, pk >> 24);
        if (e) { p += e >> 8; return e & 255u; }
        for (uint32_t l = 9; l <= 16; l++) {
            const int32_t code = (int32_t)(pk >> (32u - l));
            if (code <= m.maxcode(s, l)) { p += l; return m.val(s, (uint32_t)(code + m.valoff(s, l)) & 255u); }
        }
        p += 16; // JWRN_HUFF_BAD_CODE: libjpeg carries on with a zero symbol
        return 0u;
    }
};

LP_PHD int32_t lp_prog_extend(uint32_t v, uint32_t s) // HUFF_EXTEND
{
    return v < (1u << (s - 1u)) ? (int32_t)v - (int32_t)((1u << s) - 1u) : (int32_t)v;
}

template <class P>
LP_PHD void lp_prog_scan(P& m, const LpProgScan& sc, uint32_t total_bits, uint32_t n_rst)
{
    LpProgBits<P> b(m, total_bits);
    const uint32_t Ss = sc.Ss, Se = sc.Se, Ah = sc.Ah, Al = sc.Al;
    const int32_t p1 = 1 << Al, m1 = -(1 << Al);
    int32_t pred[4] = {0, 0, 0, 0};
    uint32_t eobrun = 0, rst_left = sc.dri, rst_k = 0;
    for (uint32_t my = 0; my < sc.mcuy; my++)
        for (uint32_t mx = 0; mx < sc.mcux; mx++) {
            if (sc.dri && rst_left == 0) { // process_restart: the rest of the interval's bits are dropped, predictors and the EOB run start over
                b.p = rst_k < n_rst ? m.rst_bit(rst_k) : total_bits;
                rst_k++;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                eobrun = 0;
                rst_left = sc.dri;
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
                                if (k < 64u) m.st(blk, m.tz(k), (int32_t)((uint32_t)val << Al));
                            } else if (r == 15u)
                                k += 15u;
                            else {
                                eobrun = (1u << r) + b.get(r) - 1u;
                                break;
                            }
                        }
                } else { // decode_mcu_AC_refine
                    m.open(blk);
                    uint32_t k = Ss;
                    if (eobrun == 0) {
                        for (; k <= Se; k++) {
                            const uint32_t rs = b.sym(0);
                            int32_t r = (int32_t)(rs >> 4), t = (int32_t)(rs & 15u);
                            if (t) t = b.get(1) ? p1 : m1; // a new coefficient: its size is 1 whatever the symbol says (JWRN_HUFF_BAD_CODE otherwise)
                            else if (r != 15) {
                                eobrun = (1u << r) + b.get((uint32_t)r);
                                break; // the rest of this band is handled as the first block of the run
                            }
                            do { // pass r still-zero coefficients, correcting the non-zero ones on the way
                                const uint32_t e = m.tz(k);
                                const int32_t co = m.get(e);
                                if (co != 0) {
                                    if (b.get(1) && (co & p1) == 0) m.set(e, co >= 0 ? co + p1 : co + m1);
                                } else if (--r < 0)
                                    break;
                                k++;
                            } while (k <= Se);
                            if (t && k < 64u) m.set(m.tz(k), t);
                        }
                    }
                    if (eobrun > 0) {
                        for (; k <= Se; k++) {
                            const uint32_t e = m.tz(k);
                            const int32_t co = m.get(e);
                            if (co != 0 && b.get(1) && (co & p1) == 0) m.set(e, co >= 0 ? co + p1 : co + m1);
                        }
                        eobrun--;
                    }
                    m.close(blk);
                }
            }
            if (sc.dri) rst_left--;
        }
}
