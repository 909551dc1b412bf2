// lp_inflate.cpp -- a one-shot inflater for the ordinary case of a PNG's image data: the whole zlib stream in one buffer, the whole
// filtered image as the output, and nothing unusual about either.
//
// Why: in a mixed stream of sources the PNG items are bound by the host's inflate (lp_png.cpp: 41 of the 46 ms of a 2048 x 2048 RGB
// file), and the system zlib decodes literal-heavy photographic data at ~110 MB/s. This decoder keeps 56+ bits in a 64-bit buffer,
// resolves a code in one table load (11 bits for literals / lengths, 8 for distances, second-level tables behind them) and stays in
// its loop from the first block to the last.
//
// What it is NOT: an arbiter. It answers 1 only for a stream that zlib's inflate (as libpng 1.6 drives it: window size from the stream
// header, no preset dictionary) accepts too, decodes to exactly `out_len` bytes, ends at the last input byte and carries the right
// Adler-32 -- for such a stream the output of any conforming inflater is the same bytes. Everything else -- a code set that is not
// complete (zlib tolerates some), a distance beyond the declared window or the output so far, a reserved symbol, trailing or missing
// bytes -- answers 0 and the caller runs zlib, whose verdict and partial output are then what counts (lp_png.cpp, the reference's
// png_read_IDAT_data pattern). tests/test_png.py compares both routes on every fixture and on mutated streams.
#include "lp_inflate.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h> // adler32()

#include <immintrin.h>

// Checksums of the PNG path at memory speed: the system zlib's adler32 / crc32 run at 1.7 / 1.1 GB/s and were a fifth of a PNG's host
// time (12.6 MB of filtered rows, 8 MB of IDAT bytes for a 2048 x 2048 photograph). Both fall back to zlib when the CPU lacks the
// instructions; tests/test_inflate.py checks them against zlib on many lengths and alignments.
__attribute__((target("avx2"))) static inline uint32_t hsum8x32(__m256i v) // the sum of eight 32-bit lanes
{
    __m128i x = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    x = _mm_add_epi32(x, _mm_shuffle_epi32(x, 0x4e));
    x = _mm_add_epi32(x, _mm_shuffle_epi32(x, 0xb1));
    return (uint32_t)_mm_cvtsi128_si32(x);
}
__attribute__((target("avx2"))) static uint32_t adler32_avx2(uint32_t adler, const uint8_t* p, size_t n)
{
    uint32_t s1 = adler & 0xffffu, s2 = adler >> 16;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m256i ones16 = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
    while (n >= 32) {
        // a run of B blocks of 32 bytes; within it s1 and s2 are kept relative to the run's start and reduced once at its end:
        //   s2 += 32 * (s1 before the block) + sum (32 - i) * b[i];  s1 += sum b[i]
        const size_t B = (n < 5536 ? n : 5536) / 32;
        __m256i vs1 = zero, vs2 = zero, vs1_before = zero;
        for (size_t k = 0; k < B; k++, p += 32) {
            const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
            vs1_before = _mm256_add_epi32(vs1_before, vs1);
            vs1 = _mm256_add_epi32(vs1, _mm256_sad_epu8(b, zero));
            vs2 = _mm256_add_epi32(vs2, _mm256_madd_epi16(_mm256_maddubs_epi16(b, weights), ones16));
        }
        n -= B * 32;
        const uint64_t t2 = (uint64_t)s2 + (uint64_t)32 * B * s1 + (uint64_t)32 * hsum8x32(vs1_before) + hsum8x32(vs2);
        s1 = (uint32_t)(((uint64_t)s1 + hsum8x32(vs1)) % 65521u);
        s2 = (uint32_t)(t2 % 65521u);
    }
    for (; n; n--) { s1 += *p++; s2 += s1; } // fewer than 32 bytes: no overflow
    return ((s2 % 65521u) << 16) | (s1 % 65521u);
}

uint32_t lp_adler32(uint32_t adler, const uint8_t* p, size_t n)
{
    static const bool fast = __builtin_cpu_supports("avx2");
    if (fast && n >= 64) return adler32_avx2(adler, p, n);
    uLong a = adler;
    while (n) { const size_t c = n < (1u << 30) ? n : (1u << 30); a = adler32(a, p, (uInt)c); p += c; n -= c; }
    return (uint32_t)a;
}

// CRC-32 (IEEE 802.3, reflected) by carry-less multiplication: four 128-bit lanes folded 64 bytes at a time, then to one lane, to 64
// bits and Barrett-reduced (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009;
// the constants are x^(512+64), x^512, x^(128+64), x^128, x^64 mod P, P and floor(x^64 / P), bit-reflected).
// n >= 64 and a multiple of 16; crc is the raw register (already inverted by the caller).
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_fold(__m128i x, __m128i k, __m128i next)
{
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
}
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_ld(const uint8_t* q) { return _mm_loadu_si128(reinterpret_cast<const __m128i*>(q)); }
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc, const uint8_t* p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_xor_si128(crc_ld(p), _mm_cvtsi32_si128((int)crc)), x2 = crc_ld(p + 16), x3 = crc_ld(p + 32), x4 = crc_ld(p + 48);
    p += 64; n -= 64;
    for (; n >= 64; p += 64, n -= 64) {
        x1 = crc_fold(x1, k1k2, crc_ld(p)); x2 = crc_fold(x2, k1k2, crc_ld(p + 16)); x3 = crc_fold(x3, k1k2, crc_ld(p + 32)); x4 = crc_fold(x4, k1k2, crc_ld(p + 48));
    }
    x1 = crc_fold(x1, k3k4, x2);
    x1 = crc_fold(x1, k3k4, x3);
    x1 = crc_fold(x1, k3k4, x4);
    for (; n >= 16; p += 16, n -= 16) x1 = crc_fold(x1, k3k4, crc_ld(p));
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(-1, 0, -1, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, mask32), poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
}

uint32_t lp_crc32(uint32_t crc, const uint8_t* p, size_t n)
{
    static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (fast && n >= 64) {
        const size_t chunk = n & ~(size_t)15;
        crc = ~crc32_clmul(~crc, p, chunk);
        p += chunk; n -= chunk;
    }
    uLong c = crc;
    while (n) { const size_t k = n < (1u << 30) ? n : (1u << 30); c = crc32(c, p, (uInt)k); p += k; n -= k; }
    return (uint32_t)c;
}

namespace {

// kinds: bit 2 (bit 15 of the entry, E_NOT_LITERAL) is clear for the two literal kinds and set for everything else, so the test the loop
// makes after every lookup is one bit -- and not "one literal or two?", which is a coin toss on photographic data and cost a branch
// miss on a third of the lookups when it was a branch; the count of literals is bit 13
enum : uint32_t { K_LIT = 0, K_LIT2 = 1, K_BASE = 4, K_EOB = 5, K_SUB = 6, K_BAD = 7 };
constexpr uint32_t E_NOT_LITERAL = 1u << 15;
// entry: bits 0..7 code length (K_SUB: bits of the second-level index), 8..12 extra bits, 13..15 kind, 16..31 value
inline uint32_t mk(uint32_t len, uint32_t extra, uint32_t kind, uint32_t value) { return len | (extra << 8) | (kind << 13) | (value << 16); }
inline uint32_t e_len(uint32_t e) { return e & 255u; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7u; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr int LL_CAP = (1 << LL_BITS) + 1024, D_CAP = (1 << D_BITS) + 512; // zlib's ENOUGH figures bound the second level: 852 / 592 for 9 / 6 root bits

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t rev_bits(uint32_t c, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

// Canonical code -> lookup table. kind_of(sym) yields the finished entry without its length. Only a COMPLETE code set is accepted
// (Kraft sum exactly one): incomplete sets that zlib lets through (a lone distance code) go to zlib.
template <class F>
bool build_table(const uint8_t* lens, int n, int root, uint32_t* table, int cap, F entry_of)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    uint32_t kraft = 0;
    for (int l = 1; l <= 15; l++) kraft += (uint32_t)count[l] << (15 - l);
    if (kraft != (1u << 15)) return false;
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    next[0] = 0;
    const int nroot = 1 << root;
    // second-level tables: one per root prefix that long codes share, as wide as the longest of them
    static thread_local uint8_t sub_bits[1 << LL_BITS];
    memset(sub_bits, 0, (size_t)nroot);
    uint32_t codes[320];
    {
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int i = 0; i < n; i++) {
            const int l = lens[i];
            if (!l) continue;
            const uint32_t r = rev_bits(nx[l]++, l);
            codes[i] = r;
            if (l > root) {
                uint8_t& sb = sub_bits[r & (uint32_t)(nroot - 1)];
                if (l - root > sb) sb = (uint8_t)(l - root);
            }
        }
    }
    int used = nroot;
    for (int p = 0; p < nroot; p++) {
        if (!sub_bits[p]) continue;
        if (used + (1 << sub_bits[p]) > cap) return false;
        table[p] = mk(sub_bits[p], 0, K_SUB, (uint32_t)used);
        used += 1 << sub_bits[p];
    }
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) continue;
        const uint32_t r = codes[i], e = entry_of(i) | (uint32_t)l;
        if (l <= root) {
            for (uint32_t k = r; k < (uint32_t)nroot; k += 1u << l) table[k] = e;
        } else {
            const uint32_t p = r & (uint32_t)(nroot - 1), sb = sub_bits[p], base = e_value(table[p]);
            for (uint32_t k = r >> root; k < (1u << sb); k += 1u << (l - root)) table[base + k] = e;
        }
    }
    return true; // complete code: every entry has been written
}

// Two literals in one lookup: where a literal's code leaves room in the root index for the whole code of the literal that follows,
// the entry carries both (value = first | second << 8, length = both codes). The symbol-to-symbol chain -- index, load, shift -- is
// what bounds a table-driven inflater on literal-heavy data (photographs: five bits per byte), and this halves its length there.
void pair_literals(uint32_t* table)
{
    static thread_local uint32_t base[1 << LL_BITS];
    memcpy(base, table, sizeof(base));
    for (uint32_t i = 0; i < (1u << LL_BITS); i++) {
        const uint32_t e1 = base[i];
        if (e_kind(e1) != K_LIT || e_len(e1) >= (uint32_t)LL_BITS) continue;
        const uint32_t room = (uint32_t)LL_BITS - e_len(e1), e2 = base[i >> e_len(e1)]; // the index bits behind the first code, zeros above them
        if (e_kind(e2) != K_LIT || e_len(e2) > room) continue; // (an entry whose code fits the known bits does not depend on the unknown ones)
        table[i] = mk(e_len(e1) + e_len(e2), 0, K_LIT2, e_value(e1) | (e_value(e2) << 8));
    }
}

inline uint32_t ll_entry(int sym)
{
    if (sym < 256) return mk(0, 0, K_LIT, (uint32_t)sym);
    if (sym == 256) return mk(0, 0, K_EOB, 0);
    if (sym > 285) return mk(0, 0, K_BAD, 0);
    return mk(0, kLenExtra[sym - 257], K_BASE, kLenBase[sym - 257]);
}
inline uint32_t d_entry(int sym)
{
    if (sym > 29) return mk(0, 0, K_BAD, 0);
    return mk(0, kDistExtra[sym], K_BASE, kDistBase[sym]);
}

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; } // little-endian host (x86-64)

} // namespace

int lp_inflate_exact(const uint8_t* in0, size_t in_len, uint8_t* out0, size_t out_len)
{
    // in0 must be readable up to in_len + LP_INFLATE_PAD bytes (zeros behind the stream); out0 up to out_len bytes exactly
    if (in_len < 2 + 4) return 0;
    const uint32_t cmf = in0[0], flg = in0[1];
    if ((cmf & 15u) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31u != 0 || (flg & 0x20u)) return 0;
    const size_t window = (size_t)1 << ((cmf >> 4) + 8);
    const uint8_t* in = in0 + 2;
    const uint8_t* const in_stop = in0 + in_len + 8; // the refill may run a few bytes ahead of what has been consumed; beyond this the stream is too short
    uint8_t* out = out0;
    uint8_t* const out_end = out0 + out_len;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    static thread_local uint32_t ll[LL_CAP], dt[D_CAP];
    static thread_local bool fixed_ready = false;
    static thread_local uint32_t fll[LL_CAP], fdt[D_CAP];
#define REFILL() do { bitbuf |= load64(in) << bitcnt; in += (63u - bitcnt) >> 3; bitcnt |= 56u; } while (0)
#define DROP(n) do { bitbuf >>= (n); bitcnt -= (n); } while (0)
    for (;;) {
        if (in > in_stop) return 0;
        REFILL();
        const uint32_t last = (uint32_t)bitbuf & 1u, type = ((uint32_t)bitbuf >> 1) & 3u;
        DROP(3);
        const uint32_t* LL;
        const uint32_t* DT;
        if (type == 0) { // stored: to the next byte boundary, LEN / NLEN, the bytes
            DROP(bitcnt & 7u);
            // the buffer holds whole bytes now: hand them back
            in -= bitcnt >> 3;
            bitbuf = 0; bitcnt = 0;
            if ((size_t)(in0 + in_len - in) < 4 || in > in0 + in_len) return 0;
            const uint32_t len = in[0] | ((uint32_t)in[1] << 8), nlen = in[2] | ((uint32_t)in[3] << 8);
            if ((len ^ nlen) != 0xffffu) return 0;
            in += 4;
            if ((size_t)(in0 + in_len - in) < len || (size_t)(out_end - out) < len) return 0;
            memcpy(out, in, len);
            in += len; out += len;
            if (last) break;
            continue;
        } else if (type == 1) {
            if (!fixed_ready) {
                uint8_t l[288];
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                uint8_t d[32];
                for (int i = 0; i < 32; i++) d[i] = 5;
                if (!build_table(l, 288, LL_BITS, fll, LL_CAP, ll_entry) || !build_table(d, 32, D_BITS, fdt, D_CAP, d_entry)) return 0;
                pair_literals(fll);
                fixed_ready = true;
            }
            LL = fll; DT = fdt;
        } else if (type == 2) {
            const uint32_t hlit = ((uint32_t)bitbuf & 31u) + 257, hdist = (((uint32_t)bitbuf >> 5) & 31u) + 1, hclen = (((uint32_t)bitbuf >> 10) & 15u) + 4;
            DROP(14);
            if (hlit > 286 || hdist > 30) return 0; // zlib: "too many length or distance symbols"
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            REFILL();
            for (uint32_t i = 0; i < hclen; i++) {
                if (bitcnt < 3) REFILL();
                cl[order[i]] = (uint8_t)(bitbuf & 7u);
                DROP(3);
            }
            uint32_t clt[1 << 7];
            if (!build_table(cl, 19, 7, clt, 1 << 7, [](int s) { return mk(0, 0, K_LIT, (uint32_t)s); })) return 0;
            uint8_t lens[320];
            uint32_t i = 0;
            while (i < hlit + hdist) {
                if (in > in_stop) return 0;
                REFILL();
                const uint32_t e = clt[bitbuf & 127u];
                DROP(e_len(e));
                const uint32_t s = e_value(e);
                if (s < 16) { lens[i++] = (uint8_t)s; continue; }
                uint32_t rep, val = 0;
                if (s == 16) { if (!i) return 0; val = lens[i - 1]; rep = 3 + ((uint32_t)bitbuf & 3u); DROP(2); }
                else if (s == 17) { rep = 3 + ((uint32_t)bitbuf & 7u); DROP(3); }
                else { rep = 11 + ((uint32_t)bitbuf & 127u); DROP(7); }
                if (i + rep > hlit + hdist) return 0; // "invalid bit length repeat"
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) return 0; // "invalid code -- missing end-of-block"
            if (!build_table(lens, (int)hlit, LL_BITS, ll, LL_CAP, ll_entry) || !build_table(lens + hlit, (int)hdist, D_BITS, dt, D_CAP, d_entry)) return 0;
            pair_literals(ll);
            LL = ll; DT = dt;
        } else
            return 0;
        // the block's symbols
        for (;;) {
            if (in > in_stop) return 0;
            REFILL();
            uint32_t e = LL[bitbuf & ((1u << LL_BITS) - 1u)];
            // up to three lookups on one refill (each at most LL_BITS bits: longer codes leave through the second-level branch below),
            // each one or two literals
            if (!(e & E_NOT_LITERAL)) {
                if (out_end - out < 6) { // close to the end of the image: one lookup at a time, with the room checked
                    const size_t nlit = 1 + ((e >> 13) & 1u);
                    if ((size_t)(out_end - out) < nlit) return 0;
                    DROP(e_len(e));
                    *out++ = (uint8_t)e_value(e);
                    if (nlit == 2) *out++ = (uint8_t)(e_value(e) >> 8);
                    continue;
                }
#define PUT_LITERALS() do { const uint16_t v = (uint16_t)e_value(e); memcpy(out, &v, 2); out += 1 + ((e >> 13) & 1u); DROP(e_len(e)); } while (0)
#define NEXT_ENTRY() do { e = LL[bitbuf & ((1u << LL_BITS) - 1u)]; } while (0)
                PUT_LITERALS();
                NEXT_ENTRY();
                if (!(e & E_NOT_LITERAL)) {
                    PUT_LITERALS();
                    NEXT_ENTRY();
                    if (!(e & E_NOT_LITERAL)) {
                        PUT_LITERALS();
                        continue;
                    }
                }
#undef PUT_LITERALS
#undef NEXT_ENTRY
                // a length, a long code or the end of the block behind the literals: at most 33 bits are gone; top the buffer up
                REFILL();
            }
            if (e_kind(e) == K_SUB) {
                e = LL[e_value(e) + ((bitbuf >> LL_BITS) & ((1u << e_len(e)) - 1u))];
                if (!(e & E_NOT_LITERAL)) { // a literal with a code longer than LL_BITS
                    if (out == out_end) return 0;
                    DROP(e_len(e));
                    *out++ = (uint8_t)e_value(e);
                    continue;
                }
            }
            const uint32_t kind = e_kind(e);
            if (kind == K_EOB) { DROP(e_len(e)); break; }
            if (kind != K_BASE) return 0; // reserved length symbol
            DROP(e_len(e));
            const uint32_t xl = e_extra(e);
            const uint32_t len = e_value(e) + ((uint32_t)bitbuf & ((1u << xl) - 1u));
            DROP(xl);
            // at most 15 + 5 bits went since the refill: 36 left, enough for the distance code (15) and its extra bits (13)
            uint32_t d = DT[bitbuf & ((1u << D_BITS) - 1u)];
            if (e_kind(d) == K_SUB) d = DT[e_value(d) + ((bitbuf >> D_BITS) & ((1u << e_len(d)) - 1u))];
            if (e_kind(d) != K_BASE) return 0; // reserved distance symbol
            DROP(e_len(d));
            const uint32_t xd = e_extra(d);
            const size_t dist = e_value(d) + ((uint32_t)bitbuf & ((1u << xd) - 1u));
            DROP(xd);
            if (dist > (size_t)(out - out0) || dist > window) return 0; // "invalid distance too far back"
            if ((size_t)(out_end - out) < len) return 0;
            const uint8_t* src = out - dist;
            if (dist >= 8 && (size_t)(out_end - out) >= (size_t)len + 8) {
                uint8_t* o = out;
                const uint8_t* const oe = out + len;
                do { memcpy(o, src, 8); o += 8; src += 8; } while (o < oe);
                out = const_cast<uint8_t*>(oe);
            } else if (dist == 1) {
                memset(out, *src, len);
                out += len;
            } else {
                for (uint32_t k = 0; k < len; k++) out[k] = src[k];
                out += len;
            }
        }
        if (last) break;
    }
#undef REFILL
#undef DROP
    // behind the last block: to the byte boundary, then the Adler-32 of the output, and nothing else
    const uint8_t* p = in - (bitcnt >> 3); // whole bytes still in the buffer have not been consumed
    if (p + 4 != in0 + in_len) return 0;
    if (out != out_end) return 0;
    const uint32_t want = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    return lp_adler32(1u, out0, out_len) == want ? 1 : 0;
}
