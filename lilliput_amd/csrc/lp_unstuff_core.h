// lp_unstuff_core.h -- byte classification of the unstuff kernels (k_unstuff_count, k_unstuff_scatter), 16 bytes per lane, as word
// arithmetic. Byte classes inside the entropy-coded segment (T.81 B.1.1.5, F.1.2.3; libjpeg jdhuff.c jpeg_fill_bit_buffer):
//   data byte (kept); FF followed by 00 = data FF (kept, the 00 dropped); FF FF .. = fill (dropped); FF Dn = RSTn (dropped, a
//   restart boundary); FF + anything else = a real marker inside the scan (error bit 1).
// The first version walked the 16 bytes one by one (about 20 VALU instructions per byte: 324 per wave in k_unstuff_count, PMC);
// here every test is done on four bytes at a time with the exact zero-byte detector
//   zero(v) = ~(((v & 0x7f7f7f7f) + 0x7f7f7f7f) | v | 0x7f7f7f7f)      (0x80 in every byte of v that is 0, no carries between bytes)
// and the neighbour relations are byte shifts across the four words. Host + device so that tests/emu can hold it against the
// byte-by-byte definition on every neighbour / length combination.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LP_UHD __host__ __device__ __forceinline__
#else
#define LP_UHD inline
#endif

LP_UHD uint32_t lp_zero_bytes(uint32_t v) { return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu); }
// bit j of the result = bit (8 * j + 7) of m, j = 0..3
LP_UHD uint32_t lp_movemask4(uint32_t m)
{
    const uint32_t x = m >> 7;
    return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 15u;
}

// w[0..3]: 16 raw bytes (byte j = bits 8 * (j & 3) of w[j >> 2]) at stream position pos0; prev / next: the bytes around them;
// raw_len: length of the segment. K / R: bit 8 * (j & 3) + 7 of word j >> 2 set = byte j is kept / is the Dn of a restart marker
// (so a count is a popcount and the test for byte j is one bit-field extract -- no 16 x movemask).
// TAIL = false: the caller knows that these 16 bytes AND the byte after them lie inside the segment (every lane of a chunk that is not
// the segment's last): the range masks fall away.
// HEAD = true (the first 16 bytes of a segment that does not start on a 16-byte boundary of the raw arena -- see LpJpeg::raw_skip): the
// first `head` bytes of this group lie BEFORE the segment; they are dropped and their values never looked at.
template <bool TAIL, bool HEAD = false>
LP_UHD void lp_unstuff_classify_masks(const uint32_t w[4], uint32_t prev, uint32_t next, uint32_t pos0, uint32_t raw_len, uint32_t K[4],
                                      uint32_t R[4], uint32_t& err, uint32_t head = 0)
{
    // bytes of this lane that lie inside the segment: the first n_in
    const uint32_t n_in = pos0 >= raw_len ? 0u : (raw_len - pos0 >= 16u ? 16u : raw_len - pos0);
    // the last byte of the segment has no successor: an FF there is not data. Index relative to this lane (wraps when it is elsewhere).
    const uint32_t last = raw_len - 1u - pos0;
    uint32_t F[4], Z[4], H[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int32_t hn = (int32_t)head - 4 * i;                      // bytes of this word before the segment
        H[i] = !HEAD || hn <= 0 ? 0u : hn >= 4 ? 0x80808080u : 0x80808080u >> (8u * (4u - (uint32_t)hn));
        F[i] = lp_zero_bytes(~w[i]) & ~H[i];                           // == FF (never for a byte before the segment)
        Z[i] = lp_zero_bytes(w[i]);                                    // == 00
    }
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t d = lp_zero_bytes((w[i] ^ 0xd0d0d0d0u) & 0xf8f8f8f8u); // D0..D7
        const uint32_t f_prev = i == 0 ? ((prev & 0xffu) == 0xffu ? 0x80u : 0u) : F[i - 1] >> 24;
        const uint32_t z_next = i == 3 ? ((next & 0xffu) == 0u ? 0x80000000u : 0u) : Z[i + 1] << 24;
        const uint32_t Fp = (F[i] << 8) | f_prev;                      // the byte before is FF
        const uint32_t Zn = (Z[i] >> 8) | z_next;                      // the byte after is 00
        const uint32_t after_ff = ~F[i] & Fp;                          // a non-FF byte that follows an FF: 00 (stuffing), Dn, or a marker
        const int32_t n = (int32_t)n_in - 4 * i;                       // bytes of this word inside the segment
        const uint32_t in = (!TAIL || n >= 4 ? 0x80808080u : n <= 0 ? 0u : 0x80808080u >> (8u * (4u - (uint32_t)n))) & ~H[i];
        const uint32_t last_ff = (TAIL && raw_len != 0u && (last >> 2) == (uint32_t)i) ? (0x80u << (8u * (last & 3u))) & F[i] : 0u;
        K[i] = ((F[i] & Zn & ~last_ff) | (~F[i] & ~Fp & 0x80808080u)) & in;
        R[i] = after_ff & d & in;
        bad |= after_ff & ~Z[i] & ~d & in;
    }
    if (bad) err |= 1u;
}

// The same as 16-bit masks (bit j = byte j); used by tests/emu.
LP_UHD void lp_unstuff_classify(const uint32_t w[4], uint32_t prev, uint32_t next, uint32_t pos0, uint32_t raw_len, uint32_t& keep_mask,
                                uint32_t& rst_mask, uint32_t& err)
{
    uint32_t K[4], R[4];
    lp_unstuff_classify_masks<true>(w, prev, next, pos0, raw_len, K, R, err);
    keep_mask = 0; rst_mask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        keep_mask |= lp_movemask4(K[i]) << (4 * i);
        rst_mask |= lp_movemask4(R[i]) << (4 * i);
    }
}
