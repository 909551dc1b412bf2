// lp_jbits.h -- how the reference's JPEG decoder reads the bytes of an entropy-coded scan, restated (host only).
//
// Behind opencv_decoder_read_data (/root/reference/opencv.cpp:166-171) sits cv::JpegDecoder::readData over libjpeg-turbo 3.1.0, fed
// by OpenCV's own memory source manager. Three layers decide what a damaged or short stream decodes to, and all three are restated here:
//   * jdmarker.c next_marker / read_restart_marker / jpeg_resync_to_restart: which bytes count as a marker, what a restart marker with
//     the wrong number, a missing one or a byte pair that only looks like a marker does to the following intervals;
//   * jdhuff.c jpeg_fill_bit_buffer / HUFF_DECODE / jpeg_huff_decode (every Huffman scan, sequential and progressive) and
//     decode_mcu_fast's GET_BYTE / FILL_BIT_BUFFER_FAST (sequential scans without a restart interval, while more than 512 bytes per
//     block of the MCU are left): the 64-bit holding register, when it is refilled and how far ahead of the decoder it reads;
//   * the source manager of cv::JpegDecoder (grfmt_jpeg.cpp in the reference's libopencv_imgcodecs.a; read from its disassembly:
//     fill_input_buffer is `return FALSE`, there is no fake EOI): a refill that finds the buffer empty SUSPENDS the decoder, readData's
//     jpeg_read_scanlines then returns 0 rows and the image FAILS -- where jpeg_mem_src (what rounds 1-4 pinned against) would have
//     warned and painted the rest grey. Because libjpeg refills its register up to eight bytes ahead of the bits it needs, whether a
//     stream that stops without a marker still decodes depends on where exactly the refills fell: hence the register is modelled bit
//     for bit. A stream that stops AT a marker (any marker: a file cut short and closed with EOI) never suspends: zero bits, a warning.
// Pinned by tests/test_damaged.py against oracle/_ref/librefjpegcv.so: the reference's own cv::JpegDecoder object code linked with the
// reference's own libjpeg.a (verdicts and pixels of several thousand cut, padded and damaged streams).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "lp_types.h"

// The byte source with its marker state: jdmarker.c over cv::JpegDecoder's source manager.
struct LpJSrc {
    const uint8_t* p;       // next_input_byte
    const uint8_t* end;     // end of the FILE: bytes_in_buffer = end - p
    int marker;             // cinfo->unread_marker
    bool suspended;         // a read found the buffer empty: fill_input_buffer answered FALSE (the Huffman decoder suspends and readData
                            // gives up; the arithmetic decoder raises JERR_CANT_SUSPEND) -- either way the image fails

    void init(const uint8_t* at, const uint8_t* file_end) { p = at; end = file_end; marker = 0; suspended = false; }

    // jdmarker.c next_marker: on to the next FF xx with xx neither 00 (a stuffed data byte) nor FF (fill); false = ran out of bytes
    bool next_marker()
    {
        for (;;) {
            int c;
            do {
                if (p == end) { suspended = true; return false; }
                c = *p++;
            } while (c != 0xFF);
            do {
                if (p == end) { suspended = true; return false; }
                c = *p++;
            } while (c == 0xFF);
            if (c != 0) { marker = c; return true; }
        }
    }
    // jdmarker.c read_restart_marker + jpeg_resync_to_restart for the restart number `num`; false = ran out of bytes.
    // On return marker == 0 means decoding resumes behind a restart marker; a marker left pending makes the interval an empty segment.
    bool restart_marker(uint32_t num)
    {
        if (marker == 0 && !next_marker()) return false;
        if (marker == 0xD0 + (int)num) { marker = 0; return true; } // the expected one: swallowed
        for (;;) {
            const int m = marker;
            int action;
            if (m < 0xC0) action = 2;                                                                               // not a marker libjpeg knows: look further
            else if (m < 0xD0 || m > 0xD7) action = 3;                                                              // a real marker: leave it
            else if (m == 0xD0 + (int)((num + 1u) & 7u) || m == 0xD0 + (int)((num + 2u) & 7u)) action = 3;          // one of the next two: this interval is missing
            else if (m == 0xD0 + (int)((num - 1u) & 7u) || m == 0xD0 + (int)((num - 2u) & 7u)) action = 2;          // one of the last two: skip it
            else action = 1;                                                                                        // too far away to tell: take it
            if (action == 1) { marker = 0; return true; }
            if (action == 3) return true;
            if (!next_marker()) return false;
        }
    }
    // What jdmarker.c read_markers meets right behind the scan, up to the marker that ends the scan for the header walk (lp_jpeg_parse.cpp:
    // the first marker from 0xC0 up that is no RSTn -- or any marker at all when the scan has no restart interval). Only files that
    // libjpeg reads to the end before it returns pixels care (several scans; jdapimin.c jpeg_start_decompress): RSTn and TEM are
    // parameterless and skipped, any other code below 0xC0 is JERR_UNKNOWN_MARKER, the end of the buffer is a suspension that
    // cv::JpegDecoder turns into a failure. Returns 0, LP_SCAN_BAD_MARKER or LP_SCAN_OUT_OF_DATA.
    int after_scan(bool has_dri);
};

enum { LP_SCAN_OK = 0, LP_SCAN_WARNED = 1, LP_SCAN_BAD_MARKER = 2, LP_SCAN_OUT_OF_DATA = 3 };

inline int LpJSrc::after_scan(bool has_dri)
{
    for (;;) {
        if (marker == 0 && !next_marker()) return LP_SCAN_OUT_OF_DATA;
        const int m = marker;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { marker = 0; continue; }
        if (m < 0xC0) return has_dri ? LP_SCAN_BAD_MARKER : LP_SCAN_OK; // without a restart interval the header walk meets (and judges) it itself
        return LP_SCAN_OK;
    }
}

// The Huffman bit reader (what lp_prog_core.h's scan loop asks of a reader: get / sym / insufficient / restart / mcu_begin / mcu_redo / failed).
struct LpJBits {
    LpJSrc src;
    const LpProgHuff* ht;
    uint64_t buf;           // get_buffer: the unread bits are the low `bits` ones
    int bits;               // bits_left
    bool insufficient;      // jdhuff.c insufficient_data: a read went past the data (JWRN_HIT_MARKER); zero bits from there on
    bool fast;              // the MCU at hand is being decoded with decode_mcu_fast's refill pattern
    bool fast_hit;          // ... and met a marker: decode_mcu_fast gives the MCU up, decode_mcu_slow decodes it again
    uint32_t next_num;      // marker->next_restart_num
    struct Snap { const uint8_t* p; uint64_t buf; int bits; } snap;

    LpJBits(const uint8_t* at, const uint8_t* file_end, const LpProgHuff* tables) : ht(tables), buf(0), bits(0), insufficient(false), fast(false), fast_hit(false), next_num(0)
    {
        src.init(at, file_end);
        snap.p = at; snap.buf = 0; snap.bits = 0;
    }
    bool failed() const { return src.suspended; }

    // jdhuff.c jpeg_fill_bit_buffer: at least MIN_GET_BITS = 57 bits unless a marker is in the way
    bool fill(int nbits)
    {
        if (src.marker == 0) {
            while (bits < 57) {
                if (src.p == src.end) { src.suspended = true; return false; }
                int c = *src.p++;
                if (c == 0xFF) {
                    do {
                        if (src.p == src.end) { src.suspended = true; return false; }
                        c = *src.p++;
                    } while (c == 0xFF);
                    if (c == 0) c = 0xFF;
                    else { src.marker = c; goto no_more_bytes; }
                }
                buf = (buf << 8) | (uint64_t)c;
                bits += 8;
            }
            return true;
        }
    no_more_bytes:
        if (nbits > bits) {
            insufficient = true;
            buf <<= 57 - bits;
            bits = 57;
        }
        return true;
    }
    // decode_mcu_fast FILL_BIT_BUFFER_FAST: six GET_BYTEs once 16 bits or fewer are left
    void load6()
    {
        for (int i = 0; i < 6; i++) {
            if (src.p + 1 >= src.end) { fast_hit = true; buf <<= 8; bits += 8; continue; } // (cannot happen: the fast path needs 512 bytes per block in the buffer)
            const int c0 = src.p[0], c1 = src.p[1];
            src.p++;
            buf = (buf << 8) | (uint64_t)c0;
            bits += 8;
            if (c0 == 0xFF) {
                src.p++;
                if (c1 != 0) { fast_hit = true; src.p -= 2; buf &= ~(uint64_t)0xFF; }
            }
        }
    }
    uint32_t take(int n) { bits -= n; return (uint32_t)(buf >> bits) & ((1u << n) - 1u); }

    uint32_t get(uint32_t n) // CHECK_BIT_BUFFER + GET_BITS, n <= 16 (a progressive EOB run or a band of correction bits is read in pieces by the caller)
    {
        if (!n) return 0u;
        if (fast) { if (bits <= 16) load6(); }
        else if (bits < (int)n && !fill((int)n)) return 0u;
        return take((int)n);
    }
    uint32_t get_each(uint32_t n) // n x (CHECK_BIT_BUFFER(1) + GET_BITS(1)): how jdphuff.c reads correction bits; first bit on top
    {
        if (bits >= (int)n) { // the register holds them all: the n single reads are one shift (no refill falls between them either way)
            bits -= (int)n;
            return (uint32_t)(buf >> bits) & (n >= 32u ? 0xffffffffu : (1u << n) - 1u);
        }
        uint32_t v = 0;
        for (; n; n--) {
            if (bits < 1 && !fill(1)) return 0u;
            v = (v << 1) | take(1);
        }
        return v;
    }
    uint32_t sym(uint32_t s) // HUFF_DECODE / HUFF_DECODE_FAST
    {
        int nb;
        if (fast) {
            if (bits <= 16) load6();
            const uint32_t e = ht->lut8[s][(buf >> (bits - 8)) & 255u];
            if (e) { bits -= (int)(e >> 8); return e & 255u; }
            nb = 9;
            int32_t code = (int32_t)take(9);
            while (code > ht->maxcode[s][nb]) { code = (code << 1) | (int32_t)take(1); nb++; }
            return nb > 16 ? 0u : ht->vals[s][(uint32_t)(code + ht->valoff[s][nb]) & 255u];
        }
        if (bits < 8) {
            if (!fill(0)) return 0u;
            if (bits < 8) { nb = 1; goto slow; }
        }
        {
            const uint32_t e = ht->lut8[s][(buf >> (bits - 8)) & 255u];
            if (e) { bits -= (int)(e >> 8); return e & 255u; }
        }
        nb = 9;
    slow: // jpeg_huff_decode(min_bits = nb)
        if (bits < nb && !fill(nb)) return 0u;
        {
            int32_t code = (int32_t)take(nb);
            while (code > ht->maxcode[s][nb]) {
                if (bits < 1 && !fill(1)) return 0u;
                code = (code << 1) | (int32_t)take(1);
                nb++;
            }
            return nb > 16 ? 0u : ht->vals[s][(uint32_t)(code + ht->valoff[s][nb]) & 255u]; // JWRN_HUFF_BAD_CODE: a zero, 17 bits gone
        }
    }
    // jdhuff.c / jdphuff.c process_restart; false = the decoder ran out of bytes looking for the marker
    bool restart()
    {
        bits = 0; // the rest of the holding register is thrown away
        if (!src.restart_marker(next_num)) return false;
        next_num = (next_num + 1u) & 7u;
        if (src.marker == 0) insufficient = false; // "unless read_restart_marker left us smack up against a marker"
        return true;
    }
    // jdhuff.c decode_mcu: the fast decoder while the buffer holds 512 bytes per block of the MCU and no marker is pending -- never with
    // a restart interval, never in a progressive scan (jdphuff.c has no fast path)
    void mcu_begin(uint32_t blocks, bool may_fast)
    {
        fast = may_fast && src.marker == 0 && (size_t)(src.end - src.p) >= (size_t)512 * blocks;
        fast_hit = false;
        snap.p = src.p; snap.buf = buf; snap.bits = bits;
    }
    bool mcu_redo()
    {
        if (!fast) return false;
        fast = false;
        if (!fast_hit) return false;
        src.p = snap.p; buf = snap.buf; bits = snap.bits; // decode_mcu_fast saves nothing when it gives up
        fast_hit = false;
        return true;
    }
};
