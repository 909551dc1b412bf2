// lp_jpeg_parse.h -- host-side JPEG marker parsing (S0) and table construction.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

#include "lp_types.h"
#include "lp_arith_host.h"

enum {
    LP_PARSE_OK = 0,
    LP_PARSE_NOT_JPEG = 1,      // no SOI / broken marker structure      -> ErrInvalidImage
    LP_PARSE_UNSUPPORTED = 2,   // lossless, 12-bit, fractional sampling ratios
    LP_PARSE_TRUNCATED = 3
};

struct LpProgScanHost {         // one scan of a file that is decoded scan by scan
    LpProgScan s;               // img / stream / huff indices are filled in by the engine
    LpProgHuff tables;
    size_t ecs_off, ecs_len;    // this scan's entropy-coded bytes in the file
    bool arith = false;         // an arithmetic-coded scan (SOF9 / SOF10): `tables` is unused, `ar` holds the conditioning (lp_arith_host.h)
    LpArithScan ar = {};
};

struct LpJpegHeader {
    LpJpeg j;                   // geometry + table slots filled; arena offsets left 0
    LpHuffSet huff;
    size_t ecs_off = 0;         // first entropy-coded byte in the file
    size_t ecs_len = 0;         // bytes up to (not including) the terminating marker / end of file (progressive: through the last scan)
    int saw_eoi = 0;
    bool arith = false;         // an arithmetic-coded file (always decoded scan by scan, on host threads)
    bool one_pass = true;       // libjpeg returns pixels while it reads the one scan (a sequential file whose first scan holds every
                                // component: jdinput.c has_multiple_scans false); otherwise the whole file is read first, to EOI
    bool decode_fails = false;  // the header walk already knows that the reference's read_data fails: a file of several scans without
                                // its EOI (jpeg_start_decompress suspends at the end of the buffer, cv::JpegDecoder gives up)
    bool open_end = false;      // the entropy-coded data of a one-scan file runs to the end of the buffer, no marker behind it: whether
                                // the reference decodes it depends on where libjpeg's read-ahead falls (lp_jbits.h): the serial route
    bool ref_smooths = false;   // a progressive file whose scan script leaves one of the first nine AC coefficients of some component short
                                // of full precision (never sent, or last sent with Al > 0): libjpeg then estimates those coefficients, where
                                // they are zero, from the neighbouring blocks' DC values (jdcoefct.c smoothing_ok / decompress_smooth_data,
                                // do_block_smoothing is on under cv::JpegDecoder): restated by lp_prog_smooth (lp_prog_host.h), which such an image's
                                // coefficients pass on the host before they reach the IDCT
    int8_t coef_bits[4][10] = {}; // with ref_smooths: libjpeg's coef_bits[component][zigzag 0..9] after the last scan (Al of the last scan that
                                // carried the coefficient, -1: none did)
    bool scan_path = false;     // decoded scan by scan (progressive, multi-scan sequential, four components, unusual sampling):
                                // `scans` lists the scans in file order, `huff` is unused
    std::vector<LpProgScanHost> scans;
};

// Parses up to SOS, locates the end of the scan, builds the decode tables.
int lp_jpeg_parse(const uint8_t* data, size_t len, LpJpegHeader* out);
// force_scans: take a sequential file scan by scan (LpProgScan::sequential, libjpeg's serial decode_mcu with its end-of-data rule) even
// when the subsequence-parallel baseline kernels could take it -- how a baseline stream that came up short of blocks on the device
// (truncated, damaged) is decoded again, to the pixels libjpeg returns for it.
int lp_jpeg_parse_opts(const uint8_t* data, size_t len, LpJpegHeader* out, bool force_scans);

// Build one table slot of an LpHuffSet from DHT counts/values.
void lp_build_huff_slot(LpHuffSet* hs, int slot, const uint8_t bits[17], const uint8_t* vals);
// After the four slots: the multi-symbol entries of the counting passes (LpHuffSet::lutm). ac_of_dc[d] = the AC slot (2 / 3) every
// block with DC slot d decodes its coefficients with, or -1 (no such block, or two different ones: DC entries then stay one symbol).
void lp_build_huff_multi(LpHuffSet* hs, const int ac_of_dc[2]);

// T.81 Annex K.3 tables (DC luma, AC luma, DC chroma, AC chroma): encoder output and the decoder's fallback for undefined ids 0/1.
extern const uint8_t lp_std_huff_bits[4][17];
extern const uint8_t lp_std_huff_dc_vals[12];
extern const uint8_t lp_std_huff_ac_luma[162];
extern const uint8_t lp_std_huff_ac_chroma[162];

// frame-header sniff (no entropy-coded byte read): a progressive (SOF2) Huffman-coded file?
bool lp_jpeg_sniff_progressive(const uint8_t* data, size_t len, uint64_t* coef_bytes = nullptr); // coef_bytes: the int16 coefficient bytes its frame header asks for
