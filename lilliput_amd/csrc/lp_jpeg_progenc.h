// lp_jpeg_progenc.h -- progressive (SOF2) JPEG OUTPUT: lilliput's EncodeOptions{JpegProgressive: 1} (opencv.go:47), which
// cv::JpegEncoder turns into jpeg_simple_progression() on libjpeg-turbo. Colour conversion, chroma downsampling, FDCT and
// quantisation stay on the device (k_enc_fdct, the same coefficients the baseline encoder codes); the multi-scan entropy coding
// with per-scan optimal Huffman tables (libjpeg forces optimize_coding in progressive mode) runs on the host: it is serial per
// scan and a thumbnail has a few thousand blocks -- microseconds of work on a few hundred kilobytes.
// The output is byte-identical to libjpeg-turbo 3.1.0's (tests compare with the reference's own library).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

// coef: quantised coefficients as k_enc_fdct leaves them -- MCU order (4:2:0: Y00 Y01 Y10 Y11 Cb Cr per MCU; grey: one block per
// MCU), 64 values per block in zigzag order, dummy blocks of the padded MCU grid included. ncomp 1 or 3. Returns false when the
// image cannot be coded (a coefficient outside the 8-bit JPEG range).
bool lp_jpeg_encode_progressive(int width, int height, int ncomp, int quality, const int16_t* coef, std::vector<uint8_t>& out);
