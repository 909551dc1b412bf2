// lp_arith_host.h -- arithmetic-coded JPEG scans (SOF9 sequential, SOF10 progressive), entropy-decoded on host threads.
//
// The reference's libjpeg-turbo decodes arithmetic-coded files (jdarith.c behind opencv_decoder_read_data,
// /root/reference/opencv.cpp:166-171); round 3 rejected them. A QM-coded scan is one adaptive binary decision after the other --
// every decision reads the statistics bin the previous ones updated, so there is no subsequence to cut and nothing for a GPU
// lane to win (the progressive Huffman scans were already measured at 1 / 30 of a host core per lane). The scans
// therefore take the route of the progressive sources: host threads (lp_prog_host.cpp) fill the pinned int16 coefficient arena
// (zigzag order per block), the device takes over at the IDCT. Not a fallback: without a device the image fails like any other.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "lp_types.h"

// Conditioning of one scan (T.81 F.1.4.4.1.4 / F.1.4.4.2: the DAC marker's L, U and Kx; defaults 0, 1, 5) and the statistics areas
// its components use: scan components with the same table number share one area (jdarith.c dc_stats[tbl] / ac_stats[tbl]).
struct LpArithScan {
    uint8_t dc_tbl[4], ac_tbl[4];   // table numbers 0..15 of scan component s
    uint8_t dc_L[4], dc_U[4], ac_K[4]; // their conditioning values as the DAC segments before the scan left them
};

// Decodes one scan from its raw entropy-coded bytes (stuffed zeros and restart markers still in: the QM decoder's byte-in does the
// unstuffing, T.81 D.2.6) into the image's coefficient arena (blocks in raster order per component, 64 zigzag-ordered values each).
// The decoder reads up to the end of the FILE the way libjpeg does (the marker that ends the scan stops it; lp_jbits.h). whole_file:
// libjpeg reads this file to its end before it returns pixels (several scans), so what follows the scan matters.
// Returns LP_SCAN_OK, LP_SCAN_WARNED when the decoder met an impossible code (jdarith.c JWRN_ARITH_BAD_CODE: the rest of the scan is left
// alone), LP_SCAN_BAD_MARKER when the scan ended with a marker code pending that libjpeg does not know (read_markers:
// JERR_UNKNOWN_MARKER) or LP_SCAN_OUT_OF_DATA when the decoder needed a byte past the end of the file (JERR_CANT_SUSPEND under
// cv::JpegDecoder's source manager): the image fails in both.
int lp_arith_scan(const uint8_t* ecs, const uint8_t* file_end, const LpProgScan& sc, const LpArithScan& ar, int16_t* coef, bool whole_file);
