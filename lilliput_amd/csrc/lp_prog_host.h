// lp_prog_host.h -- hybrid mode for progressive (SOF2) sources: the scans' entropy decode on host threads.
//
// A progressive scan is serial by construction (a refinement scan parses differently depending on which coefficients of the
// block at hand are already non-zero, so nothing downstream of an unknown block index can be decoded), and one GPU lane runs
// such a chain ~30x slower than a CPU core (measured: 8.3 s per 4096 x 4096 image with one lane per scan against ~0.2 s here).
// So by default the scans are decoded by host threads -- the same lane logic, lp_prog_core.h -- straight into a pinned int16
// coefficient buffer, like the PNG path's inflate and the GIF path's LZW ("host inflate"); dequantisation,
// IDCT, upsampling, colour, resample and encode stay on the device. This is part of the designed path, not a fallback: it feeds
// k_idct<true> and fails like everything else when there is no device. LILLIPUT_HIP_PROG_ENTROPY=device (or
// lilliput_hip_set_progressive_entropy(1)) keeps the scans on the device instead (k_prog_scan); both modes are tested against
// the oracle and against each other.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "lp_jpeg_parse.h"

struct LpProgHostTask {
    const uint8_t* data;            // the file
    size_t len;                     // ... all of it: the reader stops where libjpeg's would (lp_jbits.h), not at a precomputed end
    const LpProgScanHost* scan;
    int16_t* coef;                  // the image's first block (zeroed before the first scan)
    uint32_t level;                 // dependency level: a task runs after every task of a lower level
    uint32_t* error;                // |= 8: the decoder needed bytes the file does not hold (cv::JpegDecoder's source manager suspends: the
                                    // reference fails the image); |= 16: a marker code libjpeg does not know is pending behind the scan
                                    // of a file that is read to its end before pixels are returned (JERR_UNKNOWN_MARKER)
    bool whole_file;                // several scans: libjpeg reads on to EOI before it returns (jdapimin.c jpeg_start_decompress)
};

// Runs the tasks level by level on up to `nthreads` threads (0 = LILLIPUT_HIP_PROG_THREADS, default max(min(16, cores), cores / 4) capped at 64).
void lp_prog_host_run(std::vector<LpProgHostTask>& tasks, int nthreads);

// Dependency level of every scan of one image: scans that touch the same coefficients of the same component keep file order.
void lp_prog_levels(const std::vector<LpProgScanHost>& scans, std::vector<uint32_t>& level);

// 1 = scans decoded on the device, 0 = on host threads (default)
bool lp_prog_entropy_on_device();
