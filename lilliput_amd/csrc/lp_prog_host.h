// lp_prog_host.h -- progressive (SOF2) and other scan-by-scan sources: the scans' entropy decode on host threads, and the switch between
// this home and the device's (lp_kernels_prog.hip: a wave per scan).
//
// A progressive scan is serial by construction (a refinement scan parses differently depending on which coefficients of the
// block at hand are already non-zero, so nothing downstream of an unknown block index can be decoded). Host threads decode the scans --
// lp_prog_core.h over the raw bytes (lp_jbits.h: libjpeg's reader restated, damaged data included) -- straight into a pinned int16
// coefficient buffer, like the PNG path's inflate and the GIF path's LZW ("host inflate"); dequantisation,
// IDCT, upsampling, colour, resample and encode stay on the device. This is part of the designed path, not a fallback: it feeds
// k_idct<true> and fails like everything else when there is no device. It is the home of small sets (lower latency: a host core walks one
// chain ~3x faster than a wave), of QM-coded files, of sequential multi-scan files in auto mode, and of every image the device decoders
// give up on; large sets of progressive files go to the device (see lp_prog_entropy_mode below). All homes are tested against the oracle
// and against each other.
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
    size_t coef_elems;              // int16 elements of the image's coefficients behind `coef` (all components): zeroed again when the image is
                                    // decoded a second time in file order (a scan stored outside its band, lp_prog_host_run)
};
// bit of *LpProgHostTask::error while lp_prog_host_run is at work (cleared before it returns): a scan of the image stored outside its band
#define LP_PROG_HOST_STRAY 0x40000000u

// Runs the tasks level by level on up to `nthreads` threads (0 = LILLIPUT_HIP_PROG_THREADS, default max(min(16, cores), cores / 4) capped at 64).
void lp_prog_host_run(std::vector<LpProgHostTask>& tasks, int nthreads);

// Dependency level of every scan of one image: scans that touch the same coefficients of the same component keep file order.
void lp_prog_levels(const std::vector<LpProgScanHost>& scans, std::vector<uint32_t>& level);

// Where the scans of a scan-path image are entropy-decoded. LILLIPUT_HIP_PROG_ENTROPY = auto | host | device | lanes, or
// lilliput_hip_set_progressive_entropy(-1 | 0 | 1 | 2):
//   0 host   host threads (this file), whatever the set holds.
//   1 device every Huffman-coded scan-path image of an upload set on the device: a WAVE per progressive scan (lp_kernels_prog.hip), a lane per
//            sequential scan (k_prog_scan). QM-coded files stay on host threads (a set may mix both).
//   2 lanes  as 1 with the generic one-lane-per-scan kernel for every scan: the wave decoder's tested reference (30-50x slower).
//  -1 auto   (default) device for the progressive images of a set that holds at least lp_prog_device_min_images() of them
//            (LILLIPUT_HIP_PROG_DEVICE_MIN, default five per usable host CPU: 80 on the 16-CPU box), host threads otherwise: a scan is a serial chain, a wave walks it ~3x slower than
//            a host core, and a set only offers (images x independent scans) chains -- few images are faster on the host's cores, many
//            on the device's thousands of wave slots.
// An image the device decoders give up on (damaged data; LpEngine::scan_gave_up) is decoded again by the host threads in every mode.
int lp_prog_entropy_mode();
uint32_t lp_prog_device_min_images();
// process-wide counters (lilliput_hip_progressive_stats): scan-path images whose scans ran on the device, those of them the device decoders
// gave up on (decoded again by the host threads), scans launched on the device
void lp_prog_count(uint64_t device_images, uint64_t gave_up, uint64_t device_scans);


// libjpeg-turbo 3.1.0's interblock smoothing (jdcoefct.c decompress_smooth_data, on by default and under cv::JpegDecoder) for a progressive
// image whose header walk set LpJpegHeader::ref_smooths: every one of the first nine AC coefficients (zigzag 1..9) of a block that is
// still zero and whose precision is not final (coef_bits != 0) is replaced by an estimate from the 5 x 5 neighbourhood of DC values; when
// NO AC data of a component was sent at all (coef_bits[1..9] all -1) the Gaussian-like kernel set is used and the DC itself is
// re-estimated. `coef` = the image's first block, components back to back, [bh][bw][64] in zigzag order over the MCU-padded grid
// (what the scan decoders write). In place; the neighbours' DC values are the ones from before the pass. The weights and the row
// rules at the bottom of the image were pinned against the reference's libjpeg.a (oracle.ref_jpeg_decode): tests/test_progressive.py.
void lp_prog_smooth(const LpJpegHeader& h, int16_t* coef);
