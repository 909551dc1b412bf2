// lp_types.h -- descriptors shared by the host orchestration and the HIP kernels.
//
// One LpJpeg per source image describes a baseline JPEG (the input side of lilliput's
// ImageOps.Transform, /root/reference/ops.go:352-444 -> opencv.cpp:126-171) after the host has
// parsed its markers (S0 in SURVEY.md 2a). Everything below the markers -- unstuffing, Huffman
// decode, IDCT, upsampling, colour conversion -- runs on the device.
#pragma once
#include <stdint.h>

#define LP_MAX_COMP 3           // components the subsequence-parallel baseline kernels take (grey, YCbCr, RGB)
#define LP_GEOM_COMP 4          // components an image may have: CMYK / YCCK files go through the per-scan path (LpProgScan)
#define LP_MAX_BPM 6            // blocks per MCU: 4:2:0 = 6, 4:2:2/4:4:0 = 4, 4:4:4 = 3, gray = 1
#ifndef LP_LUT_BITS
#define LP_LUT_BITS 10          // first-level Huffman lookup width. 9 (with 128-entry slices) was measured: it saves 3 KB of LDS per workgroup and
                                // lets a fifth WRITE workgroup onto each CU, which buys nothing (27.5 us per image either way: the kernel is
                                // not occupancy-bound at four) while the extra second-level lookups cost SPEC / VERIFY / WRITE 2-4 % each
#endif
#define LP_LUT_SIZE (1 << LP_LUT_BITS)
#define LP_LUT2_BITS (16 - LP_LUT_BITS) // second-level lookup: slices indexed by the bits that follow the first-level prefix (codes up to 16 bits)
#define LP_LUT2_SUBS 16          // slices shared by the four tables, one per first-level prefix of long codes (Annex-K tables need 1 + 5 + 5)
#define LP_LUT2_POOL (LP_LUT2_SUBS << LP_LUT2_BITS)
#define LP_MAX_CKPT 16          // checkpoints per subsequence

// Huffman decode tables of one image: 2 DC + 2 AC (baseline allows ids 0..1).
// lut[t][i]  : indexed by the next LP_LUT_BITS bits. A code of length <= LP_LUT_BITS: lp_lut_entry(t, len, symbol) =
//              (len + size) | size << 5 | run << 9 | ends_block << 15 (size / run = the symbol's nibbles; the first field is what the
//              counting passes advance by -- they never need the code length alone; ends_block marks the AC symbols that finish a
//              block: size 0, run != 15 -- it sits 64 x above the run so that one shift yields "run, + 64 at the end of a block", see
//              LpLane::step). The prefix of longer codes: first field 0 and bits 5..12 = the slice of lut2 that decodes them (0xff:
//              none -- not a prefix of any code, or the pool was exhausted -> canonical search through maxcode / valoff / vals).
// lut2[(slice << LP_LUT2_BITS) | j] : same encoding (len LP_LUT_BITS + 1 .. 16) for the codes that start with the slice's prefix, j = the
//              LP_LUT2_BITS bits after the prefix; 0 = no such code -> canonical search.
// Table slot t: 0 = DC0, 1 = DC1, 2 = AC0, 3 = AC1.
#define LP_HUFF_LDS_BYTES ((4 * LP_LUT_SIZE + LP_LUT2_POOL) * 2)   // the lookup part the kernels stage in LDS; the canonical part stays in HBM
#define LP_E_BITS(e) ((e) & 31u)            // code length + extra bits: what the symbol consumes; 0 = no short code here
#define LP_E_SIZE(e) (((e) >> 5) & 15u)
#define LP_E_RUNX(e) ((e) >> 9)              // run, + 64 when the symbol ends the block (entries are 16 bits wide)
#define LP_E_SLICE(e) (((e) >> 5) & 0xffu)   // of a first-level entry with LP_E_BITS == 0
#define LP_E_NO_SLICE ((uint16_t)(0xffu << 5))
// entry for a code of length len (1..17: 17 = libjpeg's "no code matches" sentinel) of table slot t decoding to symbol v
#if defined(__HIP__)
__host__ __device__
#endif
static inline uint16_t lp_lut_entry(int t, int len, unsigned v)
{
    const bool ends_block = t >= 2 && (v & 15u) == 0 && (v >> 4) != 15; // AC symbol of size 0 that is not ZRL: EOB (jdhuff.c decode_mcu)
    return (uint16_t)(((unsigned)len + (v & 15u)) | ((v & 15u) << 5) | (((v >> 4) & 15u) << 9) | (ends_block ? 0x8000u : 0u));
}
// lutm[t][i] : the counting passes' MULTI-SYMBOL entry for the same index (round 5; SPEC / VERIFY only need where the next code starts,
//              how far the zigzag index moves and whether the block ended -- not the values): every symbol whose CODE lies inside the
//              LP_LUT_BITS window, first to last, as one entry of the same layout -- field 0 = the bits all of them consume, run field =
//              (zigzag advance of all of them) - 1, or the advance of the symbols before the EOB that ends the run, + 64 (ends_block).
//              A lane uses it when z + (run field & 63) < 64 -- then no symbol of the group is read at or beyond coefficient 64, where
//              jdhuff.c's decode_mcu has left the block -- and the one-symbol entry otherwise. Equal to lut[t][i] where only one
//              symbol fits (and for every prefix of a long code). DC slots: the DC symbol and the AC symbols behind it, when every block
//              that uses DC slot d uses one and the same AC slot (lp_build_huff_multi).
struct LpHuffSet {
    uint16_t lut[4][LP_LUT_SIZE];
    uint16_t lut2[LP_LUT2_POOL];
    uint32_t lut2_used;         // slices handed out so far (host-side bookkeeping while the four slots are built)
    uint32_t pad[3];
    uint16_t lutm[4][LP_LUT_SIZE];
    int32_t maxcode[4][18];     // maxcode[l] = largest code of length l, -1 if none; [17] = sentinel
    int32_t valoff[4][17];      // valptr[l] - mincode[l]
    uint8_t vals[4][256];
};

struct LpJpeg {
    // ---- stream
    uint64_t raw_off;           // byte offset of the entropy-coded segment inside the raw arena, rounded DOWN to 16 (the unstuff kernels load 16 bytes per lane)
    uint32_t raw_len;           // bytes of ECS (markers inside: FF00, RSTn; ends before EOI)
    uint32_t raw_skip;          // 0..15: bytes between raw_off and the segment's first byte. Non-zero when several pinned sources that lie next
                                // to each other in host memory were fetched with ONE copy: their spacing in the arena is their spacing on the host
    uint64_t clean_off;         // word (uint32) offset of this image's unstuffed stream in the clean arena
    uint32_t clean_cap_words;
    uint32_t huff_idx;          // index into the LpHuffSet array
    // ---- geometry
    uint32_t width, height;
    uint32_t mcus_x, mcus_y;
    uint32_t total_blocks;
    uint32_t dri;               // MCUs per restart interval, 0 = none
    uint8_t ncomp, hmax, vmax, bpm;
    uint8_t colorspace;         // 1 gray, 2 YCbCr, 3 RGB, 4 CMYK, 5 YCCK (four components: decoded to BGR the way cv::JpegDecoder does)
    uint8_t orientation;        // EXIF 1..8
    uint8_t scan_path;          // 1: decoded scan by scan (LpProgScan: progressive files, sequential ones in several scans, four components,
                                // unusual sampling) -- the subsequence-parallel Huffman stages skip the image and coef_off counts
                                // int16 elements in the progressive arena (see LpProgScan)
    uint8_t generic_sampling;   // 1: sampling factors other than luma 1x1 / 2x1 / 1x2 / 2x2 over 1x1 chroma (4:1:1, 4:1:0, chroma larger
                                // than luma, CMYK ...): every component goes through its own upsampler in k_ycc_to_frame, no fused resample
    uint8_t hs[LP_GEOM_COMP], vs[LP_GEOM_COMP];
    uint8_t dc_tbl[LP_GEOM_COMP], ac_tbl[LP_GEOM_COMP];   // slots into LpHuffSet (0..1 / 2..3)
    uint8_t blk_comp[8], blk_h[8], blk_v[8];            // per block-in-MCU
    uint8_t blk_first[LP_GEOM_COMP], pad2;               // index inside the MCU of a component's first block
    uint32_t pad3;
    uint64_t blkpack;           // 4 bits per block-in-MCU b (bits 4b..4b+3): component (2) | DC table id (1) << 2 | AC table id (1) << 3
    uint32_t bw[LP_GEOM_COMP], bh[LP_GEOM_COMP];          // blocks per row / column (MCU padded)
    uint64_t coef_off;          // element offset of this image's blocks in the coefficient arenas (int8 blocks, int16 wide slots; /64 =
                                // block offset); blocks are stored in DECODE order (MCU by MCU, blocks of an MCU in scan order),
                                // 64 natural-order coefficients each
    uint64_t plane_off[LP_GEOM_COMP];                    // byte offset into the plane arena
    uint32_t plane_stride[LP_GEOM_COMP];                 // = bw*8
    uint16_t qt[LP_GEOM_COMP][64];                       // dequantisation table per component, natural order
    // ---- subsequence bookkeeping
    uint32_t sub_bits;          // subsequence size S of this image in bits (multiple of 32): chosen so that the subsequence
                                // count lands just below a multiple of 256 -- whole Huffman workgroups, no nearly-empty tail block
    uint32_t sub_off;           // index of this image's first subsequence in the per-subsequence arrays
    uint32_t sub_cap;           // capacity (from raw_len, an upper bound of the clean length)
    uint32_t rst_off;           // index of this image's first entry in the restart-position array
    uint32_t rst_cap;
    uint32_t chunk_off;         // index of first unstuff chunk in the chunk-count array
    uint32_t nchunks;
};

// ---- progressive JPEG (SOF2): one entry per scan. A scan's entropy-coded data is decoded by ONE lane from start to end:
// DC / AC "first" scans would self-synchronise like a baseline stream, but a refinement scan's parse depends on which
// coefficients of the block at hand are already non-zero, so a lane that does not know its block index cannot decode it at all.
// Parallelism comes from the images of a batch and from the scans of an image that touch different (component, band) pairs:
// the engine sorts the scans into dependency levels and launches one kernel per level (lane = one scan of one image).
// Coefficients accumulate in an int16 arena: per image, per component, blocks in raster order over the MCU-padded grid,
// 64 values per block in ZIGZAG order (see lp_prog_core.h: a refinement scan then works on a 64-bit non-zero mask).
#define LP_MAX_SCANS 1024        // libjpeg has no limit; real files stay below 20 scans, hostile ones are cut off here ("unsupported")
struct LpProgHuff {             // the Huffman tables one scan uses. Progressive scans: slot = position of the component in the scan (its
                                // DC or its AC table, whichever the scan codes). Sequential scans (see LpProgScan::sequential) need both:
                                // slot s = DC table, slot 4 + s = AC table of scan component s.
    uint16_t lut8[8][256];      // (length << 8) | symbol for codes of up to 8 bits, indexed by the next 8 bits; 0 = longer code
    int32_t maxcode[8][18];     // canonical decode of the longer codes (T.81 F.2.2.3)
    int32_t valoff[8][17];
    uint8_t vals[8][256];
};
struct LpProgScan {
    uint32_t img;               // index of the image in the current decode range
    uint32_t stream;            // index of this scan's pseudo stream (LpJpeg entry used by the unstuff kernels) and of its LpJpegState
    uint32_t huff;              // index into the LpProgHuff array
    uint32_t ns;                // components in the scan
    uint8_t comp[4];            // component index of scan position s
    uint8_t Ss, Se, Ah, Al;
    uint32_t dri;               // restart interval in MCUs of THIS scan, 0 = none
    uint32_t mcux, mcuy;        // MCU grid of the scan: the padded grid for interleaved scans, the component's own blocks otherwise
    uint32_t cblk[4];           // first block of scan component s, relative to the image's first block in the arena
    uint32_t bw[4];             // blocks per row of scan component s (MCU padded)
    uint8_t hs[4], vs[4];       // blocks of scan component s per MCU (1 x 1 in a single-component scan)
    uint8_t sequential;         // 1: a scan of a SEQUENTIAL (SOF0 / SOF1) file the baseline kernels do not take -- several scans, components
                                // in another order than the frame's, table numbers above 1, four components: whole blocks, DC then AC
                                // (jdhuff.c decode_mcu), through the same per-scan machinery as progressive scans
    uint8_t pad8[7];
    uint64_t coef_off;          // element offset of the image in the int16 coefficient arena
};

// Which earlier scans of its image a scan must stay behind when all dependency levels of a decode range run in ONE launch
// (lp_kernels_prog.hip "pipelined"): the scans that wrote the coefficients it refines. Progress is published per MCU row of the
// producer's own grid; vs_self / vs_dep = block rows of the shared component per MCU row of the two scans (1 in a single-component scan).
#define LP_PROG_MAX_DEPS 3
struct LpProgDep {
    uint32_t ndep;
    uint32_t scan[LP_PROG_MAX_DEPS];    // indices into the launch's scan array (always lower than the scan's own)
    uint8_t vs_self[LP_PROG_MAX_DEPS];
    uint8_t vs_dep[LP_PROG_MAX_DEPS];
    uint8_t pad[2];
};

// Per-image results produced on the device.
struct LpJpegState {
    uint32_t clean_bytes;       // unstuffed length
    uint32_t n_rst;             // restart markers found
    uint32_t nsub;              // ceil(clean_bytes*8 / S)
    uint32_t error;             // bit 0: a marker other than RSTn inside the entropy-coded data, bit 1: the stream came up short of blocks,
                                // bit 2: more restart markers than the list holds, bit 3: the restart markers are not RST0..7 in turn, one per
                                // interval, or an interval does not hold exactly its MCUs. Any of them sends a baseline image through the
                                // serial decoder, which does with such a stream what libjpeg does (lp_jbits.h)
    uint32_t end_marker_pos;    // raw position of the first non-RST marker (or raw_len)
    uint32_t blocks_decoded;
    uint32_t n_wide;            // blocks that needed a 16-bit (wide) slot
    uint32_t pad;
};

// LpJpegState::error of a progressive scan's pseudo stream: the wave decoder (lp_kernels_prog.hip) met something a well-formed stream does
// not hold; the image is decoded by the host route, which does what libjpeg does with it
#define LP_PROG_IRREGULAR 32u

// Decoder state at a symbol boundary.
struct LpSubState {
    uint32_t p;                 // bit position in the clean stream
    uint32_t bz;                // (b << 8) | z : block-in-MCU, zigzag index of next coefficient
};

// Summary of the blocks STARTING inside one subsequence (or inside its prefix up to a checkpoint). DC predictors are not
// tracked by the entropy passes at all: the WRITE pass stores DC *differences* and k_dc_scan turns them into absolute
// values afterwards (a prefix sum per component that restarts every restart interval), which removes the per-symbol DC
// bookkeeping from the three VALU-bound decode loops.
struct LpSubSum {
    uint32_t nblk;              // block starts
    uint32_t nreset;            // restart boundaries crossed
};

// Checkpoint of the speculative pass (16 B): decoder state + sums at a fixed ITERATION of the lane's decode loop.
struct LpCkptPk {
    uint32_t p;                 // 0xffffffff = not recorded
    uint32_t bz;
    uint32_t nblk;
    uint32_t nreset;
};

// Checkpoint schedule: checkpoint k is taken before iteration it[k] of the lane's decode loop. The spacing grows
// geometrically (8, 16, 24, 32, 48, 72, ... for large images): a verifying lane re-synchronises within a few hundred bits
// in the common case and can stop at the very next checkpoint, while 16 records still cover a whole subsequence.
struct LpCkSched { uint32_t K; uint32_t base; uint32_t it[LP_MAX_CKPT]; }; // it[] = the iterations; base = the first spacing (the kernels regenerate it[] from it)

// ---------------------------------------------------------------------------------------------
// Pixel frames and per-image operation descriptors (orientation, crop+resize, compositing, encode).
struct LpFrame {
    uint64_t off;               // byte offset into a frame arena
    uint32_t w, h, stride, cn;  // cn = 1 (gray), 3 (BGR), 4 (BGRA)
};

// Tables of libwebp's gamma-aware chroma down-sampling (k_webp_yuv420)
struct LpWebpYuvTab {
    uint16_t gamma_to_linear[256];
    int32_t linear_to_gamma[33];
    uint32_t pad[3];
};

struct LpOrientOp {
    LpFrame src, dst;
    uint32_t orientation;       // EXIF 1..8 (opencv.hpp:17-26)
    uint32_t pad;
};

struct LpTap { uint32_t si; float alpha; };

struct LpResizeOp {
    LpFrame src;                // source VIEW: off already includes the crop origin, w/h are the cropped size
    LpFrame dst;
    uint32_t mode;              // 0 copy, 1 area-fast (integer scale), 2 area (fractional), 3 linear with area coefficients
    uint32_t iscale_x, iscale_y;
    float inv_area;             // 1.f / (iscale_x*iscale_y)
    uint32_t xtab_off, ytab_off;     // into the tap arena (mode 2)
    uint32_t xrange_off, yrange_off; // into the range arena: mode 2 -> [dw+1]/[dh+1] tap ranges; mode 3 -> int32 triples
    uint32_t xmax;              // mode 3: first dx that only has one source column
    uint32_t fast;              // mode 2: 0 = generic kernel, else MAXT of k_resize_area3<MAXT> (3-channel, contiguous taps)
};

// Fused S3+S4+S5+S6+S7 for integer scales (resizeAreaFast_): destination pixel (dx, dy) is the mean of the
// BGR values of the source rectangle [x0 + dx*dxx + dy*dyx, +rw) x [y0 + dx*dxy + dy*dyy, +rh) of the UN-oriented
// decoded image; orientation and crop are folded into x0/y0 and the strides (a box sum does not care about order).
struct LpFusedOp {
    uint32_t img;               // index into the LpJpeg array of the current decode range
    uint32_t rw, rh;            // source rectangle size
    int32_t x0, y0, dxx, dxy, dyx, dyy;
    float inv_area;
    uint32_t round_2x2;         // 1: (sum+2)>>2 (ResizeAreaFastVec_SIMD_8u), 0: cvRound(sum * inv_area)
    uint32_t fast;              // 0: general kernel; 4 / 8 / 16: k_resample_420<fast> (chroma columns per box), set by the engine
    LpFrame dst;
};

struct LpCompositeOp {
    LpFrame src, dst;
    uint32_t kind;              // 0 alpha blend, 1 copy (channel fix-up), 2 clear
    uint32_t x0, y0, w, h;      // ROI in dst
    uint32_t pad;
};

// One GIF frame onto the persistent BGRA canvas (giflib.cpp:349-568 giflib_decoder_render_frame): background fill on the
// first frame, disposal of the previous frame's rectangle, snapshot for "restore to previous", then the palette lookup.
struct LpGifFrameOp {
    LpFrame canvas;             // BGRA, cn = 4
    uint64_t saved_off;         // device address of the snapshot canvas (tightly packed, same size)
    uint64_t index_off;         // device address of the frame's colour indices (raster_w * raster_h bytes)
    uint64_t palette_off;       // device address of 256 x {B, G, R, 255}
    uint32_t first;             // 1: paint the whole canvas with `bg` first, no disposal, no snapshot
    uint32_t dispose;           // of the previous frame: 0 keep, 1 to background, 2 to previous
    int32_t px, py, pw, ph;     // previous frame's rectangle, already clipped to the canvas (pw/ph >= 0)
    int32_t fx, fy, fw, fh;     // this frame's visible rectangle on the canvas (fw/fh may be <= 0: nothing drawn)
    int32_t skip_left, skip_top;
    int32_t raster_w;           // width of the index raster (the frame's own width, before clipping)
    int32_t transparent;        // colour index that leaves the canvas untouched, -1 = none
    int32_t color_count;        // indices >= this leave the canvas untouched as well
    uint8_t bg[4];              // B, G, R, A
};

// GIF encoder (giflib.cpp:921-1105 giflib_encoder_render_frame): BGRA frame -> palette indices. The reference walks the pixels
// in raster order with a 32 768-entry cache keyed by the colour crushed to 5 bits per channel: the FIRST pixel that lands in an
// empty bucket decides the bucket's palette entry (nearest by Manhattan distance to the bucket midpoint, or to the pixel itself
// when it is nearly white / black), every later pixel of the bucket reuses it. Three kernels reproduce that order dependence:
// k_gifenc_first (per bucket: lowest raster index among this frame's pixels, only where the cache is empty), k_gifenc_fill
// (per claimed bucket: the entry that first pixel would have chosen), k_gifenc_map (per pixel: entry, or the transparent
// index when the pixel is see-through or when the previous output frame already shows a closer colour).
struct LpGifEncOp {
    LpFrame frame;              // BGRA, the output-sized frame
    uint64_t prev_off;          // device address of the previous output frame (tightly packed BGRA), read when use_prev
    uint64_t lookup_off;        // device address of uint16[32768]: palette index, 0xffff = empty (persists while the palette stays the same)
    uint64_t first_off;         // device address of uint32[32768]: scratch, 0xffffffff = unclaimed
    uint64_t fresh_off;         // device address of uint32[32768]: distance found when the bucket was filled
    uint64_t palette_off;       // device address of 256 x {R, G, B, 0}
    uint64_t out_off;           // device address of the index raster (frame.w * frame.h bytes)
    int32_t color_count;
    int32_t transparent;        // -1 = none
    uint32_t use_prev;          // previous frame stays on screen (disposal "unspecified" / "do not dispose") and a first frame exists
    uint32_t pad;
};

// One PNG image on the device: `data_off` holds the inflated stream (per Adam7 pass, per row: filter byte + packed row);
// k_png_unfilter reconstructs it in place, k_png_convert expands it to the 8-bit BGR(A) / grey frame OpenCV's PngDecoder yields.
struct LpPngPass {
    uint64_t off;               // byte offset of the pass inside the inflated stream
    uint32_t pw, ph;            // pixels per row / rows of this pass (0 = empty pass, no data)
    uint32_t x0, y0, dx, dy;    // where pass pixel (0,0) lands in the image and the steps between pass pixels
    uint32_t row_bytes;         // packed bytes per row without the filter byte
    uint32_t pad;
};
struct LpPngOp {
    LpFrame dst;                // cn = 1, 3 or 4
    uint64_t data_off;          // device address of the inflated stream
    uint64_t palette_off;       // device address of 256 x {B, G, R, A} (palette images)
    uint64_t error_off;         // device address of a uint32 flag: set when a row carries a filter type above 4
    uint64_t sync_off;          // device address of the un-filter kernel's tickets and mailboxes (lp_png_sync_bytes(op), zeroed before the launch)
    LpPngPass pass[7];
    uint32_t npass;
    uint32_t depth, color_type; // as in IHDR
    uint32_t bpp;               // filter unit: bytes per complete pixel, at least 1
    uint32_t has_key;           // RGB / grey colour key from tRNS (only RGB makes a difference: alpha 0 where the pixel equals it)
    uint32_t key[3];            // 16-bit R, G, B
};
// k_png_unfilter's scratch (lp_kernels_pixel.hip): a ticket counter per (pass, channel), the dump slots, then per pass one 32-bit
// mailbox word per byte of a row -- the channel through which the last row of a band of 64 rows reaches the band below.
#define LP_PNG_TICKET_BYTES 256     // 7 passes x 8 channels x uint32
#define LP_PNG_DUMP_BYTES 256       // where lanes outside their row send their stores (never read)
#define LP_PNG_MARGIN 1024          // bytes the device keeps readable before and behind the inflated stream: lanes of the un-filter kernel that are
                                    // not inside their row yet (or any more) fetch up to 63 + 32 filter units beside it, and use none of it
#if defined(__HIP__)
#define LP_TYPES_HD __host__ __device__
#else
#define LP_TYPES_HD
#endif
static inline LP_TYPES_HD uint32_t lp_png_bands(const LpPngPass& q) { return (q.pw && q.ph) ? (q.ph + 63u) / 64u : 0u; }
static inline size_t lp_png_sync_bytes(const LpPngOp& op)
{
    size_t words = 0;
    for (uint32_t p = 0; p < op.npass; p++) words += lp_png_bands(op.pass[p]) ? op.pass[p].row_bytes : 0u;
    return LP_PNG_TICKET_BYTES + LP_PNG_DUMP_BYTES + words * 4;
}

// PNG output (cv::PngEncoder::write over libpng, opencv.cpp:185-194 with FileType ".png"): every row of the frame becomes a filter
// byte + the filtered row, RGB(A) byte order. libpng picks a row's filter by trying the enabled ones (NONE, SUB, UP, AVG, PAETH in this
// order) and keeping the first with the smallest sum of |signed byte| (pngwutil.c png_write_find_filter); a row only needs the
// UNFILTERED row above, so all rows are independent: one workgroup per row.
struct LpPngEncOp {
    LpFrame src;                // grey, BGR or BGRA
    uint64_t out_off;           // device address of the filtered stream: h * (1 + w * cn) bytes
    uint32_t filters;           // bit f set = filter type f may be used (libpng's do_filter after png_write_start_row's pruning)
    uint32_t pad;
};

// HDR -> SDR tone map of a decoded 8-bit frame (color_info.cpp:80-236 tonemap_rgb_to_sdr): four passes over a float copy of the
// three colour channels; the global statistics between the passes (minimum / maximum, log-luminance mean, channel means) come back as
// per-workgroup partials in `stats` (LP_TONE_STATS doubles each) and are folded on the host, which also derives the next pass's scalars.
#define LP_TONE_STATS 8         // per workgroup: min, max, sum log, sum gray, sum c0, sum c1, sum c2, -
#define LP_TONE_MAX_WG 1024
struct LpToneOp {
    LpFrame f;                  // 3 or 4 channels; a fourth channel (alpha) is left as it is
    uint64_t img;               // device address: w * h * 3 floats
    uint64_t stats;             // device address: LP_TONE_MAX_WG * LP_TONE_STATS doubles
    int32_t transfer, primaries; // cICP code points
    float a, b;                 // the pass's normalisation: v * a + b
    float glob[3];              // Reinhard: per-channel global adaptation level
    float map_key, intensity;
    uint32_t depth;             // bits per sample of src16 (tonemap_rgb_to_sdr), unused for an 8-bit frame
    uint64_t src16;             // device address of w * h * 3 uint16 samples, or 0: pass 0 reads the frame's own bytes
};

// JPEG encode job (S8-S10): pixels -> baseline 4:2:0 (or grayscale) JFIF stream with Annex-K tables.
struct LpEncJob {
    LpFrame src;
    uint32_t ncomp;             // 1 or 3
    uint32_t mcus_x, mcus_y, bpm, total_blocks;
    uint32_t wib, hib;          // real luma blocks per row / column; the rest of the MCU grid is dummy
    uint32_t blk_off;           // index of this image's first block in the per-block arrays
    uint64_t coef_off;          // int16 element offset into the encoder coefficient arena (zigzag order, MCU order)
    uint64_t bits_off;          // uint32 word offset into the bit-buffer arena (unstuffed entropy-coded segment)
    uint32_t bits_cap_words;
    uint32_t hdr_off, hdr_len;  // pre-built SOI..SOS header bytes
    uint32_t out_cap;
    uint64_t out_off;           // byte offset into the output arena
    uint16_t qt[2][64];         // luma / chroma quantisation tables, natural order
};

struct LpEncState {
    uint32_t total_bits;
    uint32_t out_len;           // 0 on failure
    uint32_t error;             // 1 = output buffer too small
    uint32_t pad;
};
