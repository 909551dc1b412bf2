/*
 * lilliput_hip.h -- C ABI of liblilliput_hip.so, the MI355X-native replacement for the hot path of
 * discord/lilliput's ImageOps.Transform (decode -> orientation/crop -> resize -> encode).
 *
 * Part A re-declares, with identical names, argument meaning and error behaviour, the subset of the
 * reference's cgo boundary /root/reference/opencv.hpp:57-145 that this path crosses; a Go build links
 * liblilliput_hip.so instead of opencv.cpp + the OpenCV/libjpeg static archives.
 * Part B is additive: a batched entry point (the synchronous one-image ABI cannot reach the throughput
 * target) -- same ownership rules, arrays of caller-owned buffers, per-item status.
 * Part C mirrors the Go-side API surface (Decoder / ImageOps / ImageOptions, ops.go + opencv.go +
 * lilliput.go) in C, because no Go toolchain exists in the build image; it drives Part A exactly the
 * way the Go code does.
 *
 * No torch / OpenCV types cross this boundary: plain pointers, sizes and ints only.
 */
#ifndef LILLIPUT_HIP_H
#define LILLIPUT_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part A -- drop-in for /root/reference/opencv.hpp
 * ---------------------------------------------------------------------------------------------- */

/* OpenCV type codes that cross the ABI as plain ints (opencv2/core/hal/interface.h; used by
 * /root/reference/opencv.go:232,241,443) */
#ifndef CV_8U
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24
#endif

/* opencv.hpp:17-26 */
typedef enum CVImageOrientation {
    CV_IMAGE_ORIENTATION_TL = 1,
    CV_IMAGE_ORIENTATION_TR = 2,
    CV_IMAGE_ORIENTATION_BR = 3,
    CV_IMAGE_ORIENTATION_BL = 4,
    CV_IMAGE_ORIENTATION_LT = 5,
    CV_IMAGE_ORIENTATION_RT = 6,
    CV_IMAGE_ORIENTATION_RB = 7,
    CV_IMAGE_ORIENTATION_LB = 8
} CVImageOrientation;

/* opencv.hpp:33-36 */
#define CV_IMWRITE_JPEG_QUALITY 1
#define CV_IMWRITE_PNG_COMPRESSION 16
#define CV_IMWRITE_WEBP_QUALITY 64
#define CV_IMWRITE_JPEG_PROGRESSIVE 2

/* opencv.hpp:53-55 (values of cv::INTER_AREA / INTER_LINEAR / INTER_CUBIC) */
extern const int CV_INTER_AREA;
extern const int CV_INTER_LINEAR;
extern const int CV_INTER_CUBIC;

/* opencv.hpp:57-59 */
typedef void* opencv_mat;
typedef void* opencv_decoder;
typedef void* opencv_encoder;

/* opencv.hpp:61-63 */
int opencv_type_depth(int type);
int opencv_type_channels(int type);
int opencv_type_convert_depth(int type, int depth);

/* opencv.hpp:65-74 -- JPEG sources are decoded on the device; PNG sources are inflated on the host and un-filtered / expanded
 * on the device; any other container yields NULL from opencv_decoder_create (the Go caller maps that to ErrInvalidImage,
 * opencv.go:453-456). */
opencv_decoder opencv_decoder_create(const opencv_mat buf);
const char* opencv_decoder_get_description(const opencv_decoder d);
void opencv_decoder_release(opencv_decoder d);
/* opencv.hpp:68: declared there, never defined or called by the reference. Here: the decoder chosen at create() now reads `buf`
 * (cv::ImageDecoder::setSource: no signature check, header state forgotten); false for a NULL argument. */
bool opencv_decoder_set_source(opencv_decoder d, const opencv_mat buf);
bool opencv_decoder_read_header(opencv_decoder d);
int opencv_decoder_get_width(const opencv_decoder d);
int opencv_decoder_get_height(const opencv_decoder d);
int opencv_decoder_get_pixel_type(const opencv_decoder d);
int opencv_decoder_get_orientation(const opencv_decoder d);
bool opencv_decoder_read_data(opencv_decoder d, opencv_mat dst);

/* opencv.hpp:75-93 */
int opencv_copy_to_region_with_alpha(opencv_mat src, opencv_mat dst, int xOffset, int yOffset, int width, int height);
int opencv_copy_to_region(opencv_mat src, opencv_mat dst, int xOffset, int yOffset, int width, int height);
void opencv_mat_set_color(opencv_mat, int red, int green, int blue, int alpha);
void opencv_mat_reset(opencv_mat mat);
int opencv_mat_clear_to_transparent(opencv_mat mat, int xOffset, int yOffset, int width, int height);

/* opencv.hpp:95-114 */
opencv_mat opencv_mat_create(int width, int height, int type);
opencv_mat opencv_mat_create_from_data(int width, int height, int type, void* data, size_t data_len);
opencv_mat opencv_mat_create_empty_from_data(int length, void* data);
bool opencv_mat_set_row_stride(opencv_mat mat, size_t stride);
void opencv_mat_release(opencv_mat mat);
void opencv_mat_resize(const opencv_mat src, opencv_mat dst, int width, int height, int interpolation);
opencv_mat opencv_mat_crop(const opencv_mat src, int x, int y, int width, int height);
void opencv_mat_orientation_transform(CVImageOrientation orientation, opencv_mat mat);
int opencv_mat_get_width(const opencv_mat mat);
int opencv_mat_get_height(const opencv_mat mat);
void* opencv_mat_get_data(const opencv_mat mat);

/* opencv.hpp:116-118 -- ".jpeg" / ".jpg" only; other extensions yield NULL. */
opencv_encoder opencv_encoder_create(const char* ext, opencv_mat dst);
void opencv_encoder_release(opencv_encoder e);
bool opencv_encoder_write(opencv_encoder e, const opencv_mat src, const int* opt, size_t opt_len);

/* opencv.hpp:118-132 -- colour metadata readers (host-side container walks; opencv.go:697-767, ops.go:306-333).
 * get_*_icc copy the embedded ICC profile into dest and return its length (0 = none / malformed / does not fit);
 * get_png_cicp returns 1 and the four H.273 code points when a valid cICP chunk precedes the image data;
 * png_insert_cicp splices a cICP chunk in after IHDR and returns the new length (the old one if nothing was done). */
int opencv_decoder_get_jpeg_icc(void* src, size_t src_len, void* dest, size_t dest_len);
int opencv_decoder_get_png_icc(void* src, size_t src_len, void* dest, size_t dest_len);
int opencv_decoder_get_png_cicp(void* src, size_t src_len, uint8_t* primaries, uint8_t* transfer, uint8_t* matrix, uint8_t* full_range);
size_t opencv_png_insert_cicp(void* png, size_t png_len, size_t png_cap, uint8_t primaries, uint8_t transfer, uint8_t matrix, uint8_t full_range);

/* opencv.hpp:135-145 */
#define OPENCV_SUCCESS 0
#define OPENCV_ERROR_INVALID_CHANNEL_COUNT 1
#define OPENCV_ERROR_OUT_OF_BOUNDS 2
#define OPENCV_ERROR_NULL_MATRIX 3
#define OPENCV_ERROR_RESIZE_FAILED 4
#define OPENCV_ERROR_COPY_FAILED 5
#define OPENCV_ERROR_CONVERSION_FAILED 6
#define OPENCV_ERROR_ALPHA_BLENDING_FAILED 7
#define OPENCV_ERROR_FINAL_CONVERSION_FAILED 8
#define OPENCV_ERROR_INVALID_DIMENSIONS 9
#define OPENCV_ERROR_UNKNOWN 10

/* ------------------------------------------------------------------------------------------------
 * Part A2 -- the decoder half of the reference's giflib.hpp C ABI (giflib.hpp:9-52; Go caller: giflib.go:56-242).
 * Container walk + LZW on the host, frame compositing (background, disposal, restore-to-previous, palette lookup) on the
 * device, on a canvas that stays in HBM for the life of the decoder.
 * ---------------------------------------------------------------------------------------------- */
struct GifAnimationInfo {   /* giflib.hpp:9-17 */
    int loop_count;
    int frame_count;
    int bg_red;
    int bg_green;
    int bg_blue;
    int bg_alpha;
    int duration_ms;
};

#define GIF_DISPOSE_NONE 0          /* giflib.hpp:19-21 */
#define GIF_DISPOSE_BACKGROUND 1
#define GIF_DISPOSE_PREVIOUS 2

typedef struct giflib_decoder_struct* giflib_decoder;   /* giflib.hpp:23 */
typedef struct giflib_encoder_struct* giflib_encoder;   /* giflib.hpp:24 */

typedef enum {              /* giflib.hpp:26-30 */
    giflib_decoder_have_next_frame,
    giflib_decoder_eof,
    giflib_decoder_error,
} giflib_decoder_frame_state;

/* giflib.hpp:33-43, 51-52 */
giflib_decoder giflib_decoder_create(const opencv_mat buf);
int giflib_decoder_get_width(const giflib_decoder d);
int giflib_decoder_get_height(const giflib_decoder d);
int giflib_decoder_get_num_frames(const giflib_decoder d);
int giflib_decoder_get_frame_width(const giflib_decoder d);
int giflib_decoder_get_frame_height(const giflib_decoder d);
int giflib_decoder_get_prev_frame_delay(const giflib_decoder d);
void giflib_decoder_release(giflib_decoder d);
giflib_decoder_frame_state giflib_decoder_decode_frame_header(giflib_decoder d);
bool giflib_decoder_decode_frame(giflib_decoder d, opencv_mat mat);   /* mat: screen-sized CV_8UC4, as gifDecoder.DecodeTo passes */
giflib_decoder_frame_state giflib_decoder_skip_frame(giflib_decoder d);
struct GifAnimationInfo giflib_decoder_get_animation_info(const giflib_decoder d);
int giflib_decoder_get_prev_frame_disposal(const giflib_decoder d);

/* giflib.hpp:45-50 -- a GIF is written from a GIF: palettes, delays and extension blocks come from the decoder `d`. The frame
 * passed to encode_frame is the screen-sized CV_8UC4 Mat the resize produced; its BGRA -> palette-index mapping runs on the
 * device, container and LZW coding on the host, byte-identical to giflib's writer. */
giflib_encoder giflib_encoder_create(void* buf, size_t buf_len);
bool giflib_encoder_init(giflib_encoder e, const giflib_decoder d, int width, int height);
bool giflib_encoder_encode_frame(giflib_encoder e, const giflib_decoder d, const opencv_mat frame);
bool giflib_encoder_flush(giflib_encoder e, const giflib_decoder d);
void giflib_encoder_release(giflib_encoder e);
int giflib_encoder_get_output_length(giflib_encoder e);

/* ------------------------------------------------------------------------------------------------
 * Part A3 -- the reference's webp.hpp C ABI (webp.hpp:13-75; Go caller: webp.go:27-261). RIFF container walk and animation writer are
 * this library's own; the VP8 / VP8L payloads are (de)coded on the host by libwebp; decoded (sub-)frames join the device path at the
 * next opencv_* call (blend / dispose on the HBM canvas, Fit, resize).
 * ---------------------------------------------------------------------------------------------- */
enum WebpEncoderOptions {   /* webp.hpp:13-23 */
    WEBP_METHOD = 1000,
    WEBP_FILTER_STRENGTH = 1001,
    WEBP_FILTER_TYPE = 1002,
    WEBP_AUTOFILTER = 1003,
    WEBP_PARTITIONS = 1004,
    WEBP_SEGMENTS = 1005,
    WEBP_PREPROCESSING = 1006,
    WEBP_THREAD_LEVEL = 1007,
    WEBP_PALETTE = 1008
};
typedef struct webp_decoder_struct* webp_decoder;   /* webp.hpp:28-29 */
typedef struct webp_encoder_struct* webp_encoder;
/* webp.hpp:35-51, 72-73 */
webp_decoder webp_decoder_create(const opencv_mat buf);
int webp_decoder_get_width(const webp_decoder d);
int webp_decoder_get_height(const webp_decoder d);
int webp_decoder_get_pixel_type(const webp_decoder d);
int webp_decoder_get_num_frames(const webp_decoder d);
int webp_decoder_get_total_duration(const webp_decoder d);
int webp_decoder_get_prev_frame_delay(const webp_decoder d);
int webp_decoder_get_prev_frame_dispose(const webp_decoder d);
int webp_decoder_get_prev_frame_blend(const webp_decoder d);
int webp_decoder_get_prev_frame_x_offset(const webp_decoder d);
int webp_decoder_get_prev_frame_y_offset(const webp_decoder d);
bool webp_decoder_get_prev_frame_has_alpha(const webp_decoder d);
uint32_t webp_decoder_get_bg_color(const webp_decoder d);
uint32_t webp_decoder_get_loop_count(const webp_decoder d);
size_t webp_decoder_get_icc(const webp_decoder d, void* buf, size_t buf_len);
void webp_decoder_release(webp_decoder d);
bool webp_decoder_decode(webp_decoder d, opencv_mat mat);   /* the Mat takes the frame's own dimensions: animation frames are sub-rectangles */
void webp_decoder_advance_frame(webp_decoder d);
int webp_decoder_has_more_frames(webp_decoder d);
/* webp.hpp:56-71 -- one frame: a still image through libwebp's simple API (quality / lossless above 100; the other options are
 * ignored for stills, as in the reference); from the second frame on an animation: every frame's changed rectangle coded with the
 * caller's options and placed without blending (the reference's WebPAnimEncoder picks rectangles and key frames its own way: the
 * files differ, the decoded frames do not beyond the lossy coder's error). flush assembles the container, ICCP chunk included. */
webp_encoder webp_encoder_create(void* buf, size_t buf_len, const void* icc, size_t icc_len, uint32_t bgcolor, int loop_count);
size_t webp_encoder_write(webp_encoder e, const opencv_mat src, const int* opt, size_t opt_len, int delay, int blend, int dispose, int x_offset, int y_offset);
void webp_encoder_release(webp_encoder e);
size_t webp_encoder_flush(webp_encoder e);

/* ------------------------------------------------------------------------------------------------
 * Part A4 -- the reference's color_info.hpp C ABI (color_info.hpp:24-106; Go callers: opencv.go:281-290, 730-812, ops.go:154-165,
 * 489-538). The tone map (PQ / HLG -> linear light -> Reinhard -> BT.709 primaries -> 8 bit) runs on the device; the ICC profiles
 * cicp_get_icc_profile hands out are derived from the standards' chromaticities at first use (ICC v4.2 matrix/TRC), not embedded blobs.
 * ---------------------------------------------------------------------------------------------- */
bool is_hdr_transfer_function(const uint8_t* icc_data, size_t icc_len);            /* color_info.cpp:17-36: ICC 'cicp' tag names PQ or HLG */
bool cicp_is_hdr_transfer(uint8_t transfer);                                       /* color_info.cpp:38-41 */
const uint8_t* cicp_get_icc_profile(uint8_t primaries, size_t* profile_size);      /* color_info.cpp:43-68 */
bool icc_header_is_sane(const uint8_t* icc, size_t icc_len);                       /* color_info.cpp:70-79 */
void tonemap_rgb_to_sdr(const uint16_t* src, uint8_t* dst, int width, int height, int src_depth, uint8_t transfer, uint8_t primaries); /* :112-204 */
void tonemap_rgb_8u_inplace(uint8_t* pixels, int width, int height, int channels, uint8_t transfer, uint8_t primaries);                /* :206-236 */
const uint8_t* lilliput_hip_srgb_icc_profile(size_t* profile_size);                /* lilliput.go:18-22 SRGBICCProfile */
/* Framebuffer.TonemapToSDR (opencv.go:794-812) on a Mat whose pixels are resident on the device. 0 = done (or not a 3/4-channel 8-bit Mat). */
int lilliput_hip_mat_tonemap(opencv_mat mat, uint8_t transfer, uint8_t primaries);

/* Test access: the host half of decode_frame for the frame whose header was just read (no device work).
 * meta = {left, top, width, height, interlace, disposal, delay, transparent, color_count, has_local_map}; returns the
 * number of indices written, -1 on a decode error, -2 when cap is too small. */
int lilliput_hip_gif_read_frame(giflib_decoder d, uint8_t* indices, size_t cap, int meta[10], uint8_t palette_rgb[768]);

/* thumbhash.hpp:12-16 -- ThumbHash of a frame (CV_8U / CV_8UC3 / CV_8UC4): at most 100 x 100 nearest-neighbour samples are gathered
 * on the device, the hash is computed on the host in the reference's summation order. Returns its length, -1 on error. */
typedef struct thumbhash_encoder_struct* thumbhash_encoder;
thumbhash_encoder thumbhash_encoder_create(void* buf, size_t buf_len);
int thumbhash_encoder_encode(thumbhash_encoder e, const opencv_mat frame);
void thumbhash_encoder_release(thumbhash_encoder e);

/* ------------------------------------------------------------------------------------------------
 * Part B -- batched extension (additive)
 * ---------------------------------------------------------------------------------------------- */

/* lilliput error values (lilliput.go:24-31) as ints, 0 = nil */
#define LILLIPUT_OK 0
#define LILLIPUT_ERR_INVALID_IMAGE 1
#define LILLIPUT_ERR_DECODING_FAILED 2
#define LILLIPUT_ERR_BUF_TOO_SMALL 3
#define LILLIPUT_ERR_UNSUPPORTED 4      /* stream feature the device path does not cover; nothing is written */
#define LILLIPUT_ERR_DEVICE 5           /* no GPU / HIP failure */
#define LILLIPUT_ERR_FRAMEBUF_NO_PIXELS 6
#define LILLIPUT_ERR_ENCODE_TIMEOUT 7
#define LILLIPUT_ERR_EOF 8
#define LILLIPUT_ERR_SKIP_NOT_SUPPORTED 9   /* ErrSkipNotSupported, lilliput.go:29 */
#define LILLIPUT_ERR_GIF_ENCODER_NEEDS_DECODER 10   /* ErrGifEncoderNeedsDecoder, giflib.go:44 */
#define LILLIPUT_ERR_OPENCV_BASE 100        /* + OPENCV_ERROR_* : handleOpenCVError, opencv.go:399-426 */

/* ops.go:18-22 */
#define LILLIPUT_OPS_NO_RESIZE 0
#define LILLIPUT_OPS_FIT 1
#define LILLIPUT_OPS_RESIZE 2

/* Hand-over of frames that were decoded outside this library (lilliput.go:136-164 sends AVIF to libavif and MP4 / MOV / WEBM to
 * libavcodec: AV1 / H.264 / VPx are serial codecs that stay on the host; SURVEY.md section 2 #6 "host decode -> GPU takes over at
 * BGR(A)"). A source buffer that starts with this header is a decoded frame: the header, then height rows of width * channels bytes
 * (B, G, R[, A], the layout avcodec_decoder_decode / avif_decoder_decode leave in the framebuffer), stride bytes apart. It is accepted
 * wherever an encoded source is -- lilliput_new_decoder, every batch / node call -- and goes through the same orientation, Fit / resize
 * and encode stages on the device; Description() is "PIXELS". */
#define LILLIPUT_HIP_PIXELS_MAGIC "LPPIXELS"
typedef struct lilliput_hip_pixels_header {
    char magic[8];          /* LILLIPUT_HIP_PIXELS_MAGIC */
    uint32_t width, height;
    uint32_t channels;      /* 1 (gray), 3 (BGR) or 4 (BGRA) */
    uint32_t stride;        /* bytes between rows; 0 = width * channels */
    uint32_t orientation;   /* EXIF orientation 1..8 the decoder reported (avcodec's display matrix, avif's irot / imir) */
    uint32_t duration_ms;   /* frame duration; 0 for a still */
} lilliput_hip_pixels_header;
/* Copy rows of pixels into a Mat of the matching shape (the host side of the hand-over). 0 = ok. */
int lilliput_hip_mat_set_pixels(opencv_mat mat, const void* pixels, size_t stride);

typedef struct lilliput_batch_item {
    const void* src;    /* encoded source image (caller-owned, like Go's []byte input) */
    size_t src_len;
    void* dst;          /* caller-owned output buffer, like the dst []byte of ImageOps.Transform */
    size_t dst_cap;
    size_t dst_len;     /* out: bytes written */
    int status;         /* out: LILLIPUT_* */
    int out_width;      /* out */
    int out_height;     /* out */
} lilliput_batch_item;

typedef struct lilliput_batch_options {
    int width, height;          /* ImageOptions.Width / Height (ops.go:31-35) */
    int resize_method;          /* ImageOptions.ResizeMethod */
    int normalize_orientation;  /* ImageOptions.NormalizeOrientation */
    int jpeg_quality;           /* EncodeOptions[JpegQuality]; 0 -> OpenCV's default 95 */
    int chunk;                  /* images in flight on the device at once; 0 = automatic */
    int jpeg_progressive;       /* EncodeOptions[JpegProgressive]: non-zero -> progressive output (device FDCT, multi-scan entropy coding on host threads) */
} lilliput_batch_options;

typedef void* lilliput_hip_batch;

lilliput_hip_batch lilliput_hip_batch_create(int device);
void lilliput_hip_batch_destroy(lilliput_hip_batch b);
/* What n calls of ImageOps.Transform do in the reference (ops.go:352-444; every call starts from the caller's encoded bytes,
 * opencv.cpp:99-171, and ends with the encoded result in the caller's dst, opencv.go:872-900), as ONE call: JPEG (and PNG / GIF)
 * sources -> (orientation, Fit/Resize) -> JPEG for n independent images, host bytes in, host bytes out. Ingest is pipelined with
 * the device work: per engine a stager thread walks the headers of chunk k + 1, copies its entropy-coded bytes into pinned memory
 * and enqueues the H2D copy on a copy stream while chunk k is decoded. Returns the number of failed items. */
int lilliput_hip_batch_transform(lilliput_hip_batch b, lilliput_batch_item* items, size_t n, const lilliput_batch_options* opt);
/* ONE image, from any thread, through the process-wide dispatchers that turn concurrent calls into shared batch launches
 * (lilliput_amd/csrc/lp_coalesce.h): what a Go build's ImageOps.Transform calls for a static source with JPEG output when the
 * service runs one ImageOps per goroutine (README.md:82-85). Blocks until the item is done; returns its LILLIPUT_*
 * status, *dst_len = bytes written. Same results as n such items in one lilliput_hip_batch_transform. Part C's
 * lilliput_image_ops_transform takes this route by itself once LILLIPUT_HIP_COALESCE (3) calls are in flight. */
int lilliput_hip_transform_one(int device, const void* src, size_t src_len, const lilliput_batch_options* opt, void* dst, size_t dst_cap, size_t* dst_len);
/* Of the last transform: out[0] entropy-coded bytes that reached the device, out[1] host ms the ingest threads spent (header walk,
 * registration, staging copies, copy enqueue; summed over the threads), out[2] ms the compute threads waited for a chunk, out[3] wall
 * ms of the call. ingest_stats2 adds: out[4] bytes that were copied through the engines' pinned slots (a host memcpy each), out[5]
 * bytes the DMA engine read from the caller's own pages (pinned arena, registered buffers, or pages registered for the call),
 * out[6] host ms inside hipHostRegister, out[7] NUMA node the ingest threads were bound to (-1: none / unknown). */
void lilliput_hip_batch_ingest_stats(lilliput_hip_batch b, double out[4]);
void lilliput_hip_batch_ingest_stats2(lilliput_hip_batch b, double out[8]);

/* Where the caller's encoded bytes live (lilliput_amd/csrc/lp_hostmem.h). The reference decodes from the caller's []byte in place
 * (opencv.cpp:99-171); so does the batched path whenever the DMA engine can read those pages:
 *   - lilliput_hip_host_alloc / _free: a pinned, device-mapped arena on the NUMA node next to `device` (-1: the current one). A
 *     service that reads its network bytes into such an arena (Go: unsafe.Slice over the pointer) has nothing copied or registered
 *     per call.
 *   - lilliput_hip_host_register / _unregister: pin a long-lived buffer of the caller's once (a receive-buffer pool).
 *   - anything else is copied through the engines' pinned slots by the ingest threads (one host memcpy per byte) -- or, with
 *     LILLIPUT_HIP_INGEST=register, has its page range registered for the duration of the call (each distinct range once, however many
 *     items name it; sources below LILLIPUT_HIP_REGISTER_MIN bytes, default 64 KiB, ranges that share a page with a live registration
 *     and ranges the driver refuses are still copied). Per-call registration is opt-in: a first-time hipHostRegister of 4 MB costs
 *     0.3 - 2 ms on this driver, several times the memcpy it saves (profiles/r03_a_ingest.md).
 * LILLIPUT_HIP_INGEST = auto (default: pinned sources in place, the rest through the slots) | register | staged (everything through the
 * slots). Ingest threads and their pinned slots are placed on the NUMA node the device hangs off
 * (/sys/bus/pci/devices/<bdf>/numa_node); LILLIPUT_HIP_NUMA=0 switches that off.
 * Why it matters: host memcpy into pinned memory does not scale on a two-socket box (12 threads 299 GB/s, 96 threads 42 GB/s: eight
 * ranks' stagers together deliver a tenth of what eight links need), while sources in pinned memory cost the host nothing. */
void* lilliput_hip_host_alloc(size_t bytes, int device);
void lilliput_hip_host_free(void* p);
int lilliput_hip_host_register(void* p, size_t bytes);      /* LILLIPUT_OK, or LILLIPUT_ERR_DEVICE when the range cannot be pinned */
int lilliput_hip_host_unregister(void* p);
int lilliput_hip_host_is_pinned(const void* p, size_t bytes); /* 1: the DMA engine will read [p, p + bytes) in place */
int lilliput_hip_set_ingest_mode(const char* mode);         /* "auto" | "register" | "staged" for the transforms that start after the call (process-wide, like
                                                             * LILLIPUT_HIP_INGEST); returns the previous mode: 0 register, 1 staged, 2 auto */
/* Resident form (kernel-pipeline measurements, tests): upload parses the headers and moves the compressed bytes into HBM,
 * run executes every device stage (inputs resident), download copies the encoded results back.
 * One difference to lilliput_hip_batch_transform and the one-image ABI: a baseline stream that ends short of its blocks (a truncated
 * file) answers LILLIPUT_ERR_DECODING_FAILED here, because the second pass that decodes it the way libjpeg does (zero bits from the
 * end of data on, lp_prog_core.h) re-reads the SOURCE bytes, which the resident form no longer has after upload; transform keeps the
 * caller's buffers for the duration of the call and retries such items itself. */
int lilliput_hip_batch_upload(lilliput_hip_batch b, const lilliput_batch_item* items, size_t n);
int lilliput_hip_batch_upload2(lilliput_hip_batch b, const lilliput_batch_item* items, size_t n, int engines); /* engines: how many engines (streams) share the batch, 0 = default (LILLIPUT_HIP_STREAMS, 4) */
int lilliput_hip_batch_run(lilliput_hip_batch b, const lilliput_batch_options* opt);
int lilliput_hip_batch_download(lilliput_hip_batch b, lilliput_batch_item* items, size_t n);
/* One process, several GPUs (what a Go service links: cgo cannot be one process per GPU). The devices of a node share one chunk queue in
 * host memory -- an atomic counter claimed chunk by chunk by every engine of every device -- so the batch is sharded dynamically and a
 * device that finishes early takes more chunks; no pixel or bitstream byte crosses between devices. devices == NULL: every visible GPU.
 * The same device may be listed more than once (two engine sets on one GPU; used by the tests on a one-GPU box).
 * (Across PROCESSES -- torchrun, one rank per GPU -- lilliput_amd/dist.py keeps the queue state in step with one small RCCL all-gather.) */
typedef void* lilliput_hip_node;
lilliput_hip_node lilliput_hip_node_create(const int* devices, int n_devices);
void lilliput_hip_node_destroy(lilliput_hip_node n);
int lilliput_hip_node_device_count(lilliput_hip_node n);
int lilliput_hip_node_transform(lilliput_hip_node n, lilliput_batch_item* items, size_t n_items, const lilliput_batch_options* opt); /* like lilliput_hip_batch_transform */
void lilliput_hip_node_device_stats(lilliput_hip_node n, int k, double out[2]); /* device k in the last transform: images served, bytes staged */
void lilliput_hip_node_queue_stats(lilliput_hip_node n, double out[2]); /* the last transform: chunks in the queue, chunks a device took from another device's share.
 * The chunk list is dealt out in contiguous shares, device k (the k-th entry of `devices`) first claims from the k-th share -- keep the k-th part of a
 * batch's sources in lilliput_hip_host_alloc(.., device k) memory and every DMA read stays on the near NUMA node -- and steals from the fullest share
 * once its own is dry (static block assignment, then work stealing for the tail / for heterogeneous sizes). */

/* Per-stage device milliseconds of the last run (HIP events on the engine's stream): unstuff, huffman (total), idct,
 * colour, resize, encode, then the huffman breakdown: speculate, verify, scan, write; plus the verify rounds. */
void lilliput_hip_batch_timings(lilliput_hip_batch b, float out_ms[10], int* verify_rounds);
/* Decoder tuning: subsequence bits / checkpoint spacing (0 = automatic). */
void lilliput_hip_batch_set_subsequence(lilliput_hip_batch b, unsigned S, unsigned C);
/* Images (of at most max_src_len encoded bytes) per launch of a resident run when options.chunk is 0: one full round of the entropy
 * decoder's workgroups on this device. LILLIPUT_HIP_RESIDENT_CHUNK overrides it. */
int lilliput_hip_batch_resident_round(lilliput_hip_batch b, size_t max_src_len);

/* Stage-level access for parity tests (device results copied to host). */
int lilliput_hip_decode_jpeg(lilliput_hip_batch b, const void* src, size_t len, void* dst, size_t cap, int* w, int* h, int* channels, int* orientation);
int lilliput_hip_decode_jpeg_coefs(lilliput_hip_batch b, const void* src, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh);
int lilliput_hip_decode_jpeg_plane(lilliput_hip_batch b, const void* src, size_t len, int comp, uint8_t* dst, size_t cap, int* pw, int* ph);
const char* lilliput_hip_last_error(void);
int lilliput_hip_device_count(void);

/* Process-wide effects of loading this library, and its resource bounds:
 *   - GPU_MAX_HW_QUEUES: the library's static initialiser sets it to 8 with setenv(.., overwrite = 0) BEFORE the HIP runtime reads it
 *     (first HIP call of the process), because with the runtime's default of 4 hardware queues one of a batch's four engines ends up
 *     behind the copy stream's barrier packets (8.9 k -> 10.6 k images/s end to end). It affects every HIP user in the process. A
 *     value the host application exported itself is left alone; LILLIPUT_HIP_KEEP_RUNTIME_ENV=1 makes the library touch nothing.
 *   - one-image ABI (Parts A, A2-A4, C): every call checks an engine (one stream + grow-only device arenas) out of a process-wide
 *     pool and returns it synchronised, so consecutive calls on one handle may come from any OS thread (cgo) and the number of
 *     engines follows the number of calls in flight, not the number of threads. At most LILLIPUT_HIP_ENGINE_POOL (default 64) idle
 *     engines holding at most LILLIPUT_HIP_ENGINE_POOL_MB (default 4096) of device arenas between them are kept, least recently used
 *     out first; an engine whose arenas grew beyond LILLIPUT_HIP_ENGINE_TRIM_MB (default 1024) is destroyed on return.
 *   - concurrent ImageOps.Transform calls (Part C) on static JPEG sources with JPEG output share batch launches once
 *     LILLIPUT_HIP_COALESCE (default 3) of them are in flight (lilliput_amd/csrc/lp_coalesce.h): LILLIPUT_HIP_COALESCE_WORKERS
 *     dispatcher threads (default 4) per device, each with a batch object that it gives back after LILLIPUT_HIP_COALESCE_IDLE_MS idle.
 * lilliput_hip_engine_pool_stats: engines checked out now, idle in the pool, created so far, destroyed by the bounds.
 * lilliput_hip_mem_info: hipMemGetInfo of `device` (LILLIPUT_OK or LILLIPUT_ERR_DEVICE). */
void lilliput_hip_engine_pool_stats(size_t out[4]);
int lilliput_hip_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* Guard-page debugging mode (LILLIPUT_HIP_GUARD=<alignment>, read once; lilliput_amd/csrc/lp_guard.h): every device and pinned buffer of
 * the library ends flush against an unmapped page, so one byte read or written past a buffer by a kernel or a DMA transfer is a GPU
 * memory fault on the spot (the reference never touches a byte outside the caller's buffer, opencv.cpp:99-124), and a canary band in
 * front of each buffer is checked when it is freed. out[0] = alignment in force (0: mode off), out[1] = guarded allocations so far,
 * out[2] = canary violations found, out[3] = peak of mapped device bytes. */
void lilliput_hip_guard_stats(size_t out[4]);
/* Stage profile of the one-image entry points (measurement access; bench.py's roofline of the PNG / WebP / animated workloads): while on,
 * every probed launch (PNG un-filter + expansion, GIF frame, composite, orientation, crop + resize, WebP Y'CbCr import, JPEG decode,
 * JPEG encode, PNG filter) is bracketed by two HIP events on the engine's stream and waited for. _profile(on) returns the previous
 * setting and clears the table when switching on; _read copies "name<TAB>calls<TAB>device ms<TAB>algorithmic bytes" lines into out and
 * returns the length of the whole text. */
int lilliput_hip_stage_profile(int on);
size_t lilliput_hip_stage_profile_read(char* out, size_t cap);

/* Test access (no device work): number of inflated image-data bytes of a PNG, or -1 when libpng would reject the file. */
int lilliput_hip_webp_yuv420(const opencv_mat src, uint8_t* y, uint8_t* u, uint8_t* v); /* test access: the planes the lossy WebP encoder is handed
                                                                                         * (w x h, then two of (w+1)/2 x (h+1)/2); 1 = translucent frame, -1 = error */
int lilliput_hip_bmp_decode(const void* data, size_t len, int* w, int* h, int* channels, uint8_t* out, size_t cap); /* test access: 0 decoded, 1 header refused, 2 data refused, -1 cap */
int lilliput_hip_pxm_decode(const void* data, size_t len, int* w, int* h, int* type, uint8_t* out, size_t cap);     /* the same for PBM / PGM / PPM (cv::PxMDecoder, opencv.cpp:99-171): *type = the decoder's Mat type, out = 8-bit pixels */
long lilliput_hip_png_inflate_check(const void* data, size_t len);
long lilliput_hip_png_inflate_bytes(const void* data, size_t len, uint8_t* out, size_t cap); /* test access: the filtered rows; -1 rejected, -2 cap too small */
int lilliput_hip_png_set_inflater(int own);   /* test access / A-B: 1 = the library's one-shot inflater first (default), 0 = zlib only; returns the previous setting */
/* Test access (no device work): the kernel the batch path hands a JPEG source of this shape to between its decoded planes and the output size:
 * 0 a materialised frame, 1 k_resample_420 (8 / 16 / 32-pixel boxes), 2 k_resample_420_small (2 / 4), 3 k_resample_hv1 (4:4:4 / 4:2:2), 4 k_resample_gray,
 * 5 the area walk with float taps (fractional scales), 6 the area walk with unit taps (other integer scales), 7 a wave per destination pixel. hs / vs = the luma
 * sampling factors (2,2 = 4:2:0; 2,1 = 4:2:2; 1,1 = 4:4:4 or grey). */
int lilliput_hip_resample_route(int width, int height, int orientation, int ncomp, int hs, int vs, int out_w, int out_h, int resize_method, int normalize_orientation);
uint32_t lilliput_hip_checksum(int which, uint32_t seed, const void* p, size_t n); /* test access: 0 = Adler-32, 1 = CRC-32 of the PNG path (zlib's conventions) */
int lilliput_hip_inflate_exact(const void* in, size_t in_len, uint8_t* out, size_t out_len); /* test access: 1 = ordinary zlib stream of exactly out_len bytes, decoded; 0 = ask zlib */
/* Test access, no device work: the per-pixel walk of the fused fractional INTER_AREA kernel (lp_area_core.h), run on the host over
 * decoded YCbCr planes the caller supplies (strides multiples of 4; sampling 2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4). orientation 1-8; crop in
 * oriented coordinates; out = dst_w * dst_h * 3 bytes BGR. 0 = done, 1 = this geometry keeps the frame route. */
int lilliput_hip_area420_host(const uint8_t* py, const uint8_t* pb, const uint8_t* pr, uint32_t stride_y, uint32_t stride_c, int w, int h, int sampling,
                              int orientation, int crop_x, int crop_y, int crop_w, int crop_h, int dst_w, int dst_h, uint8_t* out);

/* Lazy host write-back for Part A. Off (the default): every opencv_* call that produces pixels copies them into
 * the caller's buffer before it returns, as cv::Mat over Go memory does (opencv.go:258-267). On: pixels stay on
 * the device until opencv_mat_get_data or lilliput_hip_mat_sync_host asks for them -- safe for ImageOps.Transform
 * (ops.go:331-446), which only hands Mats back to this ABI. Also settable with LILLIPUT_HIP_LAZY_HOST=1. */
void lilliput_hip_set_lazy_host(int on);
int lilliput_hip_mat_sync_host(opencv_mat mat);   /* 0 = host pixels are current */

/* Deferred Part A (on by default; LILLIPUT_HIP_DEFER=0 or lilliput_hip_set_deferred(0): off). What unchanged ops.go does per image --
 * opencv_decoder_read_data, opencv_mat_orientation_transform, opencv_mat_crop, opencv_mat_resize, opencv_encoder_write (ops.go:352-444
 * through opencv.go:816-839, 271-279, 326-374, 872-900) -- is RECORDED for a baseline JPEG source instead of executed call by call:
 * every dimension is known from the header, and nothing needs bytes before opencv_encoder_write(".jpeg"), which hands {source,
 * orientation, crop, size, quality} to the batched path as one item, sharing launches with the calls other goroutines have in
 * flight. Anything else that touches such a Mat (opencv_mat_get_data, a PNG / WebP / GIF / ThumbHash encoder, a composite) runs the
 * recorded calls first, the eager way. Results are the batched path's: byte-identical to the eager route for integer scales, within
 * the +-1 LSB contract otherwise. What the caller must not do while it is on: read Framebuffer.buf behind the library's back (ops.go
 * never does; the same contract as lazy write-back), or modify the encoded source bytes between DecodeTo and the decoder's Close
 * (opencv_decoder_release copies the bytes if a recorded chain still needs them). Sources that can FAIL to decode (a stream that
 * runs out of bytes: scan-path files) are decoded in opencv_decoder_read_data as before, so that ErrDecodingFailed surfaces there.
 * lilliput_hip_deferred_stats: chains recorded, chains served by the batched path, chains run the eager way after all, sources copied
 * at decoder release -- process-wide counters since start. */
void lilliput_hip_set_deferred(int on);
void lilliput_hip_deferred_stats(uint64_t out[4]);
/* A recorded chain that reaches opencv_encoder_write while no other is being served (one goroutine, a quiet moment) runs on the caller's
 * own thread through the one-image route -- the Mat then holds real pixels on the device, like the reference's framebuffer -- instead of
 * travelling to a dispatcher thread as a batch of one (round 6; on by default, LILLIPUT_HIP_DEFER_INLINE=0 or ...(0): always the batched
 * path). Returns the previous setting. */
int lilliput_hip_set_deferred_inline(int on);

/* Progressive (SOF2) JPEG sources (libjpeg-turbo jdphuff.c behind opencv_decoder_read_data, opencv.cpp:166-171): where the scans' entropy
 * decode runs. mode -1 = auto (default): on the device -- one wave per scan, lilliput_amd/csrc/lp_kernels_prog.hip -- for the progressive
 * images of an upload set that holds at least LILLIPUT_HIP_PROG_DEVICE_MIN (default: five per usable host CPU) of them, on host threads (lp_prog_host.h; thread
 * count LILLIPUT_HIP_PROG_THREADS) otherwise; 0 = host threads always; 1 = device always; 2 = the generic one-lane-per-scan device kernel
 * (k_prog_scan, the wave decoder's tested reference). LILLIPUT_HIP_PROG_ENTROPY=auto|host|device|lanes sets the process default.
 * Same results in every mode: an image whose data the device decoders find irregular is decoded again by the host threads.
 * lilliput_hip_progressive_device_lanes_built(): 1 (the device decoders were a build option in rounds 3-5). */
void lilliput_hip_set_progressive_entropy(int mode);
int lilliput_hip_progressive_device_lanes_built(void);
/* Process-wide counters since start: out[0] scan-path images whose scans were decoded on the device, out[1] those of them the device decoders
 * gave up on (irregular data: decoded again by the host threads), out[2] scans launched on the device. */
void lilliput_hip_progressive_stats(uint64_t out[3]);
/* Launches of the baseline entropy decoder whose queued verify rounds (4 behind a large launch, 6 behind a small one: LILLIPUT_HIP_VERIFY_ROUNDS /
 * LILLIPUT_HIP_SMALL_ROUNDS) did not settle every subsequence's exit state: they continued under host control and ran the stages behind the
 * verification -- and, in the batched path, the chunk -- a second time. Process-wide, since load. A service whose sources make this number grow
 * with its request count should raise the variables (an idle round costs ~5 us per launch). */
uint64_t lilliput_hip_decode_redone_count(void);
/* Requests served on their caller's own thread as a resident batch of one (deferred Part A chains and Part C calls without company:
 * LILLIPUT_HIP_DEFER_INLINE, _INLINE_MAX), process-wide, since load -- the others went through the dispatchers' shared launches. */
uint64_t lilliput_hip_lone_batch_count(void);
/* Test access (no device work): component `comp` of a progressive JPEG as the host threads decode it, [block row][block column][64]
 * natural-order coefficients over the MCU-padded grid. 0 = ok, -1 = not an accepted progressive JPEG, -2 = restart-marker overflow,
 * -3 = dst too small. nthreads 0 = default. */
/* Test access (no device work): the progressive JPEG writer behind opencv_encoder_write(..., CV_IMWRITE_JPEG_PROGRESSIVE, 1) on
 * quantised coefficients in the encoder's own layout (MCU order -- 4:2:0: Y00 Y01 Y10 Y11 Cb Cr --, 64 zigzag-order values per block,
 * dummy blocks included). Returns the byte count, 0 on failure, or minus the count needed when cap is too small. */
long lilliput_hip_progressive_encode_coefs(int width, int height, int ncomp, int quality, const int16_t* coef, uint8_t* out, size_t cap);
int lilliput_hip_progressive_coefs_host(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh, int nthreads);
/* ... and as they reach the IDCT: behind libjpeg's interblock smoothing for the files that call for it (below); the same values otherwise */
int lilliput_hip_progressive_coefs_smoothed(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh);
/* Test access: 1 when the reference's libjpeg applies its interblock smoothing to this progressive file (jdcoefct.c smoothing_ok: the scan
 * script leaves one of the first nine AC coefficients of a component short of full precision); the library applies the same filter to
 * such a file's coefficients on the host, in front of the IDCT (lp_prog_smooth); 0 otherwise; -1: not a JPEG the parser takes. */
int lilliput_hip_jpeg_reference_smooths(const void* data, size_t len);

/* ------------------------------------------------------------------------------------------------
 * Part C -- host mirror of the Go API (ops.go / opencv.go / lilliput.go)
 * ---------------------------------------------------------------------------------------------- */
typedef void* lilliput_decoder;
typedef void* lilliput_image_ops;

typedef struct lilliput_image_options {     /* ops.go:26-65 */
    const char* file_type;                  /* ".jpeg" */
    int width, height;
    int resize_method;
    int normalize_orientation;
    const int* encode_options;              /* flattened map[int]int: key, value, key, value ... */
    size_t encode_options_len;              /* number of ints */
    int max_encode_frames;
    int64_t max_encode_duration_ns;
    int64_t encode_timeout_ns;
    int disable_animated_output;
    int force_sdr;
} lilliput_image_options;

int lilliput_new_decoder(const void* buf, size_t len, lilliput_decoder* out);           /* lilliput.go:129-164 */
void lilliput_decoder_close(lilliput_decoder d);
int lilliput_decoder_header(lilliput_decoder d, int* width, int* height, int* pixel_type, int* orientation, int* num_frames, int* content_length);
const char* lilliput_decoder_description(lilliput_decoder d);
int lilliput_decoder_icc(lilliput_decoder d, void* dst, size_t cap);                    /* opencv.go:697-712 */
int lilliput_decoder_animation_info(lilliput_decoder d, int out[4]);                    /* giflib.go:126-178: loops, frames, duration ms, background ARGB */
lilliput_image_ops lilliput_new_image_ops(int max_size);                                /* ops.go:83-91 */
void lilliput_image_ops_close(lilliput_image_ops o);
void lilliput_image_ops_clear(lilliput_image_ops o);
/* ops.go:352-444: returns LILLIPUT_*; *dst_len = length of the encoded image inside dst. */
int lilliput_image_ops_transform(lilliput_image_ops o, lilliput_decoder d, const lilliput_image_options* opt, void* dst, size_t dst_cap, size_t* dst_len);
/* Pure control logic exposed for parity tests. */
void lilliput_calculate_expected_size(int orig_w, int orig_h, int req_w, int req_h, int* out_w, int* out_h);  /* ops.go:243-255 */
void lilliput_fit_crop_rect(int fw, int fh, int width, int height, int* left, int* top, int* w, int* h);      /* opencv.go:331-363 */
int lilliput_detect_content_length(const void* buf, size_t len);                                              /* opencv.go:604-614 */
int lilliput_detect_apng(const void* buf, size_t len);                                                        /* opencv.go:617-637 */

#ifdef __cplusplus
}
#endif
#endif
