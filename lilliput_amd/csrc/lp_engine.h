// lp_engine.h -- host orchestration of the device pipeline (one engine per ImageOps / per batch worker).
//
// The engine owns a HIP stream, grow-only device arenas and pinned staging, and exposes the stages of
// lilliput's ImageOps.Transform (/root/reference/ops.go:352-444) as batched device operations:
//   decode_jpegs  : opencv_decoder_read_data        (opencv.cpp:166-171)
//   orient        : opencv_mat_orientation_transform (opencv.cpp:217-221)
//   resize        : opencv_mat_crop + opencv_mat_resize (opencv.cpp:196-215)
//   encode_jpegs  : opencv_encoder_write            (opencv.cpp:185-194)
// Frames are addressed by absolute device pointers (LpFrame.off), so they may live in the engine's heap
// or in a Mat's own device mirror.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "lp_guard.h"
#include "lp_hostmem.h"
#include "lp_jpeg_parse.h"
#include "lp_prog_host.h"
#include "lp_launch.h"
#include "lp_types.h"
#include "lp_area_core.h"

// The stream priority the engines this thread constructs next ask for: 0 default, 1 the device's highest. set < 0: read only. Returns the value before.
int lp_engine_stream_priority_hint(int set);

// Frees the blocks that growing arenas have retired (see LpDevBuf::ensure in lp_engine.cpp); called at the end of a batch / node call.
void lp_retired_collect();

struct LpDevBuf {
    void* p = nullptr;
    size_t cap = 0;
    const char* tag = "arena";  // names the buffer in the guard mode's log (lp_guard.h)
    ~LpDevBuf();
    // grow-only (x 1.5 at least); contents are NOT preserved across growth; the old block is retired, not freed (lp_retired_collect)
    bool ensure(size_t bytes);
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct LpPinned {
    void* p = nullptr;
    void* dev = nullptr;        // the buffer as kernels address it (mapped pinned memory)
    size_t cap = 0;
    const char* tag = "pinned";
    ~LpPinned();
    bool ensure(size_t bytes);
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum LpStatus {
    LP_OK = 0,
    LP_ERR_INVALID_IMAGE = 1,   // lilliput.ErrInvalidImage
    LP_ERR_DECODE_FAILED = 2,   // lilliput.ErrDecodingFailed
    LP_ERR_BUF_TOO_SMALL = 3,   // lilliput.ErrBufTooSmall
    LP_ERR_UNSUPPORTED = 4,     // stream feature outside the device path (progressive, CMYK, ...)
    LP_ERR_DEVICE = 5,          // HIP failure / no device
    LP_RETRY = 100              // internal: see LpEngine::finish_decode
};

struct LpJpegSrc {
    const uint8_t* data;
    size_t len;
};

struct LpResizeReq {
    LpFrame src;                // full source frame
    uint32_t crop_x, crop_y, crop_w, crop_h;
    uint32_t dst_w, dst_h;
};

// One image of area_resample: the oriented crop (crop_w x crop_h) -> dst (BGR; off = where to write, w / h = the size asked for).
// Orientations 1-4: index si along the crop's x axis is source column x0 + xstep * si, along its y axis source row y0 + ystep * si.
// Orientations 5-8 (transposed): index si along the crop's Y axis is source column x0 + xstep * si, along its X axis source row
// y0 + ystep * si.
struct LpAreaReq {
    uint32_t img;
    int32_t x0, y0, xstep, ystep;
    uint32_t crop_w, crop_h;
    uint32_t transposed;
    uint32_t int_x, int_y;       // 0: fractional scale (cv::resize's resizeArea_); otherwise the integer scale factors of crop -> dst along the oriented
                                 // frame's x and y (resizeAreaFast_: unit taps, sum * 1 / (int_x * int_y), or (sum + 2) >> 2 for 2 x 2)
    LpFrame dst;
};

struct LpEncodeReq {
    LpFrame src;
    int quality;
    size_t out_cap;
};

struct LpTimings {
    float unstuff_ms, huff_ms, idct_ms, color_ms, resize_ms, encode_ms;
    uint32_t verify_rounds;
    float huff_spec_ms, huff_verify_ms, huff_scan_ms, huff_write_ms; // breakdown of huff_ms
};

// One staged set of sources: their descriptors, and where their entropy-coded bytes sit (pinned staging buffer, device arena).
// An engine owns LP_UPLOAD_SLOTS of them so that a batch can stage set n + 1 (memcpy into pinned memory on a host thread, H2D on
// the engine's copy stream) while set n is being decoded on the compute stream -- the ingest half of ImageOps.Transform
// (/root/reference/opencv.cpp:99-171: opencv_decoder_create / read_header / read_data start from host bytes).
#define LP_UPLOAD_SLOTS 4
struct LpUpload {
    struct ProgScanUp { LpProgScan s; uint64_t raw_off; uint32_t raw_len; uint32_t level; };
    // An entropy-coded segment: 16-byte aligned in the arena. direct: the caller's bytes are pinned (lp_hostmem.h) and the DMA engine reads
    // them where they lie; else they pass through the slot's pinned buffer at stage_off (followed by 32 zero bytes there).
    // dev_delta: what to add to src to address the same byte from a kernel (0 under unified addressing).
    // img / scan: whose segment it is (scan < 0: the image's only one), so that a re-layout can move it; pin_base: start of the pinned range
    // the bytes lie in (0 = not pinned).
    struct Piece { size_t arena_off; const uint8_t* src; size_t len; const uint8_t* item; size_t item_len; size_t stage_off; ptrdiff_t dev_delta; bool direct; int img, scan; uintptr_t pin_base; };
    std::vector<LpJpeg> src;                        // every image of the set (raw layout only)
    std::vector<LpHuffSet> huffs;
    std::vector<Piece> pieces;
    size_t raw_bytes = 0;
    size_t stage_bytes = 0;                         // extent of the staged segments in the slot's pinned buffer (with their padding)
    size_t copied_bytes = 0, direct_bytes = 0;      // entropy-coded bytes that travel through the pinned slot / straight from the caller's pages
    LpPinScope pins;                                // the page ranges registered for this set (released when the slot is reused)
    bool staged_whole = false;                      // the pinned buffer holds the whole set (upload_copy / upload_commit); else windowed
    // progressive images (SOF2) and the other scan-by-scan cases
    int prog_mode = 0;                              // how the set's device-side scans are decoded: 1 a wave per progressive scan, 2 lanes (lp_prog_host.h)
    std::vector<uint8_t> prog_dev;                  // per image: 1 = its scans are entropy-decoded on the device, 0 = on host threads
    bool any_prog_dev = false;
    uint32_t prog_in_call = 0;                      // see LpEngine::set_progressive_in_call
    bool force_host_scans = false;                  // the next layout keeps every scan on host threads (an image the device gave up on)
    LpPinned pcoef;                                 // host mode: the decoded int16 coefficients of every scan-path image of the set
    std::vector<size_t> pcoef_off;                  // element offset per image
    std::vector<uint32_t> perr;                     // per image
    std::vector<std::vector<ProgScanUp>> prog;      // per image; empty for a baseline one
    std::vector<LpProgHuff> phuffs;
    std::vector<LpProgHostTask> host_tasks;
    size_t pcoef_total = 0;
    LpDevBuf d_raw, d_huffs, d_phuffs;
    LpPinned stage;
    hipEvent_t ready = nullptr;                     // recorded behind the set's H2D copies (first copy queue)
    hipEvent_t ready_x[3] = {nullptr, nullptr, nullptr}; // ... and behind those that went to the other copy queues
    int ready_n = 0;                                // how many of ready_x the current set uses
};

class LpEngine {
public:
    explicit LpEngine(int device);
    ~LpEngine();
    bool ok() const { return ok_; }
    const std::string& last_error() const { return err_ref(); }  // of the calling thread: the stager and the compute thread of a batch part share an engine
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }

    // Subsequence size in bits (multiple of 32, 64..32768) and checkpoint spacing in bits (sets the schedule); 0 = automatic.
    void set_subsequence(uint32_t S, uint32_t C) { S_cfg_ = S; C_cfg_ = C; }
    // Progressive sources of the CALL this engine's next upload sets belong to (the batch front end counts them over all its chunks): the
    // automatic host / device choice of a set looks at the larger of this and the set's own count -- the device wins by the number of scan
    // chains in flight, and the chunks of one call are in flight together.
    void set_progressive_in_call(uint32_t n) { for (LpUpload& u : up_) u.prog_in_call = n; }

    // Heap for intermediate frames: bump-allocated, reset per batch.
    void heap_reset() { heap_used_ = 0; }
    bool heap_reserve(size_t bytes);             // make the heap at least this large (invalidates contents)
    uint8_t* heap_alloc(size_t bytes);           // nullptr when the reservation is exhausted

    // Parse + upload + decode n JPEGs into frames (BGR / gray, tightly packed) placed at dst[i].off.
    // dst[i].off must be preset by the caller (device pointers with room for w*h*cn); status[i] per image.
    // If dst == nullptr the frames are bump-allocated from the heap and returned in out_frames.
    int decode_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs, LpFrame* frames, int* status);
    // Device-resident variant: the entropy-coded bytes are uploaded first (slot 0, through a pinned window, synchronous) ...
    int upload_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs);
    // ... or staged: layout (descriptors, arena offsets; sizes the slot's pinned buffer and device arena), copy (memcpy of pieces
    // [p0, p1) into the pinned buffer -- any host thread), commit (scan-path host decode, asynchronous H2D on the copy stream, event).
    // The sources must stay valid until commit returns. select_upload makes a slot the one decode_uploaded reads; the compute
    // stream waits for the slot's event, not the host.
    int upload_layout(int slot, const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs);
    size_t upload_pieces(int slot) const { return up_[slot].pieces.size(); }
    size_t upload_bytes(int slot) const { return up_[slot].copied_bytes + up_[slot].direct_bytes; }   // entropy-coded bytes of the set
    size_t upload_staged_bytes(int slot) const { return up_[slot].copied_bytes; }   // of them: copied through the pinned slot
    size_t upload_direct_bytes(int slot) const { return up_[slot].direct_bytes; }   // read by the DMA engine from the caller's own pages
    double upload_register_ms(int slot) const { return up_[slot].pins.register_ms(); }
    void upload_release_pins();                      // after the last set's copies have completed: drop every temporary registration
    void upload_copy(int slot, size_t p0, size_t p1);
    // on: a copy stream shared by several engines (sets arrive in enqueue order); default: the engine's own. extra / n_extra: further copy
    // queues for the segments that are read from the caller's pages -- many 4 MB copies on ONE queue reach 49.8 GB/s, spread over four
    // queues 56.3 (one 135 MB copy: 57.4; scripts/ingest_micro.hip, profiles/r03_b)
    int upload_commit(int slot, hipStream_t on = nullptr, const hipStream_t* extra = nullptr, int n_extra = 0);
    void select_upload(int slot) { u_ = &up_[slot]; }
    // want_frame (optional, per image): 0 = keep only the planes (the image will go through fused_resample)
    int decode_uploaded(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame = nullptr);
    // The same in two halves, so that the caller can enqueue the stages that follow (fused_resample, encode) without a host round
    // trip in between: decode_begin enqueues, finish_decode waits and reports. LP_RETRY from finish_decode = the verification needed
    // more rounds than were enqueued and the tail of the decode ran again: whatever was enqueued behind decode_begin must be redone.
    int decode_begin(int first, int n, LpFrame* frames, const uint8_t* want_frame = nullptr);
    int finish_decode(int* status);
    float fused_resample_ms();                   // device time of the last fused_resample (waits for it)
    // planes of the current decode range -> thumbnails, see LpFusedOp
    int fused_resample(const LpFusedOp* ops, int n);
    // planes of the current decode range -> fractional INTER_AREA thumbnails (lp_area_core.h); enqueued like fused_resample and, when
    // both are used, AFTER it (fused_resample_ms then covers both). Every request must have passed lp_area420_bucket.
    int area_resample(const LpAreaReq* reqs, int n, bool after_fused = false);
    size_t uploaded_count() const { return u_->src.size(); }
    const LpJpeg& uploaded(size_t i) const { return u_->src[i]; }
    // stage-level read-back for parity tests (valid after a decode of the current range)
    int copy_coefs(int i, int comp, int16_t* dst, size_t cap_elems);
    // image i of the last collected decode: a scan-path image whose scans ran on the device and met something only the host route decodes
    // the way libjpeg does (lp_kernels_prog.hip "irregular", a marker inside a scan, restart markers out of turn): decode it again with
    // LpUpload::force_host_scans (decode_jpegs does; the batch path sends it through its one-image retry)
    bool scan_gave_up(int i) const { return (size_t)i < h_states_.size() && (h_states_[(size_t)i].error & 64u) != 0; }
    int copy_plane(int i, int comp, uint8_t* dst, size_t cap);
    const LpJpeg& decoded(int i) const { return h_imgs_[(size_t)i]; }

    int orient(const LpOrientOp* ops, int n);
    int resize(const LpResizeReq* reqs, int n, LpFrame* dsts /* off preset */, int* status);
    // Encodes into the engine's output arena; results are fetched with encoded_copy().
    int encode_jpegs(const LpEncodeReq* reqs, int n, int* status, uint32_t* out_len);
    // Progressive (SOF2) output of one frame: FDCT + quantisation on the device, multi-scan entropy coding on the host (lp_jpeg_progenc.h).
    int encode_jpeg_progressive(const LpEncodeReq& req, std::vector<uint8_t>& out);
    // The same for n frames; the host entropy coding runs on a few threads. outs[i] stays empty where status[i] != 0.
    int encode_jpegs_progressive(const LpEncodeReq* reqs, int n, int* status, std::vector<std::vector<uint8_t>>& outs);
    int encoded_copy(int i, uint8_t* dst, size_t cap);   // D2H of job i's bytes (after encode_jpegs)
    const uint8_t* encoded_device_ptr(int i) const;
    int encoded_fetch_all();                              // D2H of every job's bytes into pinned memory (one sync)
    const uint8_t* encoded_host(int i) const;

    int composite(const LpCompositeOp& op);
    // Renders one GIF frame: uploads the indices and the palette (256 x BGRA) and runs k_gif_frame; op.index_off / palette_off are filled here.
    int gif_frame(LpGifFrameOp op, const uint8_t* indices, size_t n_indices, const uint8_t* palette_bgra);
    // PNG: uploads the inflated stream and the palette, reverses the filters and expands to op.dst; LP_ERR_DECODE_FAILED on a bad filter type.
    int png_decode(LpPngOp op, const uint8_t* filtered, size_t n, const uint8_t* palette_bgra);
    // PNG output: libpng's per-row filter choice + filtering on the device; `out` receives h * (1 + w * cn) bytes (host memory).
    int png_filter(const LpFrame& src, uint32_t filters, uint8_t* out);
    // HDR -> SDR tone map in place (color_info.cpp:206-236 tonemap_rgb_8u_inplace): 3- or 4-channel 8-bit frame, cICP code points
    // images of at most max_raw_len entropy-coded bytes whose WRITE workgroups fill the device exactly once
    size_t resident_round(size_t max_raw_len);
    int tonemap(const LpFrame& f, int transfer, int primaries, const uint16_t* d_src16 = nullptr, int depth = 8);
    // the same for host buffers (the reference's own signatures): tightly packed 8-bit pixels in place; 16-bit samples -> 8-bit
    int tonemap_host8(uint8_t* pixels, int w, int h, int cn, int transfer, int primaries);
    int tonemap_host(const uint16_t* src, uint8_t* dst, int w, int h, int depth, int transfer, int primaries);
    // ThumbHash: out[(i * w + j) * cn ..] = frame(idx[w + i], idx[j]) for a w x h lattice of sample coordinates (host memory in and out).
    int gather_samples(const LpFrame& f, const uint32_t* idx, uint32_t w, uint32_t h, uint8_t* out);
    // lossy WebP front end: the frame as Y (w x h), U, V ((w + 1) / 2 x (h + 1) / 2) planes into host memory; *translucent = some alpha below 255
    int webp_yuv420(const LpFrame& f, const LpWebpYuvTab& tab, uint8_t* y, uint8_t* u, uint8_t* v, bool* translucent);
    int sync();
    // transfers between this engine's stream and host memory that may be pageable, through its pinned buffers (see h2d_any / d2h_begin)
    bool upload_any(void* dst, const void* src, size_t bytes) { return h2d_any(dst, src, bytes); }
    const uint8_t* download_begin(const void* dev, size_t bytes) { return d2h_begin(dev, bytes); }
    size_t device_bytes() const;                 // HBM held by this engine's grow-only arenas (the pool of the one-image ABI trims by it)
    const LpTimings& timings() const { return tm_; }
    void enable_timing(bool on) { timing_ = on; }
    // Stage timestamps (the durations bench.py's exclusive leg reads). LILLIPUT_HIP_TIMING_SYNC=1 drains the stream before every
    // timestamp, so that a bracket holds exactly the kernels launched inside it and nothing of its neighbours' tails (A/B against the
    // rocprofv3 kernel trace: profiles/r04_f_event_brackets.md).
    void mark(int i);
    void set_timings(const LpTimings& t) { tm_ = t; }

private:
    bool check(hipError_t e, const char* what);
    bool h2d_small(void* dst, const void* src, size_t bytes);
    // Several small uploads, downloads and zero fills as ONE launch on the compute stream: add() them, then flush() before the first kernel
    // that reads any of them (a full batch flushes itself). Copies keep h2d_small's contract (16-byte aligned destination, size rounded up
    // to 16 inside padded arenas); zero fills are exact.
    struct SmallBatch { LpSmallOps ops; uint32_t n = 0; };
    bool small_copy(SmallBatch& sb, void* dst, const void* src, size_t bytes);
    bool small_zero(SmallBatch& sb, void* dst, size_t bytes);
    bool small_d2h(SmallBatch& sb, const LpPinned& pin, void* host, const void* dev, size_t bytes);
    bool small_flush(SmallBatch& sb);
    // Host -> device from memory that may be pageable, without the runtime's own pageable path (a staged, host-blocking copy behind a
    // process-wide lock: 0.6 ms per call with eight callers, HIP API trace of round 4): small blocks through the descriptor ring
    // (h2d_small), large ones through this engine's pinned transfer buffer. dst needs room for bytes rounded up to 16.
    bool h2d_any(void* dst, const void* src, size_t bytes, bool dst_has_slack = false);
    // Device -> pinned transfer buffer (asynchronous); the bytes are at the returned host pointer once the stream has been waited for.
    const uint8_t* d2h_begin(const void* dev, size_t bytes, size_t slot_off = 0);
    void d2h_small(const LpPinned& pin, void* host, const void* dev, size_t bytes);
    int run_decode(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame, bool defer);

    int device_ = 0;
    bool ok_ = false;
    bool timing_ = false;
    static std::string& err_ref();
    hipStream_t stream_ = nullptr, copy_stream_ = nullptr;
    hipEvent_t ev_[16] = {};
    uint32_t S_cfg_ = 0, C_cfg_ = 0;
    LpTimings tm_ = {};

    // decode state for the current batch
    LpUpload up_[LP_UPLOAD_SLOTS];
    LpUpload* u_ = &up_[0];       // the set decode_uploaded reads
    std::vector<LpJpeg> h_imgs_;  // the range being decoded (working arenas laid out)
    std::vector<LpJpegState> h_states_;
    uint32_t S_ = 0, K_ = 0;
    LpCkSched sched_ = {};
    uint32_t max_chunks_ = 0, max_sub_ = 0, max_bw_ = 0, max_rows_ = 0, max_w_ = 0, max_h_ = 0, max_mcus_ = 0;
    uint32_t tot_sub_ = 0, tot_chunks_ = 0, tot_rst_ = 0;
    LpDevBuf d_imgs_, d_states_, d_clean_, d_rst_, d_chunk_, d_ckpt_, d_exit_, d_spec_exit_, d_entry_, d_tot_, d_spec_tot_, d_prefix_, d_changed_;
    LpDevBuf d_coef_, d_wide_, d_wide_id_, d_dc_, d_dcpart_, d_planes_, d_frames_desc_;
    LpPinned h_small_, h_out_, h_dstate_, h_desc_, h_xfer_in_, h_xfer_out_;
    size_t desc_used_ = 0;
    std::vector<uint32_t> h_pk_;                                    // slot offsets of the encoded streams in h_out_ (n + 1 entries)
    std::vector<std::pair<size_t, std::vector<uint8_t>>> h_big_;    // streams that outgrew their slot
    std::vector<LpFusedOp> h_fops_;
    bool fused_timed_ = false;
    struct Pending { bool active; int first, n; size_t nstreams, pcoef_elems; bool any_baseline, any_frame, any_generic, any_420; LpFrame* frames; LpHuffArgs ha; uint32_t vr; };
    uint32_t vr_ = LP_VERIFY_ROUNDS;                // verify rounds queued behind the speculative pass of the current launch
    Pending pend_ = {};
    std::vector<size_t> h_out_off_;

    // progressive images (SOF2): their scans of the current decode range (the per-set part lives in LpUpload)
    std::vector<LpProgScan> h_pscans_;              // the current range's scans, sorted by dependency level
    std::vector<uint32_t> h_plevel_first_;          // level l = h_pscans_[h_plevel_first_[l] .. h_plevel_first_[l + 1])
    std::vector<LpJpeg> h_pstreams_;                // one pseudo stream per scan (what the unstuff kernels need)
    std::vector<LpJpegState> h_pstates_;
    LpDevBuf d_pscans_, d_pstreams_, d_pstates_, d_pcoef_, d_pdeps_, d_pprog_;
    std::vector<LpProgDep> h_pdeps_;                // non-empty: the range's scans run as ONE pipelined launch (lp_kernels_prog.hip)

    // frame heap
    LpDevBuf heap_;
    size_t heap_used_ = 0;

    // resize / orient
    LpDevBuf d_ops_, d_taps_, d_ranges_, d_fops_;
    LpDevBuf d_aops_, d_ataps_, d_aranges_;     // area_resample: ops, tap and range arenas (their own: the kernel is not waited for)
    std::vector<LpArea420Op> h_aops_;
    std::vector<LpTap> h_ataps_;
    std::vector<uint32_t> h_aranges_;
    LpDevBuf d_tone_;           // staging for the host-pointer tone-map entry points
    // encode
    std::vector<LpEncJob> h_jobs_;
    LpDevBuf d_jobs_, d_estates_, d_ecoef_, d_blkbits_, d_bits_, d_hdrs_, d_out_, d_packed_, d_pkoff_;
    std::vector<LpEncState> h_estates_;
    bool enc_tables_ready_ = false;
    bool enc_fdct_only_ = false;
};

// JFIF header (SOI..SOS) + quantisation tables exactly as cv::JpegEncoder/libjpeg-turbo write them.
size_t lp_build_jpeg_header(int W, int H, int ncomp, int quality, uint8_t* out /* >= 700 */, uint16_t qt_nat[2][64]);
void lp_encode_init_tables();

// cv::resize(INTER_AREA) dispatch arithmetic (mode, integer scales) -- shared with tests.
int lp_resize_mode(int sw, int sh, int dw, int dh, int* iscale_x, int* iscale_y);
int lp_area_tab(int ssize, int dsize, std::vector<LpTap>& taps, std::vector<uint32_t>& ranges);
// k_area_420's instantiation for an x axis ssize -> dsize (6 / 10 / 18 / 34 / 66 taps), 0 = none (more taps, or a table whose
// columns are not runs of consecutive source columns)
// (the axis meant is the one that runs along source x: the crop's x axis, or its y axis for the transposing orientations, whose
// kernel stops at 34 taps)
uint32_t lp_area420_bucket(int ssize, int dsize, bool transposed = false);
// ... of an integer scale: the taps of a destination pixel are the `scale` columns of its box
uint32_t lp_area420_bucket_int(int scale, bool transposed = false);
// Would fused_resample hand this op to one of the thread-per-box kernels (k_resample_420 / _small / _hv1)? Otherwise its wave-per-pixel
// kernel takes it, which the batch path avoids where the area walk can do the box instead.
bool lp_fused_op_is_fast(const LpFusedOp& op, const LpJpeg& j, uint32_t* fast = nullptr, uint32_t* mask_bit = nullptr);
// which chroma layout of k_area_420 an image has: 2 = YCbCr 4:2:0, 1 = 4:2:2, 0 = 4:4:4, -1 = none of them (frame route)
int lp_area_sampling(const LpJpeg& j);
// fills x0 / xstep / y0 / ystep / transposed of a request from the EXIF orientation (cv::ExifTransform's inverse), the decoded size and
// the crop origin in oriented coordinates
void lp_area420_place(int orientation, int w, int h, int crop_x, int crop_y, LpAreaReq* rq);
