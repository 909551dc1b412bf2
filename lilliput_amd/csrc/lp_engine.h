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

#include "lp_jpeg_parse.h"
#include "lp_types.h"

struct LpDevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~LpDevBuf();
    // grow-only; contents are NOT preserved across growth
    bool ensure(size_t bytes);
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct LpPinned {
    void* p = nullptr;
    size_t cap = 0;
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
    LP_ERR_DEVICE = 5           // HIP failure / no device
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

class LpEngine {
public:
    explicit LpEngine(int device);
    ~LpEngine();
    bool ok() const { return ok_; }
    const std::string& last_error() const { return err_; }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }

    // Subsequence size in bits (multiple of 32, 64..32768) and checkpoint spacing in bits (sets the schedule); 0 = automatic.
    void set_subsequence(uint32_t S, uint32_t C) { S_cfg_ = S; C_cfg_ = C; }

    // Heap for intermediate frames: bump-allocated, reset per batch.
    void heap_reset() { heap_used_ = 0; }
    bool heap_reserve(size_t bytes);             // make the heap at least this large (invalidates contents)
    uint8_t* heap_alloc(size_t bytes);           // nullptr when the reservation is exhausted

    // Parse + upload + decode n JPEGs into frames (BGR / gray, tightly packed) placed at dst[i].off.
    // dst[i].off must be preset by the caller (device pointers with room for w*h*cn); status[i] per image.
    // If dst == nullptr the frames are bump-allocated from the heap and returned in out_frames.
    int decode_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs, LpFrame* frames, int* status);
    // Device-resident variant used by the bench: ECS bytes were uploaded earlier with upload_jpegs().
    int upload_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs);
    // want_frame (optional, per image): 0 = keep only the planes (the image will go through fused_resample)
    int decode_uploaded(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame = nullptr);
    // planes of the current decode range -> thumbnails, see LpFusedOp
    int fused_resample(const LpFusedOp* ops, int n);
    size_t uploaded_count() const { return h_src_.size(); }
    const LpJpeg& uploaded(size_t i) const { return h_src_[i]; }
    // stage-level read-back for parity tests (valid after a decode of the current range)
    int copy_coefs(int i, int comp, int16_t* dst, size_t cap_elems);
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
    const uint8_t* encoded_host(int i) const { return h_out_.as<uint8_t>() + h_out_off_[(size_t)i]; }

    int composite(const LpCompositeOp& op);
    // Renders one GIF frame: uploads the indices and the palette (256 x BGRA) and runs k_gif_frame; op.index_off / palette_off are filled here.
    int gif_frame(LpGifFrameOp op, const uint8_t* indices, size_t n_indices, const uint8_t* palette_bgra);
    // PNG: uploads the inflated stream and the palette, reverses the filters and expands to op.dst; LP_ERR_DECODE_FAILED on a bad filter type.
    int png_decode(LpPngOp op, const uint8_t* filtered, size_t n, const uint8_t* palette_bgra);
    // PNG output: libpng's per-row filter choice + filtering on the device; `out` receives h * (1 + w * cn) bytes (host memory).
    int png_filter(const LpFrame& src, uint32_t filters, uint8_t* out);
    // ThumbHash: out[(i * w + j) * cn ..] = frame(idx[w + i], idx[j]) for a w x h lattice of sample coordinates (host memory in and out).
    int gather_samples(const LpFrame& f, const uint32_t* idx, uint32_t w, uint32_t h, uint8_t* out);
    int sync();
    const LpTimings& timings() const { return tm_; }
    void enable_timing(bool on) { timing_ = on; }
    void set_timings(const LpTimings& t) { tm_ = t; }

private:
    bool check(hipError_t e, const char* what);
    int run_decode(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame);

    int device_ = 0;
    bool ok_ = false;
    bool timing_ = false;
    std::string err_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_[16] = {};
    uint32_t S_cfg_ = 0, C_cfg_ = 0;
    LpTimings tm_ = {};

    // decode state for the current batch
    std::vector<LpJpeg> h_src_;   // every uploaded image (raw layout only)
    std::vector<LpJpeg> h_imgs_;  // the range being decoded (working arenas laid out)
    std::vector<LpHuffSet> h_huffs_;
    std::vector<LpJpegState> h_states_;
    uint32_t S_ = 0, K_ = 0;
    LpCkSched sched_ = {};
    uint32_t max_chunks_ = 0, max_sub_ = 0, max_bw_ = 0, max_rows_ = 0, max_w_ = 0, max_h_ = 0;
    uint32_t tot_sub_ = 0, tot_chunks_ = 0, tot_rst_ = 0;
    LpDevBuf d_imgs_, d_huffs_, d_states_, d_raw_, d_clean_, d_rst_, d_chunk_, d_ckpt_, d_exit_, d_spec_exit_, d_entry_, d_tot_, d_spec_tot_, d_prefix_, d_changed_;
    LpDevBuf d_coef_, d_wide_, d_wide_id_, d_dc_, d_dcpart_, d_planes_, d_frames_desc_;
    LpPinned h_stage_, h_small_, h_out_;
    std::vector<size_t> h_out_off_;

    // progressive images (SOF2): their scans, laid out per upload and per decode range
    struct ProgScanUp { LpProgScan s; uint64_t raw_off; uint32_t raw_len; uint32_t level; };
    bool prog_on_device_ = false;                   // where this upload's scans are entropy-decoded (lp_prog_host.h)
    LpPinned h_pcoef_;                              // host mode: the decoded int16 coefficients of every progressive image of the upload
    std::vector<size_t> h_pcoef_off_;               // element offset per uploaded image
    std::vector<uint32_t> h_perr_;                  // per uploaded image
    std::vector<std::vector<ProgScanUp>> h_prog_;   // per uploaded image; empty for a baseline one
    std::vector<LpProgHuff> h_phuffs_;
    std::vector<LpProgScan> h_pscans_;              // the current range's scans, sorted by dependency level
    std::vector<uint32_t> h_plevel_first_;          // level l = h_pscans_[h_plevel_first_[l] .. h_plevel_first_[l + 1])
    std::vector<LpJpeg> h_pstreams_;                // one pseudo stream per scan (what the unstuff kernels need)
    std::vector<LpJpegState> h_pstates_;
    LpDevBuf d_phuffs_, d_pscans_, d_pstreams_, d_pstates_, d_pcoef_;

    // frame heap
    LpDevBuf heap_;
    size_t heap_used_ = 0;

    // resize / orient
    LpDevBuf d_ops_, d_taps_, d_ranges_, d_fops_;
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
