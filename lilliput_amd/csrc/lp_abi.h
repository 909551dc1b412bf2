// lp_abi.h -- internal types behind the opaque handles of include/lilliput_hip.h.
#pragma once
#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lilliput_hip.h"
#include "lp_engine.h"
#include "lp_png.h"
#include "lp_bmp.h"
#include "lp_pxm.h"

struct LpDevBlock {
    void* p = nullptr;
    size_t cap = 0;
    ~LpDevBlock();
};
std::shared_ptr<LpDevBlock> lp_dev_alloc(size_t bytes);

// Deferred Part A (lp_abi_opencv.cpp "deferred chains"): the encoded bytes a chain starts from. Shared by every Mat derived from one
// opencv_decoder_read_data; when the decoder is released before the chain has run, the bytes are copied into `keep` (the caller may
// free or reuse its buffer after Close, opencv.go:663-667).
struct LpLazySrc {
    const uint8_t* p = nullptr;     // nullptr: the decoder went away after the chain had been served (see opencv_decoder_release)
    size_t len = 0;
    std::vector<uint8_t> keep;
    bool served = false;            // opencv_encoder_write has produced a result from a chain over these bytes
    // A request "in flight" of deferred Part A: from the read_data that recorded the chain to the first time it is served (or run the eager
    // way, or dropped). The count of those is how opencv_encoder_write tells one or a few goroutines -- each served on its own thread -- from a
    // loaded service, whose requests share batch launches (lp_abi_opencv.cpp, profiles/r06_part_a.md section 4).
    std::atomic<bool> in_flight{false};
    void enter();
    void leave();
    ~LpLazySrc() { leave(); }
};
int lp_part_a_in_flight();
// One item through a resident batch of one on the calling thread (a pooled batch object: upload / run / download); the item's LILLIPUT_* status.
// What a lone deferred Part A chain and a Part C call without company run as (lp_abi_opencv.cpp).
bool lp_lone_batch_enabled();
int lp_lone_inline_max();     // LILLIPUT_HIP_DEFER_INLINE_MAX (8): that many requests in flight are served on their callers' threads while the dispatchers are idle
int lp_lone_batch_transform(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& bo, size_t* out_len);
extern "C" void lp_batch_set_stage_timing(void* batch, bool on); // lp_batch.cpp: the resident run of this batch records no stage events
// A Mat whose pixels have not been computed yet: decode [-> orientation] [-> crop] [-> resize] of a baseline JPEG, recorded call by call
// as unchanged ops.go issues them (ops.go:352-444 through opencv.go:250-374, 816-900). opencv_encoder_write(".jpeg") hands the whole
// chain to the batched path (lp_coalesce.h) -- one launch sequence shared with whatever other calls are in flight, no 48 MB frame ever
// written back; anything else that needs the pixels (a host read, another encoder, a composite) runs the chain the eager way first.
struct LpLazy {
    std::shared_ptr<LpLazySrc> src;
    int hdr_w = 0, hdr_h = 0, hdr_orientation = 1;  // of the source
    int orientation = 1;                            // what opencv_mat_orientation_transform was asked to apply (1 = nothing yet)
    bool has_crop = false;
    int cx = 0, cy = 0, cw = 0, ch = 0;             // opencv_mat_crop, on the oriented frame
    bool has_resize = false;
    int rw = 0, rh = 0;                             // opencv_mat_resize target
};

// What an `opencv_mat` handle points at (the reference's is a cv::Mat*, opencv.cpp:22-49).
struct LpMat {
    uint8_t* data = nullptr;        // first pixel of this (view of a) matrix -- host memory, usually Go-owned
    uint8_t* datastart = nullptr;
    uint8_t* datalimit = nullptr;
    int rows = 0, cols = 0, type = 0;
    size_t step = 0;
    std::vector<uint8_t> own;       // host storage when the Mat owns it (opencv_mat_create / reallocation)
    // device mirror: authoritative between ABI calls while dev_valid
    std::shared_ptr<LpDevBlock> dev;
    size_t dev_off = 0, dev_step = 0;
    bool dev_valid = false;
    bool dev_shared = false;        // a crop view sharing its parent's block
    bool host_stale = false;        // lazy write-back: the device holds newer pixels than `data`
    std::shared_ptr<LpLazy> lazy;   // non-null: neither `data` nor `dev` hold this Mat's pixels yet (see LpLazy); rows / cols / type are the result's
};

struct LpDecoder {
    const uint8_t* data = nullptr;
    size_t len = 0;
    bool is_png = false;            // which of cv::findDecoder's signatures matched
    bool is_bmp = false;
    bool is_pxm = false;            // "P1" .. "P6": cv::PxMDecoder (lp_pxm.h)
    LpBmpInfo bmp;
    LpPxmInfo pxm;
    bool parsed = false;
    int parse_rc = 0;
    LpJpegHeader hdr;
    LpPngInfo png;
    int png_channels = 0;           // channels of the Mat cv::PngDecoder::readHeader announces
    std::vector<std::weak_ptr<LpLazySrc>> lazies;   // deferred chains that still read this decoder's bytes
};

struct LpEncoder {
    LpMat* dst = nullptr;
    bool png = false; // ".png": cv::PngEncoder semantics, else cv::JpegEncoder
};

// One-image ABI calls check an engine (stream + grow-only arenas) out of a small per-device pool for the duration of the call and
// hand it back synchronised: under cgo a goroutine's consecutive calls arrive on whatever OS thread the Go scheduler picked
// (SURVEY.md 8b "no thread-affine state"), and a process with hundreds of threads must not hold hundreds of engines. Leases nest: a
// caller that wraps a whole Transform in one lease (lp_ops.cpp) keeps its calls on ONE stream, unsynchronised in between.
class LpEngineLease {
public:
    LpEngineLease();
    explicit LpEngineLease(LpEngine* own);   // the caller's own engine becomes this thread's engine for the scope (nullptr: like the default constructor)
    ~LpEngineLease();
    LpEngineLease(const LpEngineLease&) = delete;
    LpEngineLease& operator=(const LpEngineLease&) = delete;
    LpEngine* get() const { return eng_; }   // nullptr: no usable device (the error text is set)
private:
    void acquire();
    LpEngine* eng_ = nullptr;
    int dev_ = 0;
    bool owner_ = false, tls_ = false, adopted_ = false;
};
int lp_thread_device(int device); // device for this thread's one-image ABI calls (-1 = default); returns the previous setting
int lp_current_device();          // the device this thread's one-image calls run on
void lp_set_error(const std::string& s);
// Deferred Part A is for the library's callers, not for the library itself: a scope of this on the calling thread makes the opencv_*
// entry points eager (Part C and the batch workers bring their own strategy).
struct LpEagerScope { LpEagerScope(); ~LpEagerScope(); int prev; };
bool lp_mat_materialize(LpMat* m); // run a deferred chain the eager way (no-op for an ordinary Mat); false = it failed, the Mat is left without pixels
bool lp_mat_to_device(LpMat* m, LpEngine* eng);
bool lp_mat_to_host(LpMat* m, LpEngine* eng);
bool lp_mat_host_current(LpMat* m);
int lp_lazy_host_scope(int on); // per-thread override of the lazy write-back default (-1 = none); returns the previous value
LpFrame lp_mat_frame(const LpMat* m);
bool lp_mat_reshape(LpMat* m, int rows, int cols, int type); // cv::Mat::create for an output Mat: the external buffer is kept when the new shape fits
