// lp_abi.h -- internal types behind the opaque handles of include/lilliput_hip.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/lilliput_hip.h"
#include "lp_engine.h"
#include "lp_png.h"
#include "lp_bmp.h"

struct LpDevBlock {
    void* p = nullptr;
    size_t cap = 0;
    ~LpDevBlock();
};
std::shared_ptr<LpDevBlock> lp_dev_alloc(size_t bytes);

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
};

struct LpDecoder {
    const uint8_t* data = nullptr;
    size_t len = 0;
    bool is_png = false;            // which of cv::findDecoder's signatures matched
    bool is_bmp = false;
    LpBmpInfo bmp;
    bool parsed = false;
    int parse_rc = 0;
    LpJpegHeader hdr;
    LpPngInfo png;
    int png_channels = 0;           // channels of the Mat cv::PngDecoder::readHeader announces
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
bool lp_mat_to_device(LpMat* m, LpEngine* eng);
bool lp_mat_to_host(LpMat* m, LpEngine* eng);
bool lp_mat_host_current(LpMat* m);
int lp_lazy_host_scope(int on); // per-thread override of the lazy write-back default (-1 = none); returns the previous value
LpFrame lp_mat_frame(const LpMat* m);
bool lp_mat_reshape(LpMat* m, int rows, int cols, int type); // cv::Mat::create for an output Mat: the external buffer is kept when the new shape fits
