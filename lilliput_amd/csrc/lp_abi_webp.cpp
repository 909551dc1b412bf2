// lp_abi_webp.cpp -- the reference's webp.hpp C ABI (/root/reference/webp.hpp:30-75, implemented there by webp.cpp over libwebp +
// libwebpmux) over this library's Mat: the container walk and the animation writer are lp_webp.cpp, the VP8 / VP8L payloads go to
// libwebp on the host (serial entropy-coded data stays host-side, like inflate and LZW), and the decoded
// sub-frames enter the device path at the first opencv_* call that touches them -- blend / dispose on the HBM-resident canvas
// (k_composite), Fit, resize (ops.go:552-582 through opencv.cpp:508-752).
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "lp_abi.h"
#include "lp_webp.h"
#include "lp_webp_sys.h"
#include "lp_abi_guard.h"

struct webp_decoder_struct {        // webp.cpp:10-29
    LpWebpFile file;
    bool has_alpha = false, has_animation = false;
    int width = 0, height = 0;
    int total_frame_count = 0, total_duration = 0;
    uint32_t bgcolor = 0xFFFFFFFFu, loop_count = 0;
    int current_frame_index = 1;
    int prev_frame_delay_time = 0, prev_frame_x_offset = 0, prev_frame_y_offset = 0, prev_frame_dispose = 0, prev_frame_blend = 0;
    std::vector<uint8_t> bitstream;
};

struct webp_encoder_struct {        // webp.cpp:31-55
    uint8_t* dst = nullptr;
    size_t dst_len = 0;
    std::vector<uint8_t> icc;
    uint32_t bgcolor = 0;
    uint32_t loop_count = 0;
    int frame_count = 1;
    bool is_animation = false, failed = false;
    // the first frame: its still encoding (what a one-frame input becomes) and its pixels (what the animation starts from)
    LpWebpEncodedImage still;
    std::vector<uint8_t> first_px;
    int first_w = 0, first_h = 0, first_cn = 0, first_delay = 0;
    // animation
    int canvas_w = 0, canvas_h = 0, canvas_cn = 0;
    std::vector<uint8_t> canvas;    // the previous input frame (tightly packed)
    std::vector<LpWebpAnimFrame> frames;
};

static inline int cvc(int type) { return (type >> 3) + 1; }

extern "C" {

webp_decoder webp_decoder_create(const opencv_mat buf) // webp.cpp:61-134
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto m = static_cast<const LpMat*>(buf);
    if (!m || !m->data) return nullptr;
    const size_t len = (size_t)m->rows * (size_t)m->cols * (size_t)cvc(m->type);
    webp_decoder d = new (std::nothrow) webp_decoder_struct();
    if (!d) return nullptr;
    if (!lp_webp_parse(m->data, len, &d->file)) { delete d; return nullptr; }
    // the first frame's bitstream must carry readable features (WebPGetFeatures on WebPMuxGetFrame(1))
    lp_webp_frame_bitstream(d->file.frames[0], d->bitstream);
    WebPBitstreamFeatures ft;
    if (WebPGetFeaturesInternal(d->bitstream.data(), d->bitstream.size(), &ft, LP_WEBP_DECODER_ABI) != 0) { delete d; return nullptr; }
    d->has_alpha = (d->file.flags & LP_WEBP_FLAG_ALPHA) != 0;
    d->width = d->file.canvas_w;
    d->height = d->file.canvas_h;
    d->total_frame_count = (int)d->file.frames.size();
    for (const auto& f : d->file.frames) d->total_duration += f.duration;
    if (d->file.flags & LP_WEBP_FLAG_ANIM) {
        if (d->file.has_anim_chunk) { d->bgcolor = d->file.bgcolor; d->loop_count = d->file.loop_count; }
        d->has_animation = true;
    } else
        d->total_duration = 0; // static images report no duration
    // No canvas-sized scratch buffer here: the canvas size comes from an untrusted VP8X header (a 100-byte file may claim 16383 x 16383
    // -- or, through VP8X, 2^24 x 2^24), and the reference's `new uint8_t[w * h * 4]` (webp.cpp:118) at least does not touch the pages.
    // webp_decoder_decode writes the frame straight into the caller's Mat, whose size the caller has checked (opencv.go:250-267).
    return d;
}
LP_ABI_CATCH("webp_decoder_create", return nullptr)

int webp_decoder_get_width(const webp_decoder d) { return d->width; }
int webp_decoder_get_height(const webp_decoder d) { return d->height; }
int webp_decoder_get_pixel_type(const webp_decoder d) { return d->has_alpha ? CV_8UC4 : CV_8UC3; }
int webp_decoder_get_num_frames(const webp_decoder d) { return d ? d->total_frame_count : 0; }
int webp_decoder_get_total_duration(const webp_decoder d) { return d ? d->total_duration : 0; }
int webp_decoder_get_prev_frame_delay(const webp_decoder d) { return d->prev_frame_delay_time; }
int webp_decoder_get_prev_frame_dispose(const webp_decoder d) { return d->prev_frame_dispose; }
int webp_decoder_get_prev_frame_blend(const webp_decoder d) { return d->prev_frame_blend; }
int webp_decoder_get_prev_frame_x_offset(const webp_decoder d) { return d->prev_frame_x_offset; }
int webp_decoder_get_prev_frame_y_offset(const webp_decoder d) { return d->prev_frame_y_offset; }
bool webp_decoder_get_prev_frame_has_alpha(const webp_decoder d) { return d->has_alpha; } // declared in webp.hpp, never defined nor called in the reference
uint32_t webp_decoder_get_bg_color(const webp_decoder d) { return d->bgcolor; }
uint32_t webp_decoder_get_loop_count(const webp_decoder d) { return d->loop_count; }

size_t webp_decoder_get_icc(const webp_decoder d, void* dst, size_t dst_len) // webp.cpp:262-273
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (d->file.icc && d->file.icc_size > 0 && d->file.icc_size <= dst_len) {
        memcpy(dst, d->file.icc, d->file.icc_size);
        return d->file.icc_size;
    }
    return 0;
}
LP_ABI_CATCH("webp_decoder_get_icc", return 0)

int webp_decoder_has_more_frames(webp_decoder d) { return d->current_frame_index < d->total_frame_count; }
void webp_decoder_advance_frame(webp_decoder d) { d->current_frame_index++; }

bool webp_decoder_decode(webp_decoder d, opencv_mat mat) // webp.cpp:302-362
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!d || !m) return false;
    if (d->current_frame_index < 1 || d->current_frame_index > d->total_frame_count) return false; // WebPMuxGetFrame: WEBP_MUX_NOT_FOUND
    const LpWebpFrame& fr = d->file.frames[(size_t)d->current_frame_index - 1];
    lp_webp_frame_bitstream(fr, d->bitstream);
    WebPBitstreamFeatures ft;
    if (WebPGetFeaturesInternal(d->bitstream.data(), d->bitstream.size(), &ft, LP_WEBP_DECODER_ABI) != 0) return false;
    // the Mat takes the FRAME's dimensions (cvMat->create): an animation frame is a sub-rectangle of the canvas
    const int type = webp_decoder_get_pixel_type(d);
    if (!lp_mat_reshape(m, ft.height, ft.width, type)) return false;
    const int cn = cvc(type), row_size = m->cols * cn;
    d->prev_frame_delay_time = fr.duration;
    d->prev_frame_x_offset = fr.x_offset;
    d->prev_frame_y_offset = fr.y_offset;
    d->prev_frame_dispose = fr.dispose;
    d->prev_frame_blend = fr.blend;
    if (m->rows <= 0 || row_size <= 0 || m->step < (size_t)row_size) return false;
    // The reference decodes into a buffer of canvas width x height x 4 bytes allocated with the decoder (webp.cpp:129-131, 339-350): a frame
    // whose rows do not fit it -- a damaged VP8X canvas smaller than its frames -- fails in WebPDecodeBGR(A)Into's size check
    // (MIN_BUFFER_SIZE: stride x (height - 1) + width x bytes per pixel), after the Mat was re-created and the frame's properties stored.
    if ((uint64_t)row_size * (uint64_t)(m->rows - 1) + (uint64_t)row_size > (uint64_t)webp_decoder_get_width(d) * (uint64_t)webp_decoder_get_height(d) * 4u) return false;
    const size_t span = m->step * (size_t)(m->rows - 1) + (size_t)row_size; // rows m->step apart, straight into the Mat (the reference copies them there row by row)
    uint8_t* res = cn == 4 ? WebPDecodeBGRAInto(d->bitstream.data(), d->bitstream.size(), m->data, span, (int)m->step)
                           : WebPDecodeBGRInto(d->bitstream.data(), d->bitstream.size(), m->data, span, (int)m->step);
    if (!res) return false;
    m->lazy.reset();
    m->dev_valid = false;   // the host copy is the frame now; it reaches the device with the next opencv_* call
    m->host_stale = false;
    return true;
}
LP_ABI_CATCH("webp_decoder_decode", return false)

void webp_decoder_release(webp_decoder d) { delete d; }

// ------------------------------------------------------------------------------------------------ encoder
webp_encoder webp_encoder_create(void* buf, size_t buf_len, const void* icc, size_t icc_len, uint32_t bgcolor, int loop_count) // webp.cpp:395-426
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    webp_encoder e = new (std::nothrow) webp_encoder_struct();
    if (!e) return nullptr;
    e->dst = (uint8_t*)buf;
    e->dst_len = buf_len;
    e->bgcolor = bgcolor;
    e->loop_count = (uint32_t)loop_count;
    if (icc && icc_len) e->icc.assign((const uint8_t*)icc, (const uint8_t*)icc + icc_len);
    return e;
}
LP_ABI_CATCH("webp_encoder_create", return nullptr)

static bool config_from_options(WebPConfig* c, const int* opt, size_t opt_len) // webp.cpp:450-498
{
    if (!WebPConfigInitInternal(c, 0, 100.0f, LP_WEBP_ENCODER_ABI)) return false;
    for (size_t i = 0; opt && i + 1 < opt_len; i += 2) {
        const int key = opt[i], value = opt[i + 1];
        switch (key) {
        case CV_IMWRITE_WEBP_QUALITY: {
            const float q = std::max(1.0f, (float)value);
            c->quality = std::min(100.0f, q);
            c->lossless = q > 100.0f;
            break;
        }
        case 1000: c->method = value; break;             // WEBP_METHOD ... WEBP_PALETTE, webp.hpp:13-23
        case 1001: c->filter_strength = value; break;
        case 1002: c->filter_type = value; break;
        case 1003: c->autofilter = value; break;
        case 1004: c->partitions = value; break;
        case 1005: c->segments = value; break;
        case 1006: c->preprocessing = value; break;
        case 1007: c->thread_level = value; break;
        case 1008: c->use_delta_palette = value; break;
        }
    }
    return true;
}

// libwebp's tables for the gamma-aware chroma down-sampling of its RGB -> YUV 4:2:0 import (picture_csp_enc.c InitGammaTables: gamma
// 0.80, 12-bit linear values, 32 + 1 interpolation points back), computed with the library's formula by the same libm
static const LpWebpYuvTab& webp_yuv_tables()
{
    static const LpWebpYuvTab tab = [] {
        LpWebpYuvTab t;
        memset(&t, 0, sizeof(t));
        const double kGamma = 0.80, kGammaScale = (double)((1 << 12) - 1), scale = (double)(1 << 7) / kGammaScale, norm = 1. / 255.;
        for (int v = 0; v <= 255; v++) t.gamma_to_linear[v] = (uint16_t)(pow(norm * v, kGamma) * kGammaScale + .5);
        for (int v = 0; v <= 32; v++) t.linear_to_gamma[v] = (int)(255. * pow(scale * v, 1. / kGamma) + .5);
        return t;
    }();
    return tab;
}

// The lossy still path with the colour conversion on the device: Y, U, V planes of a frame the device holds, then WebPEncode on them
// with the configuration WebPEncodeBGR(A) builds (WebPConfigPreset(default, quality)): the same bitstream, the import step skipped.
// false: not applicable (translucent pixels -- the alpha-weighted import stays libwebp's -- or no device), the caller takes the host route.
static bool encode_still_from_device(LpMat* m, float quality, LpWebpEncodedImage* out)
{
    static const bool off = getenv("LILLIPUT_HIP_WEBP_YUV") && !strcmp(getenv("LILLIPUT_HIP_WEBP_YUV"), "host");
    const int cn = cvc(m->type);
    if (off || !m->dev_valid || (cn != 3 && cn != 4)) return false;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) return false;
    const int w = m->cols, h = m->rows, uvw = (w + 1) / 2, uvh = (h + 1) / 2;
    std::vector<uint8_t> planes((size_t)w * h + 2 * (size_t)uvw * uvh);
    uint8_t *y = planes.data(), *u = y + (size_t)w * h, *v = u + (size_t)uvw * uvh;
    bool translucent = false;
    if (eng->webp_yuv420(lp_mat_frame(m), webp_yuv_tables(), y, u, v, &translucent) || translucent) return false;
    WebPConfig cfg;
    WebPPicture pic;
    if (!WebPConfigInitInternal(&cfg, 0 /* WEBP_PRESET_DEFAULT */, quality, LP_WEBP_ENCODER_ABI) || !WebPPictureInitInternal(&pic, LP_WEBP_ENCODER_ABI)) return false;
    pic.use_argb = 0;
    pic.colorspace = 0; // WEBP_YUV420
    pic.width = w; pic.height = h;
    pic.y = y; pic.u = u; pic.v = v;
    pic.y_stride = w; pic.uv_stride = uvw;
    WebPMemoryWriter wr;
    WebPMemoryWriterInit(&wr);
    pic.writer = WebPMemoryWrite;
    pic.custom_ptr = &wr;
    const bool good = WebPEncode(&cfg, &pic) && lp_webp_split_encoded(wr.mem, wr.size, out);
    WebPMemoryWriterClear(&wr); // the planes are the caller's: nothing for WebPPictureFree to release
    return good;
}

// One rectangle of a frame through libwebp's advanced API with the caller's configuration (what WebPAnimEncoderAdd does per frame).
static bool encode_rect(const WebPConfig& cfg, const uint8_t* px, int stride, int w, int h, int cn, LpWebpEncodedImage* out)
{
    WebPPicture pic;
    if (!WebPPictureInitInternal(&pic, LP_WEBP_ENCODER_ABI)) return false;
    pic.use_argb = 1;
    pic.width = w;
    pic.height = h;
    const int ok_in = cn == 3 ? WebPPictureImportBGR(&pic, px, stride) : WebPPictureImportBGRA(&pic, px, stride);
    if (!ok_in) { WebPPictureFree(&pic); return false; }
    WebPMemoryWriter wr;
    WebPMemoryWriterInit(&wr);
    pic.writer = WebPMemoryWrite;
    pic.custom_ptr = &wr;
    const int ok = WebPEncode(&cfg, &pic);
    bool good = ok && lp_webp_split_encoded(wr.mem, wr.size, out);
    WebPPictureFree(&pic);
    WebPMemoryWriterClear(&wr);
    return good;
}

// The next frame of the animation (the whole canvas, as WebPAnimEncoderAdd takes it): the rectangle that differs from the previous
// frame is coded on its own and placed without blending; an identical frame extends its predecessor (as libwebp's encoder does).
static bool anim_add(webp_encoder e, const WebPConfig& cfg, const uint8_t* px, int w, int h, int cn, int duration)
{
    if (w != e->canvas_w || h != e->canvas_h) { fprintf(stderr, "webp: frame size differs from the animation canvas\n"); return false; }
    int x0 = 0, y0 = 0, x1 = w, y1 = h;
    const size_t row = (size_t)w * cn;
    if (!e->frames.empty() && cn == e->canvas_cn) {
        int top = 0, bot = h;
        while (top < h && memcmp(px + (size_t)top * row, e->canvas.data() + (size_t)top * row, row) == 0) top++;
        // an unchanged frame only lengthens the one before it -- as far as the ANMF duration field goes (24 bits); beyond that the time goes
        // into a frame of its own (one pixel of the unchanged canvas) instead of wrapping
        if (top == h && (int64_t)e->frames.back().duration + duration <= 0xFFFFFF) { e->frames.back().duration += duration; return true; }
        if (top == h) { top = 0; bot = 1; }
        while (bot > top && memcmp(px + (size_t)(bot - 1) * row, e->canvas.data() + (size_t)(bot - 1) * row, row) == 0) bot--;
        int left = w, right = 0;
        for (int y = top; y < bot; y++) {
            const uint8_t *a = px + (size_t)y * row, *b = e->canvas.data() + (size_t)y * row;
            int l = 0, r = w;
            while (l < left && memcmp(a + (size_t)l * cn, b + (size_t)l * cn, (size_t)cn) == 0) l++;
            while (r > right && r > l && memcmp(a + (size_t)(r - 1) * cn, b + (size_t)(r - 1) * cn, (size_t)cn) == 0) r--;
            left = std::min(left, l);
            right = std::max(right, r);
        }
        if (right <= left) { left = 0; right = 1; } // the unchanged-frame case above
        x0 = left & ~1; y0 = top & ~1; x1 = right; y1 = bot; // frame offsets are stored halved: even positions only
    }
    LpWebpAnimFrame f;
    if (!encode_rect(cfg, px + (size_t)y0 * row + (size_t)x0 * cn, (int)row, x1 - x0, y1 - y0, cn, &f.im)) return false;
    f.x_offset = x0; f.y_offset = y0; f.duration = std::min(std::max(duration, 0), 0xFFFFFF);
    f.dispose = 0;  // WEBP_MUX_DISPOSE_NONE
    f.blend = 1;    // WEBP_MUX_NO_BLEND: the rectangle replaces what the canvas held, alpha included
    e->frames.push_back(std::move(f));
    e->canvas.assign(px, px + row * (size_t)h);
    e->canvas_cn = cn;
    return true;
}

size_t webp_encoder_write(webp_encoder e, const opencv_mat src, const int* opt, size_t opt_len, int delay, int blend, int dispose, int x_offset, int y_offset) // webp.cpp:436-755
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    (void)blend; (void)dispose; (void)x_offset; (void)y_offset; // stored by the reference for the first frame and never used
    if (!e) return 0;
    WebPConfig config;
    if (!config_from_options(&config, opt, opt_len)) return 0;
    try {
        if (!src) { // finalisation
            if (e->frame_count == 1 || e->failed) return 0;
            std::vector<uint8_t> out;
            if (e->is_animation) lp_webp_write_animation(e->canvas_w, e->canvas_h, e->bgcolor, e->loop_count, e->frames, e->icc.empty() ? nullptr : e->icc.data(), e->icc.size(), out);
            else lp_webp_write_still(e->still, e->icc.empty() ? nullptr : e->icc.data(), e->icc.size(), out);
            if (out.size() > e->dst_len) { fprintf(stderr, "Error: Final encoded size (%zu) exceeds buffer size (%zu)\n", out.size(), e->dst_len); return 0; }
            memcpy(e->dst, out.data(), out.size());
            return out.size();
        }
        auto m = static_cast<LpMat*>(src);
        if (!m || m->rows <= 0 || m->cols <= 0 || (m->type & 7) != 0) return 0; // empty, or not 8-bit unsigned
        int cn = cvc(m->type);
        if (cn != 1 && cn != 3 && cn != 4) return 0;
        if (lilliput_hip_mat_sync_host(src)) return 0; // the frame was produced on the device: the VP8 coder reads it on the host
        std::vector<uint8_t> px((size_t)m->rows * m->cols * (cn == 1 ? 3 : cn));
        for (int y = 0; y < m->rows; y++) {
            const uint8_t* s = m->data + (size_t)y * m->step;
            uint8_t* d = px.data() + (size_t)y * m->cols * (cn == 1 ? 3 : cn);
            if (cn == 1) for (int x = 0; x < m->cols; x++) { d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x]; } // cv::COLOR_GRAY2BGR
            else memcpy(d, s, (size_t)m->cols * cn);
        }
        if (cn == 1) cn = 3;
        const int w = m->cols, h = m->rows, stride = w * cn;
        if (e->frame_count == 1) { e->first_px = px; e->first_w = w; e->first_h = h; e->first_cn = cn; }
        if (e->frame_count == 2 && !e->is_animation) { // the second frame turns the output into an animation that starts with the stored first frame
            e->is_animation = true;
            e->canvas_w = w; e->canvas_h = h;
            if (e->first_w != w || e->first_h != h || !anim_add(e, config, e->first_px.data(), w, h, e->first_cn, e->first_delay)) { e->failed = true; return 0; }
            e->first_px.clear();
            e->first_px.shrink_to_fit();
        }
        size_t size = 0;
        if (e->is_animation) {
            if (!anim_add(e, config, px.data(), w, h, cn, delay)) { e->failed = true; return 0; }
            size = 1; // WebPPictureImport's return value in the reference: non-zero
        } else {
            uint8_t* outp = nullptr;
            if (!config.lossless && encode_still_from_device(m, config.quality, &e->still)) {
                e->first_delay = delay;
                e->frame_count++;
                return 1; // the reference returns the coded size here; the Go layer only tests it against zero (webp.go: EncodeFrame)
            }
            if (config.lossless) size = cn == 3 ? WebPEncodeLosslessBGR(px.data(), w, h, stride, &outp) : WebPEncodeLosslessBGRA(px.data(), w, h, stride, &outp);
            else size = cn == 3 ? WebPEncodeBGR(px.data(), w, h, stride, config.quality, &outp) : WebPEncodeBGRA(px.data(), w, h, stride, config.quality, &outp);
            if (size == 0) return 0;
            const bool ok = lp_webp_split_encoded(outp, size, &e->still);
            WebPFree(outp);
            if (!ok) return 0;
            e->first_delay = delay;
        }
        e->frame_count++;
        return size;
    } catch (const std::bad_alloc&) {
        e->failed = true;
        return 0;
    }
}
LP_ABI_CATCH("webp_encoder_write", return 0)

// Test access: the Y, U, V planes the lossy encoder is handed for this frame (k_webp_yuv420). 0 = ok, 1 = the frame has translucent pixels
// (the product then lets libwebp import it), -1 = error.
int lilliput_hip_webp_yuv420(const opencv_mat src, uint8_t* y, uint8_t* u, uint8_t* v)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)src));
    if (!m || m->rows <= 0 || m->cols <= 0 || (cvc(m->type) != 3 && cvc(m->type) != 4)) return -1;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) return -1;
    bool translucent = false;
    if (eng->webp_yuv420(lp_mat_frame(m), webp_yuv_tables(), y, u, v, &translucent)) return -1;
    return translucent ? 1 : 0;
}
LP_ABI_CATCH("lilliput_hip_webp_yuv420", return -1)

size_t webp_encoder_flush(webp_encoder e) { return webp_encoder_write(e, nullptr, nullptr, 0, 0, 0, 0, 0, 0); } // webp.cpp:780-783
void webp_encoder_release(webp_encoder e) { delete e; }

} // extern "C"
