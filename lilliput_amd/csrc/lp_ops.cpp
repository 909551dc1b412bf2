// lp_ops.cpp -- Part C of include/lilliput_hip.h: a C++ mirror of the Go layer that drives the cgo
// boundary -- Decoder / Framebuffer / ImageOps / Encoder (/root/reference/lilliput.go:42-202,
// opencv.go:118-146, 207-463, 639-661, 816-905, ops.go:68-591). The reference is compiled Go; with no Go
// toolchain in this image the host side is written in C++ and calls the SAME opencv_* C ABI functions in
// the same order the Go code does, so the drop-in boundary is exercised exactly as a Go caller would.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <string>
#include <vector>

#include "lp_abi.h"
#include "lp_coalesce.h"
#include "lp_ops_logic.h"
#include "lp_abi_guard.h"

// ---------------------------------------------------------------- opencv.go:468-637 byte scanners
static const uint8_t kPngMagic[8] = {0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a};
static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

namespace {
struct PngIter { // opencv.go:468-512 pngChunkIter
    const uint8_t* png;
    size_t len;
    long off;
    bool has_space() const { return off + 12 <= (long)len; }
    long next_off() const { return off + (long)be32(png + off) + 12; }
    bool next()
    {
        if (off < 8) { off = 8; return has_space(); }
        if (!has_space()) return false;
        off = next_off();
        return off >= 0 && has_space();
    }
    const uint8_t* type() const { return png + off + 4; }
};
}

static int content_length_png(const uint8_t* b, size_t n) // opencv.go:514-535
{
    if (n < 8 || memcmp(b, kPngMagic, 8) != 0) return (int)n;
    PngIter it{b, n, 0};
    while (it.next())
        if (memcmp(it.type(), "IEND", 4) == 0) {
            long e = it.next_off();
            if (e > (long)n) e = (long)n;
            return (int)e;
        }
    return (int)n;
}

static int content_length_jpeg(const uint8_t* j, size_t n) // opencv.go:537-602
{
    if (n < 3 || j[0] != 0xFF || j[1] != 0xD8 || j[2] != 0xFF) return (int)n;
    size_t idx = 0;
    for (;;) {
        if (idx + 1 >= n) break;
        if (j[idx] != 0xFF) break;
        size_t next = idx + 2;
        uint8_t t = j[idx + 1];
        if (t == 0xD9) return (int)next;
        if (t == 0xFF) { idx++; continue; }
        if ((t >= 0xD0 && t <= 0xD8)) { idx = next; continue; }
        if (idx + 3 >= n) break;
        next += ((size_t)j[idx + 2] << 8) | j[idx + 3];
        if (t == 0xDA) {
            for (; next < n; next++) {
                if (j[next] != 0xFF) { // same walk as the reference's byte loop, taken a run at a time
                    const void* f = memchr(j + next, 0xFF, n - next);
                    if (!f) { next = n; break; }
                    next = (size_t)((const uint8_t*)f - j);
                }
                if (next + 1 >= n) { next = n; break; }
                uint8_t peek = j[next + 1];
                if (peek == 0xFF) continue;
                if (peek != 0 && (peek < 0xD0 || peek > 0xD7)) break;
            }
        }
        idx = next;
    }
    return (int)n;
}

int lp_detect_content_length(const uint8_t* b, size_t n) // opencv.go:604-614
{
    int a = content_length_jpeg(b, n), p = content_length_png(b, n);
    return a < p ? a : p;
}

bool lp_detect_apng(const uint8_t* b, size_t n) // opencv.go:617-637
{
    if (n < 8 || memcmp(b, kPngMagic, 8) != 0) return false;
    PngIter it{b, n, 0};
    while (it.next()) {
        const uint8_t* t = it.type();
        if (memcmp(t, "acTL", 4) == 0 || memcmp(t, "fcTL", 4) == 0 || memcmp(t, "fdAT", 4) == 0) return true;
    }
    return false;
}

// ---------------------------------------------------------------- Framebuffer (opencv.go:118-129, 207-374)
namespace {
struct Framebuffer {
    uint8_t* buf = nullptr;
    size_t buf_len = 0;
    opencv_mat mat = nullptr;
    int width = 0, height = 0, pixel_type = 0;
    int64_t duration = 0;
    int x_offset = 0, y_offset = 0, dispose = 0, blend = 0;

    void init(int w, int h) // NewFramebuffer
    {
        buf_len = (size_t)w * h * 4;
        buf = (uint8_t*)calloc(buf_len ? buf_len : 1, 1);
    }
    void close() { if (mat) { opencv_mat_release(mat); mat = nullptr; } }
    void destroy() { close(); free(buf); buf = nullptr; }
    void clear() { if (buf) memset(buf, 0, buf_len); if (mat) opencv_mat_reset(mat); }
    int resize_mat(int w, int h, int ptype) // opencv.go:250-267
    {
        if (mat) { opencv_mat_release(mat); mat = nullptr; }
        if (opencv_type_depth(ptype) > 8) ptype = opencv_type_convert_depth(ptype, CV_8U);
        opencv_mat m = opencv_mat_create_from_data(w, h, ptype, buf, buf_len);
        if (!m) return LILLIPUT_ERR_BUF_TOO_SMALL;
        mat = m; width = w; height = h; pixel_type = ptype;
        return LILLIPUT_OK;
    }
    void orientation_transform(int orientation) // opencv.go:271-279
    {
        if (!mat) return;
        opencv_mat_orientation_transform((CVImageOrientation)orientation, mat);
        width = opencv_mat_get_width(mat);
        height = opencv_mat_get_height(mat);
    }
    int resize_to(int w, int h, Framebuffer* dst) // opencv.go:294-309
    {
        if (w < 1) w = 1;
        if (h < 1) h = 1;
        int e = dst->resize_mat(w, h, pixel_type);
        if (e) return e;
        opencv_mat_resize(mat, dst->mat, w, h, CV_INTER_AREA);
        return LILLIPUT_OK;
    }
    int fit(int w, int h, Framebuffer* dst) // opencv.go:326-374
    {
        if (!mat) return LILLIPUT_ERR_FRAMEBUF_NO_PIXELS;
        int left, top, wpc, hpc;
        lp_fit_crop_rect(width, height, w, h, &left, &top, &wpc, &hpc);
        opencv_mat nm = opencv_mat_crop(mat, left, top, wpc, hpc);
        if (!nm) return LILLIPUT_ERR_INVALID_IMAGE;
        int e = dst->resize_mat(w, h, pixel_type);
        if (!e) opencv_mat_resize(nm, dst->mat, w, h, CV_INTER_AREA);
        opencv_mat_release(nm);
        return e;
    }
};

struct Decoder { // openCVDecoder (opencv.go:132-138), gifDecoder (giflib.go:14-28) or webpDecoder (webp.go:14-18)
    enum Kind { OPENCV, GIF, WEBP, PIXELS } kind = OPENCV;
    const uint8_t* buf = nullptr;
    size_t len = 0;
    opencv_mat mat = nullptr;
    opencv_decoder dec = nullptr;
    giflib_decoder gif = nullptr;
    webp_decoder webp = nullptr;
    bool has_read_header = false, has_decoded = false;
    int content_length = -1, num_frames = 0; // the buffer cannot change under a decoder: scanned once, not per Header() call
    bool anim_read = false;                  // gifDecoder.readAnimationInfo: lazily, once
    GifAnimationInfo anim = {1, 0, 255, 255, 255, 0, 0};
    int frame_index = 0;
};

struct Header { int width, height, pixel_type, orientation, num_frames, content_length; };

const int kGifMaxFrameDimension = 10000; // defaultMaxFrameDimension, giflib.go:39

int decoder_header(Decoder* d, Header* h, bool want_content_length = false)
{
    if (d->kind == Decoder::GIF) { // giflib.go:77-87
        if (!d->anim_read) { d->anim = giflib_decoder_get_animation_info(d->gif); d->anim_read = true; }
        h->width = giflib_decoder_get_width(d->gif);
        h->height = giflib_decoder_get_height(d->gif);
        h->pixel_type = CV_8UC4;
        h->orientation = 1; // OrientationTopLeft
        h->num_frames = d->anim.frame_count;
        h->content_length = (int)d->len;
        return LILLIPUT_OK;
    }
    if (d->kind == Decoder::WEBP) { // webp.go:50-59
        h->width = webp_decoder_get_width(d->webp);
        h->height = webp_decoder_get_height(d->webp);
        h->pixel_type = webp_decoder_get_pixel_type(d->webp);
        h->orientation = 1;
        h->num_frames = webp_decoder_get_num_frames(d->webp);
        h->content_length = (int)d->len;
        return LILLIPUT_OK;
    }
    if (d->kind == Decoder::PIXELS) { // the hand-over item: a frame some host decoder (AVIF, video) produced, see lilliput_hip_pixels_header
        const auto* ph = reinterpret_cast<const lilliput_hip_pixels_header*>(d->buf);
        h->width = (int)ph->width;
        h->height = (int)ph->height;
        h->pixel_type = ph->channels == 1 ? CV_8U : ph->channels == 3 ? CV_8UC3 : CV_8UC4;
        h->orientation = (int)ph->orientation;
        h->num_frames = 1;
        h->content_length = (int)d->len;
        return LILLIPUT_OK;
    }
    // opencv.go:639-661
    if (!d->has_read_header && !opencv_decoder_read_header(d->dec)) return LILLIPUT_ERR_INVALID_IMAGE;
    d->has_read_header = true;
    if (!d->num_frames) d->num_frames = lp_detect_apng(d->buf, d->len) ? 2 : 1;
    h->num_frames = d->num_frames;
    h->width = opencv_decoder_get_width(d->dec);
    h->height = opencv_decoder_get_height(d->dec);
    h->pixel_type = opencv_decoder_get_pixel_type(d->dec);
    h->orientation = opencv_decoder_get_orientation(d->dec);
    // the content length is a walk over the whole entropy-coded segment (0.3-0.4 ms for a 4 MB file that is not in the cache): made when the caller
    // asks for it (lilliput_decoder_header with a content_length pointer), not for the Header() calls inside Transform, which never look at it
    if (want_content_length && d->content_length < 0) d->content_length = lp_detect_content_length(d->buf, d->len);
    h->content_length = d->content_length;
    return LILLIPUT_OK;
}

int decoder_decode_to(Decoder* d, Framebuffer* f)
{
    Header h;
    if (d->kind == Decoder::GIF) { // giflib.go:180-219
        int e = decoder_header(d, &h);
        if (e) return e;
        e = f->resize_mat(h.width, h.height, h.pixel_type);
        if (e) return e;
        giflib_decoder_frame_state st = giflib_decoder_decode_frame_header(d->gif);
        if (st == giflib_decoder_eof) return LILLIPUT_ERR_EOF;
        if (st == giflib_decoder_error) return LILLIPUT_ERR_INVALID_IMAGE;
        if (giflib_decoder_get_frame_width(d->gif) > kGifMaxFrameDimension || giflib_decoder_get_frame_height(d->gif) > kGifMaxFrameDimension) return LILLIPUT_ERR_INVALID_IMAGE;
        if (!giflib_decoder_decode_frame(d->gif, f->mat)) return LILLIPUT_ERR_DECODING_FAILED;
        f->duration = (int64_t)giflib_decoder_get_prev_frame_delay(d->gif) * 10 * 1000000ll;
        f->blend = 1; // NoBlend
        f->dispose = giflib_decoder_get_prev_frame_disposal(d->gif);
        f->x_offset = f->y_offset = 0;
        d->frame_index++;
        return LILLIPUT_OK;
    }
    if (d->kind == Decoder::WEBP) { // webp.go:129-167
        int e = decoder_header(d, &h);
        if (e) return e;
        e = f->resize_mat(h.width, h.height, h.pixel_type);
        if (e) return e;
        if (!webp_decoder_decode(d->webp, f->mat)) return webp_decoder_has_more_frames(d->webp) == 0 ? LILLIPUT_ERR_EOF : LILLIPUT_ERR_DECODING_FAILED;
        f->duration = (int64_t)webp_decoder_get_prev_frame_delay(d->webp) * 1000000ll;
        f->x_offset = webp_decoder_get_prev_frame_x_offset(d->webp);
        f->y_offset = webp_decoder_get_prev_frame_y_offset(d->webp);
        f->dispose = webp_decoder_get_prev_frame_dispose(d->webp);
        f->blend = webp_decoder_get_prev_frame_blend(d->webp);
        webp_decoder_advance_frame(d->webp);
        return LILLIPUT_OK;
    }
    // opencv.go:816-839
    if (d->has_decoded) return LILLIPUT_ERR_EOF;
    int e = decoder_header(d, &h);
    if (e) return e;
    e = f->resize_mat(h.width, h.height, h.pixel_type);
    if (e) return e;
    if (d->kind == Decoder::PIXELS) { // what avcodecDecoder.DecodeTo / avifDecoder.DecodeTo leave behind: one BGR(A) frame in the framebuffer
        const auto* ph = reinterpret_cast<const lilliput_hip_pixels_header*>(d->buf);
        const size_t rowb = (size_t)ph->width * ph->channels, stride = ph->stride ? ph->stride : rowb;
        if (lilliput_hip_mat_set_pixels(f->mat, d->buf + sizeof(*ph), stride)) return LILLIPUT_ERR_DECODING_FAILED;
        d->has_decoded = true;
        f->blend = 1; f->dispose = 1; f->x_offset = f->y_offset = 0;
        f->duration = (int64_t)ph->duration_ms * 1000000ll;
        return LILLIPUT_OK;
    }
    if (!opencv_decoder_read_data(d->dec, f->mat)) return LILLIPUT_ERR_DECODING_FAILED;
    d->has_decoded = true;
    f->blend = 1;   // NoBlend
    f->dispose = 1; // DisposeToBackgroundColor
    f->x_offset = f->y_offset = 0;
    f->duration = 0;
    return LILLIPUT_OK;
}

int decoder_skip_frame(Decoder* d) // giflib.go:223-235 / opencv.go:841-843
{
    if (d->kind != Decoder::GIF) return LILLIPUT_ERR_SKIP_NOT_SUPPORTED;
    giflib_decoder_frame_state st = giflib_decoder_skip_frame(d->gif);
    if (st == giflib_decoder_eof) return LILLIPUT_ERR_EOF;
    if (st == giflib_decoder_error) return LILLIPUT_ERR_INVALID_IMAGE;
    return LILLIPUT_OK;
}

int skip_to_end(Decoder* d) // ops.go:337-346
{
    for (;;) {
        int e = decoder_skip_frame(d);
        if (e) return e;
    }
}

struct Encoder { // openCVEncoder (opencv.go:141-146, 847-905), or the raw frame sink used by the tests
    enum Kind { OPENCV, RAW_FRAMES, GIF, THUMBHASH, WEBP } kind = OPENCV;
    webp_encoder webp = nullptr; // webpEncoder, webp.go:20-26, 174-261
    thumbhash_encoder th = nullptr; // thumbhashEncoder, thumbhash.go:12-16
    opencv_encoder enc = nullptr;
    giflib_encoder gif = nullptr; // gifEncoder, giflib.go:30-37, 239-296
    int gif_frame_index = 0;
    opencv_mat dst = nullptr;
    uint8_t* dst_buf = nullptr;
    size_t dst_cap = 0, raw_len = 0;
    bool flushed = false;
};

int opencv_status(int code) { return code == OPENCV_SUCCESS ? LILLIPUT_OK : LILLIPUT_ERR_OPENCV_BASE + code; } // handleOpenCVError, opencv.go:399-426

struct ImageOps { // ops.go:68-106
    Framebuffer frames[2];
    Framebuffer composite; // animatedCompositeBuffer: lives for one Transform
    bool have_composite = false;
    int frame_index = 0;
    Framebuffer* active() { return &frames[frame_index]; }
    Framebuffer* secondary() { return &frames[1 - frame_index]; }
    void swap() { frame_index = 1 - frame_index; }
    void copy_props_and_swap() // ops.go:586-591
    {
        secondary()->duration = active()->duration;
        secondary()->dispose = active()->dispose;
        secondary()->blend = active()->blend;
        swap();
    }
    void drop_composite() { if (have_composite) { composite.destroy(); have_composite = false; } }
    int setup_animated(int w, int h, bool has_alpha) // ops.go:132-150
    {
        if (have_composite) return LILLIPUT_OK;
        composite = Framebuffer();
        composite.init(w, h);
        if (!composite.buf) return LILLIPUT_ERR_BUF_TOO_SMALL;
        have_composite = true;
        int e = composite.resize_mat(w, h, has_alpha ? CV_8UC4 : CV_8UC3); // Create3Channel / Create4Channel
        if (e) return e;
        composite.clear();
        return opencv_status(opencv_mat_clear_to_transparent(composite.mat, 0, 0, w, h));
    }
    int apply_blend() // ops.go:566-582
    {
        Framebuffer* a = active();
        if (a->blend == 0) return opencv_status(opencv_copy_to_region_with_alpha(a->mat, composite.mat, a->x_offset, a->y_offset, a->width, a->height));
        if (a->blend == 1) return opencv_status(opencv_copy_to_region(a->mat, composite.mat, a->x_offset, a->y_offset, a->width, a->height));
        return LILLIPUT_OK;
    }
    int apply_dispose() // ops.go:552-563
    {
        Framebuffer* a = active();
        if (a->dispose == 1) return opencv_status(opencv_mat_clear_to_transparent(composite.mat, a->x_offset, a->y_offset, a->width, a->height));
        return LILLIPUT_OK;
    }
};

int64_t now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

std::string lower(const char* s) { std::string r(s ? s : ""); for (auto& c : r) c = (char)tolower(c); return r; }
} // namespace

extern "C" {

void lilliput_calculate_expected_size(int ow, int oh, int rw, int rh, int* w, int* h) { lp_calculate_expected_size(ow, oh, rw, rh, w, h); }
void lilliput_fit_crop_rect(int fw, int fh, int width, int height, int* left, int* top, int* w, int* h) { lp_fit_crop_rect(fw, fh, width, height, left, top, w, h); }
int lilliput_detect_content_length(const void* buf, size_t len) { return lp_detect_content_length((const uint8_t*)buf, len); }
int lilliput_detect_apng(const void* buf, size_t len) { return lp_detect_apng((const uint8_t*)buf, len) ? 1 : 0; }

int lilliput_new_decoder(const void* buf, size_t len, lilliput_decoder* out) // lilliput.go:129-164 + opencv.go:442-463 + giflib.go:56-75
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    *out = nullptr;
    if (!buf || len == 0) return LILLIPUT_ERR_INVALID_IMAGE;
    const uint8_t* b = (const uint8_t*)buf;
    // AVIF sources have their own decoder in the reference (lilliput.go:146-149: libavif + dav1d / aom -- an AV1 codec); it is outside this build.
    if (len >= 12 && memcmp(b + 4, "ftyp", 4) == 0 && (memcmp(b + 8, "avif", 4) == 0 || memcmp(b + 8, "avis", 4) == 0)) return LILLIPUT_ERR_UNSUPPORTED;
    opencv_mat mat = opencv_mat_create_from_data((int)len, 1, CV_8U, (void*)buf, len);
    if (!mat) return LILLIPUT_ERR_BUF_TOO_SMALL;
    auto d = new Decoder();
    d->buf = b; d->len = len; d->mat = mat;
    if (len >= sizeof(lilliput_hip_pixels_header) && memcmp(b, LILLIPUT_HIP_PIXELS_MAGIC, 8) == 0) { // decoded elsewhere, handed over as pixels
        const auto* ph = reinterpret_cast<const lilliput_hip_pixels_header*>(b);
        const uint64_t rowb = (uint64_t)ph->width * ph->channels, stride = ph->stride ? ph->stride : rowb;
        const bool ok = ph->width && ph->height && ph->width <= 65535 && ph->height <= 65535 && (ph->channels == 1 || ph->channels == 3 || ph->channels == 4) &&
                        stride >= rowb && ph->orientation >= 1 && ph->orientation <= 8 && sizeof(*ph) + stride * (ph->height - 1) + rowb <= len;
        if (!ok) { opencv_mat_release(mat); delete d; return LILLIPUT_ERR_INVALID_IMAGE; }
        d->kind = Decoder::PIXELS;
        *out = d;
        return LILLIPUT_OK;
    }
    if (len >= 12 && memcmp(b, "RIFF", 4) == 0 && memcmp(b + 8, "WEBP", 4) == 0) { // isWebp, lilliput.go:104-115 -> newWebpDecoder, webp.go:27-48
        d->kind = Decoder::WEBP;
        d->webp = webp_decoder_create(mat);
        if (!d->webp) { opencv_mat_release(mat); delete d; return LILLIPUT_ERR_INVALID_IMAGE; }
        *out = d;
        return LILLIPUT_OK;
    }
    if (len >= 6 && (memcmp(b, "GIF87a", 6) == 0 || memcmp(b, "GIF89a", 6) == 0)) { // isGIF, lilliput.go:100-102
        d->kind = Decoder::GIF;
        d->gif = giflib_decoder_create(mat);
        if (!d->gif) { opencv_mat_release(mat); delete d; return LILLIPUT_ERR_INVALID_IMAGE; } // newGifDecoder leaves the Mat to the GC'd Go object; here it is freed
    } else {
        d->dec = opencv_decoder_create(mat);
        if (!d->dec) { opencv_mat_release(mat); delete d; return LILLIPUT_ERR_INVALID_IMAGE; }
    }
    *out = d;
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_new_decoder", return LILLIPUT_ERR_DEVICE)

void lilliput_decoder_close(lilliput_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<Decoder*>(dd);
    if (!d) return;
    if (d->gif) giflib_decoder_release(d->gif);
    if (d->webp) webp_decoder_release(d->webp);
    if (d->dec) opencv_decoder_release(d->dec);
    opencv_mat_release(d->mat);
    delete d;
}
LP_ABI_CATCH("lilliput_decoder_close", return)

int lilliput_decoder_header(lilliput_decoder dd, int* width, int* height, int* pixel_type, int* orientation, int* num_frames, int* content_length)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    Header h;
    int e = decoder_header(static_cast<Decoder*>(dd), &h, content_length != nullptr);
    if (e) return e;
    if (width) *width = h.width;
    if (height) *height = h.height;
    if (pixel_type) *pixel_type = h.pixel_type;
    if (orientation) *orientation = h.orientation;
    if (num_frames) *num_frames = h.num_frames;
    if (content_length) *content_length = h.content_length;
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_decoder_header", return LILLIPUT_ERR_DEVICE)

const char* lilliput_decoder_description(lilliput_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<Decoder*>(dd);
    if (d->kind == Decoder::PIXELS) return "PIXELS";
    return d->kind == Decoder::GIF ? "GIF" : d->kind == Decoder::WEBP ? "WEBP" : opencv_decoder_get_description(d->dec); // giflib.go:107-109, webp.go:67-69
}
LP_ABI_CATCH("lilliput_decoder_description", return nullptr)

int lilliput_decoder_icc(lilliput_decoder dd, void* dst, size_t cap) // openCVDecoder.ICC, opencv.go:697-712; gifDecoder.ICC is empty (giflib.go:122-124)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<Decoder*>(dd);
    if (!d || !dst || d->kind == Decoder::GIF || d->kind == Decoder::PIXELS) return 0;
    if (d->kind == Decoder::WEBP) return (int)webp_decoder_get_icc(d->webp, dst, cap); // webp.go:99-103
    const char* desc = opencv_decoder_get_description(d->dec);
    if (desc && strcmp(desc, "JPEG") == 0) return opencv_decoder_get_jpeg_icc((void*)d->buf, d->len, dst, cap);
    if (desc && strcmp(desc, "PNG") == 0) return opencv_decoder_get_png_icc((void*)d->buf, d->len, dst, cap);
    return 0;
}
LP_ABI_CATCH("lilliput_decoder_icc", return 0)

// gifDecoder.LoopCount / FrameCount / Duration / BackgroundColor (giflib.go:126-178): {loop_count, frame_count, duration_ms, background ARGB}
int lilliput_decoder_animation_info(lilliput_decoder dd, int out[4])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<Decoder*>(dd);
    if (d && d->kind == Decoder::WEBP) { // webp.go:71-73, 105-111: Duration, BackgroundColor, LoopCount
        out[0] = (int)webp_decoder_get_loop_count(d->webp);
        out[1] = webp_decoder_get_num_frames(d->webp);
        out[2] = webp_decoder_get_total_duration(d->webp);
        out[3] = (int)webp_decoder_get_bg_color(d->webp);
        return LILLIPUT_OK;
    }
    if (!d || d->kind != Decoder::GIF) return LILLIPUT_ERR_UNSUPPORTED;
    Header h;
    (void)decoder_header(d, &h);
    out[0] = d->anim.loop_count;
    out[1] = d->anim.frame_count;
    out[2] = d->anim.duration_ms;
    out[3] = (int)(((uint32_t)(uint8_t)d->anim.bg_red << 16) | ((uint32_t)(uint8_t)d->anim.bg_green << 8) | (uint32_t)(uint8_t)d->anim.bg_blue | ((uint32_t)(uint8_t)d->anim.bg_alpha << 24));
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_decoder_animation_info", return LILLIPUT_ERR_DEVICE)

lilliput_image_ops lilliput_new_image_ops(int max_size) // ops.go:83-91
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto o = new ImageOps();
    o->frames[0].init(max_size, max_size);
    o->frames[1].init(max_size, max_size);
    if (!o->frames[0].buf || !o->frames[1].buf) { o->frames[0].destroy(); o->frames[1].destroy(); delete o; return nullptr; }
    return o;
}
LP_ABI_CATCH("lilliput_new_image_ops", return nullptr)

void lilliput_image_ops_clear(lilliput_image_ops oo) // ops.go:111-117
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto o = static_cast<ImageOps*>(oo);
    o->frames[0].clear();
    o->frames[1].clear();
    if (o->have_composite) o->composite.clear();
}
LP_ABI_CATCH("lilliput_image_ops_clear", return)
void lilliput_image_ops_close(lilliput_image_ops oo)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto o = static_cast<ImageOps*>(oo);
    if (!o) return;
    o->frames[0].destroy();
    o->frames[1].destroy();
    o->drop_composite();
    delete o;
}
LP_ABI_CATCH("lilliput_image_ops_close", return)

// What the batched path (lp_batch.cpp) serves exactly like the loop below would: one frame of a JPEG file, decoded once, Fit or Resize, JPEG out,
// inside the bounds this ImageOps was built with (opencv.go:250-267 resizeMat answers ErrBufTooSmall beyond them -- left to the direct route).
static bool coalescible(const ImageOps* o, const Decoder* d, const Header& hdr, const lilliput_image_options* opt)
{
    if (d->kind != Decoder::OPENCV || d->has_decoded || hdr.num_frames != 1 || !d->buf || !d->len) return false;
    const char* desc = opencv_decoder_get_description(d->dec);
    if (!desc || strcmp(desc, "JPEG") != 0) return false;
    const std::string ext = lower(opt->file_type);
    if (ext != ".jpeg" && ext != ".jpg" && ext != ".jpe") return false;
    if (opt->resize_method != LILLIPUT_OPS_FIT && opt->resize_method != LILLIPUT_OPS_RESIZE) return false;
    if (opt->encode_options_len && !opt->encode_options) return false;
    for (size_t i = 0; i + 1 < opt->encode_options_len; i += 2)
        if (opt->encode_options[i] == CV_IMWRITE_JPEG_QUALITY && opt->encode_options[i + 1] <= 0) return false; // the batch options spell "default" as 0
    const size_t cn = (size_t)opencv_type_channels(hdr.pixel_type);
    int in_w = hdr.width, in_h = hdr.height;
    if (opt->normalize_orientation && lp_swaps_axes(hdr.orientation)) { in_w = hdr.height; in_h = hdr.width; }
    int nw = opt->width, nh = opt->height;
    if (opt->resize_method == LILLIPUT_OPS_FIT) lp_calculate_expected_size(in_w, in_h, opt->width, opt->height, &nw, &nh);
    if (nw < 1) nw = 1;
    if (nh < 1) nh = 1;
    return (size_t)hdr.width * hdr.height * cn <= o->frames[0].buf_len && (size_t)nw * nh * cn <= o->frames[0].buf_len;
}

int lilliput_image_ops_transform(lilliput_image_ops oo, lilliput_decoder dd, const lilliput_image_options* opt, void* dst, size_t dst_cap, size_t* dst_len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto o = static_cast<ImageOps*>(oo);
    auto d = static_cast<Decoder*>(dd);
    *dst_len = 0;
    if (!o || !d || !opt || !dst || dst_cap == 0) return LILLIPUT_ERR_INVALID_IMAGE;
    // The framebuffers are private to ImageOps (ops.go:67-81): nothing reads their pixels on the host, so decoded, composited
    // and resized frames stay on the device for the duration of the call.
    struct LazyScope { int prev = lp_lazy_host_scope(1); ~LazyScope() { lp_lazy_host_scope(prev); } } lazy_scope;
    LpEagerScope eager; // Part C brings its own strategy (the whole call is handed to the batched path below when others are in flight): the opencv_* calls it makes are executed, not recorded
    LpTransformInFlight in_flight;
    // initializeTransform (ops.go:483-546)
    Header hdr;
    int e = decoder_header(d, &hdr);
    if (e) return e;
    // Several Transform calls in flight at once (a service's goroutines, README.md:82-85): the ones the batched path answers with the
    // same bytes -- static JPEG source, JPEG output, Fit / Resize -- share its launches instead of each paying its own (lp_coalesce.h).
    // Whatever the batch does not answer with LILLIPUT_OK runs below as if nothing had happened.
    // (Deferred Part A counts its own requests in flight, lp_abi_opencv.cpp; here it is the Transform calls of any kind, LILLIPUT_HIP_COALESCE.)
    const bool can_share = coalescible(o, d, hdr, opt);
    // With company the call goes through the dispatchers; alone -- or among at most LILLIPUT_HIP_DEFER_INLINE_MAX calls while the dispatchers are
    // idle, the rule of deferred Part A -- it runs as a resident batch of one on this thread: the fused planes -> thumbnail kernels instead of a
    // 48 MB frame and its resize (one caller, 4096 x 4096 source: 0.62 -> 0.7 k images/s; four callers 1.0 -> 1.x k)
    const bool with_company = can_share && lp_coalesce_wanted(in_flight.now);
    const bool on_this_thread = can_share && lp_lone_batch_enabled() && !lp_coalesce_suppressed() &&
                                (!with_company || (in_flight.now <= lp_lone_inline_max() && lp_coalesce_busy() == 0));
    if (with_company || on_this_thread) {
        lilliput_batch_options bo;
        memset(&bo, 0, sizeof(bo));
        bo.width = opt->width; bo.height = opt->height;
        bo.resize_method = opt->resize_method;
        bo.normalize_orientation = opt->normalize_orientation;
        for (size_t i = 0; i + 1 < opt->encode_options_len; i += 2) { // as opencv_encoder_write reads them: the last one wins
            if (opt->encode_options[i] == CV_IMWRITE_JPEG_QUALITY) bo.jpeg_quality = std::min(100, opt->encode_options[i + 1]);
            else if (opt->encode_options[i] == CV_IMWRITE_JPEG_PROGRESSIVE) bo.jpeg_progressive = opt->encode_options[i + 1] != 0;
        }
        size_t n = 0;
        if (!on_this_thread ? lp_coalesce_transform(lp_current_device(), d->buf, d->len, dst, dst_cap, bo, &n)
                            : lp_lone_batch_transform(lp_current_device(), d->buf, d->len, dst, dst_cap, bo, &n) == LILLIPUT_OK) {
            d->has_decoded = true; // openCVDecoder.DecodeTo has run once (opencv.go:816-839): a second Transform on this decoder answers EOF
            *dst_len = n;
            return LILLIPUT_OK;
        }
    }
    LpEngineLease lease; // the whole Transform on one engine, one stream: the ABI calls below nest inside it
    struct CompositeScope { ImageOps* o; ~CompositeScope() { o->drop_composite(); } } composite_scope{o}; // ops.go:353-358
    // ICC override for HDR -> SDR (ops.go:489-498): ForceSdr + a source profile whose 'cicp' tag names PQ / HLG -> tag the output sRGB
    const uint8_t* icc_override = nullptr;
    size_t icc_override_len = 0;
    static thread_local std::vector<uint8_t> icc(32768); // ICCProfileBufferSize, lilliput.go:13-16
    if (opt->force_sdr) {
        const int n = lilliput_decoder_icc(dd, icc.data(), icc.size());
        if (n > 0 && is_hdr_transfer_function(icc.data(), (size_t)n)) icc_override = lilliput_hip_srgb_icc_profile(&icc_override_len);
    }
    // Container-signalled colour (ops.go:500-541): a PNG's cICP chunk. An HDR transfer function (PQ, HLG) makes every decoded frame
    // go through the tone map right after decode (ops.go:154-165); an SDR cICP is signalling only: it is carried over to a PNG output
    // (applyOutputCICP ops.go:306-333) and, for the outputs that embed a profile, replaces the source's ICC by one for its primaries.
    uint8_t out_cicp[4] = {0, 0, 0, 0};
    bool have_out_cicp = false, tonemap = false;
    uint8_t tm_transfer = 0, tm_primaries = 0;
    std::string ext = lower(opt->file_type);
    if (d->kind == Decoder::OPENCV) {
        const char* desc = opencv_decoder_get_description(d->dec);
        uint8_t prim = 0, transfer = 0, matrix = 0, range = 0;
        if (desc && strcmp(desc, "PNG") == 0 && d->len && opencv_decoder_get_png_cicp((void*)d->buf, d->len, &prim, &transfer, &matrix, &range)) {
            if (cicp_is_hdr_transfer(transfer)) {
                tonemap = true; tm_transfer = transfer; tm_primaries = prim;
            } else {
                out_cicp[0] = prim; out_cicp[1] = transfer; out_cicp[2] = matrix; out_cicp[3] = range;
                have_out_cicp = true;
                if (ext == ".webp" || ext == ".avif") { // outputTagsICC, ops.go:297-304
                    size_t n = 0;
                    const uint8_t* syn = cicp_get_icc_profile(prim, &n);
                    if (icc_header_is_sane(syn, n)) { icc_override = syn; icc_override_len = n; }
                }
            }
        }
    }
    // NewEncoder (lilliput.go:180-202) -> newOpenCVEncoder (opencv.go:847-870)
    if (ext == ".avif" || ext == ".mp4" || ext == ".webm") return LILLIPUT_ERR_UNSUPPORTED;
    Encoder enc;
    enc.dst_buf = (uint8_t*)dst;
    enc.dst_cap = dst_cap;
    if (ext == ".gif") { // newGifEncoder (giflib.go:239-256): palettes cannot be invented, the source has to be a GIF
        if (d->kind != Decoder::GIF) return LILLIPUT_ERR_GIF_ENCODER_NEEDS_DECODER;
        enc.kind = Encoder::GIF;
        enc.gif = giflib_encoder_create(dst, dst_cap);
        if (!enc.gif) return LILLIPUT_ERR_BUF_TOO_SMALL;
    } else if (ext == ".webp") { // newWebpEncoder, webp.go:174-212: the source's ICC profile (when its header is sane), background colour and loop count
        enc.kind = Encoder::WEBP;
        int icc_len;
        if (icc_override_len) { // EncodeConfig.ICCOverride, webp.go:181-185
            icc_len = (int)std::min(icc_override_len, icc.size());
            memcpy(icc.data(), icc_override, (size_t)icc_len);
        } else {
            icc_len = lilliput_decoder_icc(dd, icc.data(), icc.size());
        }
        if (icc_len > 0 && !icc_header_is_sane(icc.data(), (size_t)icc_len)) icc_len = 0; // ICCHeaderIsSane, webp.go:192-194
        if (icc_len < 0) icc_len = 0;
        uint32_t bg = 0xFFFFFFFFu; // openCVDecoder.BackgroundColor (opencv.go:665-667) and gifDecoder's (giflib.go:161-178)
        int loops = 0;
        if (d->kind == Decoder::WEBP) { bg = webp_decoder_get_bg_color(d->webp); loops = (int)webp_decoder_get_loop_count(d->webp); }
        else if (d->kind == Decoder::GIF) { int ai[4]; (void)lilliput_decoder_animation_info(dd, ai); bg = (uint32_t)ai[3]; loops = ai[0]; }
        enc.webp = webp_encoder_create(dst, dst_cap, icc_len ? icc.data() : nullptr, (size_t)icc_len, bg, loops);
        if (!enc.webp) return LILLIPUT_ERR_BUF_TOO_SMALL;
    } else if (ext == ".thumbhash") { // newThumbhashEncoder, thumbhash.go:21-32
        enc.kind = Encoder::THUMBHASH;
        enc.th = thumbhash_encoder_create(dst, dst_cap);
        if (!enc.th) return LILLIPUT_ERR_BUF_TOO_SMALL;
    } else if (ext == ".bgra-frames") {
        // Test access, not a reference format: an "animated encoder" that keeps every frame it is handed as
        // [u32 width][u32 height][u32 channels][u32 duration_ms][pixels] and, like the reference's animated encoders
        // (webp.go:220-256, giflib.go:259-292), returns content only when flushed with a nil frame.
        enc.kind = Encoder::RAW_FRAMES;
    } else {
        enc.dst = opencv_mat_create_empty_from_data((int)dst_cap, dst);
        if (!enc.dst) return LILLIPUT_ERR_BUF_TOO_SMALL;
        enc.enc = opencv_encoder_create(opt->file_type, enc.dst);
        if (!enc.enc) { opencv_mat_release(enc.dst); return LILLIPUT_ERR_INVALID_IMAGE; }
    }
    struct Guard { Encoder& e; ~Guard() { if (e.enc) opencv_encoder_release(e.enc); if (e.dst) opencv_mat_release(e.dst); if (e.gif) giflib_encoder_release(e.gif); if (e.th) thumbhash_encoder_release(e.th); if (e.webp) webp_encoder_release(e.webp); } } guard{enc};
    // newOpenCVEncoder asks the decoder for its ICC profile on every Transform (opencv.go:863); the JPEG writer then drops it
    // (cv::imencode has no ICC channel), so the read is kept for its cost profile only.
    if (enc.kind == Encoder::OPENCV) {
        static thread_local std::vector<uint8_t> icc_scratch(32768); // ICCProfileBufferSize, lilliput.go:13-16
        (void)lilliput_decoder_icc(dd, icc_scratch.data(), icc_scratch.size());
    }

    // Encoder.Encode: *n > 0 = finished content, 0 with LILLIPUT_OK = "give me another frame"
    auto encode = [&](Framebuffer* f, size_t* n) -> int {
        *n = 0;
        if (enc.kind == Encoder::RAW_FRAMES) {
            if (enc.flushed) return LILLIPUT_ERR_EOF;
            if (!f) { enc.flushed = true; *n = enc.raw_len; return enc.raw_len ? LILLIPUT_OK : LILLIPUT_ERR_EOF; }
            const int cn = opencv_type_channels(f->pixel_type);
            const size_t px = (size_t)f->width * f->height * cn;
            if (enc.raw_len + 16 + px > enc.dst_cap) return LILLIPUT_ERR_BUF_TOO_SMALL;
            if (lilliput_hip_mat_sync_host(f->mat)) return LILLIPUT_ERR_DEVICE;
            const uint32_t head[4] = {(uint32_t)f->width, (uint32_t)f->height, (uint32_t)cn, (uint32_t)(f->duration / 1000000ll)};
            memcpy(enc.dst_buf + enc.raw_len, head, 16);
            memcpy(enc.dst_buf + enc.raw_len + 16, opencv_mat_get_data(f->mat), px);
            enc.raw_len += 16 + px;
            return LILLIPUT_OK;
        }
        if (enc.kind == Encoder::THUMBHASH) { // thumbhashEncoder.Encode, thumbhash.go:37-48
            if (!f) return LILLIPUT_ERR_EOF;
            const int len = thumbhash_encoder_encode(enc.th, f->mat);
            if (len <= 0) return LILLIPUT_ERR_INVALID_IMAGE;
            *n = (size_t)len;
            return LILLIPUT_OK;
        }
        if (enc.kind == Encoder::WEBP) { // webpEncoder.Encode, webp.go:214-256
            if (enc.flushed) return LILLIPUT_ERR_EOF;
            if (!f) {
                const size_t len = webp_encoder_flush(enc.webp);
                if (!len) return LILLIPUT_ERR_INVALID_IMAGE;
                enc.flushed = true;
                *n = len;
                return LILLIPUT_OK;
            }
            const int delay_ms = (int)(f->duration / 1000000ll);
            if (!webp_encoder_write(enc.webp, f->mat, opt->encode_options, opt->encode_options_len, delay_ms, f->blend, f->dispose, 0, 0)) return LILLIPUT_ERR_INVALID_IMAGE;
            return LILLIPUT_OK;
        }
        if (enc.kind == Encoder::GIF) { // gifEncoder.Encode, giflib.go:259-292
            if (enc.flushed) return LILLIPUT_ERR_EOF;
            if (!f) {
                if (!giflib_encoder_flush(enc.gif, d->gif)) return LILLIPUT_ERR_INVALID_IMAGE;
                enc.flushed = true;
                *n = (size_t)giflib_encoder_get_output_length(enc.gif);
                return *n ? LILLIPUT_OK : LILLIPUT_ERR_EOF;
            }
            if (enc.gif_frame_index == 0) (void)giflib_encoder_init(enc.gif, d->gif, f->width, f->height);
            if (!giflib_encoder_encode_frame(enc.gif, d->gif, f->mat)) return LILLIPUT_ERR_INVALID_IMAGE;
            enc.gif_frame_index++;
            return LILLIPUT_OK;
        }
        // opencv.go:872-900
        if (!f) return LILLIPUT_ERR_EOF;
        if (!opencv_encoder_write(enc.enc, f->mat, opt->encode_options, opt->encode_options_len)) return LILLIPUT_ERR_INVALID_IMAGE;
        if (opencv_mat_get_data(enc.dst) != (void*)enc.dst_buf) return LILLIPUT_ERR_BUF_TOO_SMALL;
        *n = (size_t)opencv_mat_get_height(enc.dst);
        if (have_out_cicp && *n >= 8 && memcmp(enc.dst_buf, "\x89PNG\r\n\x1a\n", 8) == 0) // applyOutputCICP: PNG outputs only
            *n = opencv_png_insert_cicp(enc.dst_buf, *n, enc.dst_cap, out_cicp[0], out_cicp[1], out_cicp[2], out_cicp[3]);
        return LILLIPUT_OK;
    };
    auto encode_empty = [&]() -> int { // ops.go:286-292 encodeEmpty + the caller returning its result
        size_t n = 0;
        int e2 = encode(nullptr, &n);
        if (e2) return e2;
        *dst_len = n;
        return LILLIPUT_OK;
    };

    int frame_count = 0;
    int64_t duration = 0;
    const int64_t timeout_at = now_ns() + opt->encode_timeout_ns;
    const bool animated = hdr.num_frames > 1;                 // ImageHeader.IsAnimated, opencv.go:189-191
    const bool has_alpha = opencv_type_channels(hdr.pixel_type) == 4; // ImageHeader.HasAlpha, opencv.go:194-196
    for (;;) { // ops.go:371-443
        e = decoder_decode_to(d, o->active());
        bool empty_frame = false;
        if (e) {
            if (e != LILLIPUT_ERR_EOF) return e;
            empty_frame = true;
        }
        // ImageOps.decode (ops.go:154-165): HDR pixels become SDR before any resize or composite
        if (tonemap && !e && lilliput_hip_mat_tonemap(o->active()->mat, tm_transfer, tm_primaries)) return LILLIPUT_ERR_DEVICE;
        duration += o->active()->duration;
        if (opt->max_encode_duration_ns != 0 && duration > opt->max_encode_duration_ns) {
            e = skip_to_end(d);
            if (e != LILLIPUT_ERR_EOF) return e;
            return encode_empty();
        }
        o->active()->orientation_transform(hdr.orientation); // unconditional (ops.go:392)
        bool swapped = false;
        if (!empty_frame) { // transformCurrentFrame (ops.go:449-472)
            if (!(opt->resize_method == LILLIPUT_OPS_NO_RESIZE && !animated)) {
                int in_w = hdr.width, in_h = hdr.height;
                if (opt->normalize_orientation && lp_swaps_axes(hdr.orientation)) { in_w = hdr.height; in_h = hdr.width; }
                int out_w = opt->width, out_h = opt->height;
                if (opt->resize_method == LILLIPUT_OPS_NO_RESIZE) { out_w = in_w; out_h = in_h; }
                if (opt->resize_method != LILLIPUT_OPS_FIT && opt->resize_method != LILLIPUT_OPS_NO_RESIZE && opt->resize_method != LILLIPUT_OPS_RESIZE)
                    return LILLIPUT_ERR_INVALID_IMAGE;
                const bool is_fit = opt->resize_method != LILLIPUT_OPS_RESIZE;
                int nw = out_w, nh = out_h;
                if (is_fit) lp_calculate_expected_size(in_w, in_h, out_w, out_h, &nw, &nh); // ops.go:171
                if (animated) { // ops.go:173-197, 208-229: composite -> resize the composite -> dispose
                    if ((e = o->setup_animated(in_w, in_h, has_alpha))) return e;
                    if ((e = o->apply_blend())) return e;
                    e = is_fit ? o->composite.fit(nw, nh, o->secondary()) : o->composite.resize_to(out_w, out_h, o->secondary());
                    if (e) return e;
                    if ((e = o->apply_dispose())) return e;
                } else {
                    e = is_fit ? o->active()->fit(nw, nh, o->secondary()) : o->active()->resize_to(out_w, out_h, o->secondary());
                    if (e) return e;
                }
                o->copy_props_and_swap();
                swapped = true;
            }
        }
        size_t n = 0;
        e = encode(empty_frame ? nullptr : o->active(), &n);
        if (e) return e; // Encode(nil) on the OpenCV encoder: io.EOF (opencv.go:873-875)
        if (n) { *dst_len = n; return LILLIPUT_OK; }
        frame_count++;
        if (opt->disable_animated_output) return encode_empty();
        if (opt->max_encode_frames != 0 && frame_count == opt->max_encode_frames) {
            e = skip_to_end(d);
            if (e != LILLIPUT_ERR_EOF) return e;
            return encode_empty();
        }
        if (now_ns() > timeout_at) return LILLIPUT_ERR_ENCODE_TIMEOUT;
        if (swapped) o->swap();
    }
}
LP_ABI_CATCH("lilliput_image_ops_transform", return LILLIPUT_ERR_DEVICE)

} // extern "C"
