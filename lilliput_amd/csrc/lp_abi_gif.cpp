// lp_abi_gif.cpp -- the giflib_decoder_* half of the reference's giflib.hpp C ABI (/root/reference/giflib.hpp:9-52,
// implemented there by giflib.cpp:83-724 and :1308-1431 on top of giflib 5.2.2). The container walk and the LZW
// decoder run on the host (lp_gif.cpp, a serial bit-stream like the reference's); the frame compositing -- background,
// disposal of the previous frame, restore-to-previous snapshot, palette lookup with transparency -- runs on the device
// (k_gif_frame) on a canvas that stays resident in HBM for the life of the decoder. After every frame the canvas is
// handed to the caller's Mat as its device mirror (and written to its host buffer unless lazy write-back is on), so the
// next stage (Framebuffer.Fit -> opencv_mat_resize) reads it without another upload.
// Go callers: gifDecoder in /root/reference/giflib.go:56-242.
#include <limits.h>
#include <stdio.h>
#include <string.h>

#include "lp_abi.h"
#include "lp_abi_gif.h"
#include "lp_gif.h"
#include "lp_abi_guard.h"

struct giflib_decoder_struct {
    LpMat* src = nullptr; // the 1 x N CV_8U Mat over the caller's GIF bytes
    LpGifReader gif;
    std::vector<uint8_t> pixels; // colour indices of the current frame
    int prev_disposal = 0, prev_delay = 0, prev_left = 0, prev_top = 0, prev_width = 0, prev_height = 0;
    uint8_t bg[4] = {0, 0, 0, 0}; // B, G, R, A
    bool have_read_first_frame = false, seek_clear_extensions = false;
    int image_count = 0; // gif->ImageCount: DGifGetImageHeader does not count frames, so this stays 0 like the reference's
    // device state
    std::shared_ptr<LpDevBlock> canvas, saved;
};

const LpGifReader& lp_gif_reader(giflib_decoder d) { return d->gif; }
int lp_gif_bg_alpha(giflib_decoder d) { return d->bg[3]; }

namespace {
enum { DISPOSAL_UNSPECIFIED = 0, DISPOSE_DO_NOT = 1, DISPOSE_BACKGROUND = 2, DISPOSE_PREVIOUS = 3 };
const int GRAPHICS_EXT = 0xF9, APPLICATION_EXT = 0xFF;

bool read_extensions(giflib_decoder d) // giflib.cpp:203-241
{
    int fn;
    const uint8_t* ext;
    if (d->gif.get_extension(&fn, &ext) == LP_GIF_ERROR) return false;
    if (ext) d->gif.ext_blocks.push_back(LpGifExtBlock{fn, std::vector<uint8_t>(ext + 1, ext + 1 + ext[0])});
    while (ext) {
        if (d->gif.get_extension_next(&ext) == LP_GIF_ERROR) return false;
        if (ext) d->gif.ext_blocks.push_back(LpGifExtBlock{0, std::vector<uint8_t>(ext + 1, ext + 1 + ext[0])});
    }
    return true;
}

void frame_gcb(const LpGifReader& g, LpGifGcb* gcb) // giflib.cpp:243-264: the last well-formed graphic control extension wins
{
    *gcb = LpGifGcb();
    for (const LpGifExtBlock& b : g.ext_blocks)
        if (b.function == GRAPHICS_EXT) (void)LpGifReader::extension_to_gcb(b.bytes.size(), b.bytes.data(), gcb);
}

void set_frame_gcb(LpGifReader& g, const LpGifGcb& gcb) // giflib.cpp:266-284 (EGifGCBToExtension into every GCE of at least 4 bytes)
{
    for (LpGifExtBlock& b : g.ext_blocks)
        if (b.function == GRAPHICS_EXT && b.bytes.size() >= 4) {
            b.bytes[0] = (uint8_t)(((gcb.disposal & 7) << 2) | (gcb.user_input ? 2 : 0) | (gcb.transparent != -1 ? 1 : 0));
            b.bytes[1] = (uint8_t)(gcb.delay & 0xff);
            b.bytes[2] = (uint8_t)((gcb.delay >> 8) & 0xff);
            b.bytes[3] = (uint8_t)gcb.transparent; // EGifGCBToExtension stores the field as it is: "none" (-1) becomes 0xff
        }
}

giflib_decoder_frame_state seek_next_frame(giflib_decoder d) // giflib.cpp:286-324
{
    if (d->seek_clear_extensions) {
        d->gif.ext_blocks.clear();
        d->seek_clear_extensions = false;
    }
    int type;
    do {
        if (d->gif.get_record_type(&type) == LP_GIF_ERROR) return giflib_decoder_error;
        if (type == LP_GIF_REC_IMAGE) return giflib_decoder_have_next_frame;
        if (type == LP_GIF_REC_EXTENSION && !read_extensions(d)) return giflib_decoder_error;
    } while (type != LP_GIF_REC_TERMINATE);
    return giflib_decoder_eof;
}

// giflib.cpp:585-627: the colour the canvas starts from and "dispose to background" paints
void background_color(const LpGifReader& g, const LpGifGcb& gcb, uint8_t* r, uint8_t* gg, uint8_t* b, uint8_t* a)
{
    const bool in_map = g.global_map.count > 0 && g.sbackground >= 0 && g.sbackground < g.global_map.count;
    *r = in_map ? g.global_map.rgb[g.sbackground][0] : 255;
    *gg = in_map ? g.global_map.rgb[g.sbackground][1] : 255;
    *b = in_map ? g.global_map.rgb[g.sbackground][2] : 255;
    *a = gcb.transparent != -1 ? 0 : 255; // a first frame with a transparent index starts from a see-through canvas
}

void clip_prev(giflib_decoder d, int bw, int bh, int* x, int* y, int* w, int* h) // giflib.cpp:408-436
{
    int l = d->prev_left, t = d->prev_top, pw = d->prev_width, ph = d->prev_height;
    if (l < 0) { pw += l; l = 0; }
    if (t < 0) { ph += t; t = 0; }
    if (l + pw > bw) pw = bw - l;
    if (t + ph > bh) ph = bh - t;
    *x = l; *y = t; *w = pw < 0 ? 0 : pw; *h = ph < 0 ? 0 : ph;
}
} // namespace

extern "C" {

giflib_decoder giflib_decoder_create(const opencv_mat buf) // giflib.cpp:103-158
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    if (!buf) return nullptr;
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)buf));
    auto d = new giflib_decoder_struct();
    d->src = m;
    const size_t len = (size_t)m->rows * (size_t)m->cols; // cv::Mat::total() of the 1-row byte matrix
    if (!m->data || !d->gif.open(m->data, len) || d->gif.swidth <= 0 || d->gif.sheight <= 0) { delete d; return nullptr; }
    return d;
}
LP_ABI_CATCH("giflib_decoder_create", return nullptr)

int giflib_decoder_get_width(const giflib_decoder d) { return d->gif.swidth; }
int giflib_decoder_get_height(const giflib_decoder d) { return d->gif.sheight; }
int giflib_decoder_get_num_frames(const giflib_decoder d) { return d->image_count; }
int giflib_decoder_get_frame_width(const giflib_decoder d) { return d->gif.width; }
int giflib_decoder_get_frame_height(const giflib_decoder d) { return d->gif.height; }
int giflib_decoder_get_prev_frame_delay(const giflib_decoder d) { return d->prev_delay; }

int giflib_decoder_get_prev_frame_disposal(const giflib_decoder d) // giflib.cpp:190-202
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    switch (d->prev_disposal) {
    case DISPOSE_BACKGROUND: return GIF_DISPOSE_BACKGROUND;
    case DISPOSE_PREVIOUS: return GIF_DISPOSE_PREVIOUS;
    default: return GIF_DISPOSE_NONE; // "do not dispose" and "unspecified"
    }
}
LP_ABI_CATCH("giflib_decoder_get_prev_frame_disposal", return 0)

void giflib_decoder_release(giflib_decoder d) { delete d; }

giflib_decoder_frame_state giflib_decoder_decode_frame_header(giflib_decoder d) // giflib.cpp:329-342
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    giflib_decoder_frame_state seek = seek_next_frame(d);
    if (seek == giflib_decoder_eof || seek == giflib_decoder_error) return seek;
    if (d->gif.get_image_header() == LP_GIF_ERROR) return giflib_decoder_error;
    return giflib_decoder_have_next_frame;
}
LP_ABI_CATCH("giflib_decoder_decode_frame_header", return giflib_decoder_error)

giflib_decoder_frame_state giflib_decoder_skip_frame(giflib_decoder d) // giflib.cpp:563-583
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    giflib_decoder_frame_state seek = giflib_decoder_decode_frame_header(d);
    if (seek != giflib_decoder_have_next_frame) return seek;
    const uint8_t* block;
    for (;;) {
        if (d->gif.get_code_next(&block) == LP_GIF_ERROR) return giflib_decoder_error;
        if (!block) break;
    }
    return giflib_decoder_have_next_frame;
}
LP_ABI_CATCH("giflib_decoder_skip_frame", return giflib_decoder_error)

// Host half of giflib_decoder_decode_frame (giflib.cpp:632-690): the frame's colour indices, de-interlaced.
static bool read_frame_indices(giflib_decoder d)
{
    LpGifReader& g = d->gif;
    if (g.width <= 0 || g.height <= 0) { fprintf(stderr, "encountered error, gif frame has negative or zero width or height\n"); return false; }
    // the reference's guards (giflib.cpp giflib_decoder_decode_frame): the pixel count must fit an int (get_line takes one) -- a
    // 65535 x 65535 frame does not -- and the buffer must be obtainable; nothing may unwind through the C ABI
    if (g.width > INT_MAX / g.height) { fprintf(stderr, "encountered error, gif frame too large\n"); return false; }
    const size_t image_size = (size_t)g.width * (size_t)g.height;
    try {
        if (image_size > d->pixels.size()) d->pixels.resize(image_size); // never shrinks: pixels past a shorter frame keep older values, as in the reference
    } catch (const std::bad_alloc&) { fprintf(stderr, "encountered error, could not allocate gif frame\n"); return false; }
    if (g.interlace) {
        static const int offset[4] = {0, 4, 2, 1}, jump[4] = {8, 8, 4, 2};
        for (int i = 0; i < 4; i++)
            for (int j = offset[i]; j < g.height; j += jump[i])
                if (g.get_line(d->pixels.data() + (size_t)j * g.width, g.width) == LP_GIF_ERROR) {
                    fprintf(stderr, "encountered error, could not rasterize gif line\n");
                    return false;
                }
    } else if (g.get_line(d->pixels.data(), (int)image_size) == LP_GIF_ERROR) {
        fprintf(stderr, "encountered error, could not rasterize gif\n");
        return false;
    }
    return true;
}

static void after_frame(giflib_decoder d, const LpGifGcb& gcb) // giflib.cpp:713-721
{
    d->prev_disposal = gcb.disposal;
    d->prev_delay = gcb.delay;
    d->prev_left = d->gif.left;
    d->prev_top = d->gif.top;
    d->prev_width = d->gif.width;
    d->prev_height = d->gif.height;
    d->have_read_first_frame = true;
    d->seek_clear_extensions = true;
}

bool giflib_decoder_decode_frame(giflib_decoder d, opencv_mat mat) // giflib.cpp:632-724 + render_frame :349-561
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!d || !m) return false;
    if (!read_frame_indices(d)) return false;
    LpGifReader& g = d->gif;
    LpGifGcb gcb;
    frame_gcb(g, &gcb);
    if (!d->have_read_first_frame) background_color(g, gcb, &d->bg[2], &d->bg[1], &d->bg[0], &d->bg[3]);

    // ---- render_frame
    const int bw = m->cols, bh = m->rows;
    if (m->type != CV_8UC4 || bw != g.swidth || bh != g.sheight) {
        // the reference indexes its snapshot (screen-sized) with the Mat's width: anything but a screen-sized BGRA Mat is out of contract
        fprintf(stderr, "lilliput_hip: giflib_decoder_decode_frame needs a %dx%d CV_8UC4 matrix\n", g.swidth, g.sheight);
        return false;
    }
    const LpGifColorMap& map = g.local_map.count ? g.local_map : g.global_map;
    if (!map.count) { fprintf(stderr, "encountered error, gif frame has no color map\n"); return false; }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) return false;
    const size_t canvas_bytes = (size_t)bw * bh * 4;
    if (!d->canvas) {
        d->canvas = lp_dev_alloc(canvas_bytes);
        d->saved = lp_dev_alloc(canvas_bytes);
        if (!d->canvas || !d->saved) return false;
        if (hipMemsetAsync(d->saved->p, 0, canvas_bytes, eng->stream()) != hipSuccess) return false; // prev_frame_bgra starts zeroed
    }
    int skip_left = g.left < 0 ? -g.left : 0, skip_top = g.top < 0 ? -g.top : 0;
    int skip_right = g.left + g.width > bw ? g.left + g.width - bw : 0, skip_bottom = g.top + g.height > bh ? g.top + g.height - bh : 0;
    LpGifFrameOp op;
    memset(&op, 0, sizeof(op));
    op.canvas.off = (uint64_t)(uintptr_t)d->canvas->p;
    op.canvas.w = (uint32_t)bw; op.canvas.h = (uint32_t)bh; op.canvas.stride = (uint32_t)bw * 4; op.canvas.cn = 4;
    op.saved_off = (uint64_t)(uintptr_t)d->saved->p;
    op.first = d->have_read_first_frame ? 0 : 1;
    op.dispose = !d->have_read_first_frame ? 0 : d->prev_disposal == DISPOSE_BACKGROUND ? 1 : d->prev_disposal == DISPOSE_PREVIOUS ? 2 : 0;
    clip_prev(d, bw, bh, &op.px, &op.py, &op.pw, &op.ph);
    op.fx = g.left + skip_left; op.fy = g.top + skip_top;
    op.fw = g.width - skip_left - skip_right; op.fh = g.height - skip_top - skip_bottom;
    op.skip_left = skip_left; op.skip_top = skip_top;
    op.raster_w = g.width;
    op.transparent = gcb.transparent;
    op.color_count = map.count;
    memcpy(op.bg, d->bg, 4);
    uint8_t pal[1024];
    memset(pal, 0, sizeof(pal));
    for (int i = 0; i < map.count; i++) { pal[4 * i] = map.rgb[i][2]; pal[4 * i + 1] = map.rgb[i][1]; pal[4 * i + 2] = map.rgb[i][0]; pal[4 * i + 3] = 255; }
    if (eng->gif_frame(op, d->pixels.data(), (size_t)g.width * g.height, pal)) { lp_set_error(eng->last_error()); return false; }
    // hand the canvas to the caller's Mat (a copy: the decoder keeps drawing on its own)
    m->dev = lp_dev_alloc(canvas_bytes);
    if (!m->dev) return false;
    m->dev_off = 0; m->dev_step = (size_t)bw * 4; m->dev_shared = false; m->host_stale = false;
    m->lazy.reset();
    if (hipMemcpyAsync(m->dev->p, d->canvas->p, canvas_bytes, hipMemcpyDeviceToDevice, eng->stream()) != hipSuccess) return false;
    m->dev_valid = true;
    if (!lp_mat_to_host(m, eng)) return false;
    if (eng->sync()) return false;

    // a partial frame becomes a full frame with see-through surroundings, so the encoder needs a transparent index (giflib.cpp:542-558)
    const bool partial = op.fh < bh || op.fw < bw || op.fx != 0 || op.fy != 0;
    if (partial && gcb.transparent == -1) {
        gcb.transparent = map.count - 1;
        set_frame_gcb(g, gcb);
    }
    after_frame(d, gcb);
    return true;
}
LP_ABI_CATCH("giflib_decoder_decode_frame", return false)

struct GifAnimationInfo giflib_decoder_get_animation_info(const giflib_decoder d) // giflib.cpp:1308-1431
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    GifAnimationInfo info = {1, 0, 255, 255, 255, 0, 0};
    LpGifReader g; // a second walk over the same bytes, independent of the decode position
    if (!d || !d->src || !g.open(d->src->data, (size_t)d->src->rows * (size_t)d->src->cols)) return info;
    bool found_loop = false, found_gcb = false;
    LpGifGcb first_gcb;
    first_gcb.transparent = 0; // "GraphicsControlBlock gcb = {}": colour index 0 counts as transparent when no GCE is ever seen
    int type;
    while (g.get_record_type(&type) == LP_GIF_OK) {
        if (type == LP_GIF_REC_EXTENSION) {
            int fn;
            const uint8_t* ext;
            if (g.get_extension(&fn, &ext) == LP_GIF_OK && ext) {
                if (fn == GRAPHICS_EXT) {
                    LpGifGcb fg;
                    fg.transparent = 0; fg.disposal = 0; fg.delay = 0; // what an unfilled struct would most plausibly hold; only read when the block is malformed
                    const bool ok = LpGifReader::extension_to_gcb(ext[0], ext + 1, &fg) == LP_GIF_OK;
                    (void)ok;
                    info.duration_ms += (info.frame_count > 0 && fg.delay < 2) ? 20 : fg.delay * 10;
                    if (!found_gcb) {
                        found_gcb = true;
                        first_gcb = fg;
                        uint8_t r, gg, b, a;
                        background_color(g, first_gcb, &r, &gg, &b, &a);
                        info.bg_red = r; info.bg_green = gg; info.bg_blue = b; info.bg_alpha = a;
                    }
                } else if (!found_loop && fn == APPLICATION_EXT && ext[0] >= 11 && memcmp(ext + 1, "NETSCAPE2.0", 11) == 0) {
                    if (g.get_extension_next(&ext) == LP_GIF_OK && ext && ext[0] >= 3 && ext[1] == 1) {
                        info.loop_count = ext[2] | (ext[3] << 8);
                        found_loop = true;
                    }
                }
                while (ext)
                    if (g.get_extension_next(&ext) != LP_GIF_OK) return info;
            }
        } else if (type == LP_GIF_REC_IMAGE) {
            info.frame_count++;
            if (g.get_image_header() != LP_GIF_OK) return info;
            const uint8_t* block;
            if (g.get_code_next(&block) == LP_GIF_ERROR) return info;
            while (block)
                if (g.get_code_next(&block) == LP_GIF_ERROR) return info;
        } else if (type == LP_GIF_REC_TERMINATE)
            return info;
    }
    if (!found_gcb) {
        uint8_t r, gg, b, a;
        background_color(g, first_gcb, &r, &gg, &b, &a);
        info.bg_red = r; info.bg_green = gg; info.bg_blue = b; info.bg_alpha = a;
    }
    return info;
}
LP_ABI_CATCH("giflib_decoder_get_animation_info", return GifAnimationInfo{})

// Test access (no device work): the host half of decode_frame -- colour indices, frame rectangle, GCB and palette of the frame
// whose header was just read. meta = {left, top, width, height, interlace, disposal, delay, transparent, color_count, local_map}.
int lilliput_hip_gif_read_frame(giflib_decoder d, uint8_t* indices, size_t cap, int meta[10], uint8_t palette_rgb[768])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!d) return -1;
    if (!read_frame_indices(d)) return -1;
    LpGifReader& g = d->gif;
    LpGifGcb gcb;
    frame_gcb(g, &gcb);
    const LpGifColorMap& map = g.local_map.count ? g.local_map : g.global_map;
    if (!map.count) return -1; // render_frame gives up before drawing: decode_frame reports failure
    const size_t n = (size_t)g.width * g.height;
    if (n > cap) return -2;
    memcpy(indices, d->pixels.data(), n);
    const int mm[10] = {g.left, g.top, g.width, g.height, g.interlace ? 1 : 0, gcb.disposal, gcb.delay, gcb.transparent, map.count, g.local_map.count ? 1 : 0};
    memcpy(meta, mm, sizeof(mm));
    if (palette_rgb) { memset(palette_rgb, 0, 768); memcpy(palette_rgb, map.rgb, (size_t)map.count * 3); }
    after_frame(d, gcb);
    return (int)n;
}
LP_ABI_CATCH("lilliput_hip_gif_read_frame", return -1)

} // extern "C"
