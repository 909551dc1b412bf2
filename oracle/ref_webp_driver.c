/* oracle/ref_webp_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 * Drives the REFERENCE's own prebuilt libwebp 1.5.0 / libwebpmux / libwebpdemux / libsharpyuv archives
 * (/root/reference/deps/linux/amd64/lib) the way /root/reference/webp.cpp drives them, so that the product's container walk
 * (lp_webp.cpp), its use of the system libwebp for the VP8 / VP8L payloads and its animation writer can be checked against the real
 * library: webp_decoder_create / webp_decoder_decode (webp.cpp:61-134, 302-362), the still encoder with the ICCP mux
 * (webp.cpp:501-577, 707-751) and -- to read animations back, the product's included -- libwebpdemux's WebPAnimDecoder.
 * Built by oracle/Makefile into oracle/_ref/librefwebp.so when the reference mount is present. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <webp/decode.h>
#include <webp/demux.h>
#include <webp/encode.h>
#include <webp/mux.h>

/* out: width, height, has_alpha, num_frames, total_duration, bgcolor, loop_count, icc_len. Returns 1 when webp_decoder_create would succeed. */
int ref_webp_info(const uint8_t* data, size_t len, uint32_t out[8])
{
    WebPData src = {data, len};
    WebPMux* mux = WebPMuxCreate(&src, 0);
    if (!mux) return 0;
    uint32_t flags;
    WebPMuxFrameInfo frame;
    WebPBitstreamFeatures ft;
    int ok = WebPMuxGetFeatures(mux, &flags) == WEBP_MUX_OK && WebPMuxGetFrame(mux, 1, &frame) == WEBP_MUX_OK;
    if (!ok) { WebPMuxDelete(mux); return 0; }
    ok = WebPGetFeatures(frame.bitstream.bytes, frame.bitstream.size, &ft) == VP8_STATUS_OK;
    WebPDataClear(&frame.bitstream);
    int w = 0, h = 0;
    if (!ok || WebPMuxGetCanvasSize(mux, &w, &h) != WEBP_MUX_OK) { WebPMuxDelete(mux); return 0; }
    uint32_t n = 0, dur = 0;
    do {
        n++;
        dur += (uint32_t)frame.duration;
        WebPDataClear(&frame.bitstream);
    } while (WebPMuxGetFrame(mux, n + 1, &frame) == WEBP_MUX_OK);
    uint32_t bg = 0xFFFFFFFFu, loops = 0;
    if (flags & ANIMATION_FLAG) {
        WebPMuxAnimParams ap;
        if (WebPMuxGetAnimationParams(mux, &ap) == WEBP_MUX_OK) { bg = ap.bgcolor; loops = (uint32_t)ap.loop_count; }
    } else
        dur = 0;
    WebPData icc = {NULL, 0};
    if (WebPMuxGetChunk(mux, "ICCP", &icc) != WEBP_MUX_OK) icc.size = 0;
    out[0] = (uint32_t)w; out[1] = (uint32_t)h; out[2] = (flags & ALPHA_FLAG) ? 1 : 0; out[3] = n; out[4] = dur; out[5] = bg; out[6] = loops; out[7] = (uint32_t)icc.size;
    WebPMuxDelete(mux);
    return 1;
}

size_t ref_webp_icc(const uint8_t* data, size_t len, uint8_t* out, size_t cap)
{
    WebPData src = {data, len};
    WebPMux* mux = WebPMuxCreate(&src, 0);
    if (!mux) return 0;
    WebPData icc = {NULL, 0};
    size_t n = 0;
    if (WebPMuxGetChunk(mux, "ICCP", &icc) == WEBP_MUX_OK && icc.size && icc.size <= cap) { memcpy(out, icc.bytes, icc.size); n = icc.size; }
    WebPMuxDelete(mux);
    return n;
}

/* Frame `index` (1-based) as webp_decoder_decode yields it: BGR / BGRA by the container's alpha flag, tightly packed.
 * meta: width, height, channels, duration, x_offset, y_offset, dispose, blend. Returns the byte count, -1 on failure. */
long ref_webp_decode_frame(const uint8_t* data, size_t len, int index, uint8_t* out, size_t cap, int meta[8])
{
    WebPData src = {data, len};
    WebPMux* mux = WebPMuxCreate(&src, 0);
    if (!mux) return -1;
    uint32_t flags = 0;
    int cw = 0, ch = 0;
    WebPMuxFrameInfo frame;
    long res = -1;
    if (WebPMuxGetFeatures(mux, &flags) == WEBP_MUX_OK && WebPMuxGetCanvasSize(mux, &cw, &ch) == WEBP_MUX_OK && WebPMuxGetFrame(mux, (uint32_t)index, &frame) == WEBP_MUX_OK) {
        WebPBitstreamFeatures ft;
        if (WebPGetFeatures(frame.bitstream.bytes, frame.bitstream.size, &ft) == VP8_STATUS_OK) {
            const int cn = (flags & ALPHA_FLAG) ? 4 : 3, row = ft.width * cn;
            const size_t bufsz = (size_t)cw * ch * 4; /* the reference decodes into a canvas-sized buffer: larger frames fail */
            uint8_t* buf = (uint8_t*)malloc(bufsz ? bufsz : 1);
            uint8_t* r = cn == 4 ? WebPDecodeBGRAInto(frame.bitstream.bytes, frame.bitstream.size, buf, bufsz, row)
                                 : WebPDecodeBGRInto(frame.bitstream.bytes, frame.bitstream.size, buf, bufsz, row);
            const size_t need = (size_t)row * ft.height;
            if (r && need <= cap) {
                memcpy(out, buf, need);
                meta[0] = ft.width; meta[1] = ft.height; meta[2] = cn; meta[3] = frame.duration; meta[4] = frame.x_offset; meta[5] = frame.y_offset;
                meta[6] = (int)frame.dispose_method; meta[7] = (int)frame.blend_method;
                res = (long)need;
            }
            free(buf);
        }
        WebPDataClear(&frame.bitstream);
    }
    WebPMuxDelete(mux);
    return res;
}

/* The reference's still-image writer: WebPEncode(Lossless)BGR(A) by quality (> 100 = lossless), WebPMuxSetImage, optional ICCP, assemble. */
size_t ref_webp_encode_still(const uint8_t* px, int w, int h, int cn, float quality, const uint8_t* icc, size_t icc_len, uint8_t* out, size_t cap)
{
    uint8_t* pic = NULL;
    size_t size;
    const float q = quality < 1.f ? 1.f : quality;
    if (q > 100.f) size = cn == 3 ? WebPEncodeLosslessBGR(px, w, h, w * cn, &pic) : WebPEncodeLosslessBGRA(px, w, h, w * cn, &pic);
    else size = cn == 3 ? WebPEncodeBGR(px, w, h, w * cn, q, &pic) : WebPEncodeBGRA(px, w, h, w * cn, q, &pic);
    if (!size) return 0;
    WebPMux* mux = WebPMuxNew();
    WebPData d = {pic, size};
    size_t n = 0;
    if (WebPMuxSetImage(mux, &d, 1) == WEBP_MUX_OK) {
        if (icc && icc_len) { WebPData ic = {icc, icc_len}; WebPMuxSetChunk(mux, "ICCP", &ic, 1); }
        WebPData o = {NULL, 0};
        if (WebPMuxAssemble(mux, &o) == WEBP_MUX_OK) {
            if (o.size <= cap) { memcpy(out, o.bytes, o.size); n = o.size; }
            WebPDataClear(&o);
        }
    }
    WebPFree(pic);
    WebPMuxDelete(mux);
    return n;
}

/* Any WebP file played back by libwebpdemux's WebPAnimDecoder: fully composited BGRA canvases, one per frame, with end timestamps.
 * Returns the frame count (0 on failure); *w, *h = canvas. */
int ref_webp_play(const uint8_t* data, size_t len, uint8_t* out, size_t cap, int* w, int* h, int* timestamps, int max_frames, uint32_t info[2] /* loop_count, bgcolor */)
{
    WebPAnimDecoderOptions opt;
    if (!WebPAnimDecoderOptionsInit(&opt)) return 0;
    opt.color_mode = MODE_BGRA;
    WebPData src = {data, len};
    WebPAnimDecoder* dec = WebPAnimDecoderNew(&src, &opt);
    if (!dec) return 0;
    WebPAnimInfo ai;
    if (!WebPAnimDecoderGetInfo(dec, &ai)) { WebPAnimDecoderDelete(dec); return 0; }
    *w = (int)ai.canvas_width; *h = (int)ai.canvas_height;
    info[0] = ai.loop_count; info[1] = ai.bgcolor;
    const size_t fb = (size_t)ai.canvas_width * ai.canvas_height * 4;
    int n = 0;
    while (WebPAnimDecoderHasMoreFrames(dec) && n < max_frames) {
        uint8_t* buf;
        int ts;
        if (!WebPAnimDecoderGetNext(dec, &buf, &ts)) { n = 0; break; }
        if ((size_t)(n + 1) * fb > cap) { n = 0; break; }
        memcpy(out + (size_t)n * fb, buf, fb);
        timestamps[n] = ts;
        n++;
    }
    WebPAnimDecoderDelete(dec);
    return n;
}

/* The reference's animation writer (webp.cpp:631-706, 508-548): every frame a whole canvas through WebPAnimEncoderAdd with kmin 3 / kmax 4,
 * the closing NULL frame at the total duration, WebPAnimEncoderAssemble. quality > 100 = lossless, as webp.cpp:463-467 maps it.
 * frames: n canvases of w x h x cn (cn 3 = BGR, 4 = BGRA); delays in ms. Returns the file size (0 on failure). */
size_t ref_webp_encode_anim(const uint8_t* frames, int n, int w, int h, int cn, const int* delays, float quality, uint32_t loop_count, uint32_t bgcolor,
                            uint8_t* out, size_t cap)
{
    WebPConfig config;
    if (!WebPConfigInit(&config)) return 0;
    const float q = quality < 1.f ? 1.f : quality;
    config.quality = q > 100.f ? 100.f : q;
    config.lossless = q > 100.f;
    WebPAnimEncoderOptions ao;
    if (!WebPAnimEncoderOptionsInit(&ao)) return 0;
    ao.anim_params.loop_count = (int)loop_count;
    ao.anim_params.bgcolor = bgcolor;
    ao.kmin = 3;
    ao.kmax = 4;
    WebPAnimEncoder* enc = WebPAnimEncoderNew(w, h, &ao);
    if (!enc) return 0;
    int ts = 0, ok = 1;
    for (int i = 0; i < n && ok; i++) {
        WebPPicture pic;
        WebPPictureInit(&pic);
        pic.width = w; pic.height = h; pic.use_argb = 1;
        ok = WebPPictureAlloc(&pic);
        const uint8_t* px = frames + (size_t)i * w * h * cn;
        if (ok) ok = cn == 3 ? WebPPictureImportBGR(&pic, px, w * cn) : WebPPictureImportBGRA(&pic, px, w * cn);
        if (ok) ok = WebPAnimEncoderAdd(enc, &pic, ts, &config);
        WebPPictureFree(&pic);
        ts += delays[i];
    }
    size_t size = 0;
    if (ok && WebPAnimEncoderAdd(enc, NULL, ts, &config)) {
        WebPData d;
        WebPDataInit(&d);
        if (WebPAnimEncoderAssemble(enc, &d)) {
            if (d.size <= cap) { memcpy(out, d.bytes, d.size); size = d.size; }
            WebPDataClear(&d);
        }
    }
    WebPAnimEncoderDelete(enc);
    return size;
}

/* The planes the reference's lossy still writer codes (webp.cpp:707-751: WebPEncodeBGR(A) -> WebPPictureImportBGR(A) with use_argb = 0,
 * i.e. picture_csp_enc.c ImportYUVAFromRGBA): Y (w x h), U, V ((w+1)/2 x (h+1)/2). Returns 1 when the picture came out with an alpha plane. */
int ref_webp_yuv420(const uint8_t* px, int w, int h, int cn, uint8_t* y, uint8_t* u, uint8_t* v)
{
    WebPPicture pic;
    if (!WebPPictureInit(&pic)) return -1;
    pic.use_argb = 0;
    pic.width = w; pic.height = h;
    if (!(cn == 3 ? WebPPictureImportBGR(&pic, px, w * cn) : WebPPictureImportBGRA(&pic, px, w * cn))) { WebPPictureFree(&pic); return -1; }
    const int uvw = (w + 1) / 2, uvh = (h + 1) / 2;
    for (int r = 0; r < h; r++) memcpy(y + (size_t)r * w, pic.y + (size_t)r * pic.y_stride, (size_t)w);
    for (int r = 0; r < uvh; r++) {
        memcpy(u + (size_t)r * uvw, pic.u + (size_t)r * pic.uv_stride, (size_t)uvw);
        memcpy(v + (size_t)r * uvw, pic.v + (size_t)r * pic.uv_stride, (size_t)uvw);
    }
    const int alpha = pic.a != NULL;
    WebPPictureFree(&pic);
    return alpha;
}

int ref_webp_version(void) { return WebPGetDecoderVersion(); }
