// lp_webp.cpp -- see lp_webp.h. Container rules: "WebP Container Specification" (RIFF, extended format) and the checks libwebp 1.5.0's
// mux reader applies (src/mux/muxread.c WebPMuxCreateInternal / MuxImageParse, src/mux/muxinternal.c MuxValidate), which decide whether
// the reference's webp_decoder_create succeeds (/root/reference/webp.cpp:61-84).
#include "lp_webp.h"

#include <string.h>

static inline uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t le24(const uint8_t* p) { return le16(p) | ((uint32_t)p[2] << 16); }
static inline uint32_t le32(const uint8_t* p) { return le24(p) | ((uint32_t)p[3] << 24); }
static inline bool tag_is(const uint8_t* p, const char* t) { return memcmp(p, t, 4) == 0; }

static const size_t kMaxChunkPayload = ~0u - 8 - 1;     // MAX_CHUNK_PAYLOAD





// VP8GetInfo (src/dec/vp8_dec.c): key frame header of a lossy bitstream
static bool vp8_info(const uint8_t* d, size_t n, size_t chunk_size, int* w, int* h)
{
    if (n < 10) return false;
    if (d[3] != 0x9d || d[4] != 0x01 || d[5] != 0x2a) return false;
    const uint32_t bits = le24(d);
    const bool key_frame = !(bits & 1);
    const int ww = (int)(((uint32_t)d[7] << 8) | d[6]) & 0x3fff, hh = (int)(((uint32_t)d[9] << 8) | d[8]) & 0x3fff;
    if (!key_frame) return false;
    if (((bits >> 1) & 7) > 3) return false;        // unknown profile
    if (!((bits >> 4) & 1)) return false;           // first frame is invisible
    if (((bits >> 5)) >= chunk_size) return false;  // inconsistent partition size
    if (ww == 0 || hh == 0) return false;
    *w = ww; *h = hh;
    return true;
}
// VP8LGetInfo (src/dec/vp8l_dec.c)
static bool vp8l_info(const uint8_t* d, size_t n, int* w, int* h, bool* alpha)
{
    if (n < 5) return false;
    if (d[0] != 0x2f || (d[4] >> 5) != 0) return false; // signature, version 0
    const uint32_t bits = le32(d + 1);
    *w = (int)(bits & 0x3fff) + 1;
    *h = (int)((bits >> 14) & 0x3fff) + 1;
    *alpha = ((bits >> 28) & 1) != 0;
    return true;
}

namespace {
struct Wpi { // the image being collected (WebPMuxImage): optional ALPH, then the VP8 / VP8L chunk
    LpWebpFrame f;
    bool partial = false; // an ALPH chunk has been seen, the image chunk has not
};

// one image's chunks inside [p, p + n): MuxImageParse for ANMF payloads, the same per-chunk logic for top-level images
bool take_image_chunk(Wpi& w, const uint8_t* tag, const uint8_t* payload, size_t size, bool* finished)
{
    *finished = false;
    if (tag_is(tag, "ALPH")) {
        if (w.f.alph) return false; // a second ALPH
        w.f.alph = payload; w.f.alph_size = size; w.partial = true;
        return true;
    }
    if (tag_is(tag, "VP8 ") || tag_is(tag, "VP8L")) {
        if (w.f.img) return false;
        w.f.img = payload; w.f.img_size = size; w.f.lossless = tag[3] == 'L';
        bool a = false;
        if (w.f.lossless ? !vp8l_info(payload, size, &w.f.width, &w.f.height, &a) : !vp8_info(payload, size, size, &w.f.width, &w.f.height)) return false;
        w.f.has_alpha = w.f.lossless ? a : w.f.alph != nullptr; // MuxImageFinalize
        w.partial = false;
        *finished = true;
        return true;
    }
    return true; // unknown chunk inside an image: kept aside by the mux, ignored here
}
} // namespace

bool lp_webp_parse(const uint8_t* data, size_t len, LpWebpFile* out)
{
    *out = LpWebpFile();
    if (!data || len < 12 + 8) return false;                    // RIFF_HEADER_SIZE + CHUNK_HEADER_SIZE
    if (!tag_is(data, "RIFF") || !tag_is(data + 8, "WEBP")) return false;
    size_t riff = le32(data + 4);
    if (riff < 8 || riff > kMaxChunkPayload) return false;
    if (riff > len - 8) return false;                           // truncated file
    if (riff < len - 8) len = riff + 8;                         // trailing bytes are not part of the container
    // WebPMuxCreateInternal: "First chunk should be VP8, VP8L or VP8X" -- before anything else is looked at
    if (!tag_is(data + 12, "VP8 ") && !tag_is(data + 12, "VP8L") && !tag_is(data + 12, "VP8X")) return false;
    size_t pos = 12;
    const uint8_t* vp8x = nullptr;
    int n_vp8x = 0, n_iccp = 0, n_anim = 0, n_anmf = 0, n_exif = 0, n_xmp = 0;
    const uint8_t* anim = nullptr;
    size_t anim_size = 0;
    Wpi wpi;
    int n_still = 0;
    while (pos != len) {
        if (len - pos < 8) return false;
        const uint8_t* tag = data + pos;
        const size_t size = le32(data + pos + 4);
        if (size > kMaxChunkPayload) return false;
        const size_t padded = size + (size & 1);
        if (padded > len - pos - 8) return false;               // NOT_ENOUGH_DATA
        const uint8_t* payload = data + pos + 8;
        if (tag_is(tag, "ALPH") || tag_is(tag, "VP8 ") || tag_is(tag, "VP8L")) {
            bool fin;
            if (!take_image_chunk(wpi, tag, payload, size, &fin)) return false;
            if (fin) { out->frames.push_back(wpi.f); wpi = Wpi(); n_still++; }
        } else if (tag_is(tag, "ANMF")) {
            if (wpi.partial) return false;
            if (size < 16) return false;                        // ANMF_CHUNK_SIZE
            Wpi fr;
            fr.partial = true;                                  // MuxImageParse: "waiting for ALPH and/or VP8/VP8L chunks" from the frame header on
            size_t q = 16;
            while (q != size) { // MuxImageParse (libwebp 1.5.0 src/mux/muxread.c): the sub-chunks of the frame, every one of them
                if (size - q < 8) return false;
                const size_t ss = le32(payload + q + 4);
                const size_t sp = ss + (ss & 1);
                if (ss > kMaxChunkPayload || sp > size - q - 8) return false;
                const uint8_t* st = payload + q;
                if (tag_is(st, "ALPH") || tag_is(st, "VP8 ") || tag_is(st, "VP8L")) {
                    // a second ALPH or a second image chunk fails (take_image_chunk); an ALPH behind the image leaves the frame partial again
                    bool fin;
                    if (!take_image_chunk(fr, st, st + 8, ss, &fin)) return false;
                } else if (tag_is(st, "VP8X") || tag_is(st, "ICCP") || tag_is(st, "ANIM") || tag_is(st, "ANMF") || tag_is(st, "EXIF") || tag_is(st, "XMP ")) {
                    return false;                               // a chunk the container knows, but not inside a frame ("default: goto Fail")
                } else if (fr.partial) {
                    return false;                               // an unknown chunk before the frame's image chunk ("between some image chunks")
                }                                               // (an unknown chunk behind the image is kept aside by the mux: ignored here)
                q += 8 + sp;
            }
            if (!fr.f.img || fr.partial) return false;
            fr.f.x_offset = 2 * (int)le24(payload);
            fr.f.y_offset = 2 * (int)le24(payload + 3);
            // (the frame's width / height fields are not read by the mux: the bitstream's own header counts)
            fr.f.duration = (int)le24(payload + 12);
            const uint8_t bits = payload[15];
            fr.f.dispose = bits & 1;
            fr.f.blend = (bits >> 1) & 1;                        // bit set = do not blend (WEBP_MUX_NO_BLEND = 1)
            out->frames.push_back(fr.f);
            n_anmf++;
        } else {
            if (wpi.partial) return false;                      // an ALPH chunk must be followed by its image
            if (tag_is(tag, "VP8X")) { if (size < 10) return false; if (!vp8x) vp8x = payload; n_vp8x++; }
            else if (tag_is(tag, "ICCP")) { if (!out->icc) { out->icc = payload; out->icc_size = size; } n_iccp++; }
            else if (tag_is(tag, "ANIM")) { if (!anim) { anim = payload; anim_size = size; } n_anim++; }
            else if (tag_is(tag, "EXIF")) n_exif++;
            else if (tag_is(tag, "XMP ")) n_xmp++;
        }
        pos += 8 + padded;
    }
    if (wpi.partial) return false;
    // ---- MuxValidate
    if (out->frames.empty()) return false;
    const int n_images = (int)out->frames.size();
    bool any_alpha = false;
    for (const auto& f : out->frames) any_alpha = any_alpha || f.has_alpha;
    if (n_vp8x > 1 || n_iccp > 1 || n_anim > 1 || n_exif > 1 || n_xmp > 1) return false;
    if (vp8x) {
        out->has_vp8x = true;
        out->flags = le32(vp8x);
        out->canvas_w = 1 + (int)le24(vp8x + 4);
        out->canvas_h = 1 + (int)le24(vp8x + 7);
        if ((uint64_t)out->canvas_w * (uint64_t)out->canvas_h >= (1ull << 32)) return false;
    } else {
        if (n_images != 1) return false;                        // several images need a VP8X chunk
        out->flags = any_alpha ? LP_WEBP_FLAG_ALPHA : 0u;     // WebPMuxGetFeatures without a VP8X chunk: the image's alpha is all there is --
        out->canvas_w = out->frames[0].width;                   // an ICCP / EXIF / XMP / ANIM chunk or an ANMF frame then fails the checks below
        out->canvas_h = out->frames[0].height;
    }
    const uint32_t fl = out->flags;
    if (((fl & LP_WEBP_FLAG_ICCP) != 0) != (n_iccp > 0)) return false;
    if (((fl & LP_WEBP_FLAG_EXIF) != 0) != (n_exif > 0)) return false;
    if (((fl & LP_WEBP_FLAG_XMP) != 0) != (n_xmp > 0)) return false;
    const bool has_animation = (fl & LP_WEBP_FLAG_ANIM) != 0;
    if (has_animation && (n_anim == 0 || n_anmf == 0)) return false;
    if (!has_animation && (n_anim == 1 || n_anmf > 0)) return false;
    if (!has_animation) {
        if (n_images != 1) return false;
        if (out->has_vp8x && (out->frames[0].width != out->canvas_w || out->frames[0].height != out->canvas_h)) return false;
    } else if (n_still) return false;                           // an animation holds ANMF frames only
    if (any_alpha && out->has_vp8x && !(fl & LP_WEBP_FLAG_ALPHA)) return false;
    // (frame rectangles are not checked against the canvas here: libwebpmux does not, and a frame that sticks out fails later, when
    // the compositor refuses its region -- opencv_copy_to_region: OPENCV_ERROR_OUT_OF_BOUNDS)
    if (anim) {
        if (anim_size < 6) return false;                        // ANIM_CHUNK_SIZE
        out->has_anim_chunk = true;
        out->bgcolor = le32(anim);
        out->loop_count = le16(anim + 4);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ writer helpers
static void put_le(std::vector<uint8_t>& o, uint32_t v, int bytes) { for (int i = 0; i < bytes; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_chunk(std::vector<uint8_t>& o, const char* tag, const uint8_t* p, size_t n)
{
    o.insert(o.end(), tag, tag + 4);
    put_le(o, (uint32_t)n, 4);
    o.insert(o.end(), p, p + n);
    if (n & 1) o.push_back(0);
}
static void put_vp8x(std::vector<uint8_t>& o, uint32_t flags, int w, int h)
{
    uint8_t b[10];
    b[0] = (uint8_t)flags; b[1] = b[2] = b[3] = 0;
    const uint32_t ww = (uint32_t)(w - 1), hh = (uint32_t)(h - 1);
    b[4] = (uint8_t)ww; b[5] = (uint8_t)(ww >> 8); b[6] = (uint8_t)(ww >> 16);
    b[7] = (uint8_t)hh; b[8] = (uint8_t)(hh >> 8); b[9] = (uint8_t)(hh >> 16);
    put_chunk(o, "VP8X", b, 10);
}
static void begin_riff(std::vector<uint8_t>& o) { o.clear(); o.insert(o.end(), {'R', 'I', 'F', 'F', 0, 0, 0, 0, 'W', 'E', 'B', 'P'}); }
static void end_riff(std::vector<uint8_t>& o)
{
    const uint32_t n = (uint32_t)(o.size() - 8);
    o[4] = (uint8_t)n; o[5] = (uint8_t)(n >> 8); o[6] = (uint8_t)(n >> 16); o[7] = (uint8_t)(n >> 24);
}
static void put_image(std::vector<uint8_t>& o, const uint8_t* alph, size_t alph_n, const uint8_t* img, size_t img_n, bool lossless)
{
    if (alph && alph_n) put_chunk(o, "ALPH", alph, alph_n);
    put_chunk(o, lossless ? "VP8L" : "VP8 ", img, img_n);
}

void lp_webp_frame_bitstream(const LpWebpFrame& f, std::vector<uint8_t>& out)
{
    begin_riff(out);
    if (f.alph) put_vp8x(out, LP_WEBP_FLAG_ALPHA, f.width, f.height); // a VP8X chunk is needed only for the ALPH chunk
    put_image(out, f.alph, f.alph_size, f.img, f.img_size, f.lossless);
    end_riff(out);
}

bool lp_webp_split_encoded(const uint8_t* riff, size_t len, LpWebpEncodedImage* out)
{
    LpWebpFile wf;
    if (!lp_webp_parse(riff, len, &wf) || wf.frames.size() != 1) return false;
    const LpWebpFrame& f = wf.frames[0];
    out->alph.assign(f.alph, f.alph + f.alph_size);
    out->img.assign(f.img, f.img + f.img_size);
    out->lossless = f.lossless;
    out->has_alpha = f.has_alpha;
    out->width = f.width;
    out->height = f.height;
    return true;
}

void lp_webp_write_still(const LpWebpEncodedImage& im, const uint8_t* icc, size_t icc_len, std::vector<uint8_t>& out)
{
    begin_riff(out);
    const bool need_x = (icc && icc_len) || !im.alph.empty() || (im.lossless && im.has_alpha && icc && icc_len);
    if (need_x) {
        put_vp8x(out, ((icc && icc_len) ? LP_WEBP_FLAG_ICCP : 0u) | (im.has_alpha ? LP_WEBP_FLAG_ALPHA : 0u), im.width, im.height);
        if (icc && icc_len) put_chunk(out, "ICCP", icc, icc_len);
    }
    put_image(out, im.alph.empty() ? nullptr : im.alph.data(), im.alph.size(), im.img.data(), im.img.size(), im.lossless);
    end_riff(out);
}

void lp_webp_write_animation(int canvas_w, int canvas_h, uint32_t bgcolor, uint32_t loop_count, const std::vector<LpWebpAnimFrame>& frames, const uint8_t* icc,
                             size_t icc_len, std::vector<uint8_t>& out)
{
    begin_riff(out);
    bool alpha = false;
    for (const auto& f : frames) alpha = alpha || f.im.has_alpha;
    put_vp8x(out, LP_WEBP_FLAG_ANIM | (alpha ? LP_WEBP_FLAG_ALPHA : 0u) | ((icc && icc_len) ? LP_WEBP_FLAG_ICCP : 0u), canvas_w, canvas_h);
    if (icc && icc_len) put_chunk(out, "ICCP", icc, icc_len);
    uint8_t an[6];
    an[0] = (uint8_t)bgcolor; an[1] = (uint8_t)(bgcolor >> 8); an[2] = (uint8_t)(bgcolor >> 16); an[3] = (uint8_t)(bgcolor >> 24);
    an[4] = (uint8_t)loop_count; an[5] = (uint8_t)(loop_count >> 8);
    put_chunk(out, "ANIM", an, 6);
    for (const auto& f : frames) {
        std::vector<uint8_t> body;
        put_le(body, (uint32_t)(f.x_offset / 2), 3);
        put_le(body, (uint32_t)(f.y_offset / 2), 3);
        put_le(body, (uint32_t)(f.im.width - 1), 3);
        put_le(body, (uint32_t)(f.im.height - 1), 3);
        put_le(body, (uint32_t)f.duration, 3);
        body.push_back((uint8_t)((f.dispose & 1) | ((f.blend & 1) << 1)));
        put_image(body, f.im.alph.empty() ? nullptr : f.im.alph.data(), f.im.alph.size(), f.im.img.data(), f.im.img.size(), f.im.lossless);
        put_chunk(out, "ANMF", body.data(), body.size());
    }
    end_riff(out);

}
