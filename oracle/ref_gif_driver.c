/* oracle/ref_gif_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 * The GIF side of the checker: the REFERENCE's prebuilt giflib 5.2.2 (deps/linux/amd64/lib/libgif.a) does the container walk
 * and the LZW decoding, and the frame compositing on top of it is a plain-C restatement of the reference's own
 * /root/reference/giflib.cpp (decoder create :103-158, extension bookkeeping :203-284, frame seek/header :286-342,
 * render_frame :349-561, skip :563-583, background colour :585-627, decode_frame :632-724, animation info :1308-1431).
 * Built by oracle/Makefile into oracle/_ref/librefgif.so when /root/reference is present. */
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <gif_lib.h>

typedef struct {
    const uint8_t* data;
    size_t len, pos;
    GifFileType* gif;
    GifByteType* pixels;
    size_t pixel_cap;
    uint8_t* saved; /* prev_frame_bgra: canvas before the current frame was drawn */
    int prev_disposal, prev_delay, prev_left, prev_top, prev_width, prev_height;
    uint8_t bg_b, bg_g, bg_r, bg_a;
    int have_first, clear_ext;
} rg_dec;

static int rg_read(GifFileType* gif, GifByteType* buf, int len) /* giflib.cpp:73-81 */
{
    rg_dec* d = (rg_dec*)gif->UserData;
    size_t left = d->len - d->pos, n = (size_t)len < left ? (size_t)len : left;
    memmove(buf, d->data + d->pos, n);
    d->pos += n;
    return (int)n;
}

void rg_close(void* h)
{
    rg_dec* d = (rg_dec*)h;
    int err = 0;
    if (!d) return;
    if (d->gif) DGifCloseFile(d->gif, &err);
    free(d->pixels);
    free(d->saved);
    free(d);
}

void* rg_open(const uint8_t* data, size_t len, int dims[2]) /* giflib.cpp:103-158 */
{
    rg_dec* d = (rg_dec*)calloc(1, sizeof(rg_dec));
    int err = 0;
    d->data = data;
    d->len = len;
    d->gif = DGifOpen(d, rg_read, &err);
    if (err || !d->gif) { if (d->gif) DGifCloseFile(d->gif, &err); free(d); return NULL; }
    if (d->gif->SWidth <= 0 || d->gif->SHeight <= 0) { rg_close(d); return NULL; }
    d->saved = (uint8_t*)calloc((size_t)d->gif->SWidth * d->gif->SHeight, 4);
    dims[0] = d->gif->SWidth;
    dims[1] = d->gif->SHeight;
    return d;
}

static int read_extensions(rg_dec* d) /* giflib.cpp:203-241 */
{
    GifByteType* ext;
    int fn;
    if (DGifGetExtension(d->gif, &fn, &ext) == GIF_ERROR) return 0;
    if (ext && GifAddExtensionBlock(&d->gif->ExtensionBlockCount, &d->gif->ExtensionBlocks, fn, ext[0], &ext[1]) == GIF_ERROR) return 0;
    while (ext) {
        if (DGifGetExtensionNext(d->gif, &ext) == GIF_ERROR) return 0;
        if (ext && GifAddExtensionBlock(&d->gif->ExtensionBlockCount, &d->gif->ExtensionBlocks, CONTINUE_EXT_FUNC_CODE, ext[0], &ext[1]) == GIF_ERROR) return 0;
    }
    return 1;
}

static void frame_gcb(GifFileType* gif, GraphicsControlBlock* gcb) /* giflib.cpp:243-264 */
{
    gcb->DisposalMode = DISPOSAL_UNSPECIFIED;
    gcb->UserInputFlag = 0;
    gcb->DelayTime = 0;
    gcb->TransparentColor = NO_TRANSPARENT_COLOR;
    for (int i = 0; i < gif->ExtensionBlockCount; i++)
        if (gif->ExtensionBlocks[i].Function == GRAPHICS_EXT_FUNC_CODE) DGifExtensionToGCB(gif->ExtensionBlocks[i].ByteCount, gif->ExtensionBlocks[i].Bytes, gcb);
}

static void set_frame_gcb(GifFileType* gif, const GraphicsControlBlock* gcb) /* giflib.cpp:266-284 */
{
    for (int i = 0; i < gif->ExtensionBlockCount; i++)
        if (gif->ExtensionBlocks[i].Function == GRAPHICS_EXT_FUNC_CODE && gif->ExtensionBlocks[i].ByteCount >= 4) EGifGCBToExtension(gcb, gif->ExtensionBlocks[i].Bytes);
}

/* 0 = a frame follows, 1 = end of file, 2 = error (giflib.cpp:286-342) */
static int frame_header(rg_dec* d)
{
    GifRecordType rt;
    if (d->clear_ext) { GifFreeExtensions(&d->gif->ExtensionBlockCount, &d->gif->ExtensionBlocks); d->clear_ext = 0; }
    for (;;) {
        if (DGifGetRecordType(d->gif, &rt) == GIF_ERROR) return 2;
        if (rt == IMAGE_DESC_RECORD_TYPE) break;
        if (rt == EXTENSION_RECORD_TYPE && !read_extensions(d)) return 2;
        if (rt == TERMINATE_RECORD_TYPE) return 1;
    }
    return DGifGetImageHeader(d->gif) == GIF_ERROR ? 2 : 0;
}

static void background(GifFileType* gif, const GraphicsControlBlock* gcb, uint8_t* r, uint8_t* g, uint8_t* b, uint8_t* a) /* giflib.cpp:585-627 */
{
    int in_map = gif->SColorMap && gif->SColorMap->Colors && gif->SBackGroundColor >= 0 && gif->SBackGroundColor < gif->SColorMap->ColorCount;
    *r = in_map ? gif->SColorMap->Colors[gif->SBackGroundColor].Red : 255;
    *g = in_map ? gif->SColorMap->Colors[gif->SBackGroundColor].Green : 255;
    *b = in_map ? gif->SColorMap->Colors[gif->SBackGroundColor].Blue : 255;
    *a = gcb->TransparentColor != NO_TRANSPARENT_COLOR ? 0 : 255;
}

static void clip_prev(const rg_dec* d, int bw, int bh, int* x, int* y, int* w, int* h) /* giflib.cpp:408-436, 449-478 */
{
    int l = d->prev_left, t = d->prev_top, pw = d->prev_width, ph = d->prev_height;
    if (l < 0) { pw += l; l = 0; }
    if (t < 0) { ph += t; t = 0; }
    if (l + pw > bw) pw = bw - l;
    if (t + ph > bh) ph = bh - t;
    *x = l; *y = t; *w = pw < 0 ? 0 : pw; *h = ph < 0 ? 0 : ph;
}

/* canvas: SWidth x SHeight BGRA, kept by the caller between frames (the Go Framebuffer). Returns 0 ok, 1 eof, 2 header error,
 * 3 decode failed. meta = {left, top, width, height, interlace, disposal, delay, transparent (before the partial-frame fix-up),
 * color_count, has_local_map, transparent_after_fixup}. indices (optional) receives the de-interlaced colour indices. */
int rg_next(void* h, uint8_t* canvas, int meta[11], uint8_t* indices, size_t cap)
{
    rg_dec* d = (rg_dec*)h;
    int st = frame_header(d);
    if (st) return st;
    GifImageDesc desc = d->gif->Image;
    /* ---- giflib_decoder_decode_frame, giflib.cpp:632-724 */
    if (desc.Width <= 0 || desc.Height <= 0 || desc.Width > INT_MAX / desc.Height) return 3;
    size_t image_size = (size_t)desc.Width * desc.Height;
    if (image_size > d->pixel_cap) { d->pixel_cap = image_size; d->pixels = (GifByteType*)realloc(d->pixels, image_size); }
    if (desc.Interlace) {
        static const int off[4] = {0, 4, 2, 1}, jump[4] = {8, 8, 4, 2};
        for (int i = 0; i < 4; i++)
            for (int j = off[i]; j < desc.Height; j += jump[i])
                if (DGifGetLine(d->gif, d->pixels + (size_t)j * desc.Width, desc.Width) == GIF_ERROR) return 3;
    } else if (DGifGetLine(d->gif, d->pixels, (int)image_size) == GIF_ERROR)
        return 3;
    GraphicsControlBlock gcb;
    frame_gcb(d->gif, &gcb);
    if (!d->have_first) background(d->gif, &gcb, &d->bg_r, &d->bg_g, &d->bg_b, &d->bg_a);
    /* ---- giflib_decoder_render_frame, giflib.cpp:349-561 */
    const int bw = d->gif->SWidth, bh = d->gif->SHeight;
    int fl = desc.Left, ft = desc.Top, fw = desc.Width, fh = desc.Height;
    int skip_l = fl < 0 ? -fl : 0, skip_t = ft < 0 ? -ft : 0;
    int skip_r = fl + fw > bw ? fl + fw - bw : 0, skip_b = ft + fh > bh ? ft + fh - bh : 0;
    ColorMapObject* map = desc.ColorMap ? desc.ColorMap : d->gif->SColorMap;
    if (!map) return 3;
    const uint8_t bgpx[4] = {d->bg_b, d->bg_g, d->bg_r, d->bg_a};
    if (!d->have_first) {
        for (size_t p = 0; p < (size_t)bw * bh; p++) memcpy(canvas + 4 * p, bgpx, 4);
    } else {
        int px, py, pw, ph;
        clip_prev(d, bw, bh, &px, &py, &pw, &ph);
        if (d->prev_disposal == DISPOSE_BACKGROUND) {
            for (int y = py; y < py + ph; y++)
                for (int x = px; x < px + pw; x++) memcpy(canvas + ((size_t)y * bw + x) * 4, bgpx, 4);
        } else if (d->prev_disposal == DISPOSE_PREVIOUS) {
            for (int y = py; y < py + ph; y++) memcpy(canvas + ((size_t)y * bw + px) * 4, d->saved + ((size_t)y * bw + px) * 4, (size_t)pw * 4);
        }
        memcpy(d->saved, canvas, (size_t)bw * bh * 4);
    }
    size_t pi = (size_t)skip_t * desc.Width;
    fh -= skip_t; ft += skip_t;
    fw -= skip_l; fl += skip_l;
    fw -= skip_r;
    fh -= skip_b;
    const int transparent = gcb.TransparentColor;
    for (int y = ft; y < ft + fh; y++) {
        pi += (size_t)skip_l;
        for (int x = fl; x < fl + fw; x++) {
            int idx = d->pixels[pi++];
            if (idx == transparent || idx >= map->ColorCount) continue;
            uint8_t* px = canvas + ((size_t)y * bw + x) * 4;
            px[0] = map->Colors[idx].Blue; px[1] = map->Colors[idx].Green; px[2] = map->Colors[idx].Red; px[3] = 255;
        }
        pi += (size_t)skip_r;
    }
    if ((fh < bh || fw < bw || fl != 0 || ft != 0) && transparent == -1) {
        gcb.TransparentColor = map->ColorCount - 1;
        set_frame_gcb(d->gif, &gcb);
    }
    const int mm[11] = {desc.Left, desc.Top, desc.Width, desc.Height, desc.Interlace ? 1 : 0, gcb.DisposalMode, gcb.DelayTime, transparent,
                        map->ColorCount, desc.ColorMap ? 1 : 0, gcb.TransparentColor};
    memcpy(meta, mm, sizeof(mm));
    if (indices && image_size <= cap) memcpy(indices, d->pixels, image_size);
    d->prev_disposal = gcb.DisposalMode;
    d->prev_delay = gcb.DelayTime;
    d->prev_left = d->gif->Image.Left; d->prev_top = d->gif->Image.Top;
    d->prev_width = d->gif->Image.Width; d->prev_height = d->gif->Image.Height;
    d->have_first = 1;
    d->clear_ext = 1;
    return 0;
}

int rg_skip(void* h) /* giflib.cpp:563-583 */
{
    rg_dec* d = (rg_dec*)h;
    int st = frame_header(d);
    GifByteType* block;
    if (st) return st;
    for (;;) {
        if (DGifGetCodeNext(d->gif, &block) == GIF_ERROR) return 2;
        if (!block) return 0;
    }
}

/* out = {loop_count, frame_count, bg_red, bg_green, bg_blue, bg_alpha, duration_ms} (giflib.cpp:1308-1431) */
void rg_info(const uint8_t* data, size_t len, int out[7])
{
    int info[7] = {1, 0, 255, 255, 255, 0, 0};
    rg_dec rd;
    int err = 0, found_loop = 0, found_gcb = 0;
    GraphicsControlBlock gcb;
    GifRecordType rt;
    memset(&rd, 0, sizeof(rd));
    memset(&gcb, 0, sizeof(gcb));
    memcpy(out, info, sizeof(info));
    rd.data = data;
    rd.len = len;
    GifFileType* gif = DGifOpen(&rd, rg_read, &err);
    if (err || !gif) return;
    while (DGifGetRecordType(gif, &rt) == GIF_OK) {
        if (rt == EXTENSION_RECORD_TYPE) {
            GifByteType* ext;
            int fn;
            if (DGifGetExtension(gif, &fn, &ext) == GIF_OK && ext) {
                if (fn == GRAPHICS_EXT_FUNC_CODE) {
                    GraphicsControlBlock fg;
                    memset(&fg, 0, sizeof(fg));
                    DGifExtensionToGCB(ext[0], &ext[1], &fg);
                    info[6] += (info[1] > 0 && fg.DelayTime < 2) ? 20 : fg.DelayTime * 10;
                    if (!found_gcb) {
                        uint8_t r, g, b, a;
                        found_gcb = 1;
                        gcb = fg;
                        background(gif, &gcb, &r, &g, &b, &a);
                        info[2] = r; info[3] = g; info[4] = b; info[5] = a;
                    }
                } else if (!found_loop && fn == APPLICATION_EXT_FUNC_CODE && ext[0] >= 11 && memcmp(ext + 1, "NETSCAPE2.0", 11) == 0) {
                    if (DGifGetExtensionNext(gif, &ext) == GIF_OK && ext && ext[0] >= 3 && ext[1] == 1) {
                        info[0] = ext[2] | (ext[3] << 8);
                        found_loop = 1;
                    }
                }
                while (ext)
                    if (DGifGetExtensionNext(gif, &ext) != GIF_OK) goto done;
            }
        } else if (rt == IMAGE_DESC_RECORD_TYPE) {
            GifByteType* block;
            int cs;
            info[1]++;
            if (DGifGetImageDesc(gif) != GIF_OK) goto done;
            if (DGifGetCode(gif, &cs, &block) == GIF_ERROR) goto done;
            while (block)
                if (DGifGetCodeNext(gif, &block) == GIF_ERROR) goto done;
        } else if (rt == TERMINATE_RECORD_TYPE)
            goto done;
    }
    if (!found_gcb) {
        uint8_t r, g, b, a;
        background(gif, &gcb, &r, &g, &b, &a);
        info[2] = r; info[3] = g; info[4] = b; info[5] = a;
    }
done:
    DGifCloseFile(gif, &err);
    memcpy(out, info, sizeof(info));
}

/* ------------------------------------------------------------------------------------------------
 * Encoder side: giflib 5.2.2's EGif* writer driven the way the reference's giflib_encoder_* functions drive it
 * (/root/reference/giflib.cpp:762-1306): screen descriptor and global palette carried over from the decoder, per frame the
 * decoder's extension blocks and local palette, the BGRA -> palette-index mapping with its 15-bit bucket cache and the
 * "previous frame already shows this colour" transparency trick, then EGifPutLine. */
typedef struct {
    GifFileType* gif;
    uint8_t* dst;
    size_t dst_len, off;
    uint8_t lut_present[1 << 15], lut_index[1 << 15];
    GifByteType* pixels;
    size_t pixel_cap;
    ColorMapObject* frame_map;      /* this frame's local palette (a copy), or NULL */
    GifColorType prev_colors[256];  /* the palette used for the previous frame */
    int prev_count;
    int prev_disposal;
    uint8_t* prev_bgra;
    int have_first;
} rg_enc;

static int rg_write(GifFileType* gif, const GifByteType* buf, int len) /* giflib.cpp:762-771 */
{
    rg_enc* e = (rg_enc*)gif->UserData;
    if (e->off + (size_t)len > e->dst_len) return 0;
    memcpy(e->dst + e->off, buf, (size_t)len);
    e->off += (size_t)len;
    return len;
}

void* rg_enc_create(uint8_t* buf, size_t len) /* giflib.cpp:773-797 */
{
    rg_enc* e = (rg_enc*)calloc(1, sizeof(rg_enc));
    int err = 0;
    e->dst = buf;
    e->dst_len = len;
    e->gif = EGifOpen(e, rg_write, &err);
    if (err || !e->gif) { free(e); return NULL; }
    return e;
}

int rg_enc_init(void* he, void* hd, int width, int height) /* giflib.cpp:800-851 */
{
    rg_enc* e = (rg_enc*)he;
    rg_dec* d = (rg_dec*)hd;
    EGifSetGifVersion(e->gif, true);
    e->prev_bgra = (uint8_t*)malloc((size_t)width * height * 4);
    int bg = (d->gif->SColorMap && d->gif->SBackGroundColor >= 0 && d->gif->SBackGroundColor < d->gif->SColorMap->ColorCount) ? d->gif->SBackGroundColor : 0;
    e->gif->AspectByte = d->gif->AspectByte;
    return EGifPutScreenDesc(e->gif, width, height, d->gif->SColorResolution, bg, d->gif->SColorMap) != GIF_ERROR;
}

static int mdist(int r0, int g0, int b0, int r1, int g1, int b1) { return abs(r0 - r1) + abs(g0 - g1) + abs(b0 - b1); }

/* frame: width x height BGRA, tightly packed */
int rg_enc_frame(void* he, void* hd, const uint8_t* frame, int width, int height) /* giflib.cpp:853-1214 */
{
    rg_enc* e = (rg_enc*)he;
    rg_dec* d = (rg_dec*)hd;
    GifFileType* out = e->gif;
    /* ---- setup_frame */
    const int interlace = d->gif->Image.Interlace;
    if (e->frame_map) { GifFreeMapObject(e->frame_map); e->frame_map = NULL; }
    if (d->gif->Image.ColorMap) e->frame_map = GifMakeMapObject(d->gif->Image.ColorMap->ColorCount, d->gif->Image.ColorMap->Colors);
    GifFreeExtensions(&out->ExtensionBlockCount, &out->ExtensionBlocks);
    for (int i = 0; i < d->gif->ExtensionBlockCount; i++) {
        ExtensionBlock* b = &d->gif->ExtensionBlocks[i];
        GifAddExtensionBlock(&out->ExtensionBlockCount, &out->ExtensionBlocks, b->Function, b->ByteCount, b->Bytes);
    }
    GraphicsControlBlock gcb;
    {
        int ok = 1;
        gcb.DisposalMode = DISPOSAL_UNSPECIFIED; gcb.UserInputFlag = 0; gcb.DelayTime = 0; gcb.TransparentColor = NO_TRANSPARENT_COLOR;
        for (int i = 0; i < out->ExtensionBlockCount; i++)
            if (out->ExtensionBlocks[i].Function == GRAPHICS_EXT_FUNC_CODE) ok = DGifExtensionToGCB(out->ExtensionBlocks[i].ByteCount, out->ExtensionBlocks[i].Bytes, &gcb) == GIF_OK;
        if (ok && gcb.TransparentColor != NO_TRANSPARENT_COLOR) {
            ColorMapObject* cm = e->frame_map ? e->frame_map : out->SColorMap;
            if (cm && !e->frame_map && gcb.TransparentColor == out->SBackGroundColor && d->bg_a == 255) {
                gcb.TransparentColor = NO_TRANSPARENT_COLOR;
                set_frame_gcb(out, &gcb);
            }
        }
    }
    /* ---- render_frame */
    if (width > out->SWidth || height > out->SHeight) return 0;
    const size_t image_size = (size_t)width * height;
    if (image_size > e->pixel_cap) { e->pixel_cap = image_size; e->pixels = (GifByteType*)realloc(e->pixels, image_size); }
    ColorMapObject* cm = e->frame_map ? e->frame_map : out->SColorMap;
    if (!cm) return 0;
    int clear = 1;
    if (e->have_first && e->prev_count == cm->ColorCount) clear = memcmp(e->prev_colors, cm->Colors, (size_t)cm->ColorCount * sizeof(GifColorType)) != 0;
    if (clear) memset(e->lut_present, 0, sizeof(e->lut_present));
    frame_gcb(out, &gcb);
    const int tr = gcb.TransparentColor, have_tr = tr != NO_TRANSPARENT_COLOR;
    const int prev_valid = e->have_first && (e->prev_disposal == DISPOSAL_UNSPECIFIED || e->prev_disposal == DISPOSE_DO_NOT);
    GifByteType* ro = e->pixels;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const uint8_t* s = frame + ((size_t)y * width + x) * 4;
            int B = s[0], G = s[1], R = s[2], A = s[3];
            if (A < 128 && have_tr) { *ro++ = (GifByteType)tr; continue; }
            unsigned crushed = ((unsigned)(R >> 3) << 10) | ((unsigned)(G >> 3) << 5) | (unsigned)(B >> 3);
            int least = INT_MAX, best = 0;
            if (!e->lut_present[crushed]) {
                int extreme = (R > 240 && G > 240 && B > 240) || (R < 15 && G < 15 && B < 15);
                int rc = extreme ? R : (R & 0xf8) | 4, gc = extreme ? G : (G & 0xf8) | 4, bc = extreme ? B : (B & 0xf8) | 4;
                for (int i = 0; i < cm->ColorCount; i++) {
                    if (i == tr) continue;
                    int dd = mdist(rc, gc, bc, cm->Colors[i].Red, cm->Colors[i].Green, cm->Colors[i].Blue);
                    if (dd < least) { least = dd; best = i; }
                }
                e->lut_present[crushed] = 1;
                e->lut_index[crushed] = (uint8_t)best;
            } else {
                best = e->lut_index[crushed];
                least = mdist(R, G, B, cm->Colors[best].Red, cm->Colors[best].Green, cm->Colors[best].Blue);
            }
            if (prev_valid && have_tr) {
                const uint8_t* q = e->prev_bgra + ((size_t)y * out->SWidth + x) * 4;
                if (mdist(R, G, B, q[2], q[1], q[0]) < least) best = tr;
            }
            *ro++ = (GifByteType)best;
        }
    memcpy(e->prev_bgra, frame, (size_t)4 * out->SWidth * out->SHeight);
    e->prev_count = cm->ColorCount;
    memcpy(e->prev_colors, cm->Colors, (size_t)cm->ColorCount * sizeof(GifColorType));
    e->prev_disposal = gcb.DisposalMode;
    /* ---- write_extensions + image */
    for (int i = 0; i < out->ExtensionBlockCount; i++) {
        ExtensionBlock* ep = &out->ExtensionBlocks[i];
        if (ep->Function != CONTINUE_EXT_FUNC_CODE && EGifPutExtensionLeader(out, ep->Function) == GIF_ERROR) return 0;
        if (EGifPutExtensionBlock(out, ep->ByteCount, ep->Bytes) == GIF_ERROR) return 0;
        if ((i == out->ExtensionBlockCount - 1 || (ep + 1)->Function != CONTINUE_EXT_FUNC_CODE) && EGifPutExtensionTrailer(out) == GIF_ERROR) return 0;
    }
    if (EGifPutImageDesc(out, 0, 0, width, height, interlace, e->frame_map) == GIF_ERROR) return 0;
    if (interlace) {
        static const int off[4] = {0, 4, 2, 1}, jump[4] = {8, 8, 4, 2};
        for (int i = 0; i < 4; i++)
            for (int j = off[i]; j < height; j += jump[i])
                if (EGifPutLine(out, e->pixels + (size_t)j * width, width) == GIF_ERROR) return 0;
    } else
        for (int i = 0; i < height; i++)
            if (EGifPutLine(out, e->pixels + (size_t)i * width, width) == GIF_ERROR) return 0;
    e->have_first = 1;
    return 1;
}

long rg_enc_flush(void* he, void* hd) /* giflib.cpp:1216-1250; returns the output length, -1 on error */
{
    rg_enc* e = (rg_enc*)he;
    rg_dec* d = (rg_dec*)hd;
    GifFileType* out = e->gif;
    GifFreeExtensions(&out->ExtensionBlockCount, &out->ExtensionBlocks);
    for (int i = 0; i < d->gif->ExtensionBlockCount; i++) {
        ExtensionBlock* b = &d->gif->ExtensionBlocks[i];
        GifAddExtensionBlock(&out->ExtensionBlockCount, &out->ExtensionBlocks, b->Function, b->ByteCount, b->Bytes);
    }
    for (int i = 0; i < out->ExtensionBlockCount; i++) {
        ExtensionBlock* ep = &out->ExtensionBlocks[i];
        if (ep->Function != CONTINUE_EXT_FUNC_CODE && EGifPutExtensionLeader(out, ep->Function) == GIF_ERROR) return -1;
        if (EGifPutExtensionBlock(out, ep->ByteCount, ep->Bytes) == GIF_ERROR) return -1;
        if ((i == out->ExtensionBlockCount - 1 || (ep + 1)->Function != CONTINUE_EXT_FUNC_CODE) && EGifPutExtensionTrailer(out) == GIF_ERROR) return -1;
    }
    if (EGifCloseFile(out, NULL) == GIF_ERROR) { e->gif = NULL; return -1; }
    e->gif = NULL;
    return (long)e->off;
}

void rg_enc_release(void* he)
{
    rg_enc* e = (rg_enc*)he;
    int err = 0;
    if (!e) return;
    if (e->gif) EGifCloseFile(e->gif, &err);
    if (e->frame_map) GifFreeMapObject(e->frame_map);
    free(e->pixels);
    free(e->prev_bgra);
    free(e);
}
