/* oracle/ref_png_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 * PNG decode through the REFERENCE's prebuilt libpng 1.6.47 + zlib-ng, driven the way the reference's decoder drives it:
 * /root/reference/opencv.cpp:99-171 hands the buffer to OpenCV 4.11's PngDecoder (imgcodecs grfmt_png.cpp; source not in the
 * reference tree), whose readHeader picks the Mat type from colour type / tRNS / bit depth and whose readData sets
 * strip_16 / strip_alpha or tRNS_to_alpha / palette_to_rgb / expand_gray_1_2_4_to_8 / bgr or gray_to_rgb / interlace handling.
 * lilliput asks for the decoder's own type with 16-bit depths demoted to 8 (opencv.go:250-267). The call sequence is pinned by
 * the reference's ThumbHash known answers for its five PNG fixtures (thumbhash_test.go:72-81) in tests/test_png.py. */
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libpng16/png.h>

struct src { const uint8_t* p; size_t left; };
static void rd(png_structp png, png_bytep out, png_size_t n)
{
    struct src* s = (struct src*)png_get_io_ptr(png);
    if (s->left < n) png_error(png, "read past the end");
    memcpy(out, s->p, n);
    s->p += n;
    s->left -= n;
}
static void quiet(png_structp png, png_const_charp msg) { (void)png; if (getenv("REF_META_VERBOSE")) fprintf(stderr, "libpng warning: %s\n", msg); }

/* info = {width, height, channels of the 8-bit Mat lilliput decodes into, bit_depth, color_type, interlace}; returns 0 ok,
 * -1 header error, -2 decode error, -3 cap too small. out may be NULL to read the header only. */
int ref_png_decode(const uint8_t* data, size_t n, uint8_t* out, size_t cap, int info[6])
{
    struct src s = {data, n};
    png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, NULL, NULL, quiet);
    png_infop pi = png_create_info_struct(png);
    png_infop end_info = png_create_info_struct(png); /* cv::PngDecoder keeps a second info struct for png_read_end */
    png_bytep* volatile rows = NULL;
    volatile int stage = -1;
    if (setjmp(png_jmpbuf(png))) { free((void*)rows); png_destroy_read_struct(&png, &pi, &end_info); return stage; }
    png_set_read_fn(png, &s, rd);
    png_read_info(png, pi);
    png_uint_32 w, h;
    int depth, ct, il;
    png_get_IHDR(png, pi, &w, &h, &depth, &ct, &il, NULL, NULL);
    int cn;
    if (ct == PNG_COLOR_TYPE_RGB || ct == PNG_COLOR_TYPE_PALETTE) {
        png_bytep trans; int num_trans = 0; png_color_16p tv;
        png_get_tRNS(png, pi, &trans, &num_trans, &tv);
        cn = num_trans > 0 ? 4 : 3;
    } else if (ct == PNG_COLOR_TYPE_GRAY_ALPHA || ct == PNG_COLOR_TYPE_RGB_ALPHA) cn = 4;
    else cn = 1;
    info[0] = (int)w; info[1] = (int)h; info[2] = cn; info[3] = depth; info[4] = ct; info[5] = il;
    if (!out) { png_destroy_read_struct(&png, &pi, &end_info); return 0; }
    if ((size_t)w * h * cn > cap) { png_destroy_read_struct(&png, &pi, &end_info); return -3; }
    stage = -2;
    if (depth == 16) png_set_strip_16(png);            /* the Mat is 8-bit */
    if (cn < 4) png_set_strip_alpha(png); else png_set_tRNS_to_alpha(png);
    if (ct == PNG_COLOR_TYPE_PALETTE) png_set_palette_to_rgb(png);
    if ((ct & PNG_COLOR_MASK_COLOR) == 0 && depth < 8) png_set_expand_gray_1_2_4_to_8(png);
    if ((ct & PNG_COLOR_MASK_COLOR) && cn > 1) png_set_bgr(png);
    else if (cn > 1) png_set_gray_to_rgb(png);
    else png_set_rgb_to_gray(png, 1, 0.299, 0.587);
    png_set_interlace_handling(png);
    png_read_update_info(png, pi);
    rows = (png_bytep*)malloc(sizeof(png_bytep) * h);
    for (png_uint_32 y = 0; y < h; y++) rows[y] = out + (size_t)y * w * cn;
    png_read_image(png, (png_bytep*)rows);
    png_read_end(png, end_info);
    free((void*)rows);
    png_destroy_read_struct(&png, &pi, &end_info);
    return 0;
}

/* ---- PNG OUTPUT: cv::PngEncoder::write (OpenCV 4.11 grfmt_png.cpp) restated call by call over the reference's libpng + zlib-ng.
 * px: the Mat (h x w x cn, 8-bit, BGR / BGRA / grey). level < 0: no IMWRITE_PNG_COMPRESSION in the options -> filter SUB only,
 * Z_BEST_SPEED, strategy RLE; else level 0..9 with libpng's default (adaptive) filters and Z_DEFAULT_STRATEGY. */
struct sink { uint8_t* p; size_t n, cap; int ovf; };
static void wr(png_structp png, png_bytep data, png_size_t n)
{
    struct sink* s = (struct sink*)png_get_io_ptr(png);
    if (s->n + n > s->cap) { s->ovf = 1; return; }
    memcpy(s->p + s->n, data, n);
    s->n += n;
}
static void fl(png_structp png) { (void)png; }

long ref_png_encode_like_opencv(const uint8_t* px, int w, int h, int cn, int level, uint8_t* out, size_t cap)
{
    struct sink s = {out, 0, cap, 0};
    png_structp png = png_create_write_struct(PNG_LIBPNG_VER_STRING, NULL, NULL, quiet);
    png_infop pi = png_create_info_struct(png);
    png_bytep* volatile rows = NULL;
    if (setjmp(png_jmpbuf(png))) { free((void*)rows); png_destroy_write_struct(&png, &pi); return -1; }
    png_set_write_fn(png, &s, wr, fl);
    int strategy = 3; /* IMWRITE_PNG_STRATEGY_RLE */
    if (level >= 0) {
        strategy = 0; /* IMWRITE_PNG_STRATEGY_DEFAULT */
        if (level > 9) level = 9;
        png_set_compression_level(png, level);
    } else {
        png_set_filter(png, PNG_FILTER_TYPE_BASE, PNG_FILTER_SUB);
        png_set_compression_level(png, 1);
    }
    png_set_compression_strategy(png, strategy);
    png_set_IHDR(png, pi, (png_uint_32)w, (png_uint_32)h, 8, cn == 1 ? PNG_COLOR_TYPE_GRAY : cn == 3 ? PNG_COLOR_TYPE_RGB : PNG_COLOR_TYPE_RGBA,
                 PNG_INTERLACE_NONE, PNG_COMPRESSION_TYPE_DEFAULT, PNG_FILTER_TYPE_DEFAULT);
    png_write_info(png, pi);
    png_set_bgr(png);
    png_set_swap(png); /* little-endian host */
    rows = (png_bytep*)malloc(sizeof(png_bytep) * (size_t)h);
    for (int y = 0; y < h; y++) rows[y] = (png_bytep)(px + (size_t)y * w * cn);
    png_write_image(png, (png_bytep*)rows);
    png_write_end(png, pi);
    free((void*)rows);
    png_destroy_write_struct(&png, &pi);
    return s.ovf ? -3 : (long)s.n;
}
