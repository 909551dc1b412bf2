/* oracle/ref_meta_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 * Drives the REFERENCE's prebuilt libjpeg-turbo 3.1.0 / libpng 1.6.47 / zlib-ng the way the reference's metadata
 * readers do (/root/reference/opencv.cpp:252-296 ICC from JPEG APP2, :314-344 ICC from PNG iCCP, :357-395 cICP),
 * so that the host-side restatements in lilliput_amd/csrc/lp_abi_meta.cpp can be compared with the real libraries. */
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <jpeglib.h>
#include <libpng16/png.h>

struct jerr { struct jpeg_error_mgr pub; jmp_buf jb; };
static void j_fail(j_common_ptr c) { longjmp(((struct jerr*)c->err)->jb, 1); }
static void j_quiet(j_common_ptr c) { (void)c; }
static void j_quiet_lvl(j_common_ptr c, int lvl) { (void)c; (void)lvl; }

int ref_jpeg_icc(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    struct jpeg_decompress_struct ci;
    struct jerr je;
    ci.err = jpeg_std_error(&je.pub);
    je.pub.error_exit = j_fail;
    je.pub.output_message = j_quiet;
    je.pub.emit_message = j_quiet_lvl;
    if (setjmp(je.jb)) { jpeg_destroy_decompress(&ci); return 0; }
    jpeg_create_decompress(&ci);
    jpeg_mem_src(&ci, (unsigned char*)src, (unsigned long)n);
    jpeg_save_markers(&ci, JPEG_APP0 + 2, 0xFFFF);
    int got = 0;
    if (jpeg_read_header(&ci, TRUE) == JPEG_HEADER_OK) {
        JOCTET* p = NULL;
        unsigned int len = 0;
        if (jpeg_read_icc_profile(&ci, &p, &len)) {
            if (len > 0 && len <= cap) { memcpy(dst, p, len); got = (int)len; }
        }
        free(p);
    }
    jpeg_destroy_decompress(&ci);
    return got;
}

struct mem_src { const uint8_t* p; size_t left; };
static void mem_read(png_structp png, png_bytep out, png_size_t n)
{
    struct mem_src* s = (struct mem_src*)png_get_io_ptr(png);
    if (s->left < n) png_error(png, "read past the end");
    memcpy(out, s->p, n);
    s->p += n;
    s->left -= n;
}
static void p_quiet(png_structp png, png_const_charp msg) { (void)png; if (getenv("REF_META_VERBOSE")) fprintf(stderr, "libpng: %s\n", msg); }

int ref_png_icc(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    struct mem_src s = {src, n};
    png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, NULL, NULL, p_quiet);
    png_infop info = png_create_info_struct(png);
    if (setjmp(png_jmpbuf(png))) { png_destroy_read_struct(&png, &info, NULL); return 0; }
    png_set_read_fn(png, &s, mem_read);
    png_read_info(png, info);
    png_charp name;
    int ctype;
    png_bytep prof;
    png_uint_32 len;
    int got = 0;
    if (png_get_iCCP(png, info, &name, &ctype, &prof, &len) && len > 0 && len <= cap) { memcpy(dst, prof, len); got = (int)len; }
    png_destroy_read_struct(&png, &info, NULL);
    return got;
}

int ref_png_cicp(const uint8_t* src, size_t n, uint8_t out[4])
{
    struct mem_src s = {src, n};
    png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, NULL, NULL, p_quiet);
    png_infop info = png_create_info_struct(png);
    if (setjmp(png_jmpbuf(png))) { png_destroy_read_struct(&png, &info, NULL); return 0; }
    png_set_read_fn(png, &s, mem_read);
    png_read_info(png, info);
    png_byte a = 0, b = 0, c = 0, d = 0;
    int found = 0;
    if (png_get_cICP(png, info, &a, &b, &c, &d)) { out[0] = a; out[1] = b; out[2] = c; out[3] = d; found = 1; }
    png_destroy_read_struct(&png, &info, NULL);
    return found;
}
