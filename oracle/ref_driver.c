/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin driver around the REFERENCE's own prebuilt libjpeg-turbo 3.1.0
 * (/root/reference/deps/linux/amd64/lib/libjpeg.a, headers in deps/linux/amd64/include),
 * calling it exactly the way the reference's patched OpenCV imgcodecs does for
 *   opencv_decoder_read_data  (/root/reference/opencv.cpp:166-171): library defaults
 *       (JDCT_ISLOW, do_fancy_upsampling), out_color_space JCS_EXT_BGR / JCS_GRAYSCALE,
 *       one jpeg_read_scanlines() per row;
 *   opencv_encoder_write      (/root/reference/opencv.cpp:185-194): jpeg_set_defaults,
 *       jpeg_set_quality(q, TRUE), in_color_space JCS_EXT_BGR / JCS_GRAYSCALE.
 * Built by oracle/Makefile into oracle/_ref/libref.so (git-ignored; travels to the GPU box).
 * No reference SOURCE is copied: this links the prebuilt archive where it lies.
 */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <setjmp.h>
#include <jpeglib.h>

struct err_mgr { struct jpeg_error_mgr pub; jmp_buf jb; };
static void on_error(j_common_ptr c) { longjmp(((struct err_mgr*)c->err)->jb, 1); }
static void on_msg(j_common_ptr c) { (void)c; }

/* set by the Python side to librefcv.so's ref_cv_cmyk2bgr when that library could be built */
static void (*ref_cmyk2bgr_hook)(const uint8_t* cmyk, uint8_t* bgr, int n) = 0;
void ref_set_cmyk2bgr(void (*f)(const uint8_t*, uint8_t*, int)) { ref_cmyk2bgr_hook = f; }

/* 1 (default, libjpeg's and cv::JpegDecoder's): progressive files whose scans leave low AC coefficients short of full precision get
 * libjpeg's interblock smoothing (jdcoefct.c decompress_smooth_data); 0: cinfo.do_block_smoothing = FALSE -- the plain pixels of the
 * coefficients, which is what the product returns for such files (it does not restate that filter: DESIGN.md 7). */
static int ref_block_smoothing = 1;
void ref_set_block_smoothing(int on) { ref_block_smoothing = on; }

int ref_jpeg_decode_pixels(const uint8_t* d, size_t n, uint8_t* out, size_t cap, int* w, int* h, int* ch)
{
    struct jpeg_decompress_struct ci;
    struct err_mgr em;
    ci.err = jpeg_std_error(&em.pub);
    em.pub.error_exit = on_error;
    em.pub.output_message = on_msg;
    if (setjmp(em.jb)) { jpeg_destroy_decompress(&ci); return -1; }
    jpeg_create_decompress(&ci);
    jpeg_mem_src(&ci, d, n);
    jpeg_read_header(&ci, TRUE);
    if (ci.num_components == 1) { ci.out_color_space = JCS_GRAYSCALE; ci.out_color_components = 1; }
    else if (ci.num_components == 3) { ci.out_color_space = JCS_EXT_BGR; ci.out_color_components = 3; }
    else if (ci.num_components == 4) { ci.out_color_space = JCS_CMYK; ci.out_color_components = 4; } /* cv::JpegDecoder::readData */
    else { jpeg_destroy_decompress(&ci); return -2; }
    if (!ref_block_smoothing) ci.do_block_smoothing = FALSE;
    jpeg_start_decompress(&ci);
    *w = (int)ci.output_width; *h = (int)ci.output_height; *ch = ci.out_color_components == 4 ? 3 : ci.out_color_components;
    size_t stride = (size_t)*w * *ch;
    if (stride * *h > cap) { jpeg_destroy_decompress(&ci); return -3; }
    if (ci.out_color_components == 4) { /* row by row through OpenCV's own icvCvt_CMYK2BGR_8u_C4C3R (ref_cv_driver.cpp, from the reference's archive) */
        if (!ref_cmyk2bgr_hook) { jpeg_destroy_decompress(&ci); return -2; }
        uint8_t* rowbuf4 = (uint8_t*)malloc((size_t)*w * 4);
        while (ci.output_scanline < ci.output_height) {
            JSAMPROW row = rowbuf4;
            uint8_t* dst = out + stride * ci.output_scanline;
            jpeg_read_scanlines(&ci, &row, 1);
            ref_cmyk2bgr_hook(rowbuf4, dst, *w);
        }
        free(rowbuf4);
    }
    while (ci.output_scanline < ci.output_height) {
        JSAMPROW row = out + stride * ci.output_scanline;
        jpeg_read_scanlines(&ci, &row, 1);
    }
    jpeg_finish_decompress(&ci);
    jpeg_destroy_decompress(&ci);
    return 0;
}

/* Quantised coefficients of one component, block raster order over the MCU-padded grid,
 * natural (row-major) order inside a block, DC absolute. */
int ref_jpeg_decode_coefs(const uint8_t* d, size_t n, int comp, int16_t* out, size_t cap_elems, int* bw, int* bh)
{
    struct jpeg_decompress_struct ci;
    struct err_mgr em;
    ci.err = jpeg_std_error(&em.pub);
    em.pub.error_exit = on_error;
    em.pub.output_message = on_msg;
    if (setjmp(em.jb)) { jpeg_destroy_decompress(&ci); return -1; }
    jpeg_create_decompress(&ci);
    jpeg_mem_src(&ci, d, n);
    jpeg_read_header(&ci, TRUE);
    jvirt_barray_ptr* arr = jpeg_read_coefficients(&ci);
    if (comp >= ci.num_components) { jpeg_destroy_decompress(&ci); return -2; }
    jpeg_component_info* cp = &ci.comp_info[comp];
    int mcus_x = (int)((ci.image_width + 8 * ci.max_h_samp_factor - 1) / (8 * ci.max_h_samp_factor));
    int mcus_y = (int)((ci.image_height + 8 * ci.max_v_samp_factor - 1) / (8 * ci.max_v_samp_factor));
    int hs = ci.num_components == 1 ? 1 : cp->h_samp_factor, vs = ci.num_components == 1 ? 1 : cp->v_samp_factor;
    *bw = mcus_x * hs; *bh = mcus_y * vs;
    if ((size_t)*bw * *bh * 64 > cap_elems) { jpeg_destroy_decompress(&ci); return -3; }
    memset(out, 0, (size_t)*bw * *bh * 128);
    /* the virtual array is allocated padded to whole MCUs (jdcoefct.c) for interleaved scans */
    for (int by = 0; by < *bh; by++) {
        if (by >= (int)cp->height_in_blocks && ci.num_components == 1) continue;
        JBLOCKARRAY rows = (*ci.mem->access_virt_barray)((j_common_ptr)&ci, arr[comp], (JDIMENSION)by, 1, FALSE);
        for (int bx = 0; bx < *bw; bx++) {
            if (bx >= (int)cp->width_in_blocks && ci.num_components == 1) continue;
            memcpy(out + ((size_t)by * *bw + bx) * 64, rows[0][bx], 128);
        }
    }
    jpeg_finish_decompress(&ci);
    jpeg_destroy_decompress(&ci);
    return 0;
}

/* Downsampled component planes straight out of the IDCT (raw_data_out), valid extent only:
 * out is [dh][dw] tightly packed. */
int ref_jpeg_decode_raw_plane(const uint8_t* d, size_t n, int comp, uint8_t* out, size_t cap, int* dw, int* dh)
{
    struct jpeg_decompress_struct ci;
    struct err_mgr em;
    ci.err = jpeg_std_error(&em.pub);
    em.pub.error_exit = on_error;
    em.pub.output_message = on_msg;
    uint8_t* bufs[3] = {0, 0, 0};
    if (setjmp(em.jb)) { jpeg_destroy_decompress(&ci); for (int c = 0; c < 3; c++) free(bufs[c]); return -1; }
    jpeg_create_decompress(&ci);
    jpeg_mem_src(&ci, d, n);
    jpeg_read_header(&ci, TRUE);
    ci.raw_data_out = TRUE;
    ci.out_color_space = ci.jpeg_color_space;
    jpeg_start_decompress(&ci);
    if (comp >= ci.num_components) { jpeg_destroy_decompress(&ci); return -2; }
    int nc = ci.num_components;
    int rows_per_imcu = ci.max_v_samp_factor * 8;
    JSAMPROW rowptr[3][16];
    JSAMPARRAY planes[3];
    size_t pw[3], ph[3];
    for (int c = 0; c < nc; c++) {
        pw[c] = (size_t)ci.comp_info[c].width_in_blocks * 8;
        ph[c] = (size_t)((ci.image_height + rows_per_imcu - 1) / rows_per_imcu) * ci.comp_info[c].v_samp_factor * 8;
        if (nc == 1) ph[c] = (size_t)((ci.image_height + 7) / 8) * 8;
        bufs[c] = (uint8_t*)malloc(pw[c] * ph[c]);
        planes[c] = rowptr[c];
    }
    while (ci.output_scanline < ci.output_height) {
        size_t imcu = ci.output_scanline / (nc == 1 ? 8 : rows_per_imcu);
        for (int c = 0; c < nc; c++) {
            int vr = (nc == 1 ? 1 : ci.comp_info[c].v_samp_factor) * 8;
            for (int r = 0; r < vr; r++) rowptr[c][r] = bufs[c] + (imcu * vr + r) * pw[c];
        }
        jpeg_read_raw_data(&ci, planes, nc == 1 ? 8 : rows_per_imcu);
    }
    *dw = (int)ci.comp_info[comp].downsampled_width;
    *dh = (int)ci.comp_info[comp].downsampled_height;
    int rc = 0;
    if ((size_t)*dw * *dh > cap) rc = -3;
    else for (int y = 0; y < *dh; y++) memcpy(out + (size_t)y * *dw, bufs[comp] + (size_t)y * pw[comp], *dw);
    jpeg_finish_decompress(&ci);
    jpeg_destroy_decompress(&ci);
    for (int c = 0; c < 3; c++) free(bufs[c]);
    return rc;
}

long ref_jpeg_encode(const uint8_t* px, int W, int H, int ch, size_t stride, int quality, uint8_t* out, size_t cap)
{
    struct jpeg_compress_struct ci;
    struct err_mgr em;
    unsigned char* mem = NULL;
    unsigned long memlen = 0;
    uint8_t* rowbuf = NULL;
    ci.err = jpeg_std_error(&em.pub);
    em.pub.error_exit = on_error;
    em.pub.output_message = on_msg;
    if (setjmp(em.jb)) { jpeg_destroy_compress(&ci); free(mem); free(rowbuf); return -1; }
    jpeg_create_compress(&ci);
    jpeg_mem_dest(&ci, &mem, &memlen);
    ci.image_width = (JDIMENSION)W;
    ci.image_height = (JDIMENSION)H;
    ci.input_components = ch > 1 ? 3 : 1;
    ci.in_color_space = ch > 1 ? JCS_EXT_BGR : JCS_GRAYSCALE;
    jpeg_set_defaults(&ci);
    if (quality < 0) quality = 0;
    if (quality > 100) quality = 100;
    jpeg_set_quality(&ci, quality, TRUE);
    jpeg_start_compress(&ci, TRUE);
    if (ch == 4) rowbuf = (uint8_t*)malloc((size_t)W * 3);
    while (ci.next_scanline < ci.image_height) {
        const uint8_t* src = px + stride * ci.next_scanline;
        JSAMPROW row = (JSAMPROW)src;
        if (ch == 4) {
            for (int x = 0; x < W; x++) { rowbuf[3 * x] = src[4 * x]; rowbuf[3 * x + 1] = src[4 * x + 1]; rowbuf[3 * x + 2] = src[4 * x + 2]; }
            row = rowbuf;
        }
        jpeg_write_scanlines(&ci, &row, 1);
    }
    jpeg_finish_compress(&ci);
    jpeg_destroy_compress(&ci);
    free(rowbuf);
    long rc = (long)memlen;
    if (memlen > cap) rc = -3; else memcpy(out, mem, memlen);
    free(mem);
    return rc;
}

/* Fixture generator (tests/golden/make_exotic_jpegs.py): libjpeg-turbo's compressor with the knobs Pillow does not expose.
 * px: W x H x 3 RGB (or W x H grey when ncomp == 1). mode: 0 YCbCr + JFIF, 1 RGB data kept as RGB (Adobe marker, transform 0),
 * 2 YCbCr without JFIF but with an Adobe marker (transform 1), 3 YCbCr with neither marker.
 * samp = {h0, v0, h1, v1, h2, v2}; force_baseline 0 lets low qualities produce 16-bit quantisation tables (SOF1). */
long ref_jpeg_encode_ex(const uint8_t* px, int W, int H, int ncomp, int mode, const int samp[6], int quality, int force_baseline, int restart_interval,
                        int optimize, uint8_t* out, size_t cap)
{
    struct jpeg_compress_struct ci;
    struct err_mgr em;
    unsigned char* mem = NULL;
    unsigned long memlen = 0;
    ci.err = jpeg_std_error(&em.pub);
    em.pub.error_exit = on_error;
    em.pub.output_message = on_msg;
    if (setjmp(em.jb)) { jpeg_destroy_compress(&ci); free(mem); return -1; }
    jpeg_create_compress(&ci);
    jpeg_mem_dest(&ci, &mem, &memlen);
    ci.image_width = (JDIMENSION)W;
    ci.image_height = (JDIMENSION)H;
    ci.input_components = ncomp;
    ci.in_color_space = ncomp == 1 ? JCS_GRAYSCALE : ncomp == 4 ? JCS_CMYK : JCS_RGB;
    jpeg_set_defaults(&ci);
    /* four components: px is CMYK. mode 0: stored as CMYK with an Adobe marker (transform 0); 1: as YCCK (transform 2); 2: CMYK, no
     * Adobe marker; 3: YCCK data without the marker (a decoder then takes it for CMYK) */
    if (ncomp == 4 && (mode == 1 || mode == 3)) jpeg_set_colorspace(&ci, JCS_YCCK);
    if (ncomp == 4 && mode >= 2) ci.write_Adobe_marker = FALSE;
    if (ncomp == 3 && mode == 1) jpeg_set_colorspace(&ci, JCS_RGB);
    if (ncomp == 3 && mode >= 2) { ci.write_JFIF_header = FALSE; ci.write_Adobe_marker = mode == 2; }
    jpeg_set_quality(&ci, quality, force_baseline);
    for (int c = 0; c < ncomp; c++) { /* the fourth component (K) samples like the first */
        ci.comp_info[c].h_samp_factor = samp[2 * (c == 3 ? 0 : c)];
        ci.comp_info[c].v_samp_factor = samp[2 * (c == 3 ? 0 : c) + 1];
    }
    ci.restart_interval = (unsigned)restart_interval;
    ci.optimize_coding = optimize & 1;
    /* optimize bit 1: jpeg_simple_progression; bit 2: spectral selection only, one DC scan per component;
     * bit 3: deep successive approximation (Al = 3 down to 0) with the AC band of every component split in two */
    static jpeg_scan_info script[64];
    if (optimize & 2) jpeg_simple_progression(&ci);
    if (optimize & 12) {
        int n = 0;
        if (optimize & 4) {
            for (int c = 0; c < ncomp; c++) { jpeg_scan_info si = {1, {c, 0, 0, 0}, 0, 0, 0, 0}; script[n++] = si; }
            for (int c = 0; c < ncomp; c++) {
                jpeg_scan_info a = {1, {c, 0, 0, 0}, 1, 9, 0, 0}, b = {1, {c, 0, 0, 0}, 10, 63, 0, 0};
                script[n++] = a; script[n++] = b;
            }
        } else {
            jpeg_scan_info dc = {ncomp, {0, 1, 2, 0}, 0, 0, 0, 3};
            script[n++] = dc;
            for (int c = 0; c < ncomp; c++) {
                jpeg_scan_info a = {1, {c, 0, 0, 0}, 1, 20, 0, 3}, b = {1, {c, 0, 0, 0}, 21, 63, 0, 3};
                script[n++] = a; script[n++] = b;
            }
            for (int al = 2; al >= 0; al--) {
                jpeg_scan_info dr = {ncomp, {0, 1, 2, 0}, 0, 0, al + 1, al};
                script[n++] = dr;
                for (int c = 0; c < ncomp; c++) {
                    jpeg_scan_info a = {1, {c, 0, 0, 0}, 1, 20, al + 1, al}, b = {1, {c, 0, 0, 0}, 21, 63, al + 1, al};
                    script[n++] = a; script[n++] = b;
                }
            }
        }
        ci.scan_info = script;
        ci.num_scans = n;
    }
    /* bit 4: SEQUENTIAL file with one scan per component (non-interleaved); bit 5: a Huffman table pair of its own for every
     * component (table numbers 0, 1, 2 -> SOF1, needs optimize bit 0); bit 6 (with bit 4): first scan holds components 0 and 1
     * interleaved, the second the last one */
    if (optimize & 16) {
        int n = 0;
        if ((optimize & 64) && ncomp == 3) {
            jpeg_scan_info a = {2, {0, 1, 0, 0}, 0, 63, 0, 0}, b = {1, {2, 0, 0, 0}, 0, 63, 0, 0};
            script[n++] = a; script[n++] = b;
        } else
            for (int c = 0; c < ncomp; c++) { jpeg_scan_info si = {1, {c, 0, 0, 0}, 0, 63, 0, 0}; script[n++] = si; }
        ci.scan_info = script;
        ci.num_scans = n;
    }
    if (optimize & 32)
        for (int c = 0; c < ncomp; c++) { ci.comp_info[c].dc_tbl_no = c; ci.comp_info[c].ac_tbl_no = c; }
    /* bit 7: arithmetic coding (SOF9, or SOF10 with a progressive script); bit 8: conditioning values other than the defaults, so that
     * the file carries DAC segments (T.81 B.2.4.3); bit 9 (with bit 7): a table number of its own for every component */
    if (optimize & 128) ci.arith_code = TRUE;
    if (optimize & 256) {
        ci.arith_dc_L[0] = 1; ci.arith_dc_U[0] = 4; ci.arith_ac_K[0] = 3;
        ci.arith_dc_L[1] = 0; ci.arith_dc_U[1] = 2; ci.arith_ac_K[1] = 7;
        ci.arith_dc_L[2] = 2; ci.arith_dc_U[2] = 2; ci.arith_ac_K[2] = 1;
    }
    if (optimize & 512)
        for (int c = 0; c < ncomp; c++) { ci.comp_info[c].dc_tbl_no = c; ci.comp_info[c].ac_tbl_no = c; }
    jpeg_start_compress(&ci, TRUE);
    while (ci.next_scanline < ci.image_height) {
        JSAMPROW row = (JSAMPROW)(px + (size_t)ci.next_scanline * W * ncomp);
        jpeg_write_scanlines(&ci, &row, 1);
    }
    jpeg_finish_compress(&ci);
    jpeg_destroy_compress(&ci);
    long rc = (long)memlen;
    if (memlen > cap) rc = -3; else memcpy(out, mem, memlen);
    free(mem);
    return rc;
}
