/*
 * oracle/imgproc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the OpenCV 4.11.0 core/imgproc semantics that lilliput's hot path
 * reaches through the C shim:
 *   opencv_mat_resize                 /root/reference/opencv.cpp:196-208  -> cv::resize(INTER_AREA)
 *   opencv_mat_crop                   /root/reference/opencv.cpp:210-215  -> cv::Mat(Rect) view
 *   opencv_mat_orientation_transform  /root/reference/opencv.cpp:217-221  -> cv::ExifTransform
 *   opencv_copy_to_region(_with_alpha), opencv_mat_clear_to_transparent
 *                                     /root/reference/opencv.cpp:508-752
 * and of the ThumbHash encoder /root/reference/thumbhash.cpp:27-277 (the reference's only broad
 * pixel-level known-answer test, thumbhash_test.go:63-81, is used to pin decode + orientation).
 *
 * OpenCV itself (pinned 4.11.0 + Discord patch, deps/build-deps-linux.sh:273-275) is a third-party
 * dependency whose source and libopencv_{core,imgproc}.a are ABSENT from /root/reference
 * (.MISSING_LARGE_BLOBS), so cv::resize / flip / transpose are restated from the published upstream
 * algorithm (modules/imgproc/src/resize.cpp: resizeAreaFast_, ResizeArea_Invoker,
 * computeResizeAreaTab, resizeGeneric_/HResizeLinear/VResizeLinear; modules/core matrix_transform).
 * PARITY UNPINNED for resampled pixel values: the reference's own tests hold no golden for them
 * (SURVEY.md 8c); orientation is pinned by the sunrise.jpg ThumbHash golden (EXIF orientation 6).
 * Compiled with -ffp-contract=off: float taps are plain mul then add.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int cv_round_f(float v) { return (int)lrintf(v); } /* cvRound: round-half-even (SSE cvtss2si) */
static inline uint8_t sat_u8_f(float v)
{
    if (!(v == v)) return 0; /* NaN -> INT_MIN -> saturates to 0 */
    if (v <= -1.0f) return 0;
    if (v >= 256.0f) return 255;
    int i = cv_round_f(v);
    return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}

typedef struct { int si, di; float alpha; } lo_tap;

/* resize.cpp computeResizeAreaTab (cn folded in by caller) */
static int area_tab(int ssize, int dsize, double scale, lo_tap* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; sx++) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.) a = 1.;
            if (a > cell) a = cell;
            tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(a / cell);
        }
    }
    return k;
}

/* Exported so tests can compare the tap tables the HIP host code builds. */
int lo_area_tab(int ssize, int dsize, int* si, int* di, float* alpha)
{
    double scale = 1. / ((double)dsize / ssize);
    lo_tap* t = (lo_tap*)malloc(sizeof(lo_tap) * (size_t)(ssize * 2 + 2));
    int n = area_tab(ssize, dsize, scale, t);
    for (int i = 0; i < n; i++) { si[i] = t[i].si; di[i] = t[i].di; alpha[i] = t[i].alpha; }
    free(t);
    return n;
}

/* returns which OpenCV branch ran: 0 copy, 1 area-fast, 2 area-general, 3 linear(area-mode) */
int lo_resize_area(const uint8_t* src, int sw, int sh, size_t sstep, int cn, uint8_t* dst, int dw, int dh, size_t dstep)
{
    if (sw == dw && sh == dh) {
        for (int y = 0; y < sh; y++) memcpy(dst + y * dstep, src + y * sstep, (size_t)sw * cn);
        return 0;
    }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (scale_x >= 1 && scale_y >= 1) {
        int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y); /* saturate_cast<int>(double) = cvRound */
        int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
        if (fast) {
            int area = iscale_x * iscale_y;
            float scale = 1.f / area;
            for (int dy = 0; dy < dh; dy++) {
                uint8_t* D = dst + dy * dstep;
                int sy0 = dy * iscale_y;
                for (int dx = 0; dx < dw; dx++)
                    for (int c = 0; c < cn; c++) {
                        int sx0 = dx * iscale_x;
                        if (iscale_x == 2 && iscale_y == 2 && (cn == 1 || cn == 3 || cn == 4) && sx0 + 2 <= sw && sy0 + 2 <= sh) {
                            const uint8_t* S = src + sy0 * sstep + (size_t)sx0 * cn + c;
                            D[dx * cn + c] = (uint8_t)((S[0] + S[cn] + S[sstep] + S[sstep + cn] + 2) >> 2);
                            continue;
                        }
                        if (sy0 + iscale_y <= sh && sx0 + iscale_x <= sw) {
                            int sum = 0;
                            for (int sy = 0; sy < iscale_y; sy++)
                                for (int sx = 0; sx < iscale_x; sx++) sum += src[(sy0 + sy) * sstep + (size_t)(sx0 + sx) * cn + c];
                            D[dx * cn + c] = sat_u8_f(sum * scale);
                        } else {
                            int sum = 0, count = 0;
                            if (sy0 >= sh || sx0 >= sw) { D[dx * cn + c] = 0; continue; }
                            for (int sy = 0; sy < iscale_y && sy0 + sy < sh; sy++)
                                for (int sx = 0; sx < iscale_x && sx0 + sx < sw; sx++) { sum += src[(sy0 + sy) * sstep + (size_t)(sx0 + sx) * cn + c]; count++; }
                            D[dx * cn + c] = sat_u8_f((float)sum / count);
                        }
                    }
            }
            return 1;
        }
        lo_tap* xt = (lo_tap*)malloc(sizeof(lo_tap) * (size_t)(sw * 2 + 2));
        lo_tap* yt = (lo_tap*)malloc(sizeof(lo_tap) * (size_t)(sh * 2 + 2));
        int nx = area_tab(sw, dw, scale_x, xt), ny = area_tab(sh, dh, scale_y, yt);
        float* buf = (float*)malloc(sizeof(float) * (size_t)dw * cn);
        float* sum = (float*)calloc((size_t)dw * cn, sizeof(float));
        int prev_dy = yt[0].di;
        for (int j = 0; j < ny; j++) {
            float beta = yt[j].alpha;
            int dy = yt[j].di;
            const uint8_t* S = src + yt[j].si * sstep;
            for (int i = 0; i < dw * cn; i++) buf[i] = 0.f;
            for (int k = 0; k < nx; k++)
                for (int c = 0; c < cn; c++) buf[xt[k].di * cn + c] += S[xt[k].si * cn + c] * xt[k].alpha;
            if (dy != prev_dy) {
                uint8_t* D = dst + prev_dy * dstep;
                for (int i = 0; i < dw * cn; i++) { D[i] = sat_u8_f(sum[i]); sum[i] = beta * buf[i]; }
                prev_dy = dy;
            } else
                for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
        }
        uint8_t* D = dst + prev_dy * dstep;
        for (int i = 0; i < dw * cn; i++) D[i] = sat_u8_f(sum[i]);
        free(xt); free(yt); free(buf); free(sum);
        return 2;
    }
    /* INTER_AREA with an up-scaling axis: bilinear with area-style coefficients, 11-bit fixed point */
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ia = (short*)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        int sx = (int)floor(dx * scale_x);
        float fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);
        fx = fx <= 0 ? 0.f : fx - floorf(fx);
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ia[2 * dx] = (short)cv_round_f(c0 * 2048);
        ia[2 * dx + 1] = (short)cv_round_f(c1 * 2048);
    }
    int* r0 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    int* r1 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        int sy = (int)floor(dy * scale_y);
        float fy = (float)((dy + 1) - (sy + 1) * inv_scale_y);
        fy = fy <= 0 ? 0.f : fy - floorf(fy);
        short b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
        for (int k = 0; k < 2; k++) {
            int y = sy + k;
            if (y < 0) y = 0;
            if (y > sh - 1) y = sh - 1;
            const uint8_t* S = src + y * sstep;
            int* R = k ? r1 : r0;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sx = xofs[dx] * cn + c;
                    R[dx * cn + c] = dx < xmax ? S[sx] * ia[2 * dx] + S[sx + cn] * ia[2 * dx + 1] : S[sx] * 2048;
                }
        }
        uint8_t* D = dst + dy * dstep;
        for (int i = 0; i < dw * cn; i++) D[i] = (uint8_t)((((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ia); free(r0); free(r1);
    return 3;
}

/* cv::ExifTransform: 1 none, 2 flip-h, 3 flip-both, 4 flip-v, 5 transpose, 6 transpose+flip-h,
 * 7 transpose+flip-both, 8 transpose+flip-v. dst is tightly packed; *dw,*dh receive the new size. */
void lo_orientation(const uint8_t* src, int w, int h, size_t sstep, int cn, int orientation, uint8_t* dst, int* dw, int* dh)
{
    int swap = orientation >= 5 && orientation <= 8;
    int W = swap ? h : w, H = swap ? w : h;
    *dw = W; *dh = H;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int tx = x, ty = y; /* coordinates in the (possibly transposed) intermediate */
            switch (orientation) {
            case 2: case 6: tx = W - 1 - x; break;
            case 3: case 7: tx = W - 1 - x; ty = H - 1 - y; break;
            case 4: case 8: ty = H - 1 - y; break;
            default: break;
            }
            int sx = swap ? ty : tx, sy = swap ? tx : ty;
            memcpy(dst + ((size_t)y * W + x) * cn, src + sy * sstep + (size_t)sx * cn, cn);
        }
}

/* opencv_copy_to_region_with_alpha (opencv.cpp:556-667), src size == roi size. In place on dst ROI. */
int lo_blend_alpha(const uint8_t* src, size_t sstep, int scn, uint8_t* dst, size_t dstep, int dcn, int w, int h)
{
    if ((scn != 1 && scn != 3 && scn != 4) || (dcn != 3 && dcn != 4)) return 1;
    const float k = (float)(1.0 / 255.0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + y * sstep + (size_t)x * scn;
            uint8_t* d = dst + y * dstep + (size_t)x * dcn;
            uint8_t s4[4] = {s[0], scn == 1 ? s[0] : s[1], scn == 1 ? s[0] : s[2], scn == 4 ? s[3] : 255};
            uint8_t d4[4] = {d[0], d[1], d[2], dcn == 4 ? d[3] : 255};
            float sa = s4[3] * k, da = d4[3] * k;
            float om = 1.0f - sa;
            float oa = sa + da * om;
            for (int c = 0; c < 3; c++) {
                float sc = s4[c] * k, dc = d4[c] * k;
                float t1 = sc * sa, t2 = dc * da, t3 = t2 * om;
                float bl = (t1 + t3) / oa;
                d[c] = sat_u8_f(bl * 255.0f);
            }
            if (dcn == 4) d[3] = sat_u8_f(oa * 255.0f);
        }
    return 0;
}

/* ------------------------------------------------------------------ ThumbHash (thumbhash.cpp:27-277) */
static void th_channel(const float* ch, size_t nx, size_t ny, size_t w, size_t h, float* dc, float* ac, size_t* nac, float* scale)
{
    const float PI = 3.14159265f;
    float* fx = (float*)malloc(sizeof(float) * w);
    *dc = 0; *scale = 0; *nac = 0;
    for (size_t cy = 0; cy < ny; cy++)
        for (size_t cx = 0; cx * ny < nx * (ny - cy); cx++) {
            float f = 0;
            for (size_t x = 0; x < w; x++) fx[x] = (float)cos(PI / (float)w * (float)cx * ((float)x + 0.5f));
            for (size_t y = 0; y < h; y++) {
                float fy = (float)cos(PI / (float)h * (float)cy * ((float)y + 0.5f));
                for (size_t x = 0; x < w; x++) f += ch[x + y * w] * fx[x] * fy;
            }
            f /= (float)(w * h);
            if (cx > 0 || cy > 0) { ac[(*nac)++] = f; if (fabsf(f) > *scale) *scale = fabsf(f); }
            else *dc = f;
        }
    if (*scale > 0.0f) for (size_t i = 0; i < *nac; i++) ac[i] = 0.5f + 0.5f / *scale * ac[i];
    free(fx);
}

int lo_thumbhash(const uint8_t* px, int ow, int oh, size_t step, int cn, uint8_t* out, size_t cap)
{
    size_t w = ow, h = oh;
    if (ow > 100 || oh > 100) {
        float ar = (float)ow / oh;
        if (ow > oh) { w = 100; h = (size_t)(w / ar); } else { h = 100; w = (size_t)(h * ar); }
    }
    float rr = (float)oh / h, cr = (float)ow / w;
    float *l = malloc(4 * w * h), *p = malloc(4 * w * h), *q = malloc(4 * w * h), *a = malloc(4 * w * h);
    int has_alpha = 0;
    if (cn == 4) {
        float ar_ = 0, ag = 0, ab = 0, aa = 0;
        for (size_t i = 0; i < h; i++)
            for (size_t j = 0; j < w; j++) {
                const uint8_t* x = px + (size_t)(i * rr) * step + (size_t)(j * cr) * 4;
                float al = x[3] / 255.0f;
                ab += (al / 255.0f) * x[0]; ag += (al / 255.0f) * x[1]; ar_ += (al / 255.0f) * x[2]; aa += al;
            }
        if (aa > 0.0f) { ar_ /= aa; ag /= aa; ab /= aa; }
        has_alpha = aa < (float)(w * h);
        for (size_t i = 0; i < h; i++)
            for (size_t j = 0; j < w; j++) {
                const uint8_t* x = px + (size_t)(i * rr) * step + (size_t)(j * cr) * 4;
                float al = x[3] / 255.0f;
                float b = ab * (1.0f - al) + (al / 255.0f) * x[0], g = ag * (1.0f - al) + (al / 255.0f) * x[1],
                      r = ar_ * (1.0f - al) + (al / 255.0f) * x[2];
                size_t k = i * w + j;
                l[k] = (r + g + b) / 3.0f; p[k] = (r + g) / 2.0f - b; q[k] = r - g; a[k] = al;
            }
    } else if (cn == 3) {
        for (size_t i = 0; i < h; i++)
            for (size_t j = 0; j < w; j++) {
                const uint8_t* x = px + (size_t)(i * rr) * step + (size_t)(j * cr) * 3;
                float b = (1.0f / 255.0f) * x[0], g = (1.0f / 255.0f) * x[1], r = (1.0f / 255.0f) * x[2];
                size_t k = i * w + j;
                l[k] = (r + g + b) / 3.0f; p[k] = (r + g) / 2.0f - b; q[k] = r - g; a[k] = 1.0f;
            }
    } else if (cn == 1) {
        for (size_t i = 0; i < h; i++)
            for (size_t j = 0; j < w; j++) {
                size_t k = i * w + j;
                l[k] = px[(size_t)(i * rr) * step + (size_t)(j * cr)] / 255.0f; p[k] = 0; q[k] = 0; a[k] = 1.0f;
            }
    } else { free(l); free(p); free(q); free(a); return -1; }
    size_t ll = has_alpha ? 5 : 7, mx = w > h ? w : h;
    size_t lx = (size_t)roundf((float)(ll * w) / (float)mx), ly = (size_t)roundf((float)(ll * h) / (float)mx);
    if (lx < 1) lx = 1;
    if (ly < 1) ly = 1;
    float ldc, ls, pdc, ps, qdc, qs, adc = 1.0f, as = 1.0f;
    float lac[64], pac[16], qac[16], aac[32];
    size_t nl, np, nq, na = 0;
    th_channel(l, lx > 3 ? lx : 3, ly > 3 ? ly : 3, w, h, &ldc, lac, &nl, &ls);
    th_channel(p, 3, 3, w, h, &pdc, pac, &np, &ps);
    th_channel(q, 3, 3, w, h, &qdc, qac, &nq, &qs);
    if (has_alpha) th_channel(a, 5, 5, w, h, &adc, aac, &na, &as);
    int land = w > h;
    uint32_t h24 = (uint32_t)roundf(63.0f * ldc) | ((uint32_t)roundf(31.5f + 31.5f * pdc) << 6) |
                   ((uint32_t)roundf(31.5f + 31.5f * qdc) << 12) | ((uint32_t)roundf(31.0f * ls) << 18) | (has_alpha ? 1u << 23 : 0);
    uint16_t h16 = (uint16_t)((land ? ly : lx) | ((uint16_t)roundf(63.0f * ps) << 3) | ((uint16_t)roundf(63.0f * qs) << 9) | (land ? 1 << 15 : 0));
    uint8_t hash[64];
    size_t n = 0;
    hash[n++] = h24 & 255; hash[n++] = (h24 >> 8) & 255; hash[n++] = (uint8_t)(h24 >> 16);
    hash[n++] = h16 & 255; hash[n++] = (uint8_t)(h16 >> 8);
    if (has_alpha) hash[n++] = (uint8_t)roundf(15.0f * adc) | (uint8_t)((uint8_t)roundf(15.0f * as) << 4);
    int odd = 0;
    const float* lists[4] = {lac, pac, qac, aac};
    size_t cnts[4] = {nl, np, nq, has_alpha ? na : 0};
    for (int t = 0; t < 4; t++)
        for (size_t i = 0; i < cnts[t]; i++) {
            uint8_t u = (uint8_t)roundf(15.0f * lists[t][i]);
            if (odd) hash[n - 1] |= (uint8_t)(u << 4); else hash[n++] = u;
            odd = !odd;
        }
    free(l); free(p); free(q); free(a);
    if (n > cap) return -1;
    memcpy(out, hash, n);
    return (int)n;
}
