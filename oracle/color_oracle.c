/* oracle/color_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's HDR -> SDR tone map
 * (/root/reference/color_info.cpp:80-236 tonemap_rgb_to_sdr / tonemap_rgb_8u_inplace, reached from ops.go:154-165 for a PNG whose cICP
 * chunk signals PQ or HLG). The reference's own arithmetic is the PQ / HLG inverse transfer and the primaries matrices; the tone curve
 * is OpenCV 4.11's cv::TonemapReinhard(gamma 1.0, intensity 0.6, light_adapt 0.2, color_adapt 0.3) -- modules/photo/src/tonemap.cpp,
 * restated from upstream: the OpenCV sources are not in the reference tree and libopencv_core / imgproc are missing from the mount, so
 * the prebuilt libopencv_photo.a cannot be linked. PARITY UNPINNED: nothing reference-held fixes these pixels (png_cicp_test.go only
 * asserts that tone-mapping changes the bytes); cv::log / cv::pow / cv::exp use OpenCV's own polynomial kernels where this file uses
 * libm, so agreement with the real library is expected to +-1 LSB of the 8-bit result, not bit for bit.
 * Channel order note: the reference feeds B,G,R bytes into variables it calls r,g,b and on into COLOR_RGB2GRAY and the primaries
 * matrices; this restatement keeps the channels where the reference has them. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float pq_to_linear(float x)
{
    const float m1 = 0.1593017578125f, m2 = 78.84375f, c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f;
    float xpow = powf(x, 1.0f / m2);
    float num = xpow - c1 > 0.0f ? xpow - c1 : 0.0f;
    float den = c2 - c3 * xpow;
    return powf(num / den, 1.0f / m1);
}
static float hlg_to_linear(float x)
{
    const float a = 0.17883277f, b = 0.28466892f, c = 0.55991073f;
    return x <= 0.5f ? x * x / 3.0f : (expf((x - c) / a) + b) / 12.0f;
}

/* cv::Tonemap::process with gamma 1: (src - min) / (max - min) over all three channels, as convertTo(alpha, beta) in float */
static void linear_map(float* img, size_t n3)
{
    double mn = img[0], mx = img[0];
    for (size_t i = 1; i < n3; i++) { if (img[i] < mn) mn = img[i]; if (img[i] > mx) mx = img[i]; }
    if (mx - mn > DBL_EPSILON) {
        const float a = (float)(1.0 / (mx - mn)), b = (float)(-mn / (mx - mn));
        for (size_t i = 0; i < n3; i++) img[i] = fmaf(img[i], a, b);
    }
}

static inline uint8_t sat_u8(float v)
{
    if (!(v == v)) return 0; /* cvRound(NaN) = INT_MIN -> saturates to 0 */
    long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

/* color_info.cpp:112-204 tonemap_rgb_to_sdr. src: width x height x 3 samples of `depth` bits; dst: width x height pixels of `dst_cn`
 * (3 or 4) bytes of which the first three are written. Returns 0, -1 on bad arguments. */
static int tonemap_core(const uint16_t* src, uint8_t* pixels, int width, int height, int channels, int depth, int transfer, int primaries)
{
    const size_t n = (size_t)width * height;
    float* img = (float*)malloc(n * 3 * sizeof(float));
    if (!img) return -1;
    const float scale = 1.0f / ((1 << depth) - 1);
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            float v = src[i * 3 + c] * scale;
            if (transfer == 16) v = pq_to_linear(v);
            else if (transfer == 18) v = hlg_to_linear(v);
            img[i * 3 + c] = v;
        }
    /* ---- cv::TonemapReinhard::process */
    const float gamma = 1.0f, light_adapt = 0.2f, color_adapt = 0.3f;
    float intensity = 0.6f;
    (void)gamma;
    linear_map(img, n * 3);
    double sum_log = 0, sum_c[3] = {0, 0, 0}, sum_gray = 0, log_min = 0, log_max = 0;
    for (size_t i = 0; i < n; i++) {
        const float g = img[i * 3] * 0.299f + img[i * 3 + 1] * 0.587f + img[i * 3 + 2] * 0.114f; /* COLOR_RGB2GRAY on 32F */
        const float lg = logf(g > 1e-4f ? g : 1e-4f);
        sum_log += lg; sum_gray += g;
        for (int c = 0; c < 3; c++) sum_c[c] += img[i * 3 + c];
        if (i == 0 || lg < log_min) log_min = lg;
        if (i == 0 || lg > log_max) log_max = lg;
    }
    const float log_mean = (float)(sum_log / (double)n);
    const double key = (float)((log_max - log_mean) / (log_max - log_min));
    const float map_key = 0.3f + 0.7f * powf((float)key, 1.4f);
    intensity = expf(-intensity);
    const float gray_mean = (float)(sum_gray / (double)n);
    float glob[3];
    for (int c = 0; c < 3; c++) glob[c] = color_adapt * (float)(sum_c[c] / (double)n) + (1.0f - color_adapt) * gray_mean;
    for (size_t i = 0; i < n; i++) {
        const float g = img[i * 3] * 0.299f + img[i * 3 + 1] * 0.587f + img[i * 3 + 2] * 0.114f;
        for (int c = 0; c < 3; c++) {
            const float v = img[i * 3 + c];
            float adapt = color_adapt * v + (1.0f - color_adapt) * g;
            adapt = light_adapt * adapt + (1.0f - light_adapt) * glob[c];
            adapt = powf(intensity * adapt, map_key);
            img[i * 3 + c] = v * (1.0f / (adapt + v));
        }
    }
    linear_map(img, n * 3);
    /* ---- primaries -> BT.709 (cv::transform), gamma for linear-light input, 8-bit */
    static const float m2020[9] = {1.6605f, -0.5876f, -0.0728f, -0.1246f, 1.1329f, -0.0083f, -0.0182f, -0.1006f, 1.1187f};
    static const float mp3[9] = {1.2249f, -0.2247f, -0.0002f, -0.0420f, 1.0419f, 0.0001f, -0.0197f, 0.0754f, 0.9443f};
    static const float m601[9] = {1.0440f, -0.0440f, 0.0000f, -0.0000f, 1.0000f, 0.0000f, 0.0000f, 0.0000f, 1.0000f};
    static const float mxyz[9] = {1.0569715f, -0.2039770f, 0.0556301f, 0.0415551f, 1.8759675f, -0.9692436f, -0.4986108f, -1.5373832f, 3.2409699f};
    const float* m = primaries == 9 ? m2020 : (primaries == 12 || primaries == 11) ? mp3 : primaries == 6 ? m601 : primaries == 10 ? mxyz : NULL;
    for (size_t i = 0; i < n; i++) {
        float v[3] = {img[i * 3], img[i * 3 + 1], img[i * 3 + 2]}, o[3];
        for (int j = 0; j < 3; j++) o[j] = m ? m[j * 3] * v[0] + m[j * 3 + 1] * v[1] + m[j * 3 + 2] * v[2] : v[j];
        for (int j = 0; j < 3; j++) {
            float t = o[j];
            if (transfer == 8) t = powf(t, 1.0f / 2.2f);
            pixels[i * channels + j] = sat_u8(t * 255.0f);
        }
    }
    free(img);
    return 0;
}

int lo_tonemap_16(const uint16_t* src, uint8_t* dst, int width, int height, int depth, int transfer, int primaries)
{
    if (!src || !dst || width <= 0 || height <= 0 || depth < 1 || depth > 16) return -1;
    return tonemap_core(src, dst, width, height, 3, depth, transfer, primaries);
}

/* color_info.cpp:206-236 tonemap_rgb_8u_inplace: pixels of 3 or 4 bytes, tightly packed; alpha untouched. */
int lo_tonemap_8u_inplace(uint8_t* pixels, int width, int height, int channels, int transfer, int primaries)
{
    if (!pixels || width <= 0 || height <= 0 || (channels != 3 && channels != 4)) return -1;
    const size_t n = (size_t)width * height;
    uint16_t* wide = (uint16_t*)malloc(n * 3 * sizeof(uint16_t));
    if (!wide) return -1;
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) wide[i * 3 + c] = pixels[i * channels + c];
    const int rc = tonemap_core(wide, pixels, width, height, channels, 8, transfer, primaries);
    free(wide);
    return rc;
}
