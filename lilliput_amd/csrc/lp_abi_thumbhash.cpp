// lp_abi_thumbhash.cpp -- the reference's thumbhash.hpp C ABI (/root/reference/thumbhash.hpp:12-16, implemented by
// thumbhash.cpp:17-282; Go caller thumbhash.go:21-54). ThumbHash looks at no more than 100 x 100 nearest-neighbour samples of
// the frame: the samples are gathered on the device from wherever the decode / orientation / resize left the frame
// (k_gather_samples: at most 40 KB come back), and the hash itself -- a few hundred thousand float operations whose summation
// ORDER decides the low bits -- is computed on the host in the reference's order. The reference's own known answers
// (/root/reference/thumbhash_test.go:63-81) are reproduced through this path in tests/test_thumbhash.py.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "lp_abi.h"
#include "lp_launch.h"

struct thumbhash_encoder_struct {
    uint8_t* dst;
    size_t dst_len;
};

namespace {
const size_t kMaxDimension = 100;
const float kPi = 3.14159265f;

// thumbhash.cpp:27-72 encode_channel
void encode_channel(const std::vector<float>& ch, size_t nx, size_t ny, size_t w, size_t h, float* dc, std::vector<float>* ac, float* scale)
{
    *dc = 0.0f;
    *scale = 0.0f;
    ac->clear();
    std::vector<float> fx(w, 0.0f);
    for (size_t cy = 0; cy < ny; ++cy)
        for (size_t cx = 0; cx * ny < nx * (ny - cy); ++cx) {
            float f = 0.0f;
            for (size_t x = 0; x < w; ++x) fx[x] = (float)cos(kPi / (float)w * (float)cx * ((float)x + 0.5f));
            for (size_t y = 0; y < h; ++y) {
                const float fy = (float)cos(kPi / (float)h * (float)cy * ((float)y + 0.5f));
                for (size_t x = 0; x < w; ++x) f += ch[x + y * w] * fx[x] * fy;
            }
            f /= (float)(w * h);
            if (cx > 0 || cy > 0) {
                ac->push_back(f);
                *scale = std::max(fabsf(f), *scale);
            } else
                *dc = f;
        }
    if (*scale > 0.0f)
        for (float& v : *ac) v = 0.5f + 0.5f / *scale * v;
}
}

extern "C" {

thumbhash_encoder thumbhash_encoder_create(void* buf, size_t buf_len) // thumbhash.cpp:17-25
{
    auto e = new thumbhash_encoder_struct();
    e->dst = (uint8_t*)buf;
    e->dst_len = buf_len;
    return e;
}

int thumbhash_encoder_encode(thumbhash_encoder e, const opencv_mat opaque_frame) // thumbhash.cpp:86-277
{
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)opaque_frame));
    if (!e || !m || m->rows <= 0 || m->cols <= 0) return -1;
    const int cn = m->type == CV_8UC4 ? 4 : m->type == CV_8UC3 ? 3 : m->type == CV_8U ? 1 : 0;
    if (!cn) return -1; // "Unsupported format"
    const size_t orig_w = (size_t)m->cols, orig_h = (size_t)m->rows;
    size_t w = orig_w, h = orig_h;
    if (orig_w > kMaxDimension || orig_h > kMaxDimension) {
        const float aspect = (float)orig_w / orig_h;
        if (orig_w > orig_h) { w = kMaxDimension; h = (size_t)(w / aspect); }
        else { h = kMaxDimension; w = (size_t)(h * aspect); }
    }
    if (!w || !h) return -1; // an aspect ratio beyond 100:1 leaves no samples (the reference would divide by zero further down)
    const float row_ratio = (float)orig_h / h, col_ratio = (float)orig_w / w;
    // the sample coordinates exactly as the reference computes them (float product, truncated)
    std::vector<uint32_t> idx(w + h);
    for (size_t j = 0; j < w; j++) idx[j] = (uint32_t)std::min((size_t)((int)j * col_ratio), orig_w - 1);
    for (size_t i = 0; i < h; i++) idx[w + i] = (uint32_t)std::min((size_t)((int)i * row_ratio), orig_h - 1);
    LpEngine* eng = lp_thread_engine();
    if (!eng || !lp_mat_to_device(m, eng)) return -1;
    std::vector<uint8_t> px(w * h * (size_t)cn);
    if (eng->gather_samples(lp_mat_frame(m), idx.data(), (uint32_t)w, (uint32_t)h, px.data())) return -1;

    bool has_alpha = false;
    std::vector<float> l, p, q, a;
    l.reserve(w * h); p.reserve(w * h); q.reserve(w * h); a.reserve(w * h);
    if (cn == 4) {
        float avg_r = 0.0f, avg_g = 0.0f, avg_b = 0.0f, avg_a = 0.0f;
        for (size_t k = 0; k < w * h; k++) {
            const uint8_t* s = &px[4 * k];
            const float alpha = (float)s[3] / 255.0f;
            avg_b += (alpha / 255.0f) * (float)s[0];
            avg_g += (alpha / 255.0f) * (float)s[1];
            avg_r += (alpha / 255.0f) * (float)s[2];
            avg_a += alpha;
        }
        if (avg_a > 0.0f) { avg_r /= avg_a; avg_g /= avg_a; avg_b /= avg_a; }
        has_alpha = avg_a < (float)(w * h);
        for (size_t k = 0; k < w * h; k++) {
            const uint8_t* s = &px[4 * k];
            const float alpha = (float)s[3] / 255.0f;
            const float b = avg_b * (1.0f - alpha) + (alpha / 255.0f) * (float)s[0];
            const float g = avg_g * (1.0f - alpha) + (alpha / 255.0f) * (float)s[1];
            const float r = avg_r * (1.0f - alpha) + (alpha / 255.0f) * (float)s[2];
            l.push_back((r + g + b) / 3.0f);
            p.push_back((r + g) / 2.0f - b);
            q.push_back(r - g);
            a.push_back(alpha);
        }
    } else if (cn == 3) {
        for (size_t k = 0; k < w * h; k++) {
            const uint8_t* s = &px[3 * k];
            const float b = (1.0f / 255.0f) * (float)s[0], g = (1.0f / 255.0f) * (float)s[1], r = (1.0f / 255.0f) * (float)s[2];
            l.push_back((r + g + b) / 3.0f);
            p.push_back((r + g) / 2.0f - b);
            q.push_back(r - g);
            a.push_back(1.0f);
        }
    } else {
        for (size_t k = 0; k < w * h; k++) {
            l.push_back((float)px[k] / 255.0f);
            p.push_back(0.0f);
            q.push_back(0.0f);
            a.push_back(1.0f);
        }
    }
    const size_t l_limit = has_alpha ? 5 : 7;
    const size_t lx = std::max((size_t)roundf((float)(l_limit * w) / (float)std::max(w, h)), (size_t)1);
    const size_t ly = std::max((size_t)roundf((float)(l_limit * h) / (float)std::max(w, h)), (size_t)1);
    float l_dc, l_scale, p_dc, p_scale, q_dc, q_scale, a_dc = 1.0f, a_scale = 1.0f;
    std::vector<float> l_ac, p_ac, q_ac, a_ac;
    encode_channel(l, std::max(lx, (size_t)3), std::max(ly, (size_t)3), w, h, &l_dc, &l_ac, &l_scale);
    encode_channel(p, 3, 3, w, h, &p_dc, &p_ac, &p_scale);
    encode_channel(q, 3, 3, w, h, &q_dc, &q_ac, &q_scale);
    if (has_alpha) encode_channel(a, 5, 5, w, h, &a_dc, &a_ac, &a_scale);
    const bool landscape = w > h;
    const uint32_t header24 = (uint32_t)roundf(63.0f * l_dc) | ((uint32_t)roundf(31.5f + 31.5f * p_dc) << 6) | ((uint32_t)roundf(31.5f + 31.5f * q_dc) << 12) |
                              ((uint32_t)roundf(31.0f * l_scale) << 18) | (has_alpha ? 1u << 23 : 0u);
    const uint16_t header16 = (uint16_t)((uint16_t)(landscape ? ly : lx) | ((uint16_t)roundf(63.0f * p_scale) << 3) | ((uint16_t)roundf(63.0f * q_scale) << 9) |
                                         (landscape ? 1 << 15 : 0));
    std::vector<uint8_t> hash;
    hash.reserve(25);
    hash.push_back((uint8_t)(header24 & 255));
    hash.push_back((uint8_t)((header24 >> 8) & 255));
    hash.push_back((uint8_t)(header24 >> 16));
    hash.push_back((uint8_t)(header16 & 255));
    hash.push_back((uint8_t)(header16 >> 8));
    if (has_alpha) hash.push_back((uint8_t)((uint8_t)roundf(15.0f * a_dc) | ((uint8_t)roundf(15.0f * a_scale) << 4)));
    bool odd = false;
    auto pack = [&](const std::vector<float>& ac) {
        for (float f : ac) {
            const uint8_t u = (uint8_t)roundf(15.0f * f);
            if (odd) hash.back() |= (uint8_t)(u << 4);
            else hash.push_back(u);
            odd = !odd;
        }
    };
    pack(l_ac); pack(p_ac); pack(q_ac);
    if (has_alpha) pack(a_ac);
    if (hash.size() > e->dst_len) return -1;
    memcpy(e->dst, hash.data(), hash.size());
    return (int)hash.size();
}

void thumbhash_encoder_release(thumbhash_encoder e) { delete e; }

} // extern "C"
