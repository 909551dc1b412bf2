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
#include "lp_abi_guard.h"

struct thumbhash_encoder_struct {
    uint8_t* dst;
    size_t dst_len;
};

namespace {
const size_t kMaxSamples = 100; // per axis

// ThumbHash (the public algorithm; the reference's float evaluation order is what fixes the low bits of its known answers) restated
// around three pieces of its own: planar opponent-colour channels, one cosine table per axis and channel shape, and a nibble writer.
// What must not change is the ORDER of the float operations: a coefficient is ONE accumulator running over the samples row by row,
// each term (sample * cos_x) * cos_y, divided by the sample count at the end (thumbhash.cpp:27-72); the alpha-weighted means are
// running sums in sample order (thumbhash.cpp:118-193).

// cos(pi / n * k * (i + 0.5)) for k < kmax, i < n -- evaluated in float exactly as the reference spells it, once per axis
// (the reference re-evaluates the row of x factors for every coefficient and the y factor for every row of every coefficient)
struct CosTable {
    size_t n = 0, kmax = 0;
    std::vector<float> v; // [k][i]
    void build(size_t n_, size_t kmax_)
    {
        n = n_; kmax = kmax_;
        v.resize(n * kmax);
        const float pi = 3.14159265f;
        for (size_t k = 0; k < kmax; k++)
            for (size_t i = 0; i < n; i++) v[k * n + i] = (float)cos(pi / (float)n * (float)k * ((float)i + 0.5f));
    }
    const float* row(size_t k) const { return &v[k * n]; }
};

struct ChannelCode {
    float dc = 0.0f, scale = 0.0f;
    std::vector<float> ac; // the triangle cx * ny < nx * (ny - cy), row by row, DC left out; mapped to [0, 1] once the scale is known
};

// The low-frequency cosine coefficients of one w x h channel: nx x ny triangle.
ChannelCode code_channel(const float* plane, size_t w, size_t h, size_t nx, size_t ny, const CosTable& tx, const CosTable& ty)
{
    ChannelCode out;
    const float count = (float)(w * h);
    for (size_t ky = 0; ky < ny; ky++) {
        const float* cy = ty.row(ky);
        for (size_t kx = 0; kx * ny < nx * (ny - ky); kx++) {
            const float* cx = tx.row(kx);
            float acc = 0.0f;
            const float* s = plane;
            for (size_t y = 0; y < h; y++, s += w) {
                const float fy = cy[y];
                for (size_t x = 0; x < w; x++) acc += s[x] * cx[x] * fy;
            }
            acc /= count;
            if (kx == 0 && ky == 0) { out.dc = acc; continue; }
            out.ac.push_back(acc);
            out.scale = std::max(fabsf(acc), out.scale);
        }
    }
    if (out.scale > 0.0f)
        for (float& v : out.ac) v = 0.5f + 0.5f / out.scale * v;
    return out;
}

// Four-bit values, low nibble first.
struct NibbleWriter {
    std::vector<uint8_t>& bytes;
    bool high = false;
    void put(const std::vector<float>& unit_values)
    {
        for (float f : unit_values) {
            const uint8_t u = (uint8_t)roundf(15.0f * f);
            if (high) bytes.back() |= (uint8_t)(u << 4);
            else bytes.push_back(u);
            high = !high;
        }
    }
};

// L (luminance), P (yellow - blue), Q (red - green), A planes of the sampled pixels; returns whether any sample is not opaque.
// A translucent sample is composited over the alpha-weighted mean colour of the image.
bool opponent_planes(const uint8_t* px, size_t n, int cn, float* L, float* P, float* Q, float* A)
{
    if (cn == 1) {
        for (size_t k = 0; k < n; k++) { L[k] = (float)px[k] / 255.0f; P[k] = 0.0f; Q[k] = 0.0f; A[k] = 1.0f; }
        return false;
    }
    float mean_b = 0.0f, mean_g = 0.0f, mean_r = 0.0f, alpha_sum = (float)n;
    const bool rgba = cn == 4;
    if (rgba) { // running sums in sample order
        alpha_sum = 0.0f;
        for (size_t k = 0; k < n; k++) {
            const uint8_t* s = px + 4 * k;
            const float al = (float)s[3] / 255.0f;
            mean_b += (al / 255.0f) * (float)s[0];
            mean_g += (al / 255.0f) * (float)s[1];
            mean_r += (al / 255.0f) * (float)s[2];
            alpha_sum += al;
        }
        if (alpha_sum > 0.0f) { mean_r /= alpha_sum; mean_g /= alpha_sum; mean_b /= alpha_sum; }
    }
    for (size_t k = 0; k < n; k++) {
        const uint8_t* s = px + (size_t)cn * k;
        float b, g, r, al = 1.0f;
        if (rgba) {
            al = (float)s[3] / 255.0f;
            b = mean_b * (1.0f - al) + (al / 255.0f) * (float)s[0];
            g = mean_g * (1.0f - al) + (al / 255.0f) * (float)s[1];
            r = mean_r * (1.0f - al) + (al / 255.0f) * (float)s[2];
        } else {
            b = (1.0f / 255.0f) * (float)s[0]; g = (1.0f / 255.0f) * (float)s[1]; r = (1.0f / 255.0f) * (float)s[2];
        }
        L[k] = (r + g + b) / 3.0f;
        P[k] = (r + g) / 2.0f - b;
        Q[k] = r - g;
        A[k] = al;
    }
    return rgba && alpha_sum < (float)n;
}
}

extern "C" {

thumbhash_encoder thumbhash_encoder_create(void* buf, size_t buf_len) // thumbhash.cpp:17-25
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto e = new thumbhash_encoder_struct();
    e->dst = (uint8_t*)buf;
    e->dst_len = buf_len;
    return e;
}
LP_ABI_CATCH("thumbhash_encoder_create", return nullptr)

int thumbhash_encoder_encode(thumbhash_encoder e, const opencv_mat opaque_frame) // thumbhash.cpp:86-277
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)opaque_frame));
    if (!e || !m || m->rows <= 0 || m->cols <= 0) return -1;
    const int cn = m->type == CV_8UC4 ? 4 : m->type == CV_8UC3 ? 3 : m->type == CV_8U ? 1 : 0;
    if (!cn) return -1; // "Unsupported format"
    // at most 100 x 100 nearest-neighbour samples, the longer side pinned to 100 (thumbhash.cpp:100-116)
    const size_t src_w = (size_t)m->cols, src_h = (size_t)m->rows;
    size_t w = src_w, h = src_h;
    if (src_w > kMaxSamples || src_h > kMaxSamples) {
        const float aspect = (float)src_w / src_h;
        if (src_w > src_h) { w = kMaxSamples; h = (size_t)(w / aspect); }
        else { h = kMaxSamples; w = (size_t)(h * aspect); }
    }
    if (!w || !h) return -1; // an aspect ratio beyond 100:1 leaves no samples (the reference would divide by zero further down)
    // sample coordinates: float product, truncated, clamped -- gathered on the device from wherever the frame lives
    std::vector<uint32_t> pick(w + h);
    const float step_x = (float)src_w / w, step_y = (float)src_h / h;
    for (size_t j = 0; j < w; j++) pick[j] = (uint32_t)std::min((size_t)((int)j * step_x), src_w - 1);
    for (size_t i = 0; i < h; i++) pick[w + i] = (uint32_t)std::min((size_t)((int)i * step_y), src_h - 1);
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) return -1;
    const size_t n = w * h;
    std::vector<uint8_t> px(n * (size_t)cn);
    if (eng->gather_samples(lp_mat_frame(m), pick.data(), (uint32_t)w, (uint32_t)h, px.data())) return -1;

    std::vector<float> planes(4 * n);
    float *L = planes.data(), *P = L + n, *Q = P + n, *A = Q + n;
    const bool translucent = opponent_planes(px.data(), n, cn, L, P, Q, A);

    // luminance keeps up to 7 (5 with alpha) coefficients along the longer side, at least 3 per axis; chroma 3 x 3; alpha 5 x 5
    const size_t longest = std::max(w, h), budget = translucent ? 5 : 7;
    const size_t lum_x = std::max((size_t)roundf((float)(budget * w) / (float)longest), (size_t)1);
    const size_t lum_y = std::max((size_t)roundf((float)(budget * h) / (float)longest), (size_t)1);
    const size_t nx = std::max(lum_x, (size_t)3), ny = std::max(lum_y, (size_t)3);
    CosTable tx, ty;
    tx.build(w, std::max(nx, (size_t)5));
    ty.build(h, std::max(ny, (size_t)5));
    const ChannelCode lum = code_channel(L, w, h, nx, ny, tx, ty);
    const ChannelCode yb = code_channel(P, w, h, 3, 3, tx, ty);
    const ChannelCode rg = code_channel(Q, w, h, 3, 3, tx, ty);
    ChannelCode alpha;
    alpha.dc = 1.0f; alpha.scale = 1.0f;
    if (translucent) alpha = code_channel(A, w, h, 5, 5, tx, ty);

    // 24-bit header: L dc (6 bits), P dc (6), Q dc (6), L scale (5), alpha flag; 16-bit header: the shorter side's coefficient count
    // (3 bits), P scale (6), Q scale (6), landscape flag (thumbhash.cpp:230-246)
    const bool landscape = w > h;
    const uint32_t head24 = (uint32_t)roundf(63.0f * lum.dc) | ((uint32_t)roundf(31.5f + 31.5f * yb.dc) << 6) | ((uint32_t)roundf(31.5f + 31.5f * rg.dc) << 12) |
                            ((uint32_t)roundf(31.0f * lum.scale) << 18) | (translucent ? 1u << 23 : 0u);
    const uint16_t head16 = (uint16_t)((uint16_t)(landscape ? lum_y : lum_x) | ((uint16_t)roundf(63.0f * yb.scale) << 3) | ((uint16_t)roundf(63.0f * rg.scale) << 9) |
                                       (landscape ? 1 << 15 : 0));
    std::vector<uint8_t> hash;
    hash.reserve(32);
    for (int k = 0; k < 3; k++) hash.push_back((uint8_t)(head24 >> (8 * k)));
    hash.push_back((uint8_t)(head16 & 255));
    hash.push_back((uint8_t)(head16 >> 8));
    if (translucent) hash.push_back((uint8_t)((uint8_t)roundf(15.0f * alpha.dc) | ((uint8_t)roundf(15.0f * alpha.scale) << 4)));
    NibbleWriter nib{hash};
    nib.put(lum.ac); nib.put(yb.ac); nib.put(rg.ac);
    if (translucent) nib.put(alpha.ac);
    if (hash.size() > e->dst_len) return -1;
    memcpy(e->dst, hash.data(), hash.size());
    return (int)hash.size();
}
LP_ABI_CATCH("thumbhash_encoder_encode", return -1)

void thumbhash_encoder_release(thumbhash_encoder e) { delete e; }

} // extern "C"
