// lp_kernels_pixel.hip -- gfx950 pixel kernels of the ImageOps.Transform path:
//   k_ycc_to_frame : fancy chroma upsampling (S3) + YCbCr->BGR (S4), the tail of opencv_decoder_read_data
//                    (/root/reference/opencv.cpp:166-171; libjpeg-turbo jdsample.c / jdcolor.c)
//   k_orient       : cv::ExifTransform index permutation (S5), opencv_mat_orientation_transform (opencv.cpp:217-221)
//   k_resize_*     : cv::resize(..., INTER_AREA) (S7), opencv_mat_resize (opencv.cpp:196-208); the crop (S6,
//                    opencv.cpp:210-215) is folded into the source offset exactly like cv::Mat(Rect) is a view
//   k_blend / k_clear : opencv_copy_to_region(_with_alpha), opencv_mat_clear_to_transparent (opencv.cpp:508-752)
// Byte/integer work, HBM-bound: no MFMA. Float taps use explicit non-fused mul/add to follow OpenCV's
// accumulation order.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <stdint.h>

#include "lp_area_core.h"
#include "lp_launch.h"
#include "lp_types.h"

#define FIX16(x) ((int32_t)((x)*65536.0 + 0.5))

__device__ __forceinline__ uint32_t clamp8(int32_t v) { return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

__device__ __forceinline__ uint32_t sat_round_u8(float v)
{
    int32_t i = __float2int_rn(v); // cvRound: round half to even
    return (uint32_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}

// One chroma sample of the upsampled plane at full-resolution (x, y); hr/vr in {1,2}.
__device__ __forceinline__ int32_t upsampled(const uint8_t* __restrict__ P, uint32_t stride, int32_t dw, int32_t dh, int32_t hr, int32_t vr,
                                             int32_t x, int32_t y)
{
    if (hr == 1 && vr == 1) return P[(size_t)y * stride + x];
    if (hr > 2 || vr > 2) return P[(size_t)(y / vr) * stride + x / hr]; // int_upsample: every ratio but 2:1 / 1:2 / 2:2 is plain replication
    // jdsample.c jinit_upsampler picks the fancy h2v1 / h2v2 routines only when downsampled_width > 2: the chroma of an image
    // up to 4 pixels wide is plainly replicated (vertically as well in the h2v2 case)
    if (hr == 2 && dw <= 2) return P[(size_t)(vr == 2 ? y >> 1 : y) * stride + (x >> 1)];
    if (hr == 2 && vr == 2) { // h2v2_fancy_upsample
        int32_t cy = y >> 1, ny = (y & 1) ? cy + 1 : cy - 1;
        ny = ny < 0 ? 0 : ny > dh - 1 ? dh - 1 : ny;
        int32_t cx = x >> 1, nx = (x & 1) ? cx + 1 : cx - 1;
        const uint8_t* r0 = P + (size_t)cy * stride;
        const uint8_t* r1 = P + (size_t)ny * stride;
        int32_t cs = 3 * r0[cx] + r1[cx];
        int32_t bias = (x & 1) ? 7 : 8;
        if (nx < 0 || nx > dw - 1) return (4 * cs + bias) >> 4;
        int32_t cn = 3 * r0[nx] + r1[nx];
        return (3 * cs + cn + bias) >> 4;
    }
    if (hr == 2) { // h2v1_fancy_upsample
        const uint8_t* r = P + (size_t)y * stride;
        int32_t cx = x >> 1, nx = (x & 1) ? cx + 1 : cx - 1;
        if (nx < 0 || nx > dw - 1) return r[cx];
        return (3 * r[cx] + r[nx] + ((x & 1) ? 2 : 1)) >> 2;
    }
    // h1v2_fancy_upsample
    int32_t cy = y >> 1, ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : ny > dh - 1 ? dh - 1 : ny;
    return (3 * P[(size_t)cy * stride + x] + P[(size_t)ny * stride + x] + ((y & 1) ? 2 : 1)) >> 2;
}

__device__ __forceinline__ void ycc_to_bgr(int32_t y, int32_t cb, int32_t cr, uint32_t& b, uint32_t& g, uint32_t& r)
{
    cb -= 128; cr -= 128;
    r = clamp8(y + ((FIX16(1.40200) * cr + 32768) >> 16));
    b = clamp8(y + ((FIX16(1.77200) * cb + 32768) >> 16));
    g = clamp8(y + ((-FIX16(0.34414) * cb - FIX16(0.71414) * cr + 32768) >> 16));
}

// planes -> interleaved frame. Thread = 4 horizontally adjacent pixels; block = 64x4 threads.
__global__ __launch_bounds__(256) void k_ycc_to_frame(const LpJpeg* __restrict__ imgs, const uint8_t* __restrict__ plane_arena,
                                                      const LpFrame* __restrict__ dsts, uint8_t* __restrict__ frame_arena)
{
    const LpJpeg& img = imgs[blockIdx.z];
    const LpFrame& f = dsts[blockIdx.z];
    const int32_t W = (int32_t)img.width, H = (int32_t)img.height;
    const int32_t x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H || f.off == 0) return; // off == 0: this image takes the fused path
    if (img.ncomp == 3 && !img.generic_sampling && img.colorspace == 2 && img.hs[0] == 2 && img.vs[0] == 2 && img.width > 4) return; // k_ycc_to_frame_420
    const uint8_t* PY = plane_arena + img.plane_off[0];
    uint8_t* out = frame_arena + f.off + (size_t)y * f.stride;
    if (img.ncomp == 1) {
        for (int i = 0; i < 4 && x0 + i < W; i++) out[x0 + i] = PY[(size_t)y * img.plane_stride[0] + x0 + i];
        return;
    }
    if (img.ncomp == 4 || img.generic_sampling) {
        // Every component through its own upsampler (jdsample.c picks one per component). Three components: the usual colour
        // conversion. Four: libjpeg hands cv::JpegDecoder CMYK rows -- the stored samples as they are, or jdcolor.c
        // ycck_cmyk_convert for YCCK data (C, M, Y = 255 - R, G, B of the YCbCr triple, range limited; K unchanged) -- and OpenCV
        // maps x -> k - ((255 - x) * k >> 8) (imgcodecs utils.cpp icvCvt_CMYK2BGR_8u_C4C3R).
        for (int i = 0; i < 4 && x0 + i < W; i++) {
            int32_t v[4] = {0, 0, 0, 0};
            for (int c = 0; c < (int)img.ncomp; c++) {
                const int32_t hr = img.hmax / img.hs[c], vr = img.vmax / img.vs[c];
                const int32_t dw = (W * img.hs[c] + img.hmax - 1) / img.hmax, dh = (H * img.vs[c] + img.vmax - 1) / img.vmax;
                v[c] = upsampled(plane_arena + img.plane_off[c], img.plane_stride[c], dw, dh, hr, vr, x0 + i, y);
            }
            uint8_t* o = out + (size_t)(x0 + i) * 3;
            if (img.ncomp == 3) {
                uint32_t b, g, r;
                if (img.colorspace == 3) { r = (uint32_t)v[0]; g = (uint32_t)v[1]; b = (uint32_t)v[2]; }
                else ycc_to_bgr(v[0], v[1], v[2], b, g, r);
                o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
                continue;
            }
            if (img.colorspace == 5) {
                const int32_t cb = v[1] - 128, cr = v[2] - 128;
                const uint32_t r = clamp8(255 - (v[0] + ((FIX16(1.40200) * cr + 32768) >> 16)));
                const uint32_t b = clamp8(255 - (v[0] + ((FIX16(1.77200) * cb + 32768) >> 16)));
                const uint32_t g = clamp8(255 - (v[0] + ((-FIX16(0.34414) * cb - FIX16(0.71414) * cr + 32768) >> 16)));
                v[0] = (int32_t)r; v[1] = (int32_t)g; v[2] = (int32_t)b;
            }
            const int32_t k = v[3];
            o[2] = (uint8_t)(k - ((255 - v[0]) * k >> 8));
            o[1] = (uint8_t)(k - ((255 - v[1]) * k >> 8));
            o[0] = (uint8_t)(k - ((255 - v[2]) * k >> 8));
        }
        return;
    }
    const uint8_t* PB = plane_arena + img.plane_off[1];
    const uint8_t* PR = plane_arena + img.plane_off[2];
    const int32_t hr = img.hmax / img.hs[1], vr = img.vmax / img.vs[1];
    const int32_t dw = (W * img.hs[1] + img.hmax - 1) / img.hmax, dh = (H * img.vs[1] + img.vmax - 1) / img.vmax;
    uint32_t px[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int32_t x = x0 + i < W ? x0 + i : W - 1;
        int32_t yy = PY[(size_t)y * img.plane_stride[0] + x];
        int32_t cb = upsampled(PB, img.plane_stride[1], dw, dh, hr, vr, x, y);
        int32_t cr = upsampled(PR, img.plane_stride[2], dw, dh, hr, vr, x, y);
        if (img.colorspace == 3) { px[3 * i] = (uint32_t)cr; px[3 * i + 1] = (uint32_t)cb; px[3 * i + 2] = (uint32_t)yy; } // RGB planes -> BGR
        else ycc_to_bgr(yy, cb, cr, px[3 * i], px[3 * i + 1], px[3 * i + 2]);
    }
    if (x0 + 4 <= W && ((f.off + (size_t)y * f.stride + (size_t)x0 * 3) & 3) == 0) {
        uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)x0 * 3);
        o[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        o[1] = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
        o[2] = px[8] | (px[9] << 8) | (px[10] << 16) | (px[11] << 24);
    } else {
        for (int i = 0; i < 4 && x0 + i < W; i++) {
            out[(size_t)(x0 + i) * 3 + 0] = (uint8_t)px[3 * i];
            out[(size_t)(x0 + i) * 3 + 1] = (uint8_t)px[3 * i + 1];
            out[(size_t)(x0 + i) * 3 + 2] = (uint8_t)px[3 * i + 2];
        }
    }
}

// planes -> interleaved BGR frame, YCbCr 4:2:0 fast path (what opencv_decoder_read_data hands back for almost every JPEG).
// Thread = 8 luma pixels x 2 rows (one chroma row of 4 samples + its two horizontal neighbours); a wave covers 512 x 2
// pixels, so its luma loads are one contiguous 512-byte run per row and its chroma loads 256 bytes per plane row. The
// vertical half of h2v2_fancy_upsample is computed once per chroma column and shared by the two output pixels under it.
__global__ __launch_bounds__(256) void k_ycc_to_frame_420(const LpJpeg* __restrict__ imgs, const uint8_t* __restrict__ plane_arena,
                                                          const LpFrame* __restrict__ dsts)
{
    const LpJpeg& img = imgs[blockIdx.z];
    const LpFrame& f = dsts[blockIdx.z];
    if (f.off == 0 || !(img.ncomp == 3 && !img.generic_sampling && img.colorspace == 2 && img.hs[0] == 2 && img.vs[0] == 2 && img.width > 4)) return; // generic kernel's job (incl. the non-fancy upsampling of very narrow images)
    const int32_t W = (int32_t)img.width, H = (int32_t)img.height;
    const int32_t x0 = (int32_t)(blockIdx.x * 64 + (threadIdx.x & 63)) * 8, cy = (int32_t)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (x0 >= W || 2 * cy >= H) return;
    const int32_t dw = (W + 1) >> 1, dh = (H + 1) >> 1, cx0 = x0 >> 1;
    const uint32_t sy_ = img.plane_stride[0], sc_ = img.plane_stride[1];
    const uint8_t* PY = plane_arena + img.plane_off[0] + (size_t)(2 * cy) * sy_ + x0;
    const uint8_t* PB = plane_arena + img.plane_off[1] + cx0;
    const uint8_t* PR = plane_arena + img.plane_off[2] + cx0;
    const bool has_l = cx0 > 0, has_r = cx0 + 4 <= dw - 1;
    // three chroma rows (above / this / below, replicated at the image edges), 6 samples each: left neighbour, 4, right neighbour
    int32_t cb[3][6], cr[3][6];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
        int32_t r = cy - 1 + rr;
        r = r < 0 ? 0 : r > dh - 1 ? dh - 1 : r;
        const uint8_t* b = PB + (size_t)r * sc_;
        const uint8_t* c = PR + (size_t)r * sc_;
        const uint32_t wb = *reinterpret_cast<const uint32_t*>(b), wc = *reinterpret_cast<const uint32_t*>(c); // planes are MCU padded: in bounds
#pragma unroll
        for (int i = 0; i < 4; i++) { cb[rr][i + 1] = (int32_t)((wb >> (8 * i)) & 255u); cr[rr][i + 1] = (int32_t)((wc >> (8 * i)) & 255u); }
        // the last valid chroma column replicates (jdsample.c works on downsampled_width, not on the padded plane)
#pragma unroll
        for (int i = 1; i < 4; i++)
            if (cx0 + i > dw - 1) { cb[rr][i + 1] = cb[rr][i]; cr[rr][i + 1] = cr[rr][i]; }
        cb[rr][0] = has_l ? (int32_t)b[-1] : cb[rr][1];
        cr[rr][0] = has_l ? (int32_t)c[-1] : cr[rr][1];
        cb[rr][5] = has_r ? (int32_t)b[4] : cb[rr][4];
        cr[rr][5] = has_r ? (int32_t)c[4] : cr[rr][4];
    }
    const int32_t KR = 32768 - 128 * FIX16(1.40200), KB = 32768 - 128 * FIX16(1.77200);
    const int32_t KG = 32768 + 128 * FIX16(0.34414) + 128 * FIX16(0.71414);
    uint8_t* out = reinterpret_cast<uint8_t*>((uintptr_t)f.off) + (size_t)(2 * cy) * f.stride + (size_t)x0 * 3;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        if (2 * cy + rr >= H) break;
        const uint2 yv = *reinterpret_cast<const uint2*>(PY + (size_t)rr * sy_);
        int32_t vb[6], vr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { vb[i] = 3 * cb[1][i] + cb[rr ? 2 : 0][i]; vr[i] = 3 * cr[1][i] + cr[rr ? 2 : 0][i]; }
        uint32_t px[24];
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int hx = 0; hx < 2; hx++) {
                const int32_t b_ = (3 * vb[i + 1] + vb[hx ? i + 2 : i] + (hx ? 7 : 8)) >> 4;
                const int32_t r_ = (3 * vr[i + 1] + vr[hx ? i + 2 : i] + (hx ? 7 : 8)) >> 4;
                const int p = 2 * i + hx;
                const int32_t yy = (int32_t)(((p < 4 ? yv.x : yv.y) >> (8 * (p & 3))) & 255u);
                px[3 * p + 2] = clamp8(yy + ((FIX16(1.40200) * r_ + KR) >> 16));
                px[3 * p + 0] = clamp8(yy + ((FIX16(1.77200) * b_ + KB) >> 16));
                px[3 * p + 1] = clamp8(yy + ((-FIX16(0.34414) * b_ - FIX16(0.71414) * r_ + KG) >> 16));
            }
        }
        uint8_t* o = out + (size_t)rr * f.stride;
        if (x0 + 8 <= W && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
            uint32_t w[6];
#pragma unroll
            for (int q = 0; q < 6; q++) w[q] = px[4 * q] | (px[4 * q + 1] << 8) | (px[4 * q + 2] << 16) | (px[4 * q + 3] << 24);
            reinterpret_cast<uint2*>(o)[0] = make_uint2(w[0], w[1]);
            reinterpret_cast<uint2*>(o)[1] = make_uint2(w[2], w[3]);
            reinterpret_cast<uint2*>(o)[2] = make_uint2(w[4], w[5]);
        } else {
#pragma unroll
            for (int p = 0; p < 8; p++)
                if (x0 + p < W) { o[3 * p] = (uint8_t)px[3 * p]; o[3 * p + 1] = (uint8_t)px[3 * p + 1]; o[3 * p + 2] = (uint8_t)px[3 * p + 2]; }
        }
    }
}

// K_resample: planes -> thumbnail in one pass (no full-resolution BGR frame). One wave per destination pixel.
// Fast path (4:2:0, even rectangle): a lane takes a 2x2 luma quad = one chroma site; general path: a lane takes
// single pixels through upsampled(). The three channel sums are reduced across the wave.
__global__ __launch_bounds__(256) void k_resample_fused(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops,
                                                        const uint8_t* __restrict__ plane_arena)
{
    const LpFusedOp& op = ops[blockIdx.y];
    if (op.fast) return; // taken by k_resample_420
    const LpJpeg& img = imgs[op.img];
    const uint32_t npx = op.dst.w * op.dst.h;
    const uint32_t pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pix >= npx) return;
    const int32_t dx = (int32_t)(pix % op.dst.w), dy = (int32_t)(pix / op.dst.w);
    const int32_t fx0 = op.x0 + dx * op.dxx + dy * op.dyx, fy0 = op.y0 + dx * op.dxy + dy * op.dyy;
    const uint8_t* PY = plane_arena + img.plane_off[0];
    const uint32_t sy_ = img.plane_stride[0];
    int32_t sb = 0, sg = 0, sr = 0;
    if (img.ncomp == 1) {
        const uint32_t n = op.rw * op.rh;
        for (uint32_t i = lane; i < n; i += 64) {
            uint32_t ry = i / op.rw, rx = i - ry * op.rw;
            sb += PY[(size_t)(fy0 + (int32_t)ry) * sy_ + fx0 + (int32_t)rx];
        }
    } else {
        const uint8_t* PB = plane_arena + img.plane_off[1];
        const uint8_t* PR = plane_arena + img.plane_off[2];
        const uint32_t sc_ = img.plane_stride[1];
        const int32_t W = (int32_t)img.width, H = (int32_t)img.height;
        const int32_t hr = img.hmax / img.hs[1], vr = img.vmax / img.vs[1];
        const int32_t dw = (W * img.hs[1] + img.hmax - 1) / img.hmax, dh = (H * img.vs[1] + img.vmax - 1) / img.vmax;
        const bool quad = hr == 2 && vr == 2 && dw > 2 && !((fx0 | fy0 | (int32_t)op.rw | (int32_t)op.rh) & 1) && img.colorspace == 2;
        if (quad) {
            const uint32_t qw = op.rw >> 1, nq = qw * (op.rh >> 1);
            for (uint32_t i = lane; i < nq; i += 64) {
                const uint32_t qy = i / qw, qx = i - qy * qw;
                const int32_t cx = (fx0 >> 1) + (int32_t)qx, cy = (fy0 >> 1) + (int32_t)qy;
                const int32_t cl = cx > 0 ? cx - 1 : 0, cr_ = cx < dw - 1 ? cx + 1 : dw - 1;
                const int32_t cu = cy > 0 ? cy - 1 : 0, cd = cy < dh - 1 ? cy + 1 : dh - 1;
                const uint8_t* bu = PB + (size_t)cu * sc_; const uint8_t* bm = PB + (size_t)cy * sc_; const uint8_t* bd = PB + (size_t)cd * sc_;
                const uint8_t* ru = PR + (size_t)cu * sc_; const uint8_t* rm = PR + (size_t)cy * sc_; const uint8_t* rd = PR + (size_t)cd * sc_;
                // vertical 3:1 blends for the three columns, upper (row 2cy) and lower (row 2cy+1) output rows
                const int32_t bml = 3 * bm[cl], bmc = 3 * bm[cx], bmr = 3 * bm[cr_];
                const int32_t rml = 3 * rm[cl], rmc = 3 * rm[cx], rmr = 3 * rm[cr_];
                const int32_t bul = bml + bu[cl], buc = bmc + bu[cx], bur = bmr + bu[cr_];
                const int32_t bdl = bml + bd[cl], bdc = bmc + bd[cx], bdr = bmr + bd[cr_];
                const int32_t rul = rml + ru[cl], ruc = rmc + ru[cx], rur = rmr + ru[cr_];
                const int32_t rdl = rml + rd[cl], rdc = rmc + rd[cx], rdr = rmr + rd[cr_];
                const int32_t cb[4] = {(3 * buc + bul + 8) >> 4, (3 * buc + bur + 7) >> 4, (3 * bdc + bdl + 8) >> 4, (3 * bdc + bdr + 7) >> 4};
                const int32_t crv[4] = {(3 * ruc + rul + 8) >> 4, (3 * ruc + rur + 7) >> 4, (3 * rdc + rdl + 8) >> 4, (3 * rdc + rdr + 7) >> 4};
                const uint8_t* y0p = PY + (size_t)(2 * cy) * sy_ + 2 * cx;
                const uint32_t ya = *reinterpret_cast<const uint16_t*>(y0p), yb = *reinterpret_cast<const uint16_t*>(y0p + sy_);
                const int32_t yy[4] = {(int32_t)(ya & 255), (int32_t)(ya >> 8), (int32_t)(yb & 255), (int32_t)(yb >> 8)};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t b, g, r;
                    ycc_to_bgr(yy[k], cb[k], crv[k], b, g, r);
                    sb += (int32_t)b; sg += (int32_t)g; sr += (int32_t)r;
                }
            }
        } else {
            const uint32_t n = op.rw * op.rh;
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t ry = i / op.rw, rx = i - ry * op.rw;
                const int32_t x = fx0 + (int32_t)rx, y = fy0 + (int32_t)ry;
                const int32_t yy = PY[(size_t)y * sy_ + x];
                const int32_t cb = upsampled(PB, sc_, dw, dh, hr, vr, x, y);
                const int32_t cr = upsampled(PR, img.plane_stride[2], dw, dh, hr, vr, x, y);
                uint32_t b, g, r;
                if (img.colorspace == 3) { b = (uint32_t)cr; g = (uint32_t)cb; r = (uint32_t)yy; }
                else ycc_to_bgr(yy, cb, cr, b, g, r);
                sb += (int32_t)b; sg += (int32_t)g; sr += (int32_t)r;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sb += __shfl_xor(sb, d, 64);
        sg += __shfl_xor(sg, d, 64);
        sr += __shfl_xor(sr, d, 64);
    }
    const uint32_t cn = op.dst.cn;
    if (lane < cn) {
        const int32_t sum = lane == 0 ? sb : lane == 1 ? sg : sr;
        uint32_t r;
        if (op.round_2x2) r = (uint32_t)(sum + 2) >> 2;
        else r = sat_round_u8(__fmul_rn((float)sum, op.inv_area));
        uint8_t* D = reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off) + (size_t)dy * op.dst.stride + (size_t)dx * cn;
        D[lane] = (uint8_t)r;
    }
}

// K_resample, 4:2:0 fast path (the BASELINE configs[1] shape: 4096x4096 -> 256x256 is a 16x16 box per thumbnail pixel).
// One THREAD per destination pixel; consecutive threads own boxes that are adjacent along the source x axis, so each
// luma row of a wave is one contiguous run of 16-byte loads and each chroma row a run of 8-byte loads. A thread walks its
// box two luma rows (= one chroma row) at a time with a sliding window of three chroma rows for the vertical half of
// h2v2_fancy_upsample (jdsample.c), then the horizontal half, YCbCr->BGR (jdcolor.c) and the integer box sums.
// RWC = chroma columns per box (box width / 2). Requirements (checked by the host, LpFusedOp::fast): YCbCr 4:2:0,
// box width in {8,16,32}, even box height, every box starts at a multiple of its width in x and at an even y.
// NB > 1 (round 5): SMALL boxes -- a thread still walks a tile of 2 RWC = 8 luma columns with the arithmetic above, but the tile is NB
// boxes side by side (box width 8 / NB: 4 or 2 pixels, LpFusedOp::fast = RWC / NB = 2 or 1) and the clamped pixels are summed per box:
// one v_sad_u8 per box of four pixels, two v_dot4_u32_u8 with byte masks where a word of four pixels holds two boxes. The host sends
// an op here when its boxes tile the 8-column grid exactly (x0 a multiple of 8, U a multiple of NB). Before this, 2 x 2 and 4 x 4 boxes
// -- a 512 x 512 or 1024 x 1024 source and a 256 x 256 thumbnail -- went through k_resample_fused's wave per destination pixel: 30 us
// per image where the 16 x 16 boxes of a 4096 x 4096 source take 7.4.
template <int RWC, int NB = 1>
__device__ __forceinline__ void resample_420_body(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops, const uint8_t* __restrict__ plane_arena)
{
    static_assert(NB == 1 || (RWC == 4 && (NB == 2 || NB == 4)), "small boxes: a tile of four chroma columns holds two or four boxes");
    const LpFusedOp& op = ops[blockIdx.y];
    if (op.fast != (uint32_t)(RWC / NB)) return;
    const LpJpeg& img = imgs[op.img];
    const bool swapped = op.dxy != 0;                     // orientations 5..8: destination x runs along source y
    const uint32_t UB = swapped ? op.dst.h : op.dst.w;    // boxes along source x
    const uint32_t U = UB / NB;                           // ... tiles along source x (NB > 1: the host checked that NB divides UB)
    const uint32_t V = swapped ? op.dst.w : op.dst.h;
    uint32_t v, ut;
    if (NB == 1) {
        const uint32_t ublocks = (U + 255u) / 256u;
        v = blockIdx.x / ublocks; ut = (blockIdx.x - v * ublocks) * 256u + threadIdx.x;
    } else { // a row of a small thumbnail is a fraction of a workgroup (64 tiles for 256 two-pixel boxes): rows share workgroups
        const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
        v = gid / U; ut = gid - v * U;
    }
    if (v >= V || ut >= U) return;
    // the tile's boxes in SOURCE order: box j is destination index u0 + j when the destination runs along +x, u0 + NB - 1 - j when mirrored
    const int32_t stepx = swapped ? op.dyx : op.dxx;
    const uint32_t u0 = ut * NB, u = (NB > 1 && stepx < 0) ? u0 + (NB - 1) : u0; // u: the box at the tile's lowest x
    const int32_t dx = (int32_t)(swapped ? v : u), dy = (int32_t)(swapped ? u : v);
    const int32_t fx0 = op.x0 + dx * op.dxx + dy * op.dyx, fy0 = op.y0 + dx * op.dxy + dy * op.dyy;
    const uint32_t sy_ = img.plane_stride[0], sc_ = img.plane_stride[1];
    const uint8_t* PY = plane_arena + img.plane_off[0] + (size_t)fy0 * sy_ + fx0;
    const uint8_t* PB = plane_arena + img.plane_off[1];
    const uint8_t* PR = plane_arena + img.plane_off[2];
    const int32_t W = (int32_t)img.width, H = (int32_t)img.height;
    const int32_t dw = (W + 1) >> 1, dh = (H + 1) >> 1;
    const int32_t cx0 = fx0 >> 1, cy0 = fy0 >> 1, nrows = (int32_t)(op.rh >> 1);
    const bool has_l = cx0 > 0, has_r = cx0 + RWC <= dw - 1;
    // Packed arithmetic (v_pk_*_u16): a chroma sample travels as {Cb, Cr} in the two halves of one register, so the vertical and the
    // horizontal pass of the upsampler run once for both planes (every intermediate fits 16 bits: at most 3 * 1020 + 1020 + 8); a pixel's
    // three channels are computed as 32-bit sums whose upper halves are the channel values, two pixels' halves are packed with one
    // v_perm, clamped with one v_sat_pk_u8_i16 and added to the box sum with one v_sad_u8. About 17 instead of 23 instructions per
    // source pixel (the kernel is instruction-issue bound, profiles/r02_e_final.md); the arithmetic is exactly jdsample.c / jdcolor.c's.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pk = [](uint32_t v) { return __builtin_bit_cast(u16x2, v); };
    // chroma row r of both planes -> RWC + 2 samples (index 0 = left neighbour, RWC + 1 = right neighbour; replicated at the edges)
    uint32_t P[3][RWC + 2];
    auto load_row = [&](int32_t r, uint32_t* row) {
        r = r < 0 ? 0 : r > dh - 1 ? dh - 1 : r;
        const uint8_t* b = PB + (size_t)r * sc_ + cx0;
        const uint8_t* c = PR + (size_t)r * sc_ + cx0;
        uint32_t wb[RWC / 4], wc[RWC / 4];
        if (RWC == 4) { wb[0] = *reinterpret_cast<const uint32_t*>(b); wc[0] = *reinterpret_cast<const uint32_t*>(c); }
        else if (RWC == 8) {
            const uint2 vb = *reinterpret_cast<const uint2*>(b), vc = *reinterpret_cast<const uint2*>(c);
            wb[0] = vb.x; wb[1 % (RWC / 4)] = vb.y; wc[0] = vc.x; wc[1 % (RWC / 4)] = vc.y;
        } else {
            const uint4 vb = *reinterpret_cast<const uint4*>(b), vc = *reinterpret_cast<const uint4*>(c);
            wb[0] = vb.x; wb[1 % (RWC / 4)] = vb.y; wb[2 % (RWC / 4)] = vb.z; wb[3 % (RWC / 4)] = vb.w;
            wc[0] = vc.x; wc[1 % (RWC / 4)] = vc.y; wc[2 % (RWC / 4)] = vc.z; wc[3 % (RWC / 4)] = vc.w;
        }
#pragma unroll
        for (int i = 0; i < RWC; i++) // byte i of the Cb word -> bits 0..7, byte i of the Cr word -> bits 16..23: one v_perm_b32
            row[i + 1] = __builtin_amdgcn_perm(wc[i >> 2], wb[i >> 2], 0x0c000c00u | ((4u + (i & 3)) << 16) | (uint32_t)(i & 3));
        row[0] = has_l ? (uint32_t)b[-1] | ((uint32_t)c[-1] << 16) : row[1];
        row[RWC + 1] = has_r ? (uint32_t)b[RWC] | ((uint32_t)c[RWC] << 16) : row[RWC];
    };
    load_row(cy0 - 1, P[0]);
    load_row(cy0, P[1]);
    uint32_t sbx[NB], sgx[NB], srx[NB];                  // per box of the tile (NB == 1: the one box)
#pragma unroll
    for (int j = 0; j < NB; j++) sbx[j] = sgx[j] = srx[j] = 0;
    const int32_t KR = 32768 - 128 * FIX16(1.40200), KB = 32768 - 128 * FIX16(1.77200);
    const int32_t KG = 32768 + 128 * FIX16(0.34414) + 128 * FIX16(0.71414);
#if defined(LP_RESAMPLE_R03) || defined(LP_RESAMPLE_NOSDWA)
    static_assert(NB == 1, "the timing builds of round 4 know the one-box kernels only");
    uint32_t &sr = srx[0], &sg = sgx[0], &sb = sbx[0];
    auto sat_pk = [](uint32_t x) { uint32_t d; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(x)); return d; }; // two signed halves -> two bytes clamped to 0..255
#endif
    // one step = two luma rows; A / B / C = the chroma rows above, at and below them. (Unrolling three steps so that the row buffers change
    // roles instead of being copied saves 20 of 457 instructions per step and measured the same: profiles/r04_g_resample_idct.md.)
    auto step = [&](const int32_t q, const uint32_t (&A)[RWC + 2], const uint32_t (&B)[RWC + 2], uint32_t (&C)[RWC + 2]) __attribute__((always_inline)) {
        load_row(cy0 + q + 1, C);
        // luma rows 2q and 2q+1 of the box
        uint32_t ly[2][RWC / 2];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint8_t* yp = PY + (size_t)(2 * q + rr) * sy_;
            if (RWC == 4) { const uint2 t = *reinterpret_cast<const uint2*>(yp); ly[rr][0] = t.x; ly[rr][1 % (RWC / 2)] = t.y; }
            else {
#pragma unroll
                for (int k4 = 0; k4 < RWC / 8; k4++) {
                    const uint4 t = *reinterpret_cast<const uint4*>(yp + 16 * k4);
                    ly[rr][(4 * k4 + 0) % (RWC / 2)] = t.x; ly[rr][(4 * k4 + 1) % (RWC / 2)] = t.y;
                    ly[rr][(4 * k4 + 2) % (RWC / 2)] = t.z; ly[rr][(4 * k4 + 3) % (RWC / 2)] = t.w;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            // vertical pass: 3 * this row + the nearer neighbour row (above for the upper output row, below for the lower one)
            u16x2 V[RWC + 2];
#pragma unroll
            for (int i = 0; i < RWC + 2; i++) V[i] = pk(B[i]) * (u16x2){3, 3} + pk(rr ? C[i] : A[i]);
#ifndef LP_RESAMPLE_R03
            // jdcolor.c ycc_rgb_convert with the luma kept out of the 32-bit arithmetic: (Y << 16 + t) >> 16 == Y + (t >> 16) for any t, so a
            // channel is Y + the upper half of its fixed-point chroma term (the -128 offsets folded into the constants). Each term is one
            // v_dot2_u32_u16 over the {Cb, Cr} pair: FIX(1.772) = 2 * 58065 and FIX(1.402) = 3 * 30627, so with {2 Cb, 3 Cr} the 17-bit
            // constants fit 16-bit operands; green: floor((KG - x) / 65536) == -floor((x + 65535 - KG) / 65536). Four pixels (two chroma
            // columns = the four bytes of one luma word) per group: the 16-bit sums Y + term are formed by SDWA adds that read the luma
            // byte and the term's upper half in place and write one half of a pair register, two pairs are clamped into the four bytes
            // of one word (v_sat_pk_u8_i16, the second one into the upper half) and summed by one v_sad_u8.
            static_assert(FIX16(1.77200) == 2 * 58065 && FIX16(1.40200) == 3 * 30627, "split of the colour constants");
#pragma unroll
            for (int i = 0; i < RWC; i += 2) {
                uint32_t Tr[4], Tg[4], Tb[4];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    // horizontal pass: even output column leans on the left neighbour (+8), odd on the right one (+7)
                    const u16x2 c3 = V[i + k + 1] * (u16x2){3, 3};
                    const u16x2 H[2] = {(u16x2)((c3 + V[i + k] + (u16x2){8, 8}) >> (u16x2){4, 4}), (u16x2)((c3 + V[i + k + 2] + (u16x2){7, 7}) >> (u16x2){4, 4})};
#pragma unroll
                    for (int hx = 0; hx < 2; hx++) {
                        const u16x2 hs = H[hx] * (u16x2){2, 3};
                        Tr[2 * k + hx] = __builtin_amdgcn_udot2(hs, (u16x2){0, 30627}, (uint32_t)KR, false);
                        Tb[2 * k + hx] = __builtin_amdgcn_udot2(hs, (u16x2){58065, 0}, (uint32_t)KB, false);
                        Tg[2 * k + hx] = __builtin_amdgcn_udot2(H[hx], (u16x2){(unsigned short)FIX16(0.34414), (unsigned short)FIX16(0.71414)}, 65535u - (uint32_t)KG, false);
                    }
                }
                const uint32_t y4 = ly[rr][i >> 1];
#ifndef LP_RESAMPLE_NOSDWA
                // One block so that every SDWA write of half a register is at least three instructions away from its reader (gfx940
                // family: a destination-select write needs one wait state before a VALU read, and the compiler does not look inside).
                uint32_t r0, r1, g0, g1, b0, b1;
#define LP_RS_PIXELS \
                    "v_add_u16_sdwa %[r0], %[tr0], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n\t" \
                    "v_sub_u16_sdwa %[g0], %[y], %[tg0] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\t" \
                    "v_add_u16_sdwa %[b0], %[tb0], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n\t" \
                    "v_add_u16_sdwa %[r1], %[tr2], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2\n\t" \
                    "v_sub_u16_sdwa %[g1], %[y], %[tg2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1\n\t" \
                    "v_add_u16_sdwa %[b1], %[tb2], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2\n\t" \
                    "v_add_u16_sdwa %[r0], %[tr1], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t" \
                    "v_sub_u16_sdwa %[g0], %[y], %[tg1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:WORD_1\n\t" \
                    "v_add_u16_sdwa %[b0], %[tb1], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t" \
                    "v_add_u16_sdwa %[r1], %[tr3], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t" \
                    "v_sub_u16_sdwa %[g1], %[y], %[tg3] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:WORD_1\n\t" \
                    "v_add_u16_sdwa %[b1], %[tb3], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t" \
                    "v_sat_pk_u8_i16 %[r0], %[r0]\n\t" \
                    "v_sat_pk_u8_i16 %[g0], %[g0]\n\t" \
                    "v_sat_pk_u8_i16 %[b0], %[b0]\n\t" \
                    "v_sat_pk_u8_i16_sdwa %[r0], %[r1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t" \
                    "v_sat_pk_u8_i16_sdwa %[g0], %[g1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t" \
                    "v_sat_pk_u8_i16_sdwa %[b0], %[b1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
                if constexpr (NB == 1) {        // the four pixels of the word belong to the thread's one box
                    asm(LP_RS_PIXELS
                        "v_sad_u8 %[sr], %[r0], 0, %[sr]\n\t"
                        "v_sad_u8 %[sg], %[g0], 0, %[sg]\n\t"
                        "v_sad_u8 %[sb], %[b0], 0, %[sb]"
                        : [r0] "=&v"(r0), [r1] "=&v"(r1), [g0] "=&v"(g0), [g1] "=&v"(g1), [b0] "=&v"(b0), [b1] "=&v"(b1),
                          [sr] "+v"(srx[0]), [sg] "+v"(sgx[0]), [sb] "+v"(sbx[0])
                        : [y] "v"(y4), [tr0] "v"(Tr[0]), [tr1] "v"(Tr[1]), [tr2] "v"(Tr[2]), [tr3] "v"(Tr[3]),
                      [tg0] "v"(Tg[0]), [tg1] "v"(Tg[1]), [tg2] "v"(Tg[2]), [tg3] "v"(Tg[3]),
                      [tb0] "v"(Tb[0]), [tb1] "v"(Tb[1]), [tb2] "v"(Tb[2]), [tb3] "v"(Tb[3]));
                } else if constexpr (NB == 2) { // ... to box i / 2 of the tile
                    asm(LP_RS_PIXELS
                        "v_sad_u8 %[sr], %[r0], 0, %[sr]\n\t"
                        "v_sad_u8 %[sg], %[g0], 0, %[sg]\n\t"
                        "v_sad_u8 %[sb], %[b0], 0, %[sb]"
                        : [r0] "=&v"(r0), [r1] "=&v"(r1), [g0] "=&v"(g0), [g1] "=&v"(g1), [b0] "=&v"(b0), [b1] "=&v"(b1),
                          [sr] "+v"(srx[(i >> 1) % NB]), [sg] "+v"(sgx[(i >> 1) % NB]), [sb] "+v"(sbx[(i >> 1) % NB])
                        : [y] "v"(y4), [tr0] "v"(Tr[0]), [tr1] "v"(Tr[1]), [tr2] "v"(Tr[2]), [tr3] "v"(Tr[3]),
                      [tg0] "v"(Tg[0]), [tg1] "v"(Tg[1]), [tg2] "v"(Tg[2]), [tg3] "v"(Tg[3]),
                      [tb0] "v"(Tb[0]), [tb1] "v"(Tb[1]), [tb2] "v"(Tb[2]), [tb3] "v"(Tb[3]));
                } else {                        // two boxes of two pixels: the lower and the upper two bytes of the word
                    asm(LP_RS_PIXELS
                        "v_dot4_u32_u8 %[srl], %[r0], %[mlo], %[srl]\n\t"
                        "v_dot4_u32_u8 %[sgl], %[g0], %[mlo], %[sgl]\n\t"
                        "v_dot4_u32_u8 %[sbl], %[b0], %[mlo], %[sbl]\n\t"
                        "v_dot4_u32_u8 %[srh], %[r0], %[mhi], %[srh]\n\t"
                        "v_dot4_u32_u8 %[sgh], %[g0], %[mhi], %[sgh]\n\t"
                        "v_dot4_u32_u8 %[sbh], %[b0], %[mhi], %[sbh]"
                        : [r0] "=&v"(r0), [r1] "=&v"(r1), [g0] "=&v"(g0), [g1] "=&v"(g1), [b0] "=&v"(b0), [b1] "=&v"(b1),
                          [srl] "+v"(srx[i % NB]), [sgl] "+v"(sgx[i % NB]), [sbl] "+v"(sbx[i % NB]),
                          [srh] "+v"(srx[(i + 1) % NB]), [sgh] "+v"(sgx[(i + 1) % NB]), [sbh] "+v"(sbx[(i + 1) % NB])
                        : [y] "v"(y4), [tr0] "v"(Tr[0]), [tr1] "v"(Tr[1]), [tr2] "v"(Tr[2]), [tr3] "v"(Tr[3]),
                      [tg0] "v"(Tg[0]), [tg1] "v"(Tg[1]), [tg2] "v"(Tg[2]), [tg3] "v"(Tg[3]),
                      [tb0] "v"(Tb[0]), [tb1] "v"(Tb[1]), [tb2] "v"(Tb[2]), [tb3] "v"(Tb[3]), [mlo] "s"(0x00000101u), [mhi] "s"(0x01010000u));
                }
#undef LP_RS_PIXELS
#else
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const u16x2 yp = pk(__builtin_amdgcn_perm(0u, y4, 0x0c000c00u | ((uint32_t)(2 * k + 1) << 16) | (uint32_t)(2 * k)));
                    const u16x2 pr = yp + pk(__builtin_amdgcn_perm(Tr[2 * k + 1], Tr[2 * k], 0x07060302u));
                    const u16x2 pg = yp - pk(__builtin_amdgcn_perm(Tg[2 * k + 1], Tg[2 * k], 0x07060302u));
                    const u16x2 pb = yp + pk(__builtin_amdgcn_perm(Tb[2 * k + 1], Tb[2 * k], 0x07060302u));
                    sr = __builtin_amdgcn_sad_u8(sat_pk(__builtin_bit_cast(uint32_t, pr)), 0u, sr);
                    sg = __builtin_amdgcn_sad_u8(sat_pk(__builtin_bit_cast(uint32_t, pg)), 0u, sg);
                    sb = __builtin_amdgcn_sad_u8(sat_pk(__builtin_bit_cast(uint32_t, pb)), 0u, sb);
                }
#endif
            }
#else
#pragma unroll
            for (int i = 0; i < RWC; i++) {
                // horizontal pass: even output column leans on the left neighbour (+8), odd on the right one (+7)
                const u16x2 c3 = V[i + 1] * (u16x2){3, 3};
                const u16x2 H[2] = {(u16x2)((c3 + V[i] + (u16x2){8, 8}) >> (u16x2){4, 4}), (u16x2)((c3 + V[i + 2] + (u16x2){7, 7}) >> (u16x2){4, 4})};
                uint32_t Tr[2], Tg[2], Tb[2];
#pragma unroll
                for (int hx = 0; hx < 2; hx++) {
                    const uint32_t h = __builtin_bit_cast(uint32_t, H[hx]);
                    const int32_t cb = (int32_t)(h & 0xffffu), cr = (int32_t)(h >> 16);
                    const int px = 2 * i + hx;
                    // the luma sample in bits 16..23: channel = (luma << 16 + fixed-point chroma term) >> 16, read off as the upper half
                    const uint32_t yk = __builtin_amdgcn_perm(0u, ly[rr][px >> 2], 0x0c000c0cu | ((uint32_t)(px & 3) << 16));
                    Tr[hx] = (uint32_t)(FIX16(1.40200) * cr + (int32_t)(yk + (uint32_t)KR));
                    Tb[hx] = (uint32_t)(FIX16(1.77200) * cb + (int32_t)(yk + (uint32_t)KB));
                    Tg[hx] = (uint32_t)(-FIX16(0.34414) * cb - FIX16(0.71414) * cr + (int32_t)(yk + (uint32_t)KG));
                }
                // upper halves of the two pixels side by side, clamped to bytes, summed
                sr = __builtin_amdgcn_sad_u8(sat_pk(__builtin_amdgcn_perm(Tr[1], Tr[0], 0x07060302u)), 0u, sr);
                sg = __builtin_amdgcn_sad_u8(sat_pk(__builtin_amdgcn_perm(Tg[1], Tg[0], 0x07060302u)), 0u, sg);
                sb = __builtin_amdgcn_sad_u8(sat_pk(__builtin_amdgcn_perm(Tb[1], Tb[0], 0x07060302u)), 0u, sb);
            }
#endif
        }
    };
    for (int32_t q = 0; q < nrows; q++) {
        step(q, P[0], P[1], P[2]);
#pragma unroll
        for (int i = 0; i < RWC + 2; i++) { P[0][i] = P[1][i]; P[1][i] = P[2][i]; }
    }
#pragma unroll
    for (int j = 0; j < NB; j++) { // box j of the tile in source order
        const uint32_t uj = NB == 1 ? u : stepx < 0 ? u0 + (uint32_t)(NB - 1 - j) : u0 + (uint32_t)j;
        const int32_t bdx = (int32_t)(swapped ? v : uj), bdy = (int32_t)(swapped ? uj : v);
        uint8_t* D = reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off) + (size_t)bdy * op.dst.stride + (size_t)bdx * 3;
        const int32_t sums[3] = {(int32_t)sbx[j], (int32_t)sgx[j], (int32_t)srx[j]};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t r;
            if (op.round_2x2) r = (uint32_t)(sums[c] + 2) >> 2;
            else r = sat_round_u8(__fmul_rn((float)sums[c], op.inv_area));
            D[c] = (uint8_t)r;
        }
    }
}

// The kernels proper. The register budget is part of the measurement (profiles/r04_g_resample.md): the 8- and 16-pixel-box instances
// are asked to fit seven waves per SIMD (58 / 72 VGPRs, nothing spilled; left alone the allocator takes 74 / 81 and the 16-pixel
// instance runs 10 % slower at five waves); the 32-pixel-box instance would spill at that budget and is left alone (105 VGPRs).
#ifndef LP_RESAMPLE_WPE
#define LP_RESAMPLE_WPE 7
#endif
template <int RWC>
__global__ __launch_bounds__(256) void k_resample_420(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops, const uint8_t* __restrict__ plane_arena)
{
    resample_420_body<RWC>(imgs, ops, plane_arena);
}
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LP_RESAMPLE_WPE))) void k_resample_420<4>(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops,
                                                                                                             const uint8_t* __restrict__ plane_arena)
{
    resample_420_body<4>(imgs, ops, plane_arena);
}
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LP_RESAMPLE_WPE))) void k_resample_420<8>(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops,
                                                                                                             const uint8_t* __restrict__ plane_arena)
{
    resample_420_body<8>(imgs, ops, plane_arena);
}

// Grey sources at integer scales: a thread per destination pixel sums its box of luma bytes (LpFusedOp::fast = LP_FAST_GRAY). The wave
// per destination pixel of k_resample_fused is for what nothing else takes.
#define LP_FAST_GRAY 0x1000u
__global__ __launch_bounds__(256) void k_resample_gray(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops, const uint8_t* __restrict__ plane_arena)
{
    const LpFusedOp& op = ops[blockIdx.y];
    if (op.fast != LP_FAST_GRAY) return;
    const LpJpeg& img = imgs[op.img];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= op.dst.w * op.dst.h) return;
    const int32_t dy = (int32_t)(gid / op.dst.w), dx = (int32_t)(gid - (uint32_t)dy * op.dst.w);
    const int32_t fx0 = op.x0 + dx * op.dxx + dy * op.dyx, fy0 = op.y0 + dx * op.dxy + dy * op.dyy;
    const uint32_t sy_ = img.plane_stride[0];
    const uint8_t* p = plane_arena + img.plane_off[0] + (size_t)fy0 * sy_ + fx0;
    int32_t sum = 0;
    for (uint32_t ry = 0; ry < op.rh; ry++, p += sy_)
        for (uint32_t rx = 0; rx < op.rw; rx++) sum += p[rx];
    uint32_t r;
    if (op.round_2x2) r = (uint32_t)(sum + 2) >> 2;
    else r = sat_round_u8(__fmul_rn((float)sum, op.inv_area));
    reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off)[(size_t)dy * op.dst.stride + (size_t)dx] = (uint8_t)r;
}

// Small boxes (2 x 2 and 4 x 4 pixels: NB = 4 / 2 boxes per 8-column tile), see resample_420_body
template <int NB>
__global__ __launch_bounds__(256) void k_resample_420_small(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops, const uint8_t* __restrict__ plane_arena)
{
    resample_420_body<4, NB>(imgs, ops, plane_arena);
}

// K_resample fast path for YCbCr 4:4:4 (HR = 1) and 4:2:2 (HR = 2, h2v1_fancy_upsample): same thread-per-thumbnail-pixel walk as
// k_resample_420 but the chroma rows are the luma rows (no vertical filter). RW = box width in pixels. LpFusedOp::fast =
// 0x100 * (1 + HR) + RW. Requirements: every box starts at a multiple of RW in x (aligned vector loads).
// Round 5: the arithmetic of k_resample_420 (packed {Cb, Cr} pairs, v_dot2 colour terms, SDWA adds, paired clamp, v_sad_u8 sums: four
// pixels per group) instead of round 2's scalar loop -- ~10 (4:4:4) / ~13 (4:2:2) instead of ~19 instructions per source pixel.
__device__ __forceinline__ void lp_ycc4_sum(uint32_t y4, const uint32_t (&Tr)[4], const uint32_t (&Tg)[4], const uint32_t (&Tb)[4], uint32_t& sr, uint32_t& sg, uint32_t& sb)
{
    // the block of resample_420_body: every SDWA write of half a register is at least three instructions away from its reader
    uint32_t r0, r1, g0, g1, b0, b1;
    asm("v_add_u16_sdwa %[r0], %[tr0], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_sub_u16_sdwa %[g0], %[y], %[tg0] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\t"
        "v_add_u16_sdwa %[b0], %[tb0], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %[r1], %[tr2], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
        "v_sub_u16_sdwa %[g1], %[y], %[tg2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_1\n\t"
        "v_add_u16_sdwa %[b1], %[tb2], %[y] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %[r0], %[tr1], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t"
        "v_sub_u16_sdwa %[g0], %[y], %[tg1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:WORD_1\n\t"
        "v_add_u16_sdwa %[b0], %[tb1], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %[r1], %[tr3], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t"
        "v_sub_u16_sdwa %[g1], %[y], %[tg3] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:WORD_1\n\t"
        "v_add_u16_sdwa %[b1], %[tb3], %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t"
        "v_sat_pk_u8_i16 %[r0], %[r0]\n\t"
        "v_sat_pk_u8_i16 %[g0], %[g0]\n\t"
        "v_sat_pk_u8_i16 %[b0], %[b0]\n\t"
        "v_sat_pk_u8_i16_sdwa %[r0], %[r1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %[g0], %[g1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %[b0], %[b1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_sad_u8 %[sr], %[r0], 0, %[sr]\n\t"
        "v_sad_u8 %[sg], %[g0], 0, %[sg]\n\t"
        "v_sad_u8 %[sb], %[b0], 0, %[sb]"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [g0] "=&v"(g0), [g1] "=&v"(g1), [b0] "=&v"(b0), [b1] "=&v"(b1),
          [sr] "+v"(sr), [sg] "+v"(sg), [sb] "+v"(sb)
        : [y] "v"(y4), [tr0] "v"(Tr[0]), [tr1] "v"(Tr[1]), [tr2] "v"(Tr[2]), [tr3] "v"(Tr[3]),
          [tg0] "v"(Tg[0]), [tg1] "v"(Tg[1]), [tg2] "v"(Tg[2]), [tg3] "v"(Tg[3]),
          [tb0] "v"(Tb[0]), [tb1] "v"(Tb[1]), [tb2] "v"(Tb[2]), [tb3] "v"(Tb[3]));
}

template <int RW, int HR>
__global__ __launch_bounds__(256) void k_resample_hv1(const LpJpeg* __restrict__ imgs, const LpFusedOp* __restrict__ ops,
                                                      const uint8_t* __restrict__ plane_arena)
{
    constexpr int CW = RW / HR; // chroma samples per box row
    const LpFusedOp& op = ops[blockIdx.y];
    if (op.fast != (uint32_t)(0x100 * (1 + HR) + RW)) return;
    const LpJpeg& img = imgs[op.img];
    const bool swapped = op.dxy != 0;
    const uint32_t U = swapped ? op.dst.h : op.dst.w, V = swapped ? op.dst.w : op.dst.h;
    const uint32_t ublocks = (U + 255u) / 256u;
    const uint32_t v = blockIdx.x / ublocks, u = (blockIdx.x - v * ublocks) * 256u + threadIdx.x;
    if (v >= V || u >= U) return;
    const int32_t dx = (int32_t)(swapped ? v : u), dy = (int32_t)(swapped ? u : v);
    const int32_t fx0 = op.x0 + dx * op.dxx + dy * op.dyx, fy0 = op.y0 + dx * op.dxy + dy * op.dyy;
    const uint32_t sy_ = img.plane_stride[0], sc_ = img.plane_stride[1];
    const int32_t W = (int32_t)img.width, dw = (W + HR - 1) / HR, cx0 = fx0 / HR;
    const uint8_t* PY = plane_arena + img.plane_off[0] + (size_t)fy0 * sy_ + fx0;
    const uint8_t* PB = plane_arena + img.plane_off[1] + (size_t)fy0 * sc_ + cx0;
    const uint8_t* PR = plane_arena + img.plane_off[2] + (size_t)fy0 * sc_ + cx0;
    const bool has_l = cx0 > 0, has_r = cx0 + CW <= dw - 1;
    const int32_t KR = 32768 - 128 * FIX16(1.40200), KB = 32768 - 128 * FIX16(1.77200);
    const int32_t KG = 32768 + 128 * FIX16(0.34414) + 128 * FIX16(0.71414);
    static_assert(FIX16(1.77200) == 2 * 58065 && FIX16(1.40200) == 3 * 30627, "split of the colour constants");
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pk = [](uint32_t v) { return __builtin_bit_cast(u16x2, v); };
    auto load_words = [](const uint8_t* p, int n, uint32_t* w) { // n in {4, 8, 16, 32}, p aligned to min(n, 16)
        if (n == 4) w[0] = *reinterpret_cast<const uint32_t*>(p);
        else if (n == 8) { const uint2 t = *reinterpret_cast<const uint2*>(p); w[0] = t.x; w[1] = t.y; }
        else
            for (int q = 0; q < n / 16; q++) { const uint4 t = *reinterpret_cast<const uint4*>(p + 16 * q); w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w; }
    };
    uint32_t sb = 0, sg = 0, sr = 0;
    for (uint32_t ry = 0; ry < op.rh; ry++) {
        uint32_t ly[RW / 4], wb[(CW + 3) / 4], wc[(CW + 3) / 4];
        const uint8_t* b = PB + (size_t)ry * sc_;
        const uint8_t* c = PR + (size_t)ry * sc_;
        load_words(PY + (size_t)ry * sy_, RW, ly);
        load_words(b, CW, wb);
        load_words(c, CW, wc);
        // a chroma sample travels as {Cb, Cr} in the two halves of one register (byte i of the Cb word -> bits 0..7, of the Cr word -> bits 16..23)
        uint32_t C[CW + 2];
#pragma unroll
        for (int i = 0; i < CW; i++) C[i + 1] = __builtin_amdgcn_perm(wc[i >> 2], wb[i >> 2], 0x0c000c00u | ((4u + (i & 3)) << 16) | (uint32_t)(i & 3));
        if (HR == 2) {
            C[0] = has_l ? (uint32_t)b[-1] | ((uint32_t)c[-1] << 16) : C[1];
            C[CW + 1] = has_r ? (uint32_t)b[CW] | ((uint32_t)c[CW] << 16) : C[CW];
        }
#pragma unroll
        for (int x = 0; x < RW; x += 4) {
            u16x2 H[4];
            if (HR == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) H[j] = pk(C[x + j + 1]);
            } else {
                // h2v1_fancy_upsample: even columns lean on the left neighbour (+1), odd ones on the right (+2); at an image edge the
                // output is the sample itself, which is what the replicated neighbour gives: (3c + c + k) >> 2 == c
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int ci = x / 2 + k;
                    const u16x2 c3 = pk(C[ci + 1]) * (u16x2){3, 3};
                    H[2 * k] = (u16x2)((c3 + pk(C[ci]) + (u16x2){1, 1}) >> (u16x2){2, 2});
                    H[2 * k + 1] = (u16x2)((c3 + pk(C[ci + 2]) + (u16x2){2, 2}) >> (u16x2){2, 2});
                }
            }
            // jdcolor.c ycc_rgb_convert as in resample_420_body: a channel is Y + the upper half of its fixed-point chroma term
            uint32_t Tr[4], Tg[4], Tb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u16x2 hs = H[j] * (u16x2){2, 3};
                Tr[j] = __builtin_amdgcn_udot2(hs, (u16x2){0, 30627}, (uint32_t)KR, false);
                Tb[j] = __builtin_amdgcn_udot2(hs, (u16x2){58065, 0}, (uint32_t)KB, false);
                Tg[j] = __builtin_amdgcn_udot2(H[j], (u16x2){(unsigned short)FIX16(0.34414), (unsigned short)FIX16(0.71414)}, 65535u - (uint32_t)KG, false);
            }
            lp_ycc4_sum(ly[x >> 2], Tr, Tg, Tb, sr, sg, sb);
        }
    }
    uint8_t* D = reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off) + (size_t)dy * op.dst.stride + (size_t)dx * 3;
    const int32_t sums[3] = {(int32_t)sb, (int32_t)sg, (int32_t)sr};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint32_t r;
        if (op.round_2x2) r = (uint32_t)(sums[c] + 2) >> 2;
        else r = sat_round_u8(__fmul_rn((float)sums[c], op.inv_area));
        D[c] = (uint8_t)r;
    }
}

// dst(y, x) = src(f(y, x)); dst dims are (h, w) for orientations 5..8.
__global__ __launch_bounds__(256) void k_orient(const LpOrientOp* __restrict__ ops, const uint8_t* __restrict__ src_arena,
                                                uint8_t* __restrict__ dst_arena)
{
    const LpOrientOp& op = ops[blockIdx.z];
    const uint32_t o = op.orientation;
    const bool swap = o >= 5 && o <= 8;
    const uint32_t W = swap ? op.src.h : op.src.w, H = swap ? op.src.w : op.src.h;
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    uint32_t tx = x, ty = y;
    if (o == 2 || o == 6 || o == 3 || o == 7) tx = W - 1 - x;
    if (o == 3 || o == 7 || o == 4 || o == 8) ty = H - 1 - y;
    const uint32_t sx = swap ? ty : tx, sy = swap ? tx : ty;
    const uint8_t* s = src_arena + op.src.off + (size_t)sy * op.src.stride + (size_t)sx * op.src.cn;
    uint8_t* d = dst_arena + op.dst.off + (size_t)y * op.dst.stride + (size_t)x * op.src.cn;
    for (uint32_t c = 0; c < op.src.cn; c++) d[c] = s[c];
}


// INTER_AREA, integer scale (resizeAreaFast_): one wave per destination pixel, lanes stride the
// iscale_y x (iscale_x*cn) byte box, per-channel sums reduced across the wave.
__global__ __launch_bounds__(256) void k_resize_area_fast(const LpResizeOp* __restrict__ ops, const uint8_t* __restrict__ src_arena,
                                                          uint8_t* __restrict__ dst_arena)
{
    const LpResizeOp& op = ops[blockIdx.y];
    if (op.mode != 1) return;
    const uint32_t npx = op.dst.w * op.dst.h;
    const uint32_t pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pix >= npx) return;
    const uint32_t dx = pix % op.dst.w, dy = pix / op.dst.w, cn = op.src.cn;
    const uint32_t rowb = op.iscale_x * cn, box = rowb * op.iscale_y;
    const uint8_t* S = src_arena + op.src.off + (size_t)(dy * op.iscale_y) * op.src.stride + (size_t)dx * rowb;
    uint32_t acc[4] = {0, 0, 0, 0};
    for (uint32_t i = lane; i < box; i += 64) {
        uint32_t ry = i / rowb, rx = i - ry * rowb;
        uint32_t v = S[(size_t)ry * op.src.stride + rx];
        uint32_t ch = rx % cn;
        acc[0] += ch == 0 ? v : 0; acc[1] += ch == 1 ? v : 0; acc[2] += ch == 2 ? v : 0; acc[3] += ch == 3 ? v : 0;
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc[c] += __shfl_xor(acc[c], d, 64);
    if (lane < cn) {
        uint32_t sum = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
        uint32_t r;
        if (op.iscale_x == 2 && op.iscale_y == 2) r = (sum + 2) >> 2;          // ResizeAreaFastVec_SIMD_8u
        else r = sat_round_u8(__fmul_rn((float)sum, op.inv_area));              // saturate_cast<uchar>(sum * (1.f/area))
        dst_arena[op.dst.off + (size_t)dy * op.dst.stride + (size_t)dx * cn + lane] = (uint8_t)r;
    }
}

// INTER_AREA, fractional scale (ResizeArea_Invoker): thread per destination pixel, taps in OpenCV's order.
__global__ __launch_bounds__(256) void k_resize_area(const LpResizeOp* __restrict__ ops, const LpTap* __restrict__ taps,
                                                     const uint32_t* __restrict__ ranges, const uint8_t* __restrict__ src_arena,
                                                     uint8_t* __restrict__ dst_arena)
{
    const LpResizeOp& op = ops[blockIdx.z];
    if (op.mode != 2 || op.fast) return;
    const uint32_t dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= op.dst.w || dy >= op.dst.h) return;
    const uint32_t cn = op.src.cn;
    const uint32_t x0 = ranges[op.xrange_off + dx], x1 = ranges[op.xrange_off + dx + 1];
    const uint32_t y0 = ranges[op.yrange_off + dy], y1 = ranges[op.yrange_off + dy + 1];
    const LpTap* xt = taps + op.xtab_off;
    const LpTap* yt = taps + op.ytab_off;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t j = y0; j < y1; j++) {
        const float beta = yt[j].alpha;
        const uint8_t* S = src_arena + op.src.off + (size_t)yt[j].si * op.src.stride;
        float buf[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t k = x0; k < x1; k++) {
            const float a = xt[k].alpha;
            const uint8_t* p = S + (size_t)xt[k].si * cn;
            for (uint32_t c = 0; c < cn; c++) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)p[c], a));
        }
        for (uint32_t c = 0; c < cn; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(beta, buf[c]));
    }
    uint8_t* D = dst_arena + op.dst.off + (size_t)dy * op.dst.stride + (size_t)dx * cn;
    for (uint32_t c = 0; c < cn; c++) D[c] = (uint8_t)sat_round_u8(sum[c]);
}

// INTER_AREA, fractional scale, 3-channel fast path. Same arithmetic and tap order as k_resize_area (ResizeArea_Invoker), but a
// source row of the box arrives as aligned 16-byte loads (fixed up with v_alignbit for the byte phase of the box start)
// instead of one byte load per channel and tap. MAXT = most taps per axis (floor(scale) + 2); taps beyond a pixel's count
// carry weight 0 and add exactly 0. Requires the x taps of every destination column to address consecutive source
// columns (computeResizeAreaTab always does; the host checks).
template <int MAXT>
__global__ __launch_bounds__(256) void k_resize_area3(const LpResizeOp* __restrict__ ops, const LpTap* __restrict__ taps,
                                                      const uint32_t* __restrict__ ranges, const uint8_t* __restrict__ src_arena,
                                                      uint8_t* __restrict__ dst_arena)
{
    constexpr int NB = MAXT * 3;            // bytes of a box row
    constexpr int NW = (NB + 3 + 3) / 4;    // aligned dwords that cover them at any byte phase
    constexpr int NQ = (NW + 1 + 3) / 4;    // 16-byte loads (+1 dword for the alignbit of the last one)
    const LpResizeOp& op = ops[blockIdx.z];
    if (op.mode != 2 || op.fast != (uint32_t)MAXT) return;
    const uint32_t dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= op.dst.w || dy >= op.dst.h) return;
    const uint32_t x0 = ranges[op.xrange_off + dx], x1 = ranges[op.xrange_off + dx + 1];
    const uint32_t y0 = ranges[op.yrange_off + dy], y1 = ranges[op.yrange_off + dy + 1];
    const LpTap* xt = taps + op.xtab_off + x0;
    const LpTap* yt = taps + op.ytab_off;
    const uint32_t nx = x1 - x0;
    float al[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) al[k] = (uint32_t)k < nx ? xt[k].alpha : 0.f;
    const uint8_t* S0 = src_arena + op.src.off + (size_t)xt[0].si * 3;
    float sum[3] = {0.f, 0.f, 0.f};
    for (uint32_t j = y0; j < y1; j++) {
        const float beta = yt[j].alpha;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(S0 + (size_t)yt[j].si * op.src.stride);
        const uint4* q = reinterpret_cast<const uint4*>(addr & ~(uintptr_t)3); // 4-byte aligned is all a dwordx4 load needs
        const uint32_t sh = (uint32_t)(addr & 3) * 8;
        uint32_t w[NQ * 4];
#pragma unroll
        for (int i = 0; i < NQ; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
        uint32_t a[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) a[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
        float buf[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < MAXT; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int b = 3 * k + c;
                const float v = (float)((a[b >> 2] >> (8 * (b & 3))) & 255u);
                buf[c] = __fadd_rn(buf[c], __fmul_rn(v, al[k]));
            }
#pragma unroll
        for (int c = 0; c < 3; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(beta, buf[c]));
    }
    uint8_t* D = dst_arena + op.dst.off + (size_t)dy * op.dst.stride + (size_t)dx * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) D[c] = (uint8_t)sat_round_u8(sum[c]);
}

// Fractional INTER_AREA straight from YCbCr planes (no BGR frame in between): lp_area_core.h has the per-pixel walk -- the upsampler
// and the colour conversion of k_ycc_to_frame(_420) feeding the float sums of k_resize_area3 in the same order, so the bytes equal
// the frame route's. One thread per destination pixel; a wave is 64 neighbouring columns of one destination row, so its loads of a
// source row fall into one contiguous run of the luma row and of the chroma rows. Per destination pixel the frame route read and
// wrote 3 bytes per source pixel twice over; this reads 1.5 (4:2:0). SS: 2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4.
__device__ __forceinline__ bool area_planes(const LpJpeg& img, const uint8_t* __restrict__ plane_arena, int ss, LpAreaPlanes& P)
{
    if ((img.hs[0] == 2 ? (img.vs[0] == 2 ? 2 : 1) : 0) != ss) return false;
    P.py = plane_arena + img.plane_off[0]; P.pb = plane_arena + img.plane_off[1]; P.pr = plane_arena + img.plane_off[2];
    P.sy = img.plane_stride[0]; P.sc = img.plane_stride[1];
    P.dw = ss ? (int32_t)(img.width + 1) >> 1 : (int32_t)img.width;
    P.dh = ss == 2 ? (int32_t)(img.height + 1) >> 1 : (int32_t)img.height;
    return true;
}

template <int MAXT, int SS, bool FLIPX>
__global__ __launch_bounds__(256) void k_area_420(const LpJpeg* __restrict__ imgs, const LpArea420Op* __restrict__ ops, const LpTap* __restrict__ taps,
                                                  const uint32_t* __restrict__ ranges, const uint8_t* __restrict__ plane_arena)
{
    const LpArea420Op& op = ops[blockIdx.z];
    if (op.transposed || op.maxt != (uint32_t)MAXT || (op.xstep < 0) != FLIPX) return;
    const uint32_t dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= op.dst.w || dy >= op.dst.h) return;
    LpAreaPlanes P;
    if (!area_planes(imgs[op.img], plane_arena, SS, P)) return;
    const uint32_t x0 = ranges[op.xrange_off + dx], x1 = ranges[op.xrange_off + dx + 1];
    const uint32_t y0 = ranges[op.yrange_off + dy], y1 = ranges[op.yrange_off + dy + 1];
    const LpTap* xt = taps + op.xtab_off + x0;
    const uint32_t nx = x1 - x0;
    float al[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) al[k] = (uint32_t)k < nx ? xt[k].alpha : 0.f;
    const int32_t si0 = (int32_t)xt[0].si;
    const int32_t xa = FLIPX ? op.x0 - si0 - (MAXT - 1) : op.x0 + si0;
    uint8_t* D = reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off) + (size_t)dy * op.dst.stride + (size_t)dx * 3;
    lp_area420_pixel<MAXT, SS, FLIPX>(P, xa, al, taps + op.ytab_off, y0, y1, op.y0, op.ystep, D, op.post, op.half_up);
}

// The same for the orientations that swap the axes (5-8): the lanes of a wave are 64 neighbouring destination ROWS of one destination
// column -- neighbouring source columns, one shared run of source rows -- so the plane loads coalesce exactly as above; the three
// output bytes of a lane land a destination row apart (64 small writes per wave against ~20 k instructions of work).
template <int MAXT, int SS, bool FLIPC>
__global__ __launch_bounds__(256) void k_area_420t(const LpJpeg* __restrict__ imgs, const LpArea420Op* __restrict__ ops, const LpTap* __restrict__ taps,
                                                   const uint32_t* __restrict__ ranges, const uint8_t* __restrict__ plane_arena)
{
    const LpArea420Op& op = ops[blockIdx.z];
    if (!op.transposed || op.maxt != (uint32_t)MAXT || (op.xstep < 0) != FLIPC) return;
    const uint32_t dy = blockIdx.x * 64 + threadIdx.x, dx = blockIdx.y * 4 + threadIdx.y;
    if (dx >= op.dst.w || dy >= op.dst.h) return;
    LpAreaPlanes P;
    if (!area_planes(imgs[op.img], plane_arena, SS, P)) return;
    const uint32_t x0 = ranges[op.xrange_off + dx], x1 = ranges[op.xrange_off + dx + 1];
    const uint32_t y0 = ranges[op.yrange_off + dy], y1 = ranges[op.yrange_off + dy + 1];
    const LpTap* yt = taps + op.ytab_off + y0;
    const uint32_t ny = y1 - y0;
    float be[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) be[k] = (uint32_t)k < ny ? yt[k].alpha : 0.f;
    const int32_t si0 = (int32_t)yt[0].si;
    const int32_t xa = FLIPC ? op.x0 - si0 - (MAXT - 1) : op.x0 + si0;
    uint8_t* D = reinterpret_cast<uint8_t*>((uintptr_t)op.dst.off) + (size_t)dy * op.dst.stride + (size_t)dx * 3;
    lp_area420t_pixel<MAXT, SS, FLIPC>(P, xa, be, taps + op.xtab_off, x0, x1, op.y0, op.ystep, D, op.post, op.half_up);
}

// INTER_AREA with an up-scaling axis: bilinear, area-style coefficients, 11-bit fixed point
// (resizeGeneric_<HResizeLinear, VResizeLinear<uchar,int,short,...>>).
__global__ __launch_bounds__(256) void k_resize_linear(const LpResizeOp* __restrict__ ops, const int32_t* __restrict__ itab,
                                                       const uint8_t* __restrict__ src_arena, uint8_t* __restrict__ dst_arena)
{
    const LpResizeOp& op = ops[blockIdx.z];
    if (op.mode != 3) return;
    const uint32_t dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= op.dst.w || dy >= op.dst.h) return;
    const uint32_t cn = op.src.cn;
    const int32_t* xo = itab + op.xrange_off;   // per dx: sx, a0, a1
    const int32_t* yo = itab + op.yrange_off;   // per dy: sy, b0, b1
    const int32_t sx = xo[3 * dx], a0 = xo[3 * dx + 1], a1 = xo[3 * dx + 2];
    const int32_t sy = yo[3 * dy], b0 = yo[3 * dy + 1], b1 = yo[3 * dy + 2];
    const bool edge = dx >= op.xmax;
    uint8_t* D = dst_arena + op.dst.off + (size_t)dy * op.dst.stride + (size_t)dx * cn;
    int32_t y0 = sy < 0 ? 0 : sy > (int32_t)op.src.h - 1 ? (int32_t)op.src.h - 1 : sy;
    int32_t y1 = sy + 1 < 0 ? 0 : sy + 1 > (int32_t)op.src.h - 1 ? (int32_t)op.src.h - 1 : sy + 1;
    const uint8_t* S0 = src_arena + op.src.off + (size_t)y0 * op.src.stride + (size_t)sx * cn;
    const uint8_t* S1 = src_arena + op.src.off + (size_t)y1 * op.src.stride + (size_t)sx * cn;
    for (uint32_t c = 0; c < cn; c++) {
        int32_t r0 = edge ? S0[c] * 2048 : S0[c] * a0 + S0[c + cn] * a1;
        int32_t r1 = edge ? S1[c] * 2048 : S1[c] * a0 + S1[c + cn] * a1;
        D[c] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

__global__ __launch_bounds__(256) void k_copy_rect(const LpResizeOp* __restrict__ ops, const uint8_t* __restrict__ src_arena,
                                                   uint8_t* __restrict__ dst_arena)
{
    const LpResizeOp& op = ops[blockIdx.z];
    if (op.mode != 0) return;
    const uint32_t rowb = op.dst.w * op.src.cn;
    const uint32_t xb = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (xb >= rowb || y >= op.dst.h) return;
    const uint8_t* s = src_arena + op.src.off + (size_t)y * op.src.stride + xb;
    uint8_t* d = dst_arena + op.dst.off + (size_t)y * op.dst.stride + xb;
    for (uint32_t i = 0; i < 4 && xb + i < rowb; i++) d[i] = s[i];
}

// opencv_copy_to_region_with_alpha / opencv_copy_to_region / opencv_mat_clear_to_transparent on a ROI.
__global__ __launch_bounds__(256) void k_composite(LpCompositeOp op, const uint8_t* __restrict__ src_arena, uint8_t* __restrict__ dst_arena)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= op.w || y >= op.h) return;
    uint8_t* d = dst_arena + op.dst.off + (size_t)(op.y0 + y) * op.dst.stride + (size_t)(op.x0 + x) * op.dst.cn;
    const uint32_t dcn = op.dst.cn;
    if (op.kind == 2) { // clear
        for (uint32_t c = 0; c < dcn; c++) d[c] = 0;
        return;
    }
    const uint32_t scn = op.src.cn;
    const uint8_t* s = src_arena + op.src.off + (size_t)y * op.src.stride + (size_t)x * scn;
    uint32_t s4[4] = {s[0], scn == 1 ? s[0] : s[1], scn == 1 ? s[0] : s[2], scn == 4 ? s[3] : 255u};
    if (op.kind == 1) { // copy with channel fix-up
        for (uint32_t c = 0; c < dcn; c++) d[c] = (uint8_t)s4[c];
        return;
    }
    const float k = (float)(1.0 / 255.0);
    uint32_t d4[4] = {d[0], d[1], d[2], dcn == 4 ? d[3] : 255u};
    float sa = __fmul_rn((float)s4[3], k), da = __fmul_rn((float)d4[3], k);
    float om = __fsub_rn(1.0f, sa);
    float oa = __fadd_rn(sa, __fmul_rn(da, om));
    for (uint32_t c = 0; c < 3; c++) {
        float sc = __fmul_rn((float)s4[c], k), dc = __fmul_rn((float)d4[c], k);
        float t1 = __fmul_rn(sc, sa), t3 = __fmul_rn(__fmul_rn(dc, da), om);
        float bl = __fdiv_rn(__fadd_rn(t1, t3), oa);
        float v = __fmul_rn(bl, 255.0f);
        d[c] = (v != v) ? 0 : (uint8_t)sat_round_u8(v);
    }
    if (dcn == 4) d[3] = (uint8_t)sat_round_u8(__fmul_rn(oa, 255.0f));
}

// One GIF frame, one pass over the canvas: every pixel is independent once the four steps of
// giflib_decoder_render_frame (giflib.cpp:393-540) are taken in order per pixel -- (1) first frame: background;
// (2) previous frame disposed: its rectangle goes to the background colour or back to the snapshot;
// (3) the state before drawing becomes the new snapshot; (4) the frame's colour index, unless transparent or out of range.
__global__ __launch_bounds__(256) void k_gif_frame(LpGifFrameOp op)
{
    const int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y);
    if (x >= (int)op.canvas.w || y >= (int)op.canvas.h) return;
    uint32_t* cp = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(op.canvas.off) + (size_t)y * op.canvas.stride) + x;
    uint32_t* sp = reinterpret_cast<uint32_t*>(op.saved_off) + (size_t)y * op.canvas.w + x;
    const uint32_t bg = (uint32_t)op.bg[0] | ((uint32_t)op.bg[1] << 8) | ((uint32_t)op.bg[2] << 16) | ((uint32_t)op.bg[3] << 24);
    uint32_t v;
    if (op.first) v = bg;
    else {
        v = *cp;
        if (op.dispose && x >= op.px && x < op.px + op.pw && y >= op.py && y < op.py + op.ph) v = op.dispose == 1 ? bg : *sp;
        *sp = v;
    }
    if (x >= op.fx && x < op.fx + op.fw && y >= op.fy && y < op.fy + op.fh) {
        const size_t pi = (size_t)(op.skip_top + (y - op.fy)) * (size_t)op.raster_w + (size_t)(op.skip_left + (x - op.fx));
        const int idx = reinterpret_cast<const uint8_t*>(op.index_off)[pi];
        if (idx != op.transparent && idx < op.color_count) v = reinterpret_cast<const uint32_t*>(op.palette_off)[idx];
    }
    *cp = v;
}

// ------------------------------------------------------------------------------------------------
// PNG (n2): reverse the per-row filters of the inflated stream in place (pngrutil.c png_read_filter_row: None, Sub, Up,
// Average, Paeth on bytes `bpp` apart). A reconstructed byte needs its left neighbour (a: the byte bpp before it), the byte above (b)
// and the one above-left (c). Two things follow: the bpp byte positions of a pixel ("channels") never meet -- channel k of a pass is a
// filtering problem of its own, on the bytes k, k + bpp, k + 2 bpp ... of every row -- and inside a channel the finest schedule is ONE
// DIAGONAL over all rows: row r may run exactly one byte behind row r-1. Round 5 runs that schedule:
//   * one wave per (channel, band of 64 rows), lane = row; at step s lane r reconstructs byte s - r of its row and channel. "Above"
//     is what lane r-1 produced in the step before -- one DPP wave shift, no LDS, no memory -- and "above-left" is the "above" of the
//     step before. Every step is the same short instruction sequence for every lane (the load of the byte PNG_AHEAD steps ahead into a
//     register ring, ~20 VALU operations of filter arithmetic, one store), so nothing waits on memory inside the dependency chain.
//     The channels of a band run on bpp different CUs at once (they share cache lines but no byte).
//   * a pass of more than 64 rows is a pipeline of bands: band t + 1 starts when the last row of band t has its first bytes. That row
//     travels through a MAILBOX, one 32-bit word per byte: {tag = t + 1 : 24, the byte : 8}, written and read with relaxed agent-scope
//     atomics. Data and flag are the same word, so no fence, no L2 write-back and no ordering between two stores is involved. The
//     band below fetches the words sixteen at a time (lanes 0-15, PNG_AHEAD steps before they are due), checks the tags when lane 0
//     reaches the block and re-reads what is not there yet. One mailbox per pass and channel serves every band boundary: band t + 1
//     reads word p (at its step p) before its own last row overwrites it with tag t + 2 (at step p + 63).
//   * forward progress: a band waits for the band above only, and bands are numbered by a ticket in the order their workgroups START
//     (not by blockIdx): the workgroup that holds ticket t - 1 is running when ticket t is drawn, whatever else occupies the device.
//     A wait that does not end (it cannot, short of a lost device) gives up after PNG_SPIN_MAX polls and fails the image.
// The filter arithmetic per byte: three v_sad_u16, the Paeth choice as ONE v_min3 over keys (distance << 10 | priority << 8 | value),
// the None / Sub / Up / Average predictor from per-row weights. BASELINE configs[2]'s 800 x 297 RGB image is 15 waves and 800 + 297 + 4
// band hand-overs steps. Round 4's kernel walked the diagonal in 16-byte chunks (16 bytes of serial arithmetic per step, all channels
// in one lane, bands handing rows over through memory behind agent-scope fences): ~880 us for that image.
#define PNG_IT 16               // steps of one loop iteration: the unit in which a lane's input and the row above a band are fetched
#define PNG_SPIN_MAX (1u << 22)
// every pointer of the band loop is a GLOBAL pointer (address space 1), not a generic one: the accesses are then counted by vmcnt alone
// and the compiler can wait for a load of the ring without draining the queue (flat accesses force vmcnt(0) lgkmcnt(0))
#define PNG_G __attribute__((address_space(1)))

typedef uint32_t png_u32x4 __attribute__((ext_vector_type(4)));
typedef png_u32x4 png_u32x4_any __attribute__((aligned(1))); // 16 bytes at any address (rows start on odd offsets)

template <int BPP>
__device__ void png_unfilter_band(PNG_G uint8_t* base, const LpPngPass& ps, PNG_G uint32_t* error, PNG_G uint32_t* mail, PNG_G uint8_t* dump_base, uint32_t bi, uint32_t ch)
{
    const uint32_t lane = threadIdx.x;
    const size_t stride = (size_t)ps.row_bytes + 1;
    const int nunit = (int)(ps.row_bytes / BPP);
    const uint32_t band = bi * 64u;
    const uint32_t rows_here = ps.ph - band < 64u ? ps.ph - band : 64u;
    const bool live = lane < rows_here;
    PNG_G uint8_t* cur = base + ps.off + (size_t)(band + (live ? lane : 0u)) * stride;
    const uint32_t ft = live ? cur[0] : 0u;
    if (live && ft > 4 && ch == 0) __hip_atomic_fetch_or(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // "bad adaptive filter value" (the host has already refused such a file)
    cur += 1 + ch;
    PNG_G uint8_t* dump = dump_base + lane;
    const bool fed = bi > 0;                                     // wave-uniform: lane 0's "above" comes out of the mailbox
    const bool feeds = band + 64u < ps.ph && lane == 63u;        // this lane's results go into it
    const uint32_t tag_out = (bi + 1u) << 8;
    const bool paeth = ft >= 4;
    const uint32_t wa = (0xAu >> ft) & 1u, wb = (0xCu >> ft) & 1u, sh = ft == 3 ? 1u : 0u; // None 0, Sub a, Up b, Average (a + b) >> 1
    uint32_t xa = 0, xc = 0, res = 0;
    // A lane's input arrives PNG_IT steps at a time: the PNG_IT * BPP bytes behind (row, byte s0 - lane) -- this channel's byte of the
    // next PNG_IT steps at every BPP-th position -- as BPP 16-byte loads, asked for one iteration before they are used. (64 lanes on 64
    // rows are 64 cache lines an access whatever its width: a byte a step costs the memory pipeline as much as 16 bytes every 16 steps.)
    // Lanes that are before or behind their row read what lies there -- other rows, or the margins LpEngine::png_decode keeps around
    // the stream (LP_PNG_MARGIN) -- and never use it.
    //
    // These loads and the one of the mailbox block are inline assembly with a hand-placed wait. The compiler's own bookkeeping counts
    // loads across the loop's back edge but not stores, so for a register loaded one iteration ago it waits until all but the last
    // few accesses have completed -- that is: for the byte stores of the last steps, ~1 us every iteration (SQ_WAIT_ANY was 53 % of the
    // waves' cycles). What is certain here: after these loads every step issues its store (PNG_IT of them, the dump-slot trick below
    // makes them unconditional), accesses complete in order, so "at most PNG_IT outstanding" means the loads have landed.
    png_u32x4 xn[BPP];
    uint32_t xw[4 * BPP];
    uint32_t mcur = 0, mnext = 0;
    auto load_window = [&](int s0) {
        const PNG_G uint8_t* w = cur + (ptrdiff_t)(s0 - (int)lane) * BPP;
#pragma unroll
        for (int i = 0; i < BPP; i++) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(xn[i]) : "v"(w), "n"(16 * i) : "memory");
        // the row above the band, PNG_IT bytes at a time: lanes 0 .. PNG_IT-1 hold the words of a block, lane 0 takes its byte out of
        // them step by step (v_readlane)
        const int u = s0 + (int)lane;
        if (fed && lane < PNG_IT && u < nunit) asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(mnext) : "v"(&mail[(size_t)u]) : "memory");
    };
    auto loads_landed = [&](auto outstanding) { // outstanding: an integral constant
#pragma unroll
        for (int i = 0; i < BPP; i++) asm volatile("" : "+v"(xn[i]));
        if (BPP == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xn[0]), "+v"(mnext) : "n"(decltype(outstanding)::value) : "memory");
        else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(xn[0]), "+v"(xn[BPP - 1]), "+v"(mnext) : "n"(decltype(outstanding)::value) : "memory");
#pragma unroll
        for (int i = 0; i < BPP; i++) asm volatile("" : "+v"(xn[i]));
    };
    load_window(0);
    loads_landed(std::integral_constant<int, 0>());
    const int nsteps = nunit + (int)rows_here - 1;
    for (int s0 = 0; s0 < nsteps; s0 += PNG_IT) {
        if (s0) loads_landed(std::integral_constant<int, PNG_IT>());
#pragma unroll
        for (int i = 0; i < BPP; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) xw[4 * i + j] = xn[i][j];
        }
        mcur = mnext;
        load_window(s0 + PNG_IT);
        if (fed) { // lane 0 enters a new block of the row above: every word of it must carry the tag of the band above, (bi - 1) + 1
            const int u = s0 + (int)lane;
            bool gave_up = false;
            for (uint32_t spin = 0;; spin++) {
                const bool late = (mcur >> 8) != bi && lane < PNG_IT && u < nunit;
                if (!__builtin_amdgcn_ballot_w64(late)) break;
                if (spin > PNG_SPIN_MAX) { gave_up = true; break; } // never seen; see the header comment
                __builtin_amdgcn_s_sleep(4);
                if (late) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(mcur) : "v"(&mail[(size_t)u]) : "memory");
            }
            if (gave_up && lane == 0) __hip_atomic_fetch_or(error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < PNG_IT; k++) {
            const int p = s0 + k - (int)lane;
            const bool on = live && p >= 0 && p < nunit;
            // what the row above produced one step ago: the lane above, or (lane 0 of a band below another) the mailbox
            const uint32_t top = fed ? (uint32_t)__builtin_amdgcn_readlane((int)mcur, k) & 255u : 0u;
            const uint32_t xb = (uint32_t)__builtin_amdgcn_update_dpp((int)top, (int)res, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const uint32_t x = (xw[(k * BPP) >> 2] >> (((k * BPP) & 3) * 8)) & 255u;
            // The arithmetic runs on every lane in every step. A lane that has not reached its row yet must keep a = c = result = 0: its
            // result is masked (and with it everything derived from it); a lane past its row computes what nobody reads. The store of a
            // lane outside its row goes to a dump slot instead of sitting in a branch
            const uint32_t pa = __builtin_amdgcn_sad_u16(xb, xc, 0u), pb = __builtin_amdgcn_sad_u16(xa, xc, 0u), pc = __builtin_amdgcn_sad_u16(xa + xb, xc << 1, 0u);
            // a unless b or c is strictly nearer, b unless c is strictly nearer: the smallest of (distance, priority) keys
            const uint32_t ka = (pa << 10) | xa, kb = (pb << 10) | 256u | xb, kc = (pc << 10) | 512u | xc;
            const uint32_t best = min(ka, min(kb, kc));
            const uint32_t lin = (__umul24(xa, wa) + __umul24(xb, wb)) >> sh;
            const uint32_t r = (x + (paeth ? best : lin)) & (p >= 0 ? 255u : 0u); // the key's upper bits fall to the mask
            *(on ? cur + (size_t)p * BPP : dump) = (uint8_t)r;
            xa = r;
            xc = xb;
            res = r;
            if (feeds && on) __hip_atomic_store(&mail[(size_t)p], tag_out | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(64) void k_png_unfilter(LpPngOp op)
{
    // workgroup -> pass (a pass owns bands x channels workgroups), channel, then a ticket -> band
    uint32_t wg = blockIdx.x, pass = 0;
    size_t mail_words = 0;
    for (; pass < op.npass; pass++) {
        const LpPngPass& q = op.pass[pass];
        const uint32_t nb = lp_png_bands(q) * op.bpp;
        if (wg < nb) break;
        wg -= nb;
        mail_words += lp_png_bands(q) ? q.row_bytes : 0u;
    }
    if (pass >= op.npass) return;
    const LpPngPass& ps = op.pass[pass];
    const uint32_t ch = wg % op.bpp; // neighbouring workgroups take different channels of the same bands
    PNG_G uint32_t* tickets = (PNG_G uint32_t*)(uintptr_t)op.sync_off;
    uint32_t bi = 0;
    if (threadIdx.x == 0) bi = __hip_atomic_fetch_add(&tickets[pass * 8u + ch], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bi = (uint32_t)__builtin_amdgcn_readfirstlane((int)bi);
    PNG_G uint8_t* base = (PNG_G uint8_t*)(uintptr_t)op.data_off;
    PNG_G uint32_t* err = (PNG_G uint32_t*)(uintptr_t)op.error_off;
    PNG_G uint8_t* dump = (PNG_G uint8_t*)(uintptr_t)(op.sync_off + LP_PNG_TICKET_BYTES);
    // the mailbox of (pass, channel): one word per byte of that channel in a row
    PNG_G uint32_t* mail = (PNG_G uint32_t*)(uintptr_t)(op.sync_off + LP_PNG_TICKET_BYTES + LP_PNG_DUMP_BYTES) + mail_words + (size_t)ch * (ps.row_bytes / op.bpp);
    switch (op.bpp) {
    case 1: png_unfilter_band<1>(base, ps, err, mail, dump, bi, ch); break;
    case 2: png_unfilter_band<2>(base, ps, err, mail, dump, bi, ch); break;
    case 3: png_unfilter_band<3>(base, ps, err, mail, dump, bi, ch); break;
    case 4: png_unfilter_band<4>(base, ps, err, mail, dump, bi, ch); break;
    case 6: png_unfilter_band<6>(base, ps, err, mail, dump, bi, ch); break;
    default: png_unfilter_band<8>(base, ps, err, mail, dump, bi, ch); break;
    }
}

// Reconstructed samples -> the pixels cv::PngDecoder::readData asks libpng for: 16-bit samples keep their high byte
// (png_set_strip_16), grey below 8 bits is scaled by bit replication (expand_gray_1_2_4_to_8), palette indices go through
// PLTE / tRNS, an RGB colour key becomes alpha (tRNS_to_alpha, compared at full precision before the strip), colour is
// stored B, G, R (png_set_bgr), grey + alpha is spread to B = G = R (gray_to_rgb); Adam7 passes land on their lattice.
__global__ __launch_bounds__(256) void k_png_convert(LpPngOp op)
{
    const LpPngPass& ps = op.pass[blockIdx.z];
    const uint32_t px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= ps.pw || py >= ps.ph) return;
    const uint8_t* row = reinterpret_cast<const uint8_t*>(op.data_off) + ps.off + (size_t)py * ((size_t)ps.row_bytes + 1) + 1;
    const uint32_t x = ps.x0 + px * ps.dx, y = ps.y0 + py * ps.dy;
    uint8_t* o = reinterpret_cast<uint8_t*>(op.dst.off) + (size_t)y * op.dst.stride + (size_t)x * op.dst.cn;
    const uint32_t d = op.depth, ct = op.color_type, cn = op.dst.cn;
    if (ct == 0 || ct == 3) {
        uint32_t v;
        if (d == 16) v = row[(size_t)px * 2];
        else if (d == 8) v = row[px];
        else {
            const uint32_t per = 8 / d, byte = row[px / per], sh = 8 - d * (1 + px % per);
            v = (byte >> sh) & ((1u << d) - 1);
        }
        if (ct == 0) {
            if (d < 8) v *= d == 1 ? 255u : d == 2 ? 85u : 17u;
            o[0] = (uint8_t)v; // a grey colour key is dropped together with the alpha channel (png_set_strip_alpha)
        } else {
            const uint8_t* e = reinterpret_cast<const uint8_t*>(op.palette_off) + (size_t)v * 4;
            o[0] = e[0]; o[1] = e[1]; o[2] = e[2];
            if (cn == 4) o[3] = e[3];
        }
        return;
    }
    const uint32_t bytes = d == 16 ? 2 : 1, nch = ct == 2 ? 3 : ct == 4 ? 2 : 4;
    const uint8_t* p = row + (size_t)px * nch * bytes;
    if (ct == 4) { // grey + alpha -> B = G = R = grey, A
        const uint8_t g = p[0], a = p[bytes];
        o[0] = g; o[1] = g; o[2] = g; o[3] = a;
        return;
    }
    const uint8_t r = p[0], g = p[bytes], b = p[2 * bytes];
    o[0] = b; o[1] = g; o[2] = r;
    if (cn == 4) {
        if (ct == 6) o[3] = p[3 * bytes];
        else {
            bool hit = false;
            if (op.has_key) {
                if (d == 16) hit = (((uint32_t)p[0] << 8) | p[1]) == op.key[0] && (((uint32_t)p[2] << 8) | p[3]) == op.key[1] && (((uint32_t)p[4] << 8) | p[5]) == op.key[2];
                else hit = r == (op.key[0] & 255u) && g == (op.key[1] & 255u) && b == (op.key[2] & 255u);
            }
            o[3] = hit ? 0 : 255;
        }
    }
}

void lp_launch_png(hipStream_t s, const LpPngOp& op)
{
    if (!op.npass) return;
    uint32_t bands = 0;
    for (uint32_t p = 0; p < op.npass; p++) bands += lp_png_bands(op.pass[p]) * op.bpp;
    if (bands) hipLaunchKernelGGL(k_png_unfilter, dim3(bands), dim3(64), 0, s, op);
    uint32_t mw = 0, mh = 0;
    for (uint32_t p = 0; p < op.npass; p++) { mw = op.pass[p].pw > mw ? op.pass[p].pw : mw; mh = op.pass[p].ph > mh ? op.pass[p].ph : mh; }
    if (!mw || !mh) return;
    hipLaunchKernelGGL(k_png_convert, dim3((mw + 63) / 64, (mh + 3) / 4, op.npass), dim3(64, 4), 0, s, op);
}

// ---- PNG output: filter selection + filtering (see LpPngEncOp)
__device__ __forceinline__ uint32_t png_enc_raw(const LpFrame& f, const uint8_t* row, uint32_t i)
{
    // byte i of the RGB(A) / grey row libpng sees after png_set_bgr: channel c of BGR(A) pixel x sits at 2 - c for the colours
    const uint32_t cn = f.cn, x = i / cn, c = i - x * cn;
    return row[x * cn + (cn >= 3 && c < 3 ? 2u - c : c)];
}
__device__ __forceinline__ uint32_t png_enc_filtered(uint32_t type, uint32_t x, uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t pred = 0;
    if (type == 1) pred = a;
    else if (type == 2) pred = b;
    else if (type == 3) pred = (a + b) >> 1;
    else if (type == 4) {
        const int32_t p = (int32_t)a + (int32_t)b - (int32_t)c;
        const int32_t pa = abs(p - (int32_t)a), pb = abs(p - (int32_t)b), pc = abs(p - (int32_t)c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    }
    return (x - pred) & 0xffu;
}
__global__ __launch_bounds__(256) void k_png_filter(LpPngEncOp op)
{
    __shared__ uint32_t s_sum[5][4];
    const LpFrame& f = op.src;
    const uint32_t y = blockIdx.x, cn = f.cn, rb = f.w * cn;
    const uint8_t* row = reinterpret_cast<const uint8_t*>(f.off) + (size_t)y * f.stride;
    const uint8_t* up = y ? row - f.stride : nullptr; // libpng's prev_row is all zero above the first row
    uint32_t sum[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < rb; i += 256) {
        const uint32_t x = png_enc_raw(f, row, i), a = i >= cn ? png_enc_raw(f, row, i - cn) : 0u, b = up ? png_enc_raw(f, up, i) : 0u,
                       c = (up && i >= cn) ? png_enc_raw(f, up, i - cn) : 0u;
#pragma unroll
        for (uint32_t t = 0; t < 5; t++) {
            const uint32_t v = png_enc_filtered(t, x, a, b, c);
            sum[t] += v < 128u ? v : 256u - v;
        }
    }
#pragma unroll
    for (uint32_t t = 0; t < 5; t++) {
        uint32_t v = sum[t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) s_sum[t][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    uint32_t best = 0, best_sum = 0xffffffffu;
    bool have = false;
    for (uint32_t t = 0; t < 5; t++) { // the first enabled filter with the strictly smallest sum
        if (!(op.filters & (1u << t))) continue;
        const uint32_t v = s_sum[t][0] + s_sum[t][1] + s_sum[t][2] + s_sum[t][3];
        if (!have || v < best_sum) { best = t; best_sum = v; have = true; }
    }
    uint8_t* out = reinterpret_cast<uint8_t*>(op.out_off) + (size_t)y * (rb + 1);
    if (threadIdx.x == 0) out[0] = (uint8_t)best;
    for (uint32_t i = threadIdx.x; i < rb; i += 256) {
        const uint32_t x = png_enc_raw(f, row, i), a = i >= cn ? png_enc_raw(f, row, i - cn) : 0u, b = up ? png_enc_raw(f, up, i) : 0u,
                       c = (up && i >= cn) ? png_enc_raw(f, up, i - cn) : 0u;
        out[1 + i] = (uint8_t)png_enc_filtered(best, x, a, b, c);
    }
}
void lp_launch_png_filter(hipStream_t s, const LpPngEncOp& op)
{
    if (!op.src.h || !op.src.w) return;
    hipLaunchKernelGGL(k_png_filter, dim3(op.src.h), dim3(256), 0, s, op);
}

// ThumbHash's nearest-neighbour samples (thumbhash.cpp:118-193): out[(i * w + j) * cn ..] = frame(rows[i], cols[j]).
__global__ __launch_bounds__(256) void k_gather_samples(LpFrame f, const uint32_t* __restrict__ idx, uint32_t w, uint32_t h, uint8_t* __restrict__ out)
{
    const uint32_t j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= w || i >= h) return;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(f.off) + (size_t)idx[w + i] * f.stride + (size_t)idx[j] * f.cn;
    uint8_t* o = out + ((size_t)i * w + j) * f.cn;
    for (uint32_t c = 0; c < f.cn; c++) o[c] = p[c];
}

void lp_launch_gather_samples(hipStream_t s, const LpFrame& f, const uint32_t* d_idx, uint32_t w, uint32_t h, uint8_t* d_out)
{
    hipLaunchKernelGGL(k_gather_samples, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, f, d_idx, w, h, d_out);
}

// ---- lossy WebP output: the encoder's BGR -> Y'CbCr 4:2:0 front end (libwebp picture_csp_enc.c ImportYUVAFromRGBA for opaque pictures,
// what WebPEncodeBGR / WebPPictureImportBGR + WebPEncode run first; reference call sites webp.cpp:707-751). Luma per pixel in 16-bit
// fixed point; chroma from the 2 x 2 mean taken in LINEAR light -- gamma 0.80 through a 256-entry table, back through a 33-entry
// table with linear interpolation -- with the last column / row standing in for the missing one of an odd size. One thread per 2 x 2
// block. The two tables are computed by the host with the library's own formula (lp_abi_webp.cpp) and passed in.
__global__ __launch_bounds__(256) void k_webp_yuv420(LpFrame f, const LpWebpYuvTab* __restrict__ tab, uint8_t* __restrict__ Y, uint8_t* __restrict__ U,
                                                     uint8_t* __restrict__ V, uint32_t* __restrict__ non_opaque)
{
    __shared__ uint16_t s_g2l[256];
    __shared__ int32_t s_l2g[33];
    const uint32_t t = threadIdx.y * 64 + threadIdx.x;
    s_g2l[t] = tab->gamma_to_linear[t];
    if (t < 33) s_l2g[t] = tab->linear_to_gamma[t];
    __syncthreads();
    const uint32_t bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
    const uint32_t uvw = (f.w + 1) >> 1, uvh = (f.h + 1) >> 1;
    if (bx >= uvw || by >= uvh) return;
    uint32_t lin[3] = {0, 0, 0};
    bool translucent = false;
    for (uint32_t dy = 0; dy < 2; dy++)
        for (uint32_t dx = 0; dx < 2; dx++) {
            const uint32_t x = 2 * bx + dx, y = 2 * by + dy;
            const uint32_t xc = x < f.w ? x : f.w - 1, yc = y < f.h ? y : f.h - 1;
            const uint8_t* p = reinterpret_cast<const uint8_t*>(f.off) + (size_t)yc * f.stride + (size_t)xc * f.cn;
            const uint32_t b = p[0], g = p[1], r = p[2];
            if (f.cn == 4 && p[3] != 255) translucent = true;
            if (x < f.w && y < f.h) Y[(size_t)y * f.w + x] = (uint8_t)((16839u * r + 33059u * g + 6420u * b + (1u << 15) + (16u << 16)) >> 16); // VP8RGBToY
            lin[0] += s_g2l[r]; lin[1] += s_g2l[g]; lin[2] += s_g2l[b];
        }
    int32_t c[3]; // LinearToGamma(sum, 0): the 2 x 2 sum back in gamma space, scaled by 4
    for (int k = 0; k < 3; k++) {
        const uint32_t v = lin[k], pos = v >> 9, x = v & 511u;
        c[k] = (s_l2g[pos + 1] * (int32_t)x + s_l2g[pos] * (int32_t)(512u - x) + 64) >> 7;
    }
    auto clip = [](int32_t uv) { uv = (uv + (1 << 17) + (128 << 18)) >> 18; return (uint8_t)(uv < 0 ? 0 : uv > 255 ? 255 : uv); }; // VP8ClipUV, YUV_HALF << 2
    U[(size_t)by * uvw + bx] = clip(-9719 * c[0] - 19081 * c[1] + 28800 * c[2]);
    V[(size_t)by * uvw + bx] = clip(28800 * c[0] - 24116 * c[1] - 4684 * c[2]);
    if (translucent) atomicOr(non_opaque, 1u);
}

void lp_launch_webp_yuv420(hipStream_t s, const LpFrame& f, const LpWebpYuvTab* d_tab, uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, uint32_t* d_flag)
{
    hipLaunchKernelGGL(k_webp_yuv420, dim3(((f.w + 1) / 2 + 63) / 64, ((f.h + 1) / 2 + 3) / 4), dim3(64, 4), 0, s, f, d_tab, d_y, d_u, d_v, d_flag);
}

// ---- GIF encoder: palette mapping (see LpGifEncOp)
__device__ __forceinline__ int gif_dist(int r0, int g0, int b0, int r1, int g1, int b1) { return abs(r0 - r1) + abs(g0 - g1) + abs(b0 - b1); } // giflib.cpp:921-929

__global__ __launch_bounds__(256) void k_gifenc_first(LpGifEncOp op)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= op.frame.w || y >= op.frame.h) return;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(op.frame.off) + (size_t)y * op.frame.stride + (size_t)x * 4;
    const uint32_t B = p[0], G = p[1], R = p[2], A = p[3];
    if (A < 128 && op.transparent >= 0) return; // never reaches the cache
    const uint32_t crushed = ((R >> 3) << 10) | ((G >> 3) << 5) | (B >> 3);
    if (reinterpret_cast<const uint16_t*>(op.lookup_off)[crushed] != 0xffffu) return;
    atomicMin(reinterpret_cast<uint32_t*>(op.first_off) + crushed, y * op.frame.w + x);
}

__global__ __launch_bounds__(256) void k_gifenc_fill(LpGifEncOp op)
{
    const uint32_t bucket = blockIdx.x * 256 + threadIdx.x;
    const uint32_t first = reinterpret_cast<const uint32_t*>(op.first_off)[bucket];
    if (first == 0xffffffffu) return;
    const uint32_t x = first % op.frame.w, y = first / op.frame.w;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(op.frame.off) + (size_t)y * op.frame.stride + (size_t)x * 4;
    const int B = p[0], G = p[1], R = p[2];
    const bool extreme = (R > 240 && G > 240 && B > 240) || (R < 15 && G < 15 && B < 15);
    const int rc = extreme ? R : (R & 0xf8) | 4, gc = extreme ? G : (G & 0xf8) | 4, bc = extreme ? B : (B & 0xf8) | 4;
    const uint8_t* pal = reinterpret_cast<const uint8_t*>(op.palette_off);
    int least = 0x7fffffff, best = 0;
    for (int i = 0; i < op.color_count; i++) {
        if (i == op.transparent) continue;
        const int d = gif_dist(rc, gc, bc, pal[4 * i], pal[4 * i + 1], pal[4 * i + 2]);
        if (d < least) { least = d; best = i; }
    }
    reinterpret_cast<uint16_t*>(op.lookup_off)[bucket] = (uint16_t)best;
    reinterpret_cast<uint32_t*>(op.fresh_off)[bucket] = (uint32_t)least;
}

__global__ __launch_bounds__(256) void k_gifenc_map(LpGifEncOp op)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= op.frame.w || y >= op.frame.h) return;
    const uint32_t idx = y * op.frame.w + x;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(op.frame.off) + (size_t)y * op.frame.stride + (size_t)x * 4;
    const int B = p[0], G = p[1], R = p[2], A = p[3];
    uint8_t* out = reinterpret_cast<uint8_t*>(op.out_off) + idx;
    if (A < 128 && op.transparent >= 0) { *out = (uint8_t)op.transparent; return; }
    const uint32_t crushed = (((uint32_t)R >> 3) << 10) | (((uint32_t)G >> 3) << 5) | ((uint32_t)B >> 3);
    int best = reinterpret_cast<const uint16_t*>(op.lookup_off)[crushed];
    const uint8_t* pal = reinterpret_cast<const uint8_t*>(op.palette_off);
    // the pixel that filled the bucket keeps the distance of its search; everybody else measures against the cached entry
    int least = reinterpret_cast<const uint32_t*>(op.first_off)[crushed] == idx ? (int)reinterpret_cast<const uint32_t*>(op.fresh_off)[crushed]
                                                                                 : gif_dist(R, G, B, pal[4 * best], pal[4 * best + 1], pal[4 * best + 2]);
    if (op.use_prev && op.transparent >= 0) {
        const uint8_t* q = reinterpret_cast<const uint8_t*>(op.prev_off) + (size_t)idx * 4;
        if (gif_dist(R, G, B, q[2], q[1], q[0]) < least) best = op.transparent;
    }
    *out = (uint8_t)best;
}

void lp_launch_gifenc(hipStream_t s, const LpGifEncOp& op)
{
    if (!op.frame.w || !op.frame.h) return;
    dim3 g((op.frame.w + 63) / 64, (op.frame.h + 3) / 4, 1);
    hipLaunchKernelGGL(k_gifenc_first, g, dim3(64, 4), 0, s, op);
    hipLaunchKernelGGL(k_gifenc_fill, dim3(128), dim3(256), 0, s, op);
    hipLaunchKernelGGL(k_gifenc_map, g, dim3(64, 4), 0, s, op);
}

void lp_launch_gif_frame(hipStream_t s, const LpGifFrameOp& op)
{
    if (!op.canvas.w || !op.canvas.h) return;
    dim3 g((op.canvas.w + 63) / 64, (op.canvas.h + 3) / 4, 1);
    hipLaunchKernelGGL(k_gif_frame, g, dim3(64, 4), 0, s, op);
}

// ------------------------------------------------------------------------------------------------
void lp_launch_ycc_to_frame(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_w, uint32_t max_h, bool any_generic, bool any_420,
                            const uint8_t* d_planes, const LpFrame* d_dsts, uint8_t* d_frames)
{
    if (!nimg || !max_w || !max_h) return;
    dim3 g((max_w + 255) / 256, (max_h + 3) / 4, nimg);
    if (any_generic) hipLaunchKernelGGL(k_ycc_to_frame, g, dim3(64, 4), 0, s, d_imgs, d_planes, d_dsts, d_frames);
    dim3 g2((max_w + 511) / 512, ((max_h + 1) / 2 + 3) / 4, nimg); // thread = 8 x 2 pixels
    if (any_420) hipLaunchKernelGGL(k_ycc_to_frame_420, g2, dim3(256), 0, s, d_imgs, d_planes, d_dsts);
}

void lp_launch_resample_fused(hipStream_t s, const LpJpeg* d_imgs, const LpFusedOp* d_ops, uint32_t nops, uint32_t max_px, bool general,
                              uint32_t fast_mask, uint32_t fast_grid, const uint8_t* d_planes)
{
    if (!nops) return;
    dim3 g((max_px + 3) / 4, nops);
    if (general && max_px) hipLaunchKernelGGL(k_resample_fused, g, dim3(256), 0, s, d_imgs, d_ops, d_planes);
    // fast_grid = max over the fast ops of V * ceil(U / 256)
    if (fast_mask & 1u) hipLaunchKernelGGL(k_resample_420<4>, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 2u) hipLaunchKernelGGL(k_resample_420<8>, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 4u) hipLaunchKernelGGL(k_resample_420<16>, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x4000u) hipLaunchKernelGGL(k_resample_gray, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x1000u) hipLaunchKernelGGL(k_resample_420_small<4>, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes); // 2-pixel boxes
    if (fast_mask & 0x2000u) hipLaunchKernelGGL(k_resample_420_small<2>, dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes); // 4-pixel boxes
    if (fast_mask & 0x10u) hipLaunchKernelGGL((k_resample_hv1<8, 1>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x20u) hipLaunchKernelGGL((k_resample_hv1<16, 1>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x40u) hipLaunchKernelGGL((k_resample_hv1<32, 1>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x100u) hipLaunchKernelGGL((k_resample_hv1<8, 2>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x200u) hipLaunchKernelGGL((k_resample_hv1<16, 2>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
    if (fast_mask & 0x400u) hipLaunchKernelGGL((k_resample_hv1<32, 2>), dim3(fast_grid, nops), dim3(256), 0, s, d_imgs, d_ops, d_planes);
}

void lp_launch_orient(hipStream_t s, const LpOrientOp* d_ops, uint32_t nimg, uint32_t max_w, uint32_t max_h, const uint8_t* d_src, uint8_t* d_dst)
{
    if (!nimg || !max_w || !max_h) return;
    dim3 g((max_w + 63) / 64, (max_h + 3) / 4, nimg);
    hipLaunchKernelGGL(k_orient, g, dim3(64, 4), 0, s, d_ops, d_src, d_dst);
}

void lp_launch_resize(hipStream_t s, const LpResizeOp* d_ops, uint32_t nimg, uint32_t modes_present, uint32_t area3_mask, uint32_t max_dw, uint32_t max_dh,
                      const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_src, uint8_t* d_dst)
{
    if (!nimg || !max_dw || !max_dh) return;
    dim3 g2((max_dw + 63) / 64, (max_dh + 3) / 4, nimg);
    if (modes_present & 1u) {
        dim3 gc((max_dw * 4 + 255) / 256, (max_dh + 3) / 4, nimg);
        hipLaunchKernelGGL(k_copy_rect, gc, dim3(64, 4), 0, s, d_ops, d_src, d_dst);
    }
    if (modes_present & 2u) {
        dim3 gf((max_dw * max_dh + 3) / 4, nimg);
        hipLaunchKernelGGL(k_resize_area_fast, gf, dim3(256), 0, s, d_ops, d_src, d_dst);
    }
    if (modes_present & 4u) hipLaunchKernelGGL(k_resize_area, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (area3_mask & 1u) hipLaunchKernelGGL(k_resize_area3<6>, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (area3_mask & 2u) hipLaunchKernelGGL(k_resize_area3<10>, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (area3_mask & 4u) hipLaunchKernelGGL(k_resize_area3<18>, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (area3_mask & 8u) hipLaunchKernelGGL(k_resize_area3<34>, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (area3_mask & 16u) hipLaunchKernelGGL(k_resize_area3<66>, g2, dim3(64, 4), 0, s, d_ops, d_taps, d_ranges, d_src, d_dst);
    if (modes_present & 8u)
        hipLaunchKernelGGL(k_resize_linear, g2, dim3(64, 4), 0, s, d_ops, reinterpret_cast<const int32_t*>(d_ranges), d_src, d_dst);
}

// mask[ss] (ss = 0 4:4:4, 1 4:2:2, 2 4:2:0): bit b = MAXT bucket b of {6, 10, 18, 34, 66} present, +5 for the mirrored (xstep < 0)
// instantiation; bits 10-17 the same for the axis-swapping orientations (k_area_420t; buckets up to 34: a window column there costs
// three accumulators)
template <int SS>
static void launch_area_ss(hipStream_t s, const LpJpeg* d_imgs, const LpArea420Op* d_ops, uint32_t nops, uint32_t mask, uint32_t max_dw, uint32_t max_dh,
                           const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_planes)
{
    dim3 g((max_dw + 63) / 64, (max_dh + 3) / 4, nops);
#define LP_AREA_LAUNCH(bit, T, F) if (mask & (1u << (bit))) hipLaunchKernelGGL((k_area_420<T, SS, F>), g, dim3(64, 4), 0, s, d_imgs, d_ops, d_taps, d_ranges, d_planes)
    LP_AREA_LAUNCH(0, 6, false); LP_AREA_LAUNCH(1, 10, false); LP_AREA_LAUNCH(2, 18, false); LP_AREA_LAUNCH(3, 34, false); LP_AREA_LAUNCH(4, 66, false);
    LP_AREA_LAUNCH(5, 6, true); LP_AREA_LAUNCH(6, 10, true); LP_AREA_LAUNCH(7, 18, true); LP_AREA_LAUNCH(8, 34, true); LP_AREA_LAUNCH(9, 66, true);
#undef LP_AREA_LAUNCH
    dim3 gt((max_dh + 63) / 64, (max_dw + 3) / 4, nops);
#define LP_AREA_LAUNCH(bit, T, F) if (mask & (1u << (bit))) hipLaunchKernelGGL((k_area_420t<T, SS, F>), gt, dim3(64, 4), 0, s, d_imgs, d_ops, d_taps, d_ranges, d_planes)
    LP_AREA_LAUNCH(10, 6, false); LP_AREA_LAUNCH(11, 10, false); LP_AREA_LAUNCH(12, 18, false); LP_AREA_LAUNCH(13, 34, false);
    LP_AREA_LAUNCH(14, 6, true); LP_AREA_LAUNCH(15, 10, true); LP_AREA_LAUNCH(16, 18, true); LP_AREA_LAUNCH(17, 34, true);
#undef LP_AREA_LAUNCH
}

void lp_launch_area_420(hipStream_t s, const LpJpeg* d_imgs, const LpArea420Op* d_ops, uint32_t nops, const uint32_t mask[3], uint32_t max_dw, uint32_t max_dh,
                        const LpTap* d_taps, const uint32_t* d_ranges, const uint8_t* d_planes)
{
    if (!nops || !max_dw || !max_dh) return;
    if (mask[0]) launch_area_ss<0>(s, d_imgs, d_ops, nops, mask[0], max_dw, max_dh, d_taps, d_ranges, d_planes);
    if (mask[1]) launch_area_ss<1>(s, d_imgs, d_ops, nops, mask[1], max_dw, max_dh, d_taps, d_ranges, d_planes);
    if (mask[2]) launch_area_ss<2>(s, d_imgs, d_ops, nops, mask[2], max_dw, max_dh, d_taps, d_ranges, d_planes);
}

void lp_launch_composite(hipStream_t s, const LpCompositeOp& op, const uint8_t* d_src, uint8_t* d_dst)
{
    if (!op.w || !op.h) return;
    dim3 g((op.w + 63) / 64, (op.h + 3) / 4, 1);
    hipLaunchKernelGGL(k_composite, g, dim3(64, 4), 0, s, op, d_src, d_dst);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// HDR -> SDR (color_info.cpp:80-236). PQ / HLG inverse transfer functions and the primaries matrices are the reference's own; the
// tone curve is cv::TonemapReinhard(gamma 1, intensity 0.6, light_adapt 0.2, color_adapt 0.3) as the reference configures it. The
// operator needs three global reductions (range of the linear image; log-luminance and channel statistics of the normalised image;
// range of the mapped image), so it runs as four grid-stride passes with per-workgroup partials folded on the host in between.
// One thread per pixel, 12 bytes of float state per pixel: HBM bound (8 B read + 12 B written in pass 0, 24 B in passes 1-2, 15 B in
// pass 3).
__device__ __forceinline__ float tone_pq(float x)
{
    const float m1 = 0.1593017578125f, m2 = 78.84375f, c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f;
    const float xp = powf(x, 1.0f / m2);
    const float num = fmaxf(xp - c1, 0.0f), den = c2 - c3 * xp;
    return powf(num / den, 1.0f / m1);
}
__device__ __forceinline__ float tone_hlg(float x)
{
    const float a = 0.17883277f, b = 0.28466892f, c = 0.55991073f;
    return x <= 0.5f ? x * x / 3.0f : (expf((x - c) / a) + b) / 12.0f;
}
__device__ __forceinline__ float tone_gray(float c0, float c1, float c2) { return c0 * 0.299f + c1 * 0.587f + c2 * 0.114f; } // COLOR_RGB2GRAY, 32F

__device__ __forceinline__ void tone_reduce(double mn, double mx, const double (&sum)[5], double* __restrict__ out)
{
    constexpr int NS = 5;
    __shared__ double s_part[4][2 + NS];
    for (int o = 32; o; o >>= 1) {
        mn = fmin(mn, __shfl_down(mn, o));
        mx = fmax(mx, __shfl_down(mx, o));
    }
    double acc[NS];
    for (int k = 0; k < NS; k++) {
        acc[k] = sum[k];
        for (int o = 32; o; o >>= 1) acc[k] += __shfl_down(acc[k], o);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_part[wave][0] = mn; s_part[wave][1] = mx;
        for (int k = 0; k < NS; k++) s_part[wave][2 + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = out + (size_t)blockIdx.x * LP_TONE_STATS;
        for (int w = 1; w < 4; w++) {
            s_part[0][0] = fmin(s_part[0][0], s_part[w][0]);
            s_part[0][1] = fmax(s_part[0][1], s_part[w][1]);
            for (int k = 0; k < NS; k++) s_part[0][2 + k] += s_part[w][2 + k];
        }
        o[0] = s_part[0][0]; o[1] = s_part[0][1];
        for (int k = 0; k < NS; k++) o[2 + k] = s_part[0][2 + k];
    }
}

// pass 0: bytes -> linear light, range
__global__ __launch_bounds__(256) void k_tone_linearize(LpToneOp op)
{
    const LpFrame& f = op.f;
    const uint32_t npix = f.w * f.h;
    float* __restrict__ img = reinterpret_cast<float*>(op.img);
    const uint16_t* __restrict__ src16 = reinterpret_cast<const uint16_t*>(op.src16);
    const float scale = 1.0f / (float)((1u << (src16 ? op.depth : 8u)) - 1u);
    double mn = INFINITY, mx = -INFINITY;
    const double none[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(f.off) + (size_t)(i / f.w) * f.stride + (size_t)(i % f.w) * f.cn;
        for (int c = 0; c < 3; c++) {
            float v = (src16 ? (float)src16[(size_t)i * 3 + c] : (float)p[c]) * scale;
            if (op.transfer == 16) v = tone_pq(v);
            else if (op.transfer == 18) v = tone_hlg(v);
            img[(size_t)i * 3 + c] = v;
            mn = fmin(mn, (double)v); mx = fmax(mx, (double)v);
        }
    }
    tone_reduce(mn, mx, none, reinterpret_cast<double*>(op.stats));
}

// pass 1: normalise to [0, 1]; log-luminance range and sum, gray sum, channel sums
__global__ __launch_bounds__(256) void k_tone_stats(LpToneOp op)
{
    const uint32_t npix = op.f.w * op.f.h;
    float* __restrict__ img = reinterpret_cast<float*>(op.img);
    double mn = INFINITY, mx = -INFINITY, sum[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            c[k] = fmaf(img[(size_t)i * 3 + k], op.a, op.b);
            img[(size_t)i * 3 + k] = c[k];
            sum[2 + k] += c[k];
        }
        const float g = tone_gray(c[0], c[1], c[2]);
        const float lg = logf(fmaxf(g, 1e-4f));
        sum[0] += lg; sum[1] += g;
        mn = fmin(mn, (double)lg); mx = fmax(mx, (double)lg);
    }
    tone_reduce(mn, mx, sum, reinterpret_cast<double*>(op.stats));
}

// pass 2: the Reinhard curve per channel; range of the result
__global__ __launch_bounds__(256) void k_tone_map(LpToneOp op)
{
    const uint32_t npix = op.f.w * op.f.h;
    float* __restrict__ img = reinterpret_cast<float*>(op.img);
    const float light_adapt = 0.2f, color_adapt = 0.3f;
    double mn = INFINITY, mx = -INFINITY;
    const double none[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = img[(size_t)i * 3 + k];
        const float g = tone_gray(c[0], c[1], c[2]);
        for (int k = 0; k < 3; k++) {
            float adapt = color_adapt * c[k] + (1.0f - color_adapt) * g;
            adapt = light_adapt * adapt + (1.0f - light_adapt) * op.glob[k];
            adapt = powf(op.intensity * adapt, op.map_key);
            const float v = c[k] * (1.0f / (adapt + c[k]));
            img[(size_t)i * 3 + k] = v;
            mn = fmin(mn, (double)v); mx = fmax(mx, (double)v);
        }
    }
    tone_reduce(mn, mx, none, reinterpret_cast<double*>(op.stats));
}

__constant__ float c_tone_matrix[4][9] = {
    {1.6605f, -0.5876f, -0.0728f, -0.1246f, 1.1329f, -0.0083f, -0.0182f, -0.1006f, 1.1187f},                    // BT.2020 -> BT.709
    {1.2249f, -0.2247f, -0.0002f, -0.0420f, 1.0419f, 0.0001f, -0.0197f, 0.0754f, 0.9443f},                      // P3 -> BT.709
    {1.0440f, -0.0440f, 0.0000f, -0.0000f, 1.0000f, 0.0000f, 0.0000f, 0.0000f, 1.0000f},                        // BT.601 -> BT.709
    {1.0569715f, -0.2039770f, 0.0556301f, 0.0415551f, 1.8759675f, -0.9692436f, -0.4986108f, -1.5373832f, 3.2409699f}, // XYZ -> BT.709
};

// pass 3: normalise, primaries, (gamma for linear-light input), back to bytes
__global__ __launch_bounds__(256) void k_tone_final(LpToneOp op)
{
    const LpFrame& f = op.f;
    const uint32_t npix = f.w * f.h;
    const float* __restrict__ img = reinterpret_cast<const float*>(op.img);
    const int mi = op.primaries == 9 ? 0 : (op.primaries == 11 || op.primaries == 12) ? 1 : op.primaries == 6 ? 2 : op.primaries == 10 ? 3 : -1;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        uint8_t* p = reinterpret_cast<uint8_t*>(f.off) + (size_t)(i / f.w) * f.stride + (size_t)(i % f.w) * f.cn;
        float v[3], o[3];
        for (int k = 0; k < 3; k++) v[k] = fmaf(img[(size_t)i * 3 + k], op.a, op.b);
        for (int j = 0; j < 3; j++)
            o[j] = mi < 0 ? v[j] : c_tone_matrix[mi][j * 3] * v[0] + c_tone_matrix[mi][j * 3 + 1] * v[1] + c_tone_matrix[mi][j * 3 + 2] * v[2];
        for (int j = 0; j < 3; j++) {
            float t = o[j];
            if (op.transfer == 8) t = powf(t, 1.0f / 2.2f);
            t *= 255.0f;
            p[j] = t == t ? (uint8_t)fminf(fmaxf(rintf(t), 0.0f), 255.0f) : (uint8_t)0; // saturate_cast<uchar>(cvRound(t))
        }
    }
}

void lp_launch_tonemap(hipStream_t s, const LpToneOp& op, int pass, uint32_t* n_wg)
{
    const uint64_t npix = (uint64_t)op.f.w * op.f.h;
    const uint32_t g = (uint32_t)std::min<uint64_t>(LP_TONE_MAX_WG, (npix + 255) / 256);
    if (n_wg) *n_wg = g;
    if (!g) return;
    switch (pass) {
    case 0: hipLaunchKernelGGL(k_tone_linearize, dim3(g), dim3(256), 0, s, op); break;
    case 1: hipLaunchKernelGGL(k_tone_stats, dim3(g), dim3(256), 0, s, op); break;
    case 2: hipLaunchKernelGGL(k_tone_map, dim3(g), dim3(256), 0, s, op); break;
    default: hipLaunchKernelGGL(k_tone_final, dim3(g), dim3(256), 0, s, op); break;
    }
}
