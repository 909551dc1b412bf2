// Fractional INTER_AREA straight from 4:2:0 planes: one destination pixel of cv::resize's resizeArea_ (resize.cpp ResizeArea_Invoker:
// a row buffer of alpha-weighted sums per source row, then beta-weighted into the destination row; all in float, in tap order), with
// the source pixel produced on the fly from the decoded planes -- jdsample.c h2v2_fancy_upsample, then jdcolor.c ycc_rgb_convert in
// BGR order -- instead of read from a materialised BGR frame. The arithmetic and its order are k_ycc_to_frame_420's followed by
// k_resize_area3's, so the bytes are the same as the frame route's.
//
// Host + device: the device kernel (lp_kernels_pixel.hip k_area_420) and the CPU-side test hook (lilliput_hip_area420_host, which lets
// the tests compare this restated order of operations with the oracle without a GPU) both instantiate lp_area420_pixel.
#pragma once
#include <cstdint>
#include "lp_types.h"

#if defined(__HIPCC__)
#define LPA_HD __host__ __device__ __forceinline__
#else
#define LPA_HD inline
#endif

// One image's share of the work. Orientations 1-4: tap index si of the oriented frame's x axis is source column x0 + xstep * si, of its
// y axis source row y0 + ystep * si. Orientations 5-8 (`transposed`): the oriented x axis runs down the source rows.
struct LpArea420Op {
    uint32_t img;                       // index into the LpJpeg array of the current decode range
    int32_t x0, y0, xstep, ystep;
    uint32_t xtab_off, ytab_off;        // into the tap arena
    uint32_t xrange_off, yrange_off;    // into the range arena: [dw + 1] / [dh + 1] tap ranges
    uint32_t maxt;                      // 6 / 10 / 18 / 34 / 66: which instantiation takes it
    uint32_t transposed;                // orientations 5-8 (k_area_420t): x0 / xstep place the Y taps along source x, y0 / ystep the X taps along source y; maxt counts y taps
    // Integer scales (round 5): cv::resize's resizeAreaFast_ -- the integer sum of the box, then saturate_cast<uchar>(sum * (1 / area)), or
    // (sum + 2) >> 2 for 2 x 2 boxes (ResizeAreaFastVec_SIMD_8u) -- through the same walk: every tap carries weight 1 (the float sums of at
    // most 66 x 66 bytes are exact integers) and the result is finished by `post`. Fractional scales: post = 1 (x * 1.f == x), half_up = 0.
    float post;                         // the sum is multiplied by this before it is rounded (half to even) and clamped
    uint32_t half_up;                   // 1: (sum + 2) >> 2 instead
    LpFrame dst;
};

#define LP_AREA_SLACK 128 // bytes the plane arena keeps free before its first and after its last plane (the widest window, 67 columns, overhangs a row by < 80 bytes)
struct LpAreaPlanes {
    const uint8_t* py; const uint8_t* pb; const uint8_t* pr;
    uint32_t sy, sc;                    // plane strides (luma, chroma)
    int32_t dw, dh;                     // chroma size that jdsample.c works on: 4:2:0 ceil(W / 2), ceil(H / 2); 4:2:2 ceil(W / 2), H; 4:4:4 W, H
};

LPA_HD uint32_t lpa_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) // bytes of {hi:lo} from bit sh (0 / 8 / 16 / 24)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
#endif
}
template <int B> LPA_HD uint32_t lpa_pair(uint32_t wr, uint32_t wb) // {Cb = byte B of wb, Cr = byte B of wr} in the two halves
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(wr, wb, 0x0c000c00u | ((4u + B) << 16) | (uint32_t)B);
#else
    return ((wb >> (8 * B)) & 255u) | (((wr >> (8 * B)) & 255u) << 16);
#endif
}
LPA_HD int32_t lpa_uniform(int32_t v) // a value every lane of the wave holds alike (the wave is one destination row): keep it in a scalar register
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
// One decoded plane as the walk reads it: an aligned dword at (row offset, lane offset). On the device a buffer resource built from
// wave-uniform values: the row offset rides in the scalar offset, the lane's dword offset in the 32-bit vector offset, so a load costs
// no address arithmetic (a flat load wanted a 64-bit add per load: 22 of them per source row).
struct LpaPlane {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ uint32_t word(uint32_t row_off, uint32_t lane_off) const
    {
        return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_off + LP_AREA_SLACK, (int)row_off, 0);
    }
    // N consecutive dwords from (row offset, lane offset) in as few instructions as the widths allow (dwordx4 / x2 / x1): a window's 6 luma
    // dwords are two loads instead of six, a chroma row's 4 dwords one instead of four -- 6 vector-memory instructions per source row
    // instead of 22. The offsets are NOT clamped into the row: a window that hangs over a row's end reads on into the next row (or, past
    // the last plane, into the arena's slack: LP_AREA_SLACK bytes before the first and after the last plane), and one that starts
    // left of column 0 reads the bytes before the row; those bytes only ever meet weight 0 (see LpaWindow).
    template <int N>
    __device__ __forceinline__ void words(uint32_t row_off, int32_t lane_off, uint32_t* out) const
    {
        typedef unsigned int v4_ __attribute__((ext_vector_type(4)));
        typedef unsigned int v2_ __attribute__((ext_vector_type(2)));
        int i = 0;
#pragma unroll
        for (; i + 4 <= N; i += 4) {
            const v4_ t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + LP_AREA_SLACK + 4 * i, (int)row_off, 0);
            out[i] = t.x; out[i + 1] = t.y; out[i + 2] = t.z; out[i + 3] = t.w;
        }
        if (i + 2 <= N) {
            const v2_ t = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off + LP_AREA_SLACK + 4 * i, (int)row_off, 0);
            out[i] = t.x; out[i + 1] = t.y;
            i += 2;
        }
        if (i < N) out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off + LP_AREA_SLACK + 4 * i, (int)row_off, 0);
    }
#else
    const uint8_t* p;
    uint32_t word(uint32_t row_off, uint32_t lane_off) const
    {
        uint32_t v;
        __builtin_memcpy(&v, p + row_off + lane_off, 4);
        return v;
    }
#endif
};
LPA_HD LpaPlane lpa_plane(const uint8_t* p)
{
    LpaPlane q;
#if defined(__HIP_DEVICE_COMPILE__)
    // the resource starts LP_AREA_SLACK bytes BEFORE the plane and every lane offset carries the same bias (it rides in the instruction's
    // immediate offset): buffer offsets are unsigned, and a window that starts left of column 0 has a negative one
    const uintptr_t a = reinterpret_cast<uintptr_t>(p) - LP_AREA_SLACK;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(a >> 32));
    q.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, (int)0xffffffffu, 0x00020000);
#else
    q.p = p;
#endif
    return q;
}
LPA_HD int32_t lpa_clamp(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : v > hi ? hi : v; }
LPA_HD float lpa_mul(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmul_rn(a, b);
#else
    return a * b; // built with -ffp-contract=off
#endif
}
LPA_HD float lpa_add(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fadd_rn(a, b);
#else
    return a + b;
#endif
}
LPA_HD uint32_t lpa_round_u8(float v) // saturate_cast<uchar>(float): cvRound (half to even), then clamp
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int32_t i = __float2int_rn(v);
#else
    const int32_t i = (int32_t)__builtin_lrintf(v);
#endif
    return (uint32_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}

// the sum of a destination pixel -> its byte: fractional scales round the sum itself (post == 1); integer scales see LpArea420Op::post
LPA_HD uint8_t lpa_finish(float sum, float post, uint32_t half_up)
{
    if (half_up) return (uint8_t)(((uint32_t)(int32_t)sum + 2u) >> 2); // four bytes: at most 1020, an exact float
    return (uint8_t)lpa_round_u8(lpa_mul(sum, post));
}

LPA_HD int32_t lpa_mad24(int32_t a, int32_t b, int32_t c) // a * b + c, both factors within 24 bits
{
#if defined(__HIP_DEVICE_COMPILE__)
    // spelled out: left to itself the compiler splits the green channel's two of these into two multiplies and a three-way add
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(b), "s"(a), "v"(c));
    return d;
#else
    return a * b + c;
#endif
}
typedef float lpa_f2 __attribute__((ext_vector_type(2)));

#define LPA_FIX16(x) ((int32_t)((x)*65536.0 + 0.5))

// ---- the window a lane walks along source x ------------------------------------------------------------------------------------
// MAXT = taps per axis the instantiation covers (a destination pixel with fewer carries weight 0 in the rest, which adds exactly 0).
// The taps can reach the source columns [xa, xa + MAXT); the window starts at the even column xe = xa & ~1 and is MAXT + 1 wide, so
// that which chroma column and which neighbour a window column uses is known at compile time; the weights are shifted by xa's
// parity instead (one more weight-0 term at one end of the sum). Columns outside [0, W) only ever meet weight 0.
// SS = how the chroma planes are sampled: 2 = 4:2:0 (h2v2 fancy upsampling), 1 = 4:2:2 (h2v1 fancy upsampling), 0 = 4:4:4 (none).
template <int MAXT, int SS>
struct LpaWindow {
    static_assert(MAXT % 2 == 0, "even tap counts only");
    static constexpr int NX = MAXT + 1;             // window columns
    static constexpr int NWY = (NX + 3) / 4;        // luma dwords of the window (after the byte-phase fix-up)
    static constexpr int NC = SS ? MAXT / 2 + 2 : NX; // chroma columns: subsampled, the window's own MAXT / 2 + 1 and a neighbour either side
    static constexpr int NWC = (NC + 3) / 4;
    uint32_t oy[NWY + 1], oc[NWC + 1];              // dword offsets inside a plane row (unsigned: scalar row + zero-extended lane offset is one addressing mode)
    uint32_t shy, shc;                              // byte phase of the window in its first dword, luma / chroma
    int32_t oyb, ocb;                               // the first dword's offset as it is (device: wide loads, not clamped; may be negative)
    int32_t il, ir;                                 // window index that stands for chroma column -1 / column dw
    bool edge;

    LPA_HD void init(const LpAreaPlanes& P, int32_t xe)
    {
        const int32_t xo = xe & ~3, c_lo = SS ? (xe >> 1) - 1 : xe, co = c_lo & ~3;
        shy = (uint32_t)(xe & 3) * 8; shc = (uint32_t)(c_lo & 3) * 8;
        oyb = xo; ocb = co;
        // clamped into the row: a clamped dword only feeds columns outside the image
#pragma unroll
        for (int i = 0; i <= NWY; i++) oy[i] = (uint32_t)lpa_clamp(xo + 4 * i, 0, (int32_t)P.sy - 4);
#pragma unroll
        for (int i = 0; i <= NWC; i++) oc[i] = (uint32_t)lpa_clamp(co + 4 * i, 0, (int32_t)P.sc - 4);
        // jdsample.c replicates the first and the last chroma column (of downsampled_width, not of the padded plane). Columns further
        // out are only reached by weight-0 taps.
        il = -1 - c_lo; ir = P.dw - c_lo;
        edge = SS && (il >= 0 || ir < NC);
    }
    // weight of window column c from the weights in tap order: tap k sits at column k + odd (FLIP: MAXT - 1 - k + odd)
    template <bool FLIP>
    LPA_HD static void weights(int32_t odd, const float (&tap)[MAXT], float (&w)[NX])
    {
#pragma unroll
        for (int c = 0; c < NX; c++) {
            const float w0 = c < MAXT ? tap[FLIP ? MAXT - 1 - c : c] : 0.f;   // xa even
            const float w1 = c >= 1 ? tap[FLIP ? MAXT - c : c - 1] : 0.f;     // xa odd
            w[c] = odd ? w1 : w0;
        }
    }
};

// One source row sy (the same for the whole wave) of a lane's window: f(c, b, g, r) for its NX columns, ascending or (REV) descending.
// The upsamplers' (jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample) and the colour conversion's integer arithmetic as in
// k_ycc_to_frame_420 / k_ycc_to_frame.
template <int MAXT, int SS, bool REV, class F>
LPA_HD void lpa_row(const LpAreaPlanes& P, const LpaPlane& PY, const LpaPlane& PB, const LpaPlane& PR, const LpaWindow<MAXT, SS>& W, int32_t sy, F&& f)
{
    typedef LpaWindow<MAXT, SS> Win;
    constexpr int NX = Win::NX, NWY = Win::NWY, NC = Win::NC, NWC = Win::NWC;
    const int32_t KR = 32768 - 128 * LPA_FIX16(1.40200), KB = 32768 - 128 * LPA_FIX16(1.77200);
    const int32_t KG = 32768 + 128 * LPA_FIX16(0.34414) + 128 * LPA_FIX16(0.71414);
    const int32_t cy = SS == 2 ? sy >> 1 : sy;
    const int32_t ny = SS == 2 ? lpa_clamp((sy & 1) ? cy + 1 : cy - 1, 0, P.dh - 1) : cy;
    const uint32_t ry = (uint32_t)sy * P.sy, rc0 = (uint32_t)cy * P.sc, rc1 = (uint32_t)ny * P.sc;
    uint32_t wy[NWY + 1], wb0[NWC + 1], wb1[NWC + 1], wr0[NWC + 1], wr1[NWC + 1];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LP_AREA_NARROW_LOADS)
    PY.template words<NWY + 1>(ry, W.oyb, wy);
    PB.template words<NWC + 1>(rc0, W.ocb, wb0); PR.template words<NWC + 1>(rc0, W.ocb, wr0);
    if (SS == 2) { PB.template words<NWC + 1>(rc1, W.ocb, wb1); PR.template words<NWC + 1>(rc1, W.ocb, wr1); }
#else
#pragma unroll
    for (int i = 0; i <= NWY; i++) wy[i] = PY.word(ry, W.oy[i]);
#pragma unroll
    for (int i = 0; i <= NWC; i++) {
        wb0[i] = PB.word(rc0, W.oc[i]); wr0[i] = PR.word(rc0, W.oc[i]);
        if (SS == 2) { wb1[i] = PB.word(rc1, W.oc[i]); wr1[i] = PR.word(rc1, W.oc[i]); }
    }
#endif
#pragma unroll
    for (int i = 0; i < NWY; i++) wy[i] = lpa_alignbit(wy[i + 1], wy[i], W.shy);
#pragma unroll
    for (int i = 0; i < NWC; i++) {
        wb0[i] = lpa_alignbit(wb0[i + 1], wb0[i], W.shc); wr0[i] = lpa_alignbit(wr0[i + 1], wr0[i], W.shc);
        if (SS == 2) { wb1[i] = lpa_alignbit(wb1[i + 1], wb1[i], W.shc); wr1[i] = lpa_alignbit(wr1[i + 1], wr1[i], W.shc); }
    }
    // {Cb, Cr} packed in the halves of one register. 4:2:0: the vertical half of the upsampler, 3 * nearer row + further row (<= 1020)
    uint32_t V[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        uint32_t a, b = 0;
        switch (i & 3) {
        case 0: a = lpa_pair<0>(wr0[i >> 2], wb0[i >> 2]); if (SS == 2) b = lpa_pair<0>(wr1[i >> 2], wb1[i >> 2]); break;
        case 1: a = lpa_pair<1>(wr0[i >> 2], wb0[i >> 2]); if (SS == 2) b = lpa_pair<1>(wr1[i >> 2], wb1[i >> 2]); break;
        case 2: a = lpa_pair<2>(wr0[i >> 2], wb0[i >> 2]); if (SS == 2) b = lpa_pair<2>(wr1[i >> 2], wb1[i >> 2]); break;
        default: a = lpa_pair<3>(wr0[i >> 2], wb0[i >> 2]); if (SS == 2) b = lpa_pair<3>(wr1[i >> 2], wb1[i >> 2]); break;
        }
        V[i] = SS == 2 ? 3u * a + b : a;
    }
    if (SS && W.edge) {
#pragma unroll
        for (int i = 0; i + 1 < NC; i++) V[i] = i == W.il ? V[i + 1] : V[i];
#pragma unroll
        for (int i = 1; i < NC; i++) V[i] = i == W.ir ? V[i - 1] : V[i];
    }
    // the {Cb, Cr} pair of window column c (16-bit halves): the horizontal half of the upsampler
    auto chroma = [&](int c) -> uint32_t {
        if (!SS) return V[c];
        // column xe + c: chroma window index c / 2 + 1; odd columns blend with the column to the right, even ones with the one to the
        // left -- the horizontal half of h2v2_fancy_upsample on the vertical sums, (3 * near + far + 7 or 8) >> 4, or
        // h2v1_fancy_upsample on the samples, (3 * near + far + 2 or 1) >> 2 (a replicated neighbour gives the edge rule: near itself)
        const int ic = c / 2 + 1;
        const uint32_t bias = SS == 2 ? ((c & 1) ? 0x00070007u : 0x00080008u) : ((c & 1) ? 0x00020002u : 0x00010001u);
        const uint32_t h = 3u * V[ic] + V[(c & 1) ? ic + 1 : ic - 1] + bias; // both halves at once: neither carries into the other (<= 4088)
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, (u16x2_)(__builtin_bit_cast(u16x2_, h) >> (u16x2_){SS == 2 ? 4 : 2, SS == 2 ? 4 : 2})); // one packed shift
#else
        return (h >> (SS == 2 ? 4 : 2)) & (SS == 2 ? 0x0fff0fffu : 0x3fff3fffu);
#endif
    };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LP_AREA_PLAIN_COLUMNS)
    // Device: two columns at a time. Every colour term is one v_dot2_u32_u16 over the {Cb, Cr} pair (FIX(1.772) = 2 * 58065 and
    // FIX(1.402) = 3 * 30627: with {2 Cb, 3 Cr} the constants fit 16-bit operands; green as -floor((x + 65535 - KG) / 65536)), the luma is
    // added to the terms' upper halves by SDWA adds that read the byte in place and write one half of a pair register ({B, G} of a column;
    // the R of the two columns), v_sat_pk_u8_i16 clamps a pair to two bytes, v_cvt_f32_ubyte0 / 1 turns them into the floats the taps
    // multiply: 7.5 instructions per column where a shift, an add, a median and a conversion per channel took 9 (same integers, same
    // floats). One asm block per pair keeps every half-register write three instructions away from its reader (gfx940 family: one wait
    // state, and the compiler does not look inside).
    typedef unsigned short lpa_u16x2 __attribute__((ext_vector_type(2)));
    auto terms = [&](int c, uint32_t& tb, uint32_t& tg, uint32_t& tr) {
        const lpa_u16x2 h = __builtin_bit_cast(lpa_u16x2, chroma(c));
        const lpa_u16x2 hs = h * (lpa_u16x2){2, 3};
        static_assert(LPA_FIX16(1.77200) == 2 * 58065 && LPA_FIX16(1.40200) == 3 * 30627, "split of the colour constants");
        tr = __builtin_amdgcn_udot2(hs, (lpa_u16x2){0, 30627}, (uint32_t)KR, false);
        tb = __builtin_amdgcn_udot2(hs, (lpa_u16x2){58065, 0}, (uint32_t)KB, false);
        tg = __builtin_amdgcn_udot2(h, (lpa_u16x2){(unsigned short)LPA_FIX16(0.34414), (unsigned short)LPA_FIX16(0.71414)}, 65535u - (uint32_t)KG, false);
    };
#define LPA_PAIR_ASM(B0, B1)                                                                                                              \
    asm("v_add_u16_sdwa %[x0], %[tb0], %[y0] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_" #B0 "\n\t"                 \
        "v_add_u16_sdwa %[x1], %[tb1], %[y1] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_" #B1 "\n\t"                 \
        "v_add_u16_sdwa %[xr], %[tr0], %[y0] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_" #B0 "\n\t"                 \
        "v_sub_u16_sdwa %[x0], %[y0], %[tg0] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B0 " src1_sel:WORD_1\n\t"            \
        "v_sub_u16_sdwa %[x1], %[y1], %[tg1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B1 " src1_sel:WORD_1\n\t"            \
        "v_add_u16_sdwa %[xr], %[tr1], %[y1] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_" #B1 "\n\t"            \
        "v_sat_pk_u8_i16 %[x0], %[x0]\n\t"                                                                                                \
        "v_sat_pk_u8_i16 %[x1], %[x1]\n\t"                                                                                                \
        "v_sat_pk_u8_i16 %[xr], %[xr]\n\t"                                                                                                \
        "v_cvt_f32_ubyte0 %[b0], %[x0]\n\t"                                                                                               \
        "v_cvt_f32_ubyte1 %[g0], %[x0]\n\t"                                                                                               \
        "v_cvt_f32_ubyte0 %[b1], %[x1]\n\t"                                                                                               \
        "v_cvt_f32_ubyte1 %[g1], %[x1]\n\t"                                                                                               \
        "v_cvt_f32_ubyte0 %[r0], %[xr]\n\t"                                                                                               \
        "v_cvt_f32_ubyte1 %[r1], %[xr]"                                                                                                   \
        : [x0] "=&v"(x0), [x1] "=&v"(x1), [xr] "=&v"(xr), [b0] "=&v"(fb0), [g0] "=&v"(fg0), [r0] "=&v"(fr0), [b1] "=&v"(fb1), [g1] "=&v"(fg1),    \
          [r1] "=&v"(fr1)                                                                                                                 \
        : [y0] "v"(y0), [y1] "v"(y1), [tb0] "v"(tb0), [tg0] "v"(tg0), [tr0] "v"(tr0), [tb1] "v"(tb1), [tg1] "v"(tg1), [tr1] "v"(tr1))
#pragma unroll
    for (int t = 0; t + 1 < NX; t += 2) {
        const int c0 = REV ? NX - 1 - t : t, c1 = REV ? NX - 2 - t : t + 1;
        uint32_t tb0, tg0, tr0, tb1, tg1, tr1, x0, x1, xr;
        float fb0, fg0, fr0, fb1, fg1, fr1;
        terms(c0, tb0, tg0, tr0);
        terms(c1, tb1, tg1, tr1);
        const uint32_t y0 = wy[c0 >> 2], y1 = wy[c1 >> 2];
        switch ((c0 & 3) * 4 + (c1 & 3)) { // compile-time: the loop is unrolled
        case 0 * 4 + 1: LPA_PAIR_ASM(0, 1); break;
        case 2 * 4 + 3: LPA_PAIR_ASM(2, 3); break;
        case 2 * 4 + 1: LPA_PAIR_ASM(2, 1); break;
        case 0 * 4 + 3: LPA_PAIR_ASM(0, 3); break;
        case 1 * 4 + 0: LPA_PAIR_ASM(1, 0); break;
        case 3 * 4 + 2: LPA_PAIR_ASM(3, 2); break;
        case 1 * 4 + 2: LPA_PAIR_ASM(1, 2); break;
        default: LPA_PAIR_ASM(3, 0); break;
        }
        f(c0, fb0, fg0, fr0);
        f(c1, fb1, fg1, fr1);
    }
#undef LPA_PAIR_ASM
    constexpr int T0 = (NX / 2) * 2; // the window is MAXT + 1 columns wide: one is left over
#else
    constexpr int T0 = 0;
#endif
#pragma unroll
    for (int t = T0; t < NX; t++) {
        const int c = REV ? NX - 1 - t : t;
        const uint32_t h = chroma(c);
        const int32_t cb = (int32_t)(h & 0xffffu), cr = (int32_t)(h >> 16);
        const int32_t yy = (int32_t)((wy[c >> 2] >> (8 * (c & 3))) & 255u);
        const int32_t r = lpa_clamp(yy + (lpa_mad24(LPA_FIX16(1.40200), cr, KR) >> 16), 0, 255);
        const int32_t b = lpa_clamp(yy + (lpa_mad24(LPA_FIX16(1.77200), cb, KB) >> 16), 0, 255);
        const int32_t g = lpa_clamp(yy + (lpa_mad24(-LPA_FIX16(0.34414), cb, lpa_mad24(-LPA_FIX16(0.71414), cr, KG)) >> 16), 0, 255);
        f(c, (float)b, (float)g, (float)r);
    }
}

// ---- orientations 1-4: destination x runs along source x -------------------------------------------------------------------------
// FLIPX: tap k reads column MAXT-1-k of [xa, xa + MAXT) (orientations 2 and 3) -- the sums still run in tap order.
//   al[k]  x weights in tap order
//   yt, y0..y1, ybase, ystep  the destination row's taps (wave-uniform): source row = ybase + ystep * yt[j].si
// resize.cpp ResizeArea_Invoker: per source row buf = sum over the x taps of value * alpha; sum += beta * buf.
template <int MAXT, int SS, bool FLIPX>
LPA_HD void lp_area420_pixel(const LpAreaPlanes& P, int32_t xa, const float (&al)[MAXT], const LpTap* __restrict__ yt, uint32_t y0, uint32_t y1,
                             int32_t ybase, int32_t ystep, uint8_t* __restrict__ out, float post = 1.f, uint32_t half_up = 0)
{
#pragma clang fp contract(off) // every product is rounded before it is added, as in resize.cpp's scalar loops
    constexpr int NX = LpaWindow<MAXT, SS>::NX;
    const int32_t odd = xa & 1;
    float w[NX];
    LpaWindow<MAXT, SS>::template weights<FLIPX>(odd, al, w);
    LpaWindow<MAXT, SS> W;
    W.init(P, xa - odd);
    const LpaPlane PY = lpa_plane(P.py), PB = lpa_plane(P.pb), PR = lpa_plane(P.pr);
    lpa_f2 sbg = {0.f, 0.f};
    float sum_r = 0.f;
    // the rows are the same for the whole wave: scalar loop, scalar row addresses, per-lane offsets only
    const uint32_t j0 = (uint32_t)lpa_uniform((int32_t)y0), j1 = (uint32_t)lpa_uniform((int32_t)y1);
    for (uint32_t j = j0; j < j1; j++) {
        const float beta = yt[j].alpha;
        const int32_t sy = ybase + ystep * (int32_t)yt[j].si;
        // blue and green travel as a pair (one v_pk_mul_f32 + one v_pk_add_f32 for the two: each half is the same IEEE multiply and add)
        lpa_f2 bg = {0.f, 0.f};
        float rs = 0.f;
        lpa_row<MAXT, SS, FLIPX>(P, PY, PB, PR, W, sy, [&](int c, float b, float g, float r) {
            const lpa_f2 pbg = {b, g}, ww = {w[c], w[c]};
            bg = bg + pbg * ww;
            rs = lpa_add(rs, lpa_mul(r, w[c]));
        });
        const lpa_f2 bb = {beta, beta};
        sbg = sbg + bb * bg;
        sum_r = lpa_add(sum_r, lpa_mul(beta, rs));
    }
    out[0] = lpa_finish(sbg.x, post, half_up); out[1] = lpa_finish(sbg.y, post, half_up); out[2] = lpa_finish(sum_r, post, half_up);
}

// ---- orientations 5-8: destination x runs along source y --------------------------------------------------------------------------
// A row of the oriented frame is a COLUMN of the decoded image: for destination (dx, dy) the row buffer of oriented row j is the
// alpha-weighted sum down source column c_j over the source rows the x taps name, and the rows are beta-weighted in j order. The
// planes are still read row by row (the lanes of a wave are neighbouring dy = neighbouring source columns, so the loads coalesce as in
// the row-wise kernel): every window column keeps its own row buffer while the source rows go by in tap order, and the beta pass runs
// over the window's columns at the end -- the same products added in the same order as resize.cpp's.
//   xa, be[k]  leftmost source column the y taps can reach, y weights in tap order; FLIPC: tap k reads column MAXT-1-k (orientations 7, 8)
//   xt, x0..x1, rbase, rstep  the destination column's taps (wave-uniform): source row = rbase + rstep * xt[k].si
template <int MAXT, int SS, bool FLIPC>
LPA_HD void lp_area420t_pixel(const LpAreaPlanes& P, int32_t xa, const float (&be)[MAXT], const LpTap* __restrict__ xt, uint32_t x0, uint32_t x1,
                              int32_t rbase, int32_t rstep, uint8_t* __restrict__ out, float post = 1.f, uint32_t half_up = 0)
{
#pragma clang fp contract(off)
    constexpr int NX = LpaWindow<MAXT, SS>::NX;
    const int32_t odd = xa & 1;
    float w[NX];
    LpaWindow<MAXT, SS>::template weights<FLIPC>(odd, be, w);
    LpaWindow<MAXT, SS> W;
    W.init(P, xa - odd);
    const LpaPlane PY = lpa_plane(P.py), PB = lpa_plane(P.pb), PR = lpa_plane(P.pr);
    lpa_f2 bg[NX];
    float rs[NX];
#pragma unroll
    for (int c = 0; c < NX; c++) { bg[c] = lpa_f2{0.f, 0.f}; rs[c] = 0.f; }
    const uint32_t k0 = (uint32_t)lpa_uniform((int32_t)x0), k1 = (uint32_t)lpa_uniform((int32_t)x1);
    for (uint32_t k = k0; k < k1; k++) {
        const float alpha = xt[k].alpha;
        const int32_t sy = rbase + rstep * (int32_t)xt[k].si;
        const lpa_f2 aa = {alpha, alpha};
        lpa_row<MAXT, SS, false>(P, PY, PB, PR, W, sy, [&](int c, float b, float g, float r) {
            const lpa_f2 pbg = {b, g};
            bg[c] = bg[c] + pbg * aa;
            rs[c] = lpa_add(rs[c], lpa_mul(r, alpha));
        });
    }
    lpa_f2 sbg = {0.f, 0.f};
    float sum_r = 0.f;
#pragma unroll
    for (int t = 0; t < NX; t++) {
        const int c = FLIPC ? NX - 1 - t : t;   // oriented rows in order
        const lpa_f2 ww = {w[c], w[c]};
        sbg = sbg + ww * bg[c];
        sum_r = lpa_add(sum_r, lpa_mul(w[c], rs[c]));
    }
    out[0] = lpa_finish(sbg.x, post, half_up); out[1] = lpa_finish(sbg.y, post, half_up); out[2] = lpa_finish(sum_r, post, half_up);
}
