// lp_kernels_encode.hip -- gfx950 kernels for the JPEG encode half of ImageOps.Transform (S8-S10):
// BGR -> YCbCr + h2v2 chroma downsample (jccolor.c / jcsample.c), islow FDCT + quantise
// (jfdctint.c / jcdctmgr.c), baseline Huffman coding with the Annex-K tables + byte stuffing +
// framing (jchuff.c / jcmarker.c). Replaces opencv_encoder_write (/root/reference/opencv.cpp:185-194,
// cv::JpegEncoder -> libjpeg-turbo with jpeg_set_defaults + jpeg_set_quality(q, TRUE)).
// The output must be byte-identical to libjpeg-turbo for identical input pixels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_launch.h"
#include "lp_types.h"

#define FIX16(x) ((int32_t)((x)*65536.0 + 0.5))
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

__constant__ uint8_t c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Annex-K encode tables, filled by lp_encode_init_tables() (host) before the first launch.
struct LpEncTables {
    uint16_t code[4][256]; // 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma
    uint8_t len[4][256];
};
__device__ LpEncTables g_enc_tables;

void lp_encode_upload_tables(const uint16_t code[4][256], const uint8_t len[4][256])
{
    LpEncTables t;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 256; j++) { t.code[i][j] = code[i][j]; t.len[i][j] = len[i][j]; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_tables), &t, sizeof(t));
}

__device__ __forceinline__ void load_bgr(const LpFrame& f, const uint8_t* __restrict__ base, int32_t x, int32_t y, int32_t& b, int32_t& g, int32_t& r)
{
    x = x > (int32_t)f.w - 1 ? (int32_t)f.w - 1 : x;
    y = y > (int32_t)f.h - 1 ? (int32_t)f.h - 1 : y;
    const uint8_t* p = base + f.off + (size_t)y * f.stride + (size_t)x * f.cn;
    b = p[0];
    if (f.cn == 1) { g = b; r = b; } else { g = p[1]; r = p[2]; }
}
// N consecutive pixels of row y from column x (N = 8 or 16) as packed 0x00RRGGBB words. Inside the frame, with three channels and a
// dword-aligned run (a frame whose width is a multiple of four, the usual thumbnail), the run is 3 N / 4 dword loads; otherwise the
// per-pixel route with its edge replication (the FDCT kernel issued 9 byte loads per pixel: 0.58 of the encoder's 1.0 us per image).
template <int N>
__device__ __forceinline__ void load_bgr_run(const LpFrame& f, const uint8_t* __restrict__ base, int32_t x, int32_t y, uint32_t (&px)[N])
{
    const uint8_t* p = base + f.off + (size_t)(y > (int32_t)f.h - 1 ? (int32_t)f.h - 1 : y) * f.stride + (size_t)x * 3;
    if (f.cn == 3 && x + N <= (int32_t)f.w && y < (int32_t)f.h && (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
        uint32_t w[3 * N / 4];
#pragma unroll
        for (int i = 0; i < 3 * N / 4; i++) w[i] = reinterpret_cast<const uint32_t*>(p)[i];
#pragma unroll
        for (int i = 0; i < N; i++) { // pixel i = bytes 3 i .. 3 i + 2 of the run
            const int b0 = 3 * i, q = b0 >> 2, sh = (b0 & 3) * 8;
            px[i] = (sh == 0 ? w[q] : sh == 8 ? w[q] >> 8 : __builtin_amdgcn_alignbit(w[(q + 1) % (3 * N / 4)], w[q], sh)) & 0xffffffu;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t b, g, r;
            load_bgr(f, base, x + i, y, b, g, r);
            px[i] = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16);
        }
    }
}

__device__ __forceinline__ int32_t to_y(int32_t b, int32_t g, int32_t r) { return (FIX16(0.29900) * r + FIX16(0.58700) * g + FIX16(0.11400) * b + 32768) >> 16; }
__device__ __forceinline__ int32_t to_cb(int32_t b, int32_t g, int32_t r) { return (-FIX16(0.16874) * r - FIX16(0.33126) * g + FIX16(0.50000) * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ int32_t to_cr(int32_t b, int32_t g, int32_t r) { return (FIX16(0.50000) * r - FIX16(0.41869) * g - FIX16(0.08131) * b + (128 << 16) + 32767) >> 16; }

// 1-D forward DCT (jfdctint.c); pass 1 (rows): outputs scaled by 4 / descaled by 11; pass 2 (columns): 2 / 15.
template <int PASS>
__device__ __forceinline__ void fdct_1d(int32_t p[8])
{
    int32_t t0 = p[0] + p[7], t7 = p[0] - p[7], t1 = p[1] + p[6], t6 = p[1] - p[6];
    int32_t t2 = p[2] + p[5], t5 = p[2] - p[5], t3 = p[3] + p[4], t4 = p[3] - p[4];
    int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = PASS == 1 ? 11 : 15;
    if (PASS == 1) { p[0] = (t10 + t11) * 4; p[4] = (t10 - t11) * 4; }
    else { p[0] = DESCALE(t10 + t11, 2); p[4] = DESCALE(t10 - t11, 2); }
    int32_t z1 = (t12 + t13) * 4433;
    p[2] = DESCALE(z1 + t13 * 6270, SH);
    p[6] = DESCALE(z1 - t12 * 15137, SH);
    z1 = t4 + t7;
    int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7, z5 = (z3 + z4) * 9633;
    t4 *= 2446; t5 *= 16819; t6 *= 25172; t7 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5;
    p[7] = DESCALE(t4 + z1 + z3, SH); p[5] = DESCALE(t5 + z2 + z4, SH);
    p[3] = DESCALE(t6 + z2 + z3, SH); p[1] = DESCALE(t7 + z1 + z4, SH);
}

// E1: colour convert + downsample + FDCT + quantise. 8 threads per block (thread = row, then column);
// 32 blocks per workgroup. Output: zigzag-ordered int16 coefficients, MCU order. Dummy blocks are written
// as zeros; their DC is resolved by enc_eff_dc() in the entropy stage.
__global__ __launch_bounds__(256) void k_enc_fdct(const LpEncJob* __restrict__ jobs, const uint8_t* __restrict__ frames, int16_t* __restrict__ coef_arena)
{
    __shared__ int32_t s_w[32][65];
    __shared__ int16_t s_q[32][64];
    const LpEncJob& job = jobs[blockIdx.y];
    if (blockIdx.x * 32 >= job.total_blocks) return;
    const uint32_t lb = threadIdx.x >> 3, r = threadIdx.x & 7;
    const uint32_t blk = blockIdx.x * 32 + lb;
    const bool active = blk < job.total_blocks;
    const uint32_t m = active ? blk / job.bpm : 0, k = active ? blk - m * job.bpm : 0;
    const uint32_t mx = m % job.mcus_x, my = m / job.mcus_x;
    const int32_t W = (int32_t)job.src.w, H = (int32_t)job.src.h;
    int32_t v[8];
    bool dummy = false;
    uint32_t qsel = 0;
    if (job.ncomp == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) { int32_t b, g, rr; load_bgr(job.src, frames, (int32_t)mx * 8 + i, (int32_t)my * 8 + (int32_t)r, b, g, rr); v[i] = b - 128; }
    } else if (k < 4) {
        const uint32_t bx = mx * 2 + (k & 1), by = my * 2 + (k >> 1);
        dummy = bx >= job.wib || by >= job.hib;
        uint32_t px[8];
        load_bgr_run<8>(job.src, frames, (int32_t)bx * 8, (int32_t)by * 8 + (int32_t)r, px);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = to_y((int32_t)(px[i] & 255u), (int32_t)((px[i] >> 8) & 255u), (int32_t)(px[i] >> 16)) - 128;
    } else {
        qsel = 1;
        const int32_t dh = (H + 1) / 2;
        int32_t cy = (int32_t)my * 8 + (int32_t)r;
        if (cy > dh - 1) cy = dh - 1; // downsampled rows below the image replicate the last downsampled row
        uint32_t p0[16], p1[16]; // the two source rows of downsampled row cy, 16 pixels each
        load_bgr_run<16>(job.src, frames, (int32_t)mx * 16, 2 * cy, p0);
        load_bgr_run<16>(job.src, frames, (int32_t)mx * 16, 2 * cy + 1, p1);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t cx = (int32_t)mx * 8 + i;
            int32_t s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t q = (j >> 1) ? p1[2 * i + (j & 1)] : p0[2 * i + (j & 1)];
                const int32_t b = (int32_t)(q & 255u), g = (int32_t)((q >> 8) & 255u), rr = (int32_t)(q >> 16);
                s += k == 4 ? to_cb(b, g, rr) : to_cr(b, g, rr);
            }
            v[i] = ((s + ((cx & 1) ? 2 : 1)) >> 2) - 128;
        }
    }
    (void)W;
    fdct_1d<1>(v);
#pragma unroll
    for (int i = 0; i < 8; i++) s_w[lb][r * 8 + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = s_w[lb][i * 8 + r]; // column r
    fdct_1d<2>(v);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t dv = 8 * (int32_t)job.qt[qsel][i * 8 + r];
        int32_t x = v[i], q;
        if (x < 0) { x = -x + (dv >> 1); q = -(x / dv); } else { x += dv >> 1; q = x / dv; }
        s_q[lb][i * 8 + r] = (int16_t)(dummy ? 0 : q);
    }
    __syncthreads();
    if (active) {
        int16_t o[8];
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = s_q[lb][c_zigzag[r * 8 + i]];
        uint4 pk;
        pk.x = (uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16);
        pk.y = (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16);
        pk.z = (uint16_t)o[4] | ((uint32_t)(uint16_t)o[5] << 16);
        pk.w = (uint16_t)o[6] | ((uint32_t)(uint16_t)o[7] << 16);
        *reinterpret_cast<uint4*>(coef_arena + job.coef_off + (size_t)blk * 64 + r * 8) = pk;
    }
}

// jccoefct.c compress_data: a dummy luma block carries the DC of the previous block in the MCU buffer
// (bottom row: of the last block of the row above). Returns the block whose DC this one carries.
__device__ __forceinline__ uint32_t enc_dc_source(const LpEncJob& job, uint32_t blk)
{
    if (job.ncomp == 1) return blk;
    const uint32_t m = blk / job.bpm, k = blk - m * job.bpm;
    if (k >= 4) return blk;
    const uint32_t mx = m % job.mcus_x, my = m / job.mcus_x;
    uint32_t kk = k;
    for (int it = 0; it < 3; it++) {
        const uint32_t bx = mx * 2 + (kk & 1), by = my * 2 + (kk >> 1);
        if (by >= job.hib) kk = 1;           // bottom dummy row -> last block of row 0
        else if (bx >= job.wib) kk = kk - 1; // right-edge dummy -> block to the left
        else break;
    }
    return m * job.bpm + kk;
}

__device__ __forceinline__ uint32_t enc_prev_same_comp(const LpEncJob& job, uint32_t blk, bool& none)
{
    none = false;
    if (job.ncomp == 1) { none = blk == 0; return blk - 1; }
    const uint32_t m = blk / job.bpm, k = blk - m * job.bpm;
    if (k >= 1 && k <= 3) return blk - 1;
    if (m == 0) { none = true; return 0; }
    return k == 0 ? blk - 3 : blk - 6; // k==0: previous MCU's Y3 ((m-1)*6+3); chroma: same slot of the previous MCU
}

__device__ __forceinline__ uint32_t bit_length(uint32_t a) { return a ? 32 - __clz(a) : 0; }

// Walks one block's symbols. EMIT=false: returns the bit count. EMIT=true: appends the bits to the
// big-endian word stream starting at bit offset `bitpos` (edge words via atomicOr on a zeroed buffer).
// cf = the block's 64 coefficients (the kernels stage their 256 blocks in LDS, enc_stage_blocks: a thread walking its block through
// 2-byte global loads 128 bytes apart from its neighbours' made k_enc_bitlen + k_enc_emit 2.6 of the encoder's 3.8 us per image).
template <bool EMIT>
__device__ __forceinline__ uint32_t enc_block(const LpEncJob& job, const int16_t* __restrict__ coef_arena, const int16_t* cf, const LpEncTables* tb, uint32_t blk,
                                              uint32_t* __restrict__ words, uint64_t bitpos, uint32_t pad_to_byte)
{
    bool none;
    const uint32_t pv = enc_prev_same_comp(job, blk, none);
    const int32_t dc = coef_arena[job.coef_off + (size_t)enc_dc_source(job, blk) * 64];
    const int32_t pdc = none ? 0 : coef_arena[job.coef_off + (size_t)enc_dc_source(job, pv) * 64];
    const uint32_t chroma = (job.ncomp == 3 && (blk % job.bpm) >= 4) ? 2 : 0;
    const uint16_t* dcc = tb->code[chroma];
    const uint8_t* dcl = tb->len[chroma];
    const uint16_t* acc = tb->code[chroma + 1];
    const uint8_t* acl = tb->len[chroma + 1];
    uint32_t nbits = 0;
    // emit state
    uint64_t acc64 = 0;     // pending bits, right aligned
    uint32_t accn = 0;
    uint64_t wpos = bitpos >> 5;
    uint32_t first = 1;
    const uint32_t lead = (uint32_t)(bitpos & 31);
    if (EMIT) { accn = lead; } // leading bits of the first word belong to the previous block (zeros here, OR-ed in)
    auto put = [&](uint32_t code, uint32_t n) {
        if (!EMIT) { nbits += n; return; }
        acc64 = (acc64 << n) | (code & ((1u << n) - 1u));
        accn += n;
        if (accn >= 32) {
            uint32_t w = (uint32_t)(acc64 >> (accn - 32));
            if (first) { atomicOr(&words[wpos], w); first = 0; } else words[wpos] = w;
            wpos++;
            accn -= 32;
            acc64 &= (accn ? ((1ull << accn) - 1ull) : 0ull);
        }
    };
    {
        const int32_t diff = dc - pdc;
        const uint32_t a = (uint32_t)(diff < 0 ? -diff : diff), s = bit_length(a);
        put(dcc[s], dcl[s]);
        if (s) put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
    }
    uint32_t run = 0;
    const bool is_dummy = enc_dc_source(job, blk) != blk;
#pragma unroll 8
    for (uint32_t z = 1; z < 64; z++) {
        const int32_t v = is_dummy ? 0 : cf[z];
        if (v == 0) { run++; continue; }
        while (run > 15) { put(acc[0xF0], acl[0xF0]); run -= 16; }
        const uint32_t a = (uint32_t)(v < 0 ? -v : v), s = bit_length(a);
        const uint32_t sym = (run << 4) | s;
        put(acc[sym], acl[sym]);
        put((uint32_t)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run) put(acc[0], acl[0]);
    if (EMIT) {
        if (pad_to_byte) { // last block: fill the final byte with 1-bits (jchuff.c flush_bits)
            const uint32_t used = (uint32_t)((bitpos + 0) & 7); (void)used;
            const uint32_t rem = (8 - (accn & 7)) & 7;
            if (rem) put((1u << rem) - 1u, rem);
        }
        if (accn) {
            uint32_t w = (uint32_t)(acc64 << (32 - accn));
            atomicOr(&words[wpos], w);
        }
    }
    return nbits;
}

// The 256 consecutive blocks of a workgroup (32 KB, contiguous in the coefficient arena) -> LDS with coalesced 16-byte loads; a block's
// 64 coefficients sit 66 int16 apart (33 dwords: the 64 lanes of a wave reading coefficient z of their blocks hit 64 different banks).
#define LP_ENC_BLK_PITCH 66
__device__ __forceinline__ void enc_stage_tables_and_blocks(const LpEncJob& job, const int16_t* __restrict__ coef_arena, LpEncTables* s_tb, int16_t* s_c)
{
    {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&g_enc_tables);
        uint32_t* d = reinterpret_cast<uint32_t*>(s_tb);
        for (uint32_t i = threadIdx.x; i < sizeof(LpEncTables) / 4; i += 256) d[i] = s[i];
    }
    const uint32_t blk0 = blockIdx.x * 256, nblk = job.total_blocks - blk0 < 256u ? job.total_blocks - blk0 : 256u;
    const uint4* src = reinterpret_cast<const uint4*>(coef_arena + job.coef_off + (size_t)blk0 * 64); // k_enc_fdct stores the same uint4s
    for (uint32_t q = threadIdx.x; q < nblk * 8; q += 256) {
        const uint4 v = src[q];
        uint32_t* d = reinterpret_cast<uint32_t*>(s_c + (q >> 3) * LP_ENC_BLK_PITCH + (q & 7u) * 8);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_enc_bitlen(const LpEncJob* __restrict__ jobs, const int16_t* __restrict__ coef_arena, uint32_t* __restrict__ blk_bits)
{
    __shared__ LpEncTables s_tb;
    __shared__ __attribute__((aligned(16))) int16_t s_c[256 * LP_ENC_BLK_PITCH];
    const LpEncJob& job = jobs[blockIdx.y];
    if (blockIdx.x * 256 >= job.total_blocks) return;
    enc_stage_tables_and_blocks(job, coef_arena, &s_tb, s_c);
    const uint32_t blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= job.total_blocks) return;
    blk_bits[job.blk_off + blk] = enc_block<false>(job, coef_arena, s_c + threadIdx.x * LP_ENC_BLK_PITCH, &s_tb, blk, nullptr, 0, 0);
}

// One workgroup per image: exclusive scan of the block bit lengths (in place) -> total_bits.
__global__ __launch_bounds__(256) void k_enc_scan(const LpEncJob* __restrict__ jobs, LpEncState* __restrict__ states, uint32_t* __restrict__ blk_bits)
{
    __shared__ uint32_t s_p[256];
    const LpEncJob& job = jobs[blockIdx.x];
    uint32_t* bb = blk_bits + job.blk_off;
    const uint32_t n = job.total_blocks, t = threadIdx.x, per = (n + 255) / 256;
    uint32_t b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    if (b0 > n) b0 = n;
    uint32_t s = 0;
    for (uint32_t i = b0; i < b1; i++) s += bb[i];
    s_p[t] = s;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (uint32_t i = 0; i < 256; i++) { uint32_t x = s_p[i]; if (i < t) off += x; tot += x; }
    for (uint32_t i = b0; i < b1; i++) { uint32_t x = bb[i]; bb[i] = off; off += x; }
    if (t == 0) { states[blockIdx.x].total_bits = tot; states[blockIdx.x].error = 0; states[blockIdx.x].out_len = 0; }
}

__global__ __launch_bounds__(256) void k_enc_emit(const LpEncJob* __restrict__ jobs, const LpEncState* __restrict__ states,
                                                  const int16_t* __restrict__ coef_arena, const uint32_t* __restrict__ blk_bits, uint32_t* __restrict__ bits_arena)
{
    __shared__ LpEncTables s_tb;
    __shared__ __attribute__((aligned(16))) int16_t s_c[256 * LP_ENC_BLK_PITCH];
    const LpEncJob& job = jobs[blockIdx.y];
    if (blockIdx.x * 256 >= job.total_blocks) return;
    enc_stage_tables_and_blocks(job, coef_arena, &s_tb, s_c);
    const uint32_t blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= job.total_blocks) return;
    const uint32_t total = states[blockIdx.y].total_bits;
    if (((uint64_t)total + 63) / 32 > job.bits_cap_words) return; // reported by k_enc_finish
    enc_block<true>(job, coef_arena, s_c + threadIdx.x * LP_ENC_BLK_PITCH, &s_tb, blk, bits_arena + job.bits_off, blk_bits[job.blk_off + blk], blk + 1 == job.total_blocks);
}

// One workgroup per image: header + byte-stuffed entropy-coded segment + EOI -> output buffer.
__global__ __launch_bounds__(256) void k_enc_finish(const LpEncJob* __restrict__ jobs, LpEncState* __restrict__ states, const uint32_t* __restrict__ bits_arena,
                                                    const uint8_t* __restrict__ hdrs, uint8_t* __restrict__ out_arena)
{
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_carry;
    const LpEncJob& job = jobs[blockIdx.x];
    LpEncState& st = states[blockIdx.x];
    const uint32_t t = threadIdx.x;
    const uint32_t nbytes = (st.total_bits + 7) / 8;
    uint8_t* out = out_arena + job.out_off;
    const bool cap_ok = ((uint64_t)st.total_bits + 63) / 32 <= job.bits_cap_words;
    if (!cap_ok || job.hdr_len + 2 > job.out_cap) {
        if (t == 0) { st.error = 1; st.out_len = 0; }
        return;
    }
    for (uint32_t i = t; i < job.hdr_len; i += 256) out[i] = hdrs[job.hdr_off + i];
    if (t == 0) s_carry = job.hdr_len;
    __syncthreads();
    const uint8_t* src = reinterpret_cast<const uint8_t*>(bits_arena + job.bits_off);
    bool ovf = false;
    for (uint32_t base = 0; base < nbytes; base += 256 * 16) {
        // each thread takes 16 consecutive bytes
        const uint32_t p0 = base + t * 16;
        uint32_t nff = 0, cnt = 0;
        uint8_t b[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t q = p0 + j;
            b[j] = q < nbytes ? src[q ^ 3] : 0;
            if (q < nbytes) { cnt++; nff += b[j] == 0xFF; }
        }
        uint32_t mine = cnt + nff;
        // block exclusive scan of `mine`
        uint32_t incl = mine;
        const uint32_t lane = t & 63, wv = t >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (uint32_t w = 0; w < 4; w++) { if (w < wv) woff += s_wsum[w]; tot += s_wsum[w]; }
        uint32_t o = s_carry + woff + incl - mine;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (p0 + j < nbytes) {
                if (o < job.out_cap) out[o] = b[j]; else ovf = true;
                o++;
                if (b[j] == 0xFF) { if (o < job.out_cap) out[o] = 0; else ovf = true; o++; }
            }
        }
        __syncthreads();
        if (t == 0) s_carry += tot;
        __syncthreads();
    }
    const uint32_t end = s_carry;
    if (__syncthreads_or(ovf ? 1 : 0) || end + 2 > job.out_cap) {
        if (t == 0) { st.error = 1; st.out_len = 0; }
        return;
    }
    if (t == 0) { out[end] = 0xFF; out[end + 1] = 0xD9; st.out_len = end + 2; }
}

// Gather the encoded streams (spaced out_cap apart in the output arena) into the slots pk_off[i] .. pk_off[i + 1] of `packed` -- mapped
// pinned host memory: the results of a chunk leave the device by the kernel's own stores, not through the copy engine (which is
// busy with the ingest pipeline's H2D copies). One workgroup per image; both sides are 16-byte aligned.
__global__ __launch_bounds__(256) void k_enc_pack(const LpEncJob* __restrict__ jobs, const LpEncState* __restrict__ states,
                                                  const uint32_t* __restrict__ pk_off, const uint8_t* __restrict__ out_arena, uint8_t* __restrict__ packed)
{
    const LpEncJob& job = jobs[blockIdx.x];
    const uint32_t n = states[blockIdx.x].out_len;
    if (n > pk_off[blockIdx.x + 1] - pk_off[blockIdx.x]) return; // larger than its slot (the host fetches it from the output arena)
    const uint4* src = reinterpret_cast<const uint4*>(out_arena + job.out_off);
    uint4* dst = reinterpret_cast<uint4*>(packed + pk_off[blockIdx.x]);
    for (uint32_t i = threadIdx.x; i < (n + 15) / 16; i += 256) dst[i] = src[i];
}

void lp_launch_enc_pack(hipStream_t s, const LpEncJob* d_jobs, const LpEncState* d_states, uint32_t nimg, const uint32_t* d_pk_off, const uint8_t* d_out,
                        uint8_t* d_packed)
{
    if (!nimg) return;
    hipLaunchKernelGGL(k_enc_pack, dim3(nimg), dim3(256), 0, s, d_jobs, d_states, d_pk_off, d_out, d_packed);
}

void lp_launch_enc_fdct(hipStream_t s, const LpEncJob* d_jobs, uint32_t nimg, uint32_t max_blocks, int16_t* d_coef)
{
    if (!nimg || !max_blocks) return;
    hipLaunchKernelGGL(k_enc_fdct, dim3((max_blocks + 31) / 32, nimg), dim3(256), 0, s, d_jobs, (const uint8_t*)nullptr, d_coef);
}

void lp_launch_encode(hipStream_t s, const LpEncJob* d_jobs, LpEncState* d_states, uint32_t nimg, uint32_t max_blocks, const uint8_t* d_frames,
                      int16_t* d_coef, uint32_t* d_blk_bits, uint32_t* d_bits, const uint8_t* d_hdrs, uint8_t* d_out)
{
    if (!nimg || !max_blocks) return;
    hipLaunchKernelGGL(k_enc_fdct, dim3((max_blocks + 31) / 32, nimg), dim3(256), 0, s, d_jobs, d_frames, d_coef);
    hipLaunchKernelGGL(k_enc_bitlen, dim3((max_blocks + 255) / 256, nimg), dim3(256), 0, s, d_jobs, (const int16_t*)d_coef, d_blk_bits);
    hipLaunchKernelGGL(k_enc_scan, dim3(nimg), dim3(256), 0, s, d_jobs, d_states, d_blk_bits);
    hipLaunchKernelGGL(k_enc_emit, dim3((max_blocks + 255) / 256, nimg), dim3(256), 0, s, d_jobs, (const LpEncState*)d_states, (const int16_t*)d_coef,
                       (const uint32_t*)d_blk_bits, d_bits);
    hipLaunchKernelGGL(k_enc_finish, dim3(nimg), dim3(256), 0, s, d_jobs, d_states, (const uint32_t*)d_bits, d_hdrs, d_out);
}
