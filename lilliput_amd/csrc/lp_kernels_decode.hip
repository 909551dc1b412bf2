// lp_kernels_decode.hip -- gfx950 kernels for the JPEG decode half of ImageOps.Transform:
//   unstuff (FF00 / RSTn removal + restart boundary list)  -> k_unstuff_{count,scan,scatter}
//   Huffman entropy decode (S1)                             -> k_huff_spec, k_huff_verify, k_sub_scan, k_huff_write
//   dequantise + 8x8 islow IDCT (S2)                        -> k_idct
// Replaces the libjpeg-turbo work behind opencv_decoder_read_data (/root/reference/opencv.cpp:166-171).
// All kernels are batched: blockIdx.y (or .z) selects the image, per-image descriptors live in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lp_huff_core.h"
#include "lp_prog_core.h"
#include "lp_unstuff_core.h"
#include "lp_launch.h"
#include "lp_types.h"

#define UNSTUFF_T 256
#define UNSTUFF_CHUNK 4096 // bytes per chunk (the unit of the count / scan / scatter bookkeeping), 16 per thread
#ifndef LP_UNSTUFF_CPW
#define LP_UNSTUFF_CPW 2   // chunks a workgroup of the count / scatter kernels walks: all of them are requested before the first is looked at, and
                           // the barriers in between wait for LDS only (lds_barrier) -- 3.85 -> 3.25 us per 4096 x 4096 image with 2 / 3 / 4, 3.7 with 8;
                           // walking them one after the other behind __syncthreads() gained nothing (profiles/r06_write_stream.md section 3)
#endif

// ------------------------------------------------------------------------------------------------
// block-wide helpers (256 threads = 4 waves of 64)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// A workgroup barrier for data exchanged through LDS only: __syncthreads() also waits for the wave's outstanding global loads (vmcnt(0) on
// gfx9), which is exactly what the unstuff kernels' prefetch of the next chunks must not do.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive scan of a pair of small counts (a <= 16, b <= 8 per thread: the block totals fit 16 bits each), packed into one word so
// that the wave scan's cross-lane steps are paid once
template <bool LDS_ONLY = false>
__device__ __forceinline__ void block_excl_scan2(uint32_t a, uint32_t b, uint32_t& ea, uint32_t& eb, uint32_t& ta, uint32_t& tb,
                                                 uint32_t* s_tmp /* >= 4 words */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t v = a | (b << 16);
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) s_tmp[wv] = inc;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    uint32_t o = 0, t = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = s_tmp[w];
        if (w < wv) o += x;
        t += x;
    }
    const uint32_t e = o + inc - v;
    ea = e & 0xffffu; eb = e >> 16;
    ta = t & 0xffffu; tb = t >> 16;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Unstuffing. Byte classes inside the entropy-coded segment (T.81 B.1.1.5, F.1.2.3; libjpeg jdhuff.c
// jpeg_fill_bit_buffer): data byte; FF followed by 00 = data FF (00 dropped); FF FF.. = fill; FF Dn = RSTn.
struct UnstuffBytes {
    uint32_t w[4];      // 16 raw bytes
    uint32_t prev, next; // neighbours
};

// The same in two halves, for kernels that request several chunks before they look at the first (LP_UNSTUFF_CPW > 1): the loads ...
struct UnstuffRaw { uint4 v; uint32_t ep, en; };
__device__ __forceinline__ void unstuff_request(const uint8_t* raw, uint32_t raw_len, uint32_t pos0, UnstuffRaw& r)
{
    r.v = *reinterpret_cast<const uint4*>(raw + pos0);
    const uint32_t lane = threadIdx.x & 63u;
    r.ep = 0u; r.en = 0xD9u;
    if (lane == 0u && pos0) r.ep = raw[pos0 - 1u];
    if (lane == 63u && pos0 + 16u < raw_len) r.en = raw[pos0 + 16u];
}
// ... and the neighbours' edge bytes by lane shuffle, once the data is needed
__device__ __forceinline__ void unstuff_finish(const UnstuffRaw& r, UnstuffBytes& u)
{
    u.w[0] = r.v.x; u.w[1] = r.v.y; u.w[2] = r.v.z; u.w[3] = r.v.w;
    const uint32_t lane = threadIdx.x & 63u;
    u.prev = __shfl_up(r.v.w >> 24, 1, 64);
    u.next = __shfl_down(r.v.x & 0xFFu, 1, 64);
    if (lane == 0u) u.prev = r.ep;
    if (lane == 63u) u.next = r.en;
}

__device__ __forceinline__ void unstuff_load(const uint8_t* raw, uint32_t raw_len, uint32_t pos0, UnstuffBytes& u)
{
    // raw is 16-byte aligned and the arena is padded, so the vector load is always in bounds.
    const uint4 v = *reinterpret_cast<const uint4*>(raw + pos0);
    u.w[0] = v.x; u.w[1] = v.y; u.w[2] = v.z; u.w[3] = v.w;
    // the neighbours' edge bytes travel by lane shuffle; only the two lanes at the ends of a wave read theirs from memory (the same
    // cache lines the wave is loading anyway). The first version exchanged them through LDS: two barriers per workgroup.
    const uint32_t lane = threadIdx.x & 63u;
    u.prev = __shfl_up(v.w >> 24, 1, 64);
    u.next = __shfl_down(v.x & 0xFFu, 1, 64);
    if (lane == 0u) u.prev = pos0 ? raw[pos0 - 1u] : 0u;
    if (lane == 63u) u.next = pos0 + 16u < raw_len ? raw[pos0 + 16u] : 0xD9u; // past the segment: never looked at (see lp_unstuff_classify)
}

__global__ __launch_bounds__(UNSTUFF_T) void k_unstuff_count(const LpJpeg* __restrict__ imgs, const uint8_t* __restrict__ raw_arena,
                                                             uint2* __restrict__ chunk_cnt, LpJpegState* __restrict__ states)
{
    __shared__ uint32_t s_tmp[4];
    const LpJpeg& img = imgs[blockIdx.y];
    const uint8_t* raw = raw_arena + img.raw_off;
    const uint32_t raw_end = img.raw_skip + img.raw_len; // positions count from raw_off; the first raw_skip bytes are not the segment's
    UnstuffRaw rq[LP_UNSTUFF_CPW];
#pragma unroll
    for (uint32_t c = 0; c < LP_UNSTUFF_CPW; c++) { // every chunk of the workgroup is requested before the first is looked at
        const uint32_t chunk = blockIdx.x * LP_UNSTUFF_CPW + c;
        if (chunk < img.nchunks) unstuff_request(raw, raw_end, chunk * UNSTUFF_CHUNK + threadIdx.x * 16, rq[c]);
    }
#pragma unroll
    for (uint32_t c = 0; c < LP_UNSTUFF_CPW; c++) {
        const uint32_t chunk = blockIdx.x * LP_UNSTUFF_CPW + c;
        if (chunk >= img.nchunks) break; // workgroup-uniform
        uint32_t pos0 = chunk * UNSTUFF_CHUNK + threadIdx.x * 16;
        UnstuffBytes u;
        unstuff_finish(rq[c], u);
        uint32_t K[4], R[4], err = 0;
        if (chunk == 0 && img.raw_skip) lp_unstuff_classify_masks<true, true>(u.w, u.prev, u.next, pos0, raw_end, K, R, err, threadIdx.x == 0 ? img.raw_skip : 0u); // workgroup-uniform
        else if ((chunk + 1u) * UNSTUFF_CHUNK < raw_end) lp_unstuff_classify_masks<false>(u.w, u.prev, u.next, pos0, raw_end, K, R, err);
        else lp_unstuff_classify_masks<true>(u.w, u.prev, u.next, pos0, raw_end, K, R, err);
        uint32_t ea, eb, ta, tb;
        block_excl_scan2<(LP_UNSTUFF_CPW > 1)>(__popc(K[0]) + __popc(K[1]) + __popc(K[2]) + __popc(K[3]), __popc(R[0] | R[1] >> 1 | R[2] >> 2 | R[3] >> 3), ea, eb, ta, tb, s_tmp);
        if (threadIdx.x == 0) chunk_cnt[img.chunk_off + chunk] = make_uint2(ta, tb);
        if (err) atomicOr(&states[blockIdx.y].error, err);
    }
}

// One workgroup per image: exclusive scan of the chunk counts (in place), totals -> state.
__global__ __launch_bounds__(256) void k_unstuff_scan(const LpJpeg* __restrict__ imgs, uint2* __restrict__ chunk_cnt,
                                                      LpJpegState* __restrict__ states, uint32_t* __restrict__ clean_arena)
{
    __shared__ uint32_t s_a[256], s_b[256];
    const LpJpeg& img = imgs[blockIdx.x];
    uint2* cc = chunk_cnt + img.chunk_off;
    const uint32_t n = img.nchunks, t = threadIdx.x;
    const uint32_t per = (n + 255) / 256;
    uint32_t b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    uint32_t sa = 0, sb = 0;
    for (uint32_t i = b0; i < b1; i++) { uint2 v = cc[i]; sa += v.x; sb += v.y; }
    s_a[t] = sa; s_b[t] = sb;
    __syncthreads();
    uint32_t oa = 0, ob = 0, ta = 0, tb = 0;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t xa = s_a[i], xb = s_b[i];
        if (i < t) { oa += xa; ob += xb; }
        ta += xa; tb += xb;
    }
    for (uint32_t i = b0; i < b1; i++) { uint2 v = cc[i]; cc[i] = make_uint2(oa, ob); oa += v.x; ob += v.y; }
    if (t == 0) {
        LpJpegState& st = states[blockIdx.x];
        st.clean_bytes = ta;
        st.n_rst = tb < img.rst_cap ? tb : img.rst_cap;
        if (tb > img.rst_cap) st.error |= 4u;
        // A baseline image (not a scan's pseudo stream) must hold exactly the restart markers its MCU count asks for; anything else --
        // one missing, one too many, any at all without a DRI -- is decoded by the serial route, which does what libjpeg does with it
        // (jdmarker.c jpeg_resync_to_restart, lp_jbits.h)
        if (img.total_blocks && !img.scan_path && tb != (img.dri ? (img.mcus_x * img.mcus_y + img.dri - 1u) / img.dri - 1u : 0u)) st.error |= 8u;
        uint64_t bits = (uint64_t)ta * 8;
        st.nsub = (uint32_t)((bits + img.sub_bits - 1) / img.sub_bits);
    }
    // zero the tail words so that reads past the end of the stream are deterministic
    if (t < 16) {
        uint32_t w = (ta >> 2) + t;
        if (w < img.clean_cap_words) clean_arena[img.clean_off + w] = 0;
    }
}

// Scatter pass. A workgroup's 4 KiB of raw bytes become <= 4 KiB of CONTIGUOUS clean bytes, so they are compacted in LDS
// (already in memory byte order: the clean stream is stored as big-endian 32-bit words, bit 31 of word 0 = first bit) and
// leave as coalesced dword stores; only the <= 3 bytes of a word shared with the neighbouring chunk use byte stores.
__global__ __launch_bounds__(UNSTUFF_T) void k_unstuff_scatter(const LpJpeg* __restrict__ imgs, const uint8_t* __restrict__ raw_arena,
                                                               const uint2* __restrict__ chunk_cnt, uint32_t* __restrict__ clean_arena,
                                                               uint32_t* __restrict__ rst_bits, LpJpegState* __restrict__ states)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[UNSTUFF_CHUNK + 16 + UNSTUFF_T]; // + one byte per lane where dropped bytes go
    const LpJpeg& img = imgs[blockIdx.y];
    const uint8_t* raw = raw_arena + img.raw_off;
    const uint32_t raw_end = img.raw_skip + img.raw_len; // positions count from raw_off; the first raw_skip bytes are not the segment's
    UnstuffRaw rq[LP_UNSTUFF_CPW];
    uint2 bases[LP_UNSTUFF_CPW];
#pragma unroll
    for (uint32_t c = 0; c < LP_UNSTUFF_CPW; c++) { // every chunk of the workgroup is requested before the first is looked at
        const uint32_t chunk = blockIdx.x * LP_UNSTUFF_CPW + c;
        if (chunk < img.nchunks) { unstuff_request(raw, raw_end, chunk * UNSTUFF_CHUNK + threadIdx.x * 16, rq[c]); bases[c] = chunk_cnt[img.chunk_off + chunk]; }
    }
#pragma unroll
    for (uint32_t c = 0; c < LP_UNSTUFF_CPW; c++) {
    const uint32_t chunk = blockIdx.x * LP_UNSTUFF_CPW + c;
    if (chunk >= img.nchunks) break; // workgroup-uniform
    if (c) { if (LP_UNSTUFF_CPW > 1) lds_barrier(); else __syncthreads(); } // s_out changes hands
    uint32_t pos0 = chunk * UNSTUFF_CHUNK + threadIdx.x * 16;
    UnstuffBytes u;
    unstuff_finish(rq[c], u);
    uint32_t K[4], R[4], err = 0;
    if (chunk == 0 && img.raw_skip) lp_unstuff_classify_masks<true, true>(u.w, u.prev, u.next, pos0, raw_end, K, R, err, threadIdx.x == 0 ? img.raw_skip : 0u); // workgroup-uniform
    else if ((chunk + 1u) * UNSTUFF_CHUNK < raw_end) lp_unstuff_classify_masks<false>(u.w, u.prev, u.next, pos0, raw_end, K, R, err);
    else lp_unstuff_classify_masks<true>(u.w, u.prev, u.next, pos0, raw_end, K, R, err);
    const uint32_t rany = R[0] | R[1] >> 1 | R[2] >> 2 | R[3] >> 3; // disjoint bit positions: one popcount for the four words
    uint32_t ea, eb, ta, tb;
    block_excl_scan2<(LP_UNSTUFF_CPW > 1)>(__popc(K[0]) + __popc(K[1]) + __popc(K[2]) + __popc(K[3]), __popc(rany), ea, eb, ta, tb, s_tmp);
    const uint2 base = bases[c];
    const uint32_t a0 = base.x & ~3u;           // clean position of the first (maybe shared) word
    uint32_t cpos = base.x + ea;
    if (rany) { // restart markers: a handful per image
        uint32_t rpos = base.y + eb, c = cpos;
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t bit = 0x80u << (8u * (j & 3u));
            if (R[j >> 2] & bit) {
                if (rpos < img.rst_cap) rst_bits[img.rst_off + rpos] = c * 8;
                // the k-th marker must be RST(k mod 8): libjpeg treats any other number as a lost or repeated interval (lp_jbits.h)
                if (((u.w[j >> 2] >> (8u * (j & 3u))) & 7u) != (rpos & 7u) && img.total_blocks && !img.scan_path) atomicOr(&states[blockIdx.y].error, 8u);
                rpos++;
            }
            c += (K[j >> 2] & bit) ? 1u : 0u;
        }
    }
    // Compaction, branch-free: byte j goes to its clean position (stream order; the words are byte-swapped on the way out) or, when
    // it is dropped, to the lane's dump byte -- no exec-mask pair per byte (the first version spent 2 x 16 of them per lane).
    uint32_t l = cpos - a0;
    const uint32_t dump = UNSTUFF_CHUNK + 16 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t kept = (K[j >> 2] >> (8 * (j & 3) + 7)) & 1u;
        s_out[kept ? l : dump] = (uint8_t)(u.w[j >> 2] >> (8 * (j & 3)));
        l += kept;
    }
    if (LP_UNSTUFF_CPW > 1) lds_barrier(); else __syncthreads();
    const uint32_t lo = base.x - a0, hi = lo + ta;  // owned clean positions relative to a0: [lo, hi)
    const uint32_t cap = img.clean_cap_words;
    uint32_t* out = clean_arena + img.clean_off + (a0 >> 2);
    const uint32_t nwords = (hi + 3) >> 2;
    for (uint32_t w = threadIdx.x; w < nwords; w += UNSTUFF_T) {
        if ((a0 >> 2) + w >= cap) break;
        const uint32_t q0 = w * 4;
        if (q0 >= lo && q0 + 4 <= hi) {
            out[w] = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(&s_out[q0])); // the clean stream is big-endian words: bit 31 = first bit
        } else { // word shared with a neighbouring chunk: touch only the owned bytes
            uint8_t* ob = reinterpret_cast<uint8_t*>(out + w);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t q = q0 + k;
                if (q >= lo && q < hi) ob[3u - k] = s_out[q];
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Huffman decode. Workgroup = 256 lanes = 256 consecutive subsequences of one image (blockIdx.y).
// LDS: the image's LpHuffSet (two-level code tables), one bit-reader ring per lane (word-interleaved: word j of lane l at
// [j][l], so every wave access hits 64 consecutive dwords -- conflict-free whatever the lanes' positions are).
#define HUFF_T 256
// R = ring words per lane.
template <int R, int T, int Q>
struct DevMem {
    static constexpr int kRing = R, kEvery = T, kQuads = Q, kRows = R + 1; // row R mirrors row 0, so that "slot s and slot s + 1" never wraps
    __amdgpu_buffer_rsrc_t words; // this image's clean stream as a raw buffer: a 32-bit byte offset per lane instead of 64-bit address
                                  // arithmetic in the loop, and reads past the image's region return 0
    uint32_t* ring;         // LDS, already offset by the lane: word w of the stream lives at ring[((3 - w) % R) * 64] -- descending, so that
                            // the slot a decode step asks for is a bit field of the lane's negated position (LpLane::np, fetch_np)
    uint32_t fbits;         // next stream word to load, as a BIT position (a multiple of 128): the top-up test compares it with the
                            // lane's bit position directly
    const uint16_t* l1;     // LDS: LpHuffSet::lut (the WRITE pass; the counting passes of an LP_MULTI == 0 build)
    const uint16_t* l2;     // LDS: LpHuffSet::lut2
    const uint32_t* lc;     // LDS: lut | lutm << 16, the counting passes' combined first level (stage_huff_count)
    const LpHuffSet* hsg;   // HBM: the whole set; the canonical tables are read on damaged streams only
    const uint32_t* rst;
    uint4 pend[Q];          // see reseek / topup
    static __device__ __forceinline__ DevMem make(const uint32_t* stream, uint32_t cap_words, uint32_t* ring_, const uint16_t* l1_, const uint16_t* l2_,
                                                  const uint32_t* lc_, const LpHuffSet* hsg_, const uint32_t* rst_)
    {
        DevMem m;
        m.words = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(stream), 0, (int)(cap_words * 4u), 0x00020000);
        m.ring = ring_; m.fbits = 0; m.l1 = l1_; m.l2 = l2_; m.lc = lc_; m.hsg = hsg_; m.rst = rst_;
        return m;
    }
    __device__ __forceinline__ uint32_t fetch1(uint32_t w) const { return ring[((3u - w) & (R - 1u)) << 6]; }
    __device__ __forceinline__ uint32_t peek_np(uint32_t np) const
    {
        uint32_t slot; // v_bfe + v_lshl_add; written in C the compiler re-associates it into shift, mask and add
        if (R == 8) asm("v_bfe_u32 %0, %1, 5, 3" : "=v"(slot) : "v"(np));
        else asm("v_bfe_u32 %0, %1, 5, 4" : "=v"(slot) : "v"(np));
        const uint32_t* r = ring + (slot << 6);
        return __builtin_amdgcn_alignbit(r[64], r[0], np); // words ceil(p / 32) - 1 (one slot up) and ceil(p / 32): one ds_read2st64_b32
    }
    // The next Q quads of the stream travel in registers: a top-up stores what the previous top-up loaded and issues the loads for the
    // one after, so no wave ever sits in s_waitcnt vmcnt(0) behind an HBM round trip (the first version loaded and stored in the same
    // top-up: PMC showed SPEC at 40 % and WRITE at 20 % of the VALU issue rate).
    __device__ __forceinline__ uint4 load_quad(uint32_t bits) const
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifdef LP_EXP_HOT   // timing build (wrong pixels): every lane reads inside the first 256 KB of its image's stream -- what the walks would cost if no top-up ever missed the L2
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(words, (int)((bits >> 3) & 0x3fff0u), 0, 0);
#else
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(words, (int)(bits >> 3), 0, 0);
#endif
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ void store_quad(const uint4& v)
    {
        uint32_t* r = ring + (((0u - (fbits >> 5)) & (R - 1u)) << 6); // words 4k .. 4k + 3 -> slots s + 3 .. s, s = (-4k) mod R: the quad never wraps
        r[192] = v.x;
        r[128] = v.y;
        r[64] = v.z;
        r[0] = v.w;
        if (r == ring) ring[R * 64] = v.w; // the mirror of slot 0
        fbits += 128u;
    }
    __device__ __forceinline__ void reseek(uint32_t w)
    {
        fbits = (w & ~3u) << 5;
#pragma unroll
        for (int i = 0; i < R / 4; i++) store_quad(load_quad(fbits));
#pragma unroll
        for (int i = 0; i < Q; i++) pend[i] = load_quad(fbits + 128u * i);
    }
    // p = the lane's bit position. "fill + 4 <= (p >> 5) + R" in words is "fbits <= p + 32 R - 128" in bits (both sides of the word
    // form are multiples of 32 bits).
    __device__ __forceinline__ void topup(uint32_t p)
    {
        if (fbits <= p + (32u * R - 128u)) {
            store_quad(pend[0]);
            if (Q == 2) {
                if (fbits <= p + (32u * R - 128u)) {
                    store_quad(pend[Q - 1]);
                    pend[0] = load_quad(fbits);
                } else
                    pend[0] = pend[Q - 1];
                pend[Q - 1] = load_quad(fbits + 128u);
            } else
                pend[0] = load_quad(fbits);
        }
    }
    // a value that came from a global load inside a rare branch: make the branch wait for it, so that the hot path carries no
    // s_waitcnt vmcnt for it (which would also wait for the prefetched quads, every iteration)
    __device__ __forceinline__ void settle(uint32_t& v) const { asm volatile("" : "+v"(v)); }
    // (a ballot compared with zero stays in scalar registers; __any() materialises the vote in a VGPR and compares it again: two VALU
    // instructions per vote, three votes per decode step)
    __device__ __forceinline__ bool any(bool p) const { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
    __device__ __forceinline__ bool any2(bool a, bool b) const { return (__builtin_amdgcn_ballot_w64(a) | __builtin_amdgcn_ballot_w64(b)) != 0ull; }
    __device__ __forceinline__ uint32_t lut(uint32_t t, uint32_t i) const { return l1[(t << LP_LUT_BITS) | i]; }
    __device__ __forceinline__ uint32_t lutc(uint32_t t, uint32_t i) const { return lc[(t << LP_LUT_BITS) | i]; }
    __device__ __forceinline__ uint32_t lut2(uint32_t i) const { return l2[i]; }
    __device__ __forceinline__ int32_t maxcode(uint32_t t, uint32_t l) const { return hsg->maxcode[t][l]; }
    __device__ __forceinline__ int32_t valoff(uint32_t t, uint32_t l) const { return hsg->valoff[t][l]; }
    __device__ __forceinline__ uint32_t val(uint32_t t, uint32_t i) const { return hsg->vals[t][i & 255]; }
    __device__ __forceinline__ uint32_t rst_bit(uint32_t k) const { return rst[k]; }
};
#ifndef LP_RING
#define LP_RING 8   // measured against 16 (top-up of two quads every 8 steps): 8 KB less LDS per workgroup, 3-6 % more images/s
#endif
#ifndef LP_EXP_PREFETCH_COUNT
#define LP_EXP_PREFETCH_COUNT 1   // quads in flight per lane ahead of the ring (1: the quad after next travels while the next one is consumed; 2: A/B build, profiles/r06_write_stream.md 6)
#endif
#ifndef LP_EXP_PREFETCH_WRITE
#define LP_EXP_PREFETCH_WRITE 1
#endif
#if LP_RING == 8
typedef DevMem<8, 2, LP_EXP_PREFETCH_COUNT> CountMem;   // SPEC / VERIFY: 8 words per lane, a top-up of one quad every 2 steps (2 + 3 + 3 <= 8)
typedef DevMem<8, 2, LP_EXP_PREFETCH_WRITE> WriteMemSel;
#else
typedef DevMem<16, 8, 2> CountMem;  // SPEC / VERIFY
typedef DevMem<16, 8, 2> WriteMemSel;
#endif
// WRITE: measured both ways -- DevMem<8, 4, 1> fits four workgroups per CU, DevMem<16, 8, 2> three with half the top-ups;
// three is 5 % faster alone and leaves LDS for the other parts' kernels when parts of a batch overlap.
typedef WriteMemSel WriteMem;

__device__ __forceinline__ void stage_huff(uint4* d, const LpHuffSet* src)
{
    const uint4* s = reinterpret_cast<const uint4*>(src);
    for (uint32_t i = threadIdx.x; i < LP_HUFF_LDS_BYTES / 16; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}
// The counting passes' tables: the second level as it is, the first level as one 32-bit entry per index -- the one-symbol entry in the
// lower half, the multi-symbol entry (LpHuffSet::lutm) in the upper: one LDS read serves both (LpLane::step).
#if LP_MULTI
#define LP_COUNT_LDS_BYTES (LP_LUT2_POOL * 2 + 4 * LP_LUT_SIZE * 4)
__device__ __forceinline__ void stage_huff_count(uint4* d, const LpHuffSet* src)
{
    const uint4* s2 = reinterpret_cast<const uint4*>(src->lut2);
    for (uint32_t i = threadIdx.x; i < LP_LUT2_POOL * 2 / 16; i += blockDim.x) d[i] = s2[i];
    const uint4* lo = reinterpret_cast<const uint4*>(src->lut);
    const uint4* hi = reinterpret_cast<const uint4*>(src->lutm);
    uint4* c = d + LP_LUT2_POOL * 2 / 16;
    for (uint32_t i = threadIdx.x; i < 4 * LP_LUT_SIZE / 8; i += blockDim.x) { // eight entries of each half -> eight combined entries
        const uint4 a = lo[i], b = hi[i];
        c[2 * i] = make_uint4((a.x & 0xffffu) | (b.x << 16), (a.x >> 16) | (b.x & 0xffff0000u), (a.y & 0xffffu) | (b.y << 16), (a.y >> 16) | (b.y & 0xffff0000u));
        c[2 * i + 1] = make_uint4((a.z & 0xffffu) | (b.z << 16), (a.z >> 16) | (b.z & 0xffff0000u), (a.w & 0xffffu) | (b.w << 16), (a.w >> 16) | (b.w & 0xffff0000u));
    }
    __syncthreads();
}
#define LP_COUNT_TABLES(s4) nullptr, reinterpret_cast<const uint16_t*>(s4), reinterpret_cast<const uint32_t*>(s4 + LP_LUT2_POOL * 2 / 16)
#else
#define LP_COUNT_LDS_BYTES LP_HUFF_LDS_BYTES
__device__ __forceinline__ void stage_huff_count(uint4* d, const LpHuffSet* src) { stage_huff(d, src); }
#define LP_COUNT_TABLES(s4) reinterpret_cast<const LpHuffSet*>(s4)->lut[0], reinterpret_cast<const LpHuffSet*>(s4)->lut2, nullptr
#endif

__device__ __forceinline__ LpImgCtx make_ctx(const LpJpeg& img, const LpJpegState& st)
{
    LpImgCtx ic;
    ic.blkpack = (uint32_t)img.blkpack;
    ic.bpm = img.bpm;
    ic.n_rst = st.n_rst;
    ic.total_bits = st.clean_bytes * 8;
    ic.total_blocks = img.total_blocks;
    ic.rst_blocks = img.dri * img.bpm;
    lp_ctx_tables(ic);
    return ic;
}

__device__ __forceinline__ LpSubState load_state(const LpSubState* p)
{
    const uint64_t raw = *reinterpret_cast<const volatile uint64_t*>(p); // one 8-byte access: never torn
    LpSubState s;
    s.p = (uint32_t)raw;
    s.bz = (uint32_t)(raw >> 32);
    return s;
}
__device__ __forceinline__ void store_state(LpSubState* p, const LpSubState& s)
{
    *reinterpret_cast<volatile uint64_t*>(p) = (uint64_t)s.p | ((uint64_t)s.bz << 32);
}

// Checkpoint records in HBM: [k][subsequence] so that the 64 lanes of a wave store / load 1 KiB rows.
struct DevCkSink {
    LpCkptPk* base;     // + g
    size_t stride;      // tot_sub
    bool valid;
    __device__ __forceinline__ void record(uint32_t k, const LpCkptPk& c)
    {
        if (valid) *reinterpret_cast<uint4*>(base + (size_t)k * stride) = make_uint4(c.p, c.bz, c.nblk, c.nreset);
    }
};

__global__ __launch_bounds__(HUFF_T) void k_huff_spec(const LpJpeg* __restrict__ imgs, const LpJpegState* __restrict__ states,
                                                      const LpHuffSet* __restrict__ huffs, const uint32_t* __restrict__ clean_arena,
                                                      const uint32_t* __restrict__ rst_bits, LpCkptPk* __restrict__ ckpts,
                                                      LpSubState* __restrict__ spec_exit, LpSubSum* __restrict__ spec_total,
                                                      LpSubState* __restrict__ cur_exit, LpSubSum* __restrict__ cur_total,
                                                      LpSubState* __restrict__ entry_used, LpCkSched cs, uint32_t tot_sub)
{
    typedef CountMem MEM;
    __shared__ uint4 s_hs4[LP_COUNT_LDS_BYTES / 16];
    __shared__ uint32_t s_ring[HUFF_T * MEM::kRows];
    const LpJpeg& img = imgs[blockIdx.y];
    const LpJpegState& st = states[blockIdx.y];
    const uint32_t nsub = st.nsub < img.sub_cap ? st.nsub : img.sub_cap;
    if (blockIdx.x * HUFF_T >= nsub) return;
    stage_huff_count(s_hs4, huffs + img.huff_idx);
    const uint32_t sub = blockIdx.x * HUFF_T + threadIdx.x;
    const bool valid = sub < nsub;
    const uint32_t g = img.sub_off + (valid ? sub : 0);
    const LpImgCtx ic = make_ctx(img, st);
    MEM m = MEM::make(clean_arena + img.clean_off, img.clean_cap_words, s_ring + (threadIdx.x >> 6) * (64 * MEM::kRows) + (threadIdx.x & 63), LP_COUNT_TABLES(s_hs4), huffs + img.huff_idx,
                      rst_bits + img.rst_off);
    LpSubState entry;
    const uint32_t S = img.sub_bits;
    entry.p = valid ? sub * S : 0;
    entry.bz = 0;
    uint32_t sub_end = valid ? entry.p + S : 0; // invalid lanes finish immediately but keep the wave-uniform calls company
    if (sub_end > ic.total_bits) sub_end = ic.total_bits;
    DevCkSink ck{ckpts + g, tot_sub, valid};
    LpSubState ex;
    LpSubSum tot;
    lp_spec_pass(m, ic, sub_end, entry, cs, ck, &ex, &tot);
    if (!valid) return;
    spec_exit[g] = ex;
    cur_exit[g] = ex;
    spec_total[g] = tot;
    cur_total[g] = tot;
    LpSubState none;
    none.p = 0xffffffffu; none.bz = 0xffffffffu;
    entry_used[g] = none;
}

// Checkpoint source of the verify pass: the K positions of the lane's subsequence are staged in LDS (word-interleaved
// like the ring); a whole record is fetched from HBM only when a position matches.
struct DevCkSrc {
    const uint16_t* pos_lds; // + lane; positions relative to the start of the subsequence, 0xffff = not recorded
    const LpCkptPk* base;    // + g
    size_t stride;
    uint32_t sub_begin;
    __device__ __forceinline__ uint32_t pos(uint32_t k) const
    {
        const uint32_t v = pos_lds[k << 6];
        return v == 0xffffu ? 0xffffffffu : sub_begin + v;
    }
    __device__ __forceinline__ LpCkptPk load(uint32_t k) const
    {
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)k * stride);
        LpCkptPk c;
        c.p = v.x; c.bz = v.y; c.nblk = v.z; c.nreset = v.w;
        return c;
    }
};

__global__ __launch_bounds__(HUFF_T) void k_huff_verify(const LpJpeg* __restrict__ imgs, const LpJpegState* __restrict__ states,
                                                        const LpHuffSet* __restrict__ huffs, const uint32_t* __restrict__ clean_arena,
                                                        const uint32_t* __restrict__ rst_bits, const LpCkptPk* __restrict__ ckpts,
                                                        const LpSubState* __restrict__ spec_exit, const LpSubSum* __restrict__ spec_total,
                                                        LpSubState* cur_exit, LpSubSum* __restrict__ cur_total, LpSubState* __restrict__ entry_used,
                                                        uint32_t* changed, uint32_t round, uint32_t K, uint32_t tot_sub)
{
    typedef CountMem MEM;
    __shared__ uint4 s_hs4[LP_COUNT_LDS_BYTES / 16];
    __shared__ uint32_t s_ring[HUFF_T * MEM::kRows];
    __shared__ uint16_t s_ckpos[HUFF_T * LP_MAX_CKPT];
    // changed[r] counts the exit states round r moved. The rounds of a decode are enqueued back to back without a host round
    // trip; a round that follows one which moved nothing has nothing to do (the host checks the last counter at the end).
    if (round && __hip_atomic_load(changed + round - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    changed += round;
    const LpJpeg& img = imgs[blockIdx.y];
    const LpJpegState& st = states[blockIdx.y];
    const uint32_t nsub = st.nsub < img.sub_cap ? st.nsub : img.sub_cap;
    if (blockIdx.x * HUFF_T >= nsub) return;
    const uint32_t sub = blockIdx.x * HUFF_T + threadIdx.x;
    // subsequence 0 starts at the true beginning (its SPEC result is exact); a lane whose entry state has not changed since its
    // last verification has nothing to do. A workgroup without work leaves before staging the tables (every later round).
    const uint32_t g = img.sub_off + (sub < nsub ? sub : 0);
    LpSubState entry;
    entry.p = 0; entry.bz = 0;
    bool need = sub < nsub && sub != 0;
    if (need) {
        entry = load_state(cur_exit + g - 1);
        need = !lp_state_eq(entry, entry_used[g]);
    }
    if (!__syncthreads_or(need ? 1 : 0)) return;
    stage_huff_count(s_hs4, huffs + img.huff_idx);
    if (!need) return;
    const LpImgCtx ic = make_ctx(img, st);
    MEM m = MEM::make(clean_arena + img.clean_off, img.clean_cap_words, s_ring + (threadIdx.x >> 6) * (64 * MEM::kRows) + (threadIdx.x & 63), LP_COUNT_TABLES(s_hs4), huffs + img.huff_idx,
                      rst_bits + img.rst_off);
    const uint32_t S = img.sub_bits;
    uint16_t* cp = s_ckpos + (threadIdx.x >> 6) * (64 * LP_MAX_CKPT) + (threadIdx.x & 63);
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t p = ckpts[(size_t)k * tot_sub + g].p; // recorded positions lie within 40 bits past the subsequence: 16 bits hold them
        cp[k << 6] = (uint16_t)(p == 0xffffffffu ? 0xffffu : p - sub * S);
    }
    DevCkSrc ck{cp, ckpts + g, tot_sub, sub * S};
    uint32_t sub_end = sub * S + S;
    if (sub_end > ic.total_bits) sub_end = ic.total_bits;
    const LpSubState old_exit = load_state(cur_exit + g);
    LpSubState ex = old_exit;
    LpSubSum tot;
    lp_verify_pass(m, ic, sub_end, entry, K, ck, spec_exit[g], spec_total[g], &ex, &tot);
    cur_total[g] = tot;
    entry_used[g] = entry;
    if (!lp_state_eq(ex, old_exit)) {
        store_state(cur_exit + g, ex);
        atomicAdd(changed, 1u);
    }
}

// One workgroup per image: exclusive scan (lp_sum_combine is associative, not commutative) of the
// per-subsequence sums -> prefixes[]; also validates the block count.
__global__ __launch_bounds__(256) void k_sub_scan(const LpJpeg* __restrict__ imgs, LpJpegState* __restrict__ states,
                                                  const LpSubSum* __restrict__ totals, LpSubSum* __restrict__ prefixes)
{
    __shared__ LpSubSum s_part[256];
    const LpJpeg& img = imgs[blockIdx.x];
    LpJpegState& st = states[blockIdx.x];
    uint32_t n = st.nsub < img.sub_cap ? st.nsub : img.sub_cap;
    const uint32_t t = threadIdx.x, per = (n + 255) / 256;
    uint32_t b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    if (b0 > n) b0 = n;
    LpSubSum acc;
    lp_sum_zero(acc);
    for (uint32_t i = b0; i < b1; i++) acc = lp_sum_combine(acc, totals[img.sub_off + i]);
    s_part[t] = acc;
    __syncthreads();
    LpSubSum pre;
    lp_sum_zero(pre);
    for (uint32_t i = 0; i < t; i++) pre = lp_sum_combine(pre, s_part[i]);
    for (uint32_t i = b0; i < b1; i++) {
        prefixes[img.sub_off + i] = pre;
        pre = lp_sum_combine(pre, totals[img.sub_off + i]);
    }
    if (t == 255) {
        st.blocks_decoded = pre.nblk;
        if (pre.nblk < img.total_blocks && !img.scan_path) st.error |= 2u;
    }
}

// Coefficient sink of the WRITE pass. Quantised AC coefficients almost always fit a signed byte, so a block is assembled
// and stored as 64 x int8 (natural order); a value outside [-127, 127] is stored as the escape -128 and its true 16-bit
// value goes straight to HBM into a "wide" copy of the block (one slot per block that needs it, handed out by an atomic
// counter; the slot index is recorded in wide_id[block]). Only escaped positions of a wide slot are ever read, so it needs
// no initialisation. Halving the slot (64 B per lane instead of 128 B) is what lets three or four workgroups share a CU's
// LDS -- the WRITE kernel is occupancy-bound -- and it halves the coefficient traffic of WRITE and IDCT.
// Slot layout in LDS: [16-byte chunk c][lane][16 B] per wave (chunk c = natural coefficients 16c..16c+15), so the
// ds_read_b128 / ds_write_b128 of a flush touch 1 KiB of consecutive bytes: conflict-free.
// A finished block is only queued; flush() runs at wave-uniform points so that its loads/stores execute with many lanes
// active instead of once per lane divergently. Blocks are stored in decode order: block n at coef8[n * 64], and TRANSPOSED
// (element v * 8 + u holds the coefficient of row u, column v: the write kernel feeds put() a transposed zigzag table).
// Bytes between the 16-byte chunks of a lane's slot. 1024 (= 64 lanes x 16 B) puts chunk c of every lane 256 dwords after chunk c - 1:
// the four lanes that move one block in a flush (one chunk each) then hit the same banks, a 4-way conflict on every read and clear --
// more than half of the kernel's LDS bank-conflict cycles (timing builds without the coefficient stores / without the flush:
// 170 -> 132 / 78 M conflict cycles, profiles/r04_d_write_lds.md). One 16-byte pad per chunk row (LP_SLOT_STRIDE 1040) moves the four
// lanes four banks apart.
#ifndef LP_SLOT_STRIDE
#define LP_SLOT_STRIDE 1040
#endif
#define LP_SLOT_WAVE_BYTES (4 * LP_SLOT_STRIDE)
struct DevSink {
    int8_t* slot;           // LDS: this lane's bytes of chunk 0; chunk c at slot + c * LP_SLOT_STRIDE
    int8_t* wslots;         // LDS: the wave's slots (lane 0's chunk 0)
    uint32_t* qlist;        // LDS: the wave's list of finished blocks, (block << 6) | lane -- 64 entries
    uint32_t lane;
    uint32_t qbc;           // queued block, as the lane's counter word after the block's last step (LpLane::bc; 0xffffffff = slot free)
    int8_t* coef8;          // this image's coefficient blocks
    int16_t* wide;          // this image's wide slots (64 int16 each)
    uint32_t* wide_id;      // this image's block -> wide slot
    uint32_t* n_wide;       // this image's wide-slot counter
    uint32_t wslot;         // wide slot of the current block, 0xffffffff = none
    uint32_t wide_cap;      // slots in this image's wide arena (= its block count)
    int16_t* dc16;          // this image's DC values, one per block (the DC rarely fits a byte): the WRITE pass stores the decoded
                            // DIFFERENCE, k_dc_scan turns the array into absolute values before k_idct reads it
    int32_t dcv;            // DC difference of the current block
    int32_t qdc;            // LP_X_FLUSH2: ... of the queued block (the next block's DC may be decoded before the flush)
    uint32_t dq0, dq1, dq2, dq3; // this lane's 8 most recent DC differences, newest in the top half of dq3 (a lane's blocks are consecutive, so
                            // eight of them are one aligned 16-byte store; one 2-byte store per block cost a 32-byte write transaction
                            // each -- profiles/r01_e). A 128-bit shift register in VGPRs: it was 4 KB of LDS per workgroup.
    __device__ __forceinline__ uint32_t dq_get(uint32_t i) const // 16-bit element i (rare paths only)
    {
        const uint32_t lo = (i & 2u) ? dq1 : dq0, hi = (i & 2u) ? dq3 : dq2; // four scalars, not an array: an indexed array lands in scratch
        const uint32_t w = (i & 4u) ? hi : lo;
        return (i & 1u) ? w >> 16 : w & 0xffffu;
    }
    uint32_t blk0, last;    // first block of this lane, last block flushed (0xffffffff = none yet)
    __device__ __forceinline__ void put_dc(int32_t v, bool on) { dcv = on ? v : dcv; }
    // where = the coefficient's byte offset inside the lane's slot: (nat >> 4) * LP_SLOT_STRIDE + (nat & 15), precomputed per zigzag index (s_zz)
    __device__ __forceinline__ void put(uint16_t where, int32_t v)
    {
#ifdef LP_EXP_NOPUT
        asm volatile("" :: "v"(where), "v"(v)); return; // timing experiment: what do the coefficient stores cost?
#endif
        if (LP_RARE(v < -127 || v > 127)) { // rare: strong edges at fine quantisation
            const uint32_t nat = (((uint32_t)where / LP_SLOT_STRIDE) << 4) | (((uint32_t)where % LP_SLOT_STRIDE) & 15u);
            if (wslot == 0xffffffffu) wslot = atomicAdd(n_wide, 1u);
            // A settled decode hands out at most one slot per block. A pass over UNSETTLED exit states (the deferred chunk whose verify
            // rounds were not enough: it is decoded again afterwards) lets subsequences overlap, so more slots than blocks can be asked
            // for: those writes are dropped instead of running off the image's wide arena (seen as a GPU memory fault, one run in five,
            // with noisy 1024 x 1024 sources in 3-image chunks).
            if (wslot < wide_cap) wide[(size_t)wslot * 64 + nat] = (int16_t)v;
            v = -128;
        }
        slot[where] = (int8_t)v;
    }
    // the queue entry is the lane's block counter word as it is (no arithmetic in the decode step); flush() turns it into the block index
    __device__ __forceinline__ void end_block(uint32_t bc, bool on)
    {
        qbc = on ? bc : qbc;
        if (LP_X_FLUSH2) qdc = on ? dcv : qdc;
        if (LP_RARE(on && wslot != 0xffffffffu)) { wide_id[blk0 + (bc >> 5) - 1u] = wslot; wslot = 0xffffffffu; } // rare
    }
    __device__ __forceinline__ bool stalled() const { return qbc != 0xffffffffu; }
    // Wave-cooperative: the lanes with a finished block put (block, lane) on a per-wave list; then four lanes move one block each --
    // lane 4e + c reads chunk c of the e-th listed lane's slot, stores it (64 contiguous bytes per four lanes) and clears it. The cost
    // follows the number of finished blocks (about a quarter of the lanes per flush) instead of four full-wave 16-byte reads, stores and
    // clears whatever the number (PMC / timing experiment: the flush was 30 % of the kernel).
    __device__ __forceinline__ void flush()
    {
#ifdef LP_EXP_NOFLUSH
        if (qbc != 0xffffffffu) { last = blk0 + (qbc >> 5) - 1u; qbc = 0xffffffffu; } // timing experiment: what does the block flush cost?
        return;
#endif
        const bool q = qbc != 0xffffffffu;
        const uint64_t mask = __ballot(q);
        if (!mask) return; // wave-uniform
        if (q) {
            const uint32_t qblk = blk0 + (qbc >> 5) - 1u;
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            qlist[pos] = (qblk << 6) | lane;
            const uint32_t k = qblk & 7u;
            dq0 = (dq0 >> 16) | (dq1 << 16); dq1 = (dq1 >> 16) | (dq2 << 16); // four v_alignbit
            dq2 = (dq2 >> 16) | (dq3 << 16); dq3 = (dq3 >> 16) | ((uint32_t)(LP_X_FLUSH2 ? qdc : dcv) << 16);
            if (k == 7u) { // after the eighth block of an aligned group, element j holds block g0 + j
                const uint32_t g0 = qblk - 7u;
                if (g0 >= blk0) *reinterpret_cast<uint4*>(dc16 + g0) = make_uint4(dq0, dq1, dq2, dq3); // the whole group is this lane's
                else for (uint32_t j = blk0 - g0; j < 8u; j++) dc16[g0 + j] = (int16_t)dq_get(j);           // the lane started inside the group
            }
            last = qblk;
            qbc = 0xffffffffu;
        }
        __builtin_amdgcn_wave_barrier(); // the list is read by other lanes of the same wave: LDS operations of a wave execute in order
        const uint32_t n = (uint32_t)__popcll(mask);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        for (uint32_t i = 0; i < n; i += 16u) { // wave-uniform
            const uint32_t e = i + (lane >> 2);
            if (e < n) {
                const uint32_t ent = qlist[e], c = lane & 3u;
                uint4* s = reinterpret_cast<uint4*>(wslots + c * LP_SLOT_STRIDE + ((ent & 63u) << 4));
                const uint4 r = *s;
                *s = make_uint4(0, 0, 0, 0);
                // streaming store: the block is written once and read once by k_idct; a regular store write-allocates the line
                // (PMC: 33 MB fetched per image by a kernel that reads 4 MB)
                __builtin_nontemporal_store((u32x4){r.x, r.y, r.z, r.w}, reinterpret_cast<u32x4*>(coef8 + ((size_t)(ent >> 6) << 6) + (c << 4)));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void finish() // the lane's last, incomplete group of DC values
    {
        if (last != 0xffffffffu && (last & 7u) != 7u) {
            const uint32_t g0 = last & ~7u;
            for (uint32_t b = g0 > blk0 ? g0 : blk0; b <= last; b++) dc16[b] = (int16_t)dq_get(7u - (last - b)); // the newest sits in element 7
        }
    }
};

template <class MEM>
__global__ __launch_bounds__(HUFF_T) void k_huff_write(const LpJpeg* __restrict__ imgs, LpJpegState* __restrict__ states,
                                                       const LpHuffSet* __restrict__ huffs, const uint32_t* __restrict__ clean_arena,
                                                       const uint32_t* __restrict__ rst_bits, const LpSubState* __restrict__ exits,
                                                       const LpSubSum* __restrict__ prefixes, int8_t* __restrict__ coef8_arena,
                                                       int16_t* __restrict__ wide_arena, uint32_t* __restrict__ wide_id_arena,
                                                       int16_t* __restrict__ dc_arena)
{
    __shared__ uint4 s_hs4[LP_HUFF_LDS_BYTES / 16];
    const LpHuffSet* s_hs = reinterpret_cast<const LpHuffSet*>(s_hs4);
    __shared__ uint32_t s_ring[HUFF_T * MEM::kRows];
    __shared__ __attribute__((aligned(16))) int8_t s_slots[(HUFF_T / 64) * LP_SLOT_WAVE_BYTES];
    __shared__ uint16_t s_zz[80];
    __shared__ uint32_t s_qlist[HUFF_T];
    const LpJpeg& img = imgs[blockIdx.y];
    LpJpegState& st = states[blockIdx.y];
    const uint32_t nsub = st.nsub < img.sub_cap ? st.nsub : img.sub_cap;
    if (blockIdx.x * HUFF_T >= nsub) return;
    {
        const uint8_t zz[80] = LP_ZIGZAG_INIT;
        // blocks are stored transposed (column-major) for k_idct's column pass
        if (threadIdx.x < 80) {
            const uint32_t nat = ((zz[threadIdx.x] & 7u) << 3) | (zz[threadIdx.x] >> 3);
            s_zz[threadIdx.x] = (uint16_t)((nat >> 4) * LP_SLOT_STRIDE + (nat & 15u)); // where DevSink::put stores it, relative to the lane's slot
        }
        uint4* z4 = reinterpret_cast<uint4*>(s_slots);
        for (uint32_t i = threadIdx.x; i < (HUFF_T / 64) * LP_SLOT_WAVE_BYTES / 16; i += HUFF_T) z4[i] = make_uint4(0, 0, 0, 0);
    }
    stage_huff(s_hs4, huffs + img.huff_idx);
    const uint32_t sub = blockIdx.x * HUFF_T + threadIdx.x;
    if ((sub & ~63u) >= nsub) return; // a whole wave without work
    // the flush is wave-cooperative, so the lanes past the last subsequence of a partly filled wave stay: they start at the end of
    // the stream with nothing to decode and only help moving the others' blocks
    const bool idle = sub >= nsub;
    const uint32_t g = img.sub_off + (idle ? nsub - 1u : sub);
    const LpImgCtx ic = make_ctx(img, st);
    MEM m = MEM::make(clean_arena + img.clean_off, img.clean_cap_words, s_ring + (threadIdx.x >> 6) * (64 * MEM::kRows) + (threadIdx.x & 63), s_hs->lut[0], s_hs->lut2, nullptr, huffs + img.huff_idx,
                      rst_bits + img.rst_off);
    LpSubState entry;
    entry.p = 0; entry.bz = 0;
    if (sub != 0 && !idle) entry = exits[g - 1];
    LpSubSum prefix = prefixes[g];
    uint32_t end_p = exits[g].p;
    if (idle) { entry.p = end_p = ic.total_bits; prefix.nblk = ic.total_blocks; }
    DevSink sink;
    sink.wslots = s_slots + (threadIdx.x >> 6) * LP_SLOT_WAVE_BYTES;
    sink.lane = threadIdx.x & 63;
    sink.slot = sink.wslots + sink.lane * 16;
    sink.qlist = s_qlist + (threadIdx.x >> 6) * 64;
    sink.qbc = 0xffffffffu;
    sink.coef8 = coef8_arena + img.coef_off;
    sink.wide = wide_arena + img.coef_off;
    sink.wide_id = wide_id_arena + img.coef_off / 64;
    sink.n_wide = &st.n_wide;
    sink.wslot = 0xffffffffu;
    sink.wide_cap = img.total_blocks;
    sink.dc16 = dc_arena + img.coef_off / 64;
    sink.dcv = 0;
    sink.qdc = 0;
    sink.dq0 = sink.dq1 = sink.dq2 = sink.dq3 = 0;
    sink.blk0 = prefix.nblk;
    sink.last = 0xffffffffu;
    bool irregular = false;
    lp_write_pass(m, ic, entry, end_p, prefix, s_zz, sink, &irregular);
    if (irregular && !idle) atomicOr(&st.error, 8u); // the restart intervals do not hold exactly their MCUs: the serial decoder's case (lp_jbits.h)
}

// ------------------------------------------------------------------------------------------------
// DC differences -> absolute DC (in place): jdhuff.c decode_mcu keeps last_dc_val per component and process_restart zeroes
// it every `dri` MCUs. An image is cut into 16 contiguous ranges of MCUs, one wave each; a wave walks its range 64 MCUs at
// a time, lane = MCU (so the loads/stores of a step cover one contiguous run of bytes): a segmented wave scan of the
// per-MCU component totals gives every lane its predictors. Phase 1 computes each range's (total, had-a-reset), phase 2
// rewrites the range starting from the combined totals of the ranges before it. tests/emu checks the same arithmetic
// through lp_dc_walk. ~0.8 MB per 4096x4096 image: latency-bound, hidden behind the other streams' kernels.
struct DcSeg { int32_t v[LP_MAX_COMP]; bool f; }; // sums since the last reset, reset seen

// inclusive segmented scan over the wave; carry-in applies to the lanes before the first reset of the step
__device__ __forceinline__ DcSeg dc_wave_scan(DcSeg x, const DcSeg& carry)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int32_t u[LP_MAX_COMP];
#pragma unroll
        for (int c = 0; c < LP_MAX_COMP; c++) u[c] = __shfl_up(x.v[c], d, 64);
        const bool uf = __shfl_up((int)x.f, d, 64) != 0;
        if (lane >= (uint32_t)d) {
            if (!x.f)
#pragma unroll
                for (int c = 0; c < LP_MAX_COMP; c++) x.v[c] += u[c];
            x.f = x.f || uf;
        }
    }
    if (!x.f)
#pragma unroll
        for (int c = 0; c < LP_MAX_COMP; c++) x.v[c] += carry.v[c];
    return x;
}

template <bool WRITE>
__device__ __forceinline__ DcSeg dc_walk_range(int16_t* dc, uint32_t m0, uint32_t m1, uint32_t bpm, uint32_t dri, uint32_t comps, DcSeg carry)
{
    const uint32_t lane = threadIdx.x & 63;
    // the next step's DC differences are requested before this step's scan (a range is a handful of steps, each a load latency and a
    // dozen cross-lane hops: the kernel is all latency)
    int32_t nx[LP_MAX_BPM];
    auto request = [&](uint32_t mb) {
        const uint32_t m = mb + lane;
#pragma unroll
        for (uint32_t b = 0; b < LP_MAX_BPM; b++) nx[b] = (m < m1 && b < bpm) ? (int32_t)dc[(size_t)m * bpm + b] : 0;
    };
    if (m0 < m1) request(m0);
    for (uint32_t mb = m0; mb < m1; mb += 64) {
        const uint32_t m = mb + lane;
        const bool ok = m < m1;
        int32_t df[LP_MAX_BPM];
#pragma unroll
        for (uint32_t b = 0; b < LP_MAX_BPM; b++) df[b] = nx[b];
        if (mb + 64 < m1) request(mb + 64);
        DcSeg mine;
        mine.v[0] = mine.v[1] = mine.v[2] = 0;
        mine.f = ok && dri && m % dri == 0;
#pragma unroll
        for (uint32_t b = 0; b < LP_MAX_BPM; b++) {
            const uint32_t c = (comps >> (2 * b)) & 3u; // uniform
            mine.v[0] += c == 0 ? df[b] : 0; mine.v[1] += c == 1 ? df[b] : 0; mine.v[2] += c == 2 ? df[b] : 0;
        }
        const DcSeg inc = dc_wave_scan(mine, carry);
        if (WRITE) {
            // predictors at the start of this lane's MCU: the inclusive value of the previous lane (the carry for lane 0), 0 after a reset
            int32_t pr[LP_MAX_COMP];
#pragma unroll
            for (int c = 0; c < LP_MAX_COMP; c++) {
                const int32_t up = __shfl_up(inc.v[c], 1, 64);
                pr[c] = mine.f ? 0 : (lane ? up : carry.v[c]);
            }
#pragma unroll
            for (uint32_t b = 0; b < LP_MAX_BPM; b++) {
                const uint32_t c = (comps >> (2 * b)) & 3u;
                const int32_t v = (c == 0 ? pr[0] : c == 1 ? pr[1] : pr[2]) + df[b];
                pr[0] = c == 0 ? v : pr[0]; pr[1] = c == 1 ? v : pr[1]; pr[2] = c == 2 ? v : pr[2];
                if (ok && b < bpm) dc[(size_t)m * bpm + b] = (int16_t)v;
            }
        }
        // the last lane's inclusive value is the carry of the next step
#pragma unroll
        for (int c = 0; c < LP_MAX_COMP; c++) carry.v[c] = __shfl(inc.v[c], 63, 64);
        carry.f = carry.f || __shfl((int)inc.f, 63, 64) != 0;
    }
    return carry;
}

// Two small launches instead of one 1024-thread workgroup per image: a 16-wave workgroup has to find a CU with 16 free wave
// slots, and while the other parts of a batch keep the CUs full of Huffman workgroups it waited for milliseconds. Here every
// range of an image is an independent 64-thread workgroup: k_dc_sum stores each range's (sums, had-a-reset), k_dc_apply
// combines the ranges before its own (at most DCSCAN_RANGES - 1 records) and rewrites its range.
// Round 5: the launch picks the number of ranges (gridDim.x, at most DCSCAN_RANGES) from the largest image so that a range is about four
// 64-MCU steps -- with 16 ranges whatever the size a 4096 x 4096 image was 64 serial steps per wave, 107 us per launch and kernel (1.9 us per
// image for the two: as much as the unstuff kernels' count pass).
#define DCSCAN_RANGES 64
struct DcPartial { int32_t v[LP_MAX_COMP]; int32_t f; };

__device__ __forceinline__ void dc_range(const LpJpeg& img, uint32_t w, uint32_t& m0, uint32_t& m1, uint32_t& comps)
{
    const uint32_t nmcu = img.mcus_x * img.mcus_y;
    const uint32_t nr = gridDim.x;
    const uint32_t per = ((nmcu + nr - 1) / nr + 63) / 64 * 64; // whole 64-MCU steps per range
    m0 = w * per < nmcu ? w * per : nmcu;
    m1 = m0 + per < nmcu ? m0 + per : nmcu;
    comps = 0;
    for (uint32_t b = 0; b < LP_MAX_BPM; b++) comps |= (uint32_t)(img.blk_comp[b] & 3u) << (2 * b);
}

__global__ __launch_bounds__(64) void k_dc_sum(const LpJpeg* __restrict__ imgs, int16_t* __restrict__ dc_arena, DcPartial* __restrict__ partials)
{
    const LpJpeg& img = imgs[blockIdx.y];
    if (img.scan_path) return; // its scans stored absolute DC values
    uint32_t m0, m1, comps;
    dc_range(img, blockIdx.x, m0, m1, comps);
    DcSeg zero;
    zero.v[0] = zero.v[1] = zero.v[2] = 0;
    zero.f = false;
    const DcSeg tot = dc_walk_range<false>(dc_arena + img.coef_off / 64, m0, m1, img.bpm, img.dri, comps, zero);
    if (threadIdx.x == 0) {
        DcPartial p;
        for (int c = 0; c < LP_MAX_COMP; c++) p.v[c] = tot.v[c];
        p.f = tot.f ? 1 : 0;
        partials[blockIdx.y * DCSCAN_RANGES + blockIdx.x] = p;
    }
}

__global__ __launch_bounds__(64) void k_dc_apply(const LpJpeg* __restrict__ imgs, int16_t* __restrict__ dc_arena, const DcPartial* __restrict__ partials)
{
    const LpJpeg& img = imgs[blockIdx.y];
    if (img.scan_path) return;
    uint32_t m0, m1, comps;
    dc_range(img, blockIdx.x, m0, m1, comps);
    DcSeg pre;
    pre.v[0] = pre.v[1] = pre.v[2] = 0;
    pre.f = false;
    for (uint32_t q = 0; q < blockIdx.x; q++) { // the ranges before this one
        const DcPartial p = partials[blockIdx.y * DCSCAN_RANGES + q];
        for (int c = 0; c < LP_MAX_COMP; c++) pre.v[c] = p.f ? p.v[c] : pre.v[c] + p.v[c];
    }
    (void)dc_walk_range<true>(dc_arena + img.coef_off / 64, m0, m1, img.bpm, img.dri, comps, pre);
}

// ------------------------------------------------------------------------------------------------
// Dequantise + jpeg_idct_islow. One wave = 8 horizontally adjacent blocks of one component; lane = (block j,
// row/column r). Coefficient columns arrive as 8-byte loads (the WRITE pass stores blocks transposed), the first pass's outputs cross to
// the row pass through a per-wave LDS workspace of 16-bit values, and leave as 8-byte pixel rows (64 contiguous bytes per row across
// the 8 blocks).
//
// What "jpeg_idct_islow" means here is what the reference's libjpeg-turbo computes on x86-64: its SIMD routine (jidctint-sse2 / -avx2),
// which has the C code's butterflies and constants but works in 16-bit lanes -- dequantisation wraps (pmullw), so do in0 +- in4 and the
// odd part's z3 = in7 + in3, z4 = in5 + in1 (paddw / psubw), every rotation is a 32-bit sum of two 16 x 16 products with two combined
// constants (pmaddwd), the first pass's outputs are SATURATED to 16 bits (packssdw) -- except in a block whose rows 1-7 are empty, where
// the first pass is in0 << 2 in 16 bits (psllw: wraps) -- and the second pass ends saturated to 8 bits. For the coefficients of real
// images none of that triggers and the result is the C code's; damaged streams and hostile tables do trigger it, and the reference's
// pixels are then the SIMD routine's (oracle/jpeg_oracle.c lo_idct_islow_simd, pinned against the reference's own decoder on 1 300
// damaged files, 360 of which differ between the two arithmetics).
// gfx950 has the same lane operations: v_pk_mul_lo_u16 = pmullw, v_pk_add_u16 / v_pk_sub_u16 = paddw / psubw, v_dot2_i32_i16 = pmaddwd
// (+ the add that follows it), v_cvt_pk_i16_i32 = packssdw. So the transform is written in them (round 5; rounds 1-4 computed the C
// code's 32-bit arithmetic, ~150 vector instructions per tile and lane; this is ~110 and exact for EVERY input, no checks, one path).
// A lane holds its eight values as four 16-bit pairs, paired the way the rotations need them:
//     p04 = (d0, d4)   p26 = (d2, d6)   p71 = (d7, d1)   p35 = (d3, d5)
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define IDCT_K2(lo, hi) ((s16x2){(short)(lo), (short)(hi)})
__device__ __forceinline__ int32_t idct_dot2(s16x2 a, s16x2 b, int32_t c) { return __builtin_amdgcn_sdot2(a, b, c, false); }
__device__ __forceinline__ s16x2 idct_pair(uint32_t v) { return __builtin_bit_cast(s16x2, v); }

// One 1-D pass; rnd = the pass's rounding constant (every output holds tmp0 or tmp1 exactly once, so it rides in on them).
__device__ __forceinline__ void idct_1d_pk(s16x2 p04, s16x2 p26, s16x2 p71, s16x2 p35, int32_t rnd, int32_t o[8])
{
    const int32_t tmp3 = idct_dot2(p26, IDCT_K2(10703, 4433), 0), tmp2 = idct_dot2(p26, IDCT_K2(4433, -10704), 0);
    const s16x2 p40 = __builtin_shufflevector(p04, p04, 1, 0);
    const s16x2 s = p04 + p40, d = p04 - p40;                      // (d0 + d4, ..), (d0 - d4, ..) in 16 bits
    const int32_t tmp0 = idct_dot2(s, IDCT_K2(8192, 0), rnd), tmp1 = idct_dot2(d, IDCT_K2(8192, 0), rnd);   // sign-extended << 13
    const int32_t t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
    const s16x2 z = p71 + p35;                                     // (z3, z4) = (d7 + d3, d1 + d5) in 16 bits
    const int32_t z3n = idct_dot2(z, IDCT_K2(-6436, 9633), 0), z4n = idct_dot2(z, IDCT_K2(9633, 6437), 0);
    const int32_t o0 = idct_dot2(p71, IDCT_K2(-4927, -7373), z3n), o3 = idct_dot2(p71, IDCT_K2(-7373, 4926), z4n);
    const int32_t o1 = idct_dot2(p35, IDCT_K2(-20995, -4176), z4n), o2 = idct_dot2(p35, IDCT_K2(4177, -20995), z3n);
    o[0] = t10 + o3; o[7] = t10 - o3; o[1] = t11 + o2; o[6] = t11 - o2;
    o[2] = t12 + o1; o[5] = t12 - o1; o[3] = t13 + o0; o[4] = t13 - o0;
}

// The per-wave workspace: 8 blocks x 8 rows x 8 halfwords, a block every IDCT_WPITCH halfwords (144 bytes: the eight blocks of a tile
// start four banks apart, so the halfword stores of the column pass -- lane (j, c) writes row k, slot of column c -- spread over all 32
// banks). A row's eight values sit in the order (c0, c4, c2, c6, c7, c1, c3, c5): the row pass reads them as ONE 16-byte load that is
// already the four pairs idct_1d_pk takes.
#define IDCT_WPITCH 72
__device__ __forceinline__ uint32_t idct_slot(uint32_t c) { return (0x43716250u >> (4u * c)) & 7u; } // column -> halfword slot in its row

// Column pass for lane (block j, column c): the four dequantised pairs in, the first pass's outputs (saturated to 16 bits; for a block
// whose rows 1-7 are empty the routine's shortcut value instead) out to the workspace.
__device__ __forceinline__ void idct_cols_pk(s16x2 p04, s16x2 p26, s16x2 p71, s16x2 p35, bool rows17_empty, short* s_w, uint32_t j, uint32_t slot)
{
    int32_t o[8];
    idct_1d_pk(p04, p26, p71, p35, 1 << 10, o);
    const short dc_only = (short)(p04 << (s16x2){2, 2}).x;          // psllw(in0, PASS1_BITS): wraps
    short* w = s_w + j * IDCT_WPITCH + slot;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const s16x2 v = __builtin_amdgcn_cvt_pk_i16(o[k] >> 11, o[k + 1] >> 11);   // packssdw
        w[k * 8] = rows17_empty ? dc_only : v.x;
        w[(k + 1) * 8] = rows17_empty ? dc_only : v.y;
    }
}
// (o + C) >> 18 clamped to a byte for four outputs: the upper halves of two sums are taken by one v_perm, shifted as a pair, and clamped
// two at a time by v_sat_pk_u8_i16 (the second pair straight into the upper half of the output word). C carries the DESCALE rounding and
// the + 128 (packssdw + packsswb + 128 = clamp to [-128, 127], + 128 = clamp of the sum with 128 added to [0, 255]).
__device__ __forceinline__ uint32_t idct_pack4(int32_t o0, int32_t o1, int32_t o2, int32_t o3)
{
    const s16x2 p01 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)o1, (uint32_t)o0, 0x07060302u)) >> (s16x2){2, 2};
    const s16x2 p23 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)o3, (uint32_t)o2, 0x07060302u)) >> (s16x2){2, 2};
    uint32_t d;
    // the trailing s_nop: on the gfx940 family a destination-select write needs one wait state before the register is read again, and the
    // compiler does not look inside an asm block
    asm("v_sat_pk_u8_i16 %0, %1\n\tv_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0"
        : "=&v"(d) : "v"(__builtin_bit_cast(uint32_t, p01)), "v"(__builtin_bit_cast(uint32_t, p23)));
    return d;
}
// Row pass for lane (block j, row r): the row's four pairs in one load, eight pixels out.
__device__ __forceinline__ uint2 idct_rows_pk(const short* s_w, uint32_t j, uint32_t r)
{
    int32_t o[8];
    const uint4 v = *reinterpret_cast<const uint4*>(s_w + j * IDCT_WPITCH + r * 8);
    idct_1d_pk(idct_pair(v.x), idct_pair(v.y), idct_pair(v.z), idct_pair(v.w), (int32_t)((1u << 17) + (128u << 18)), o);
    uint2 out;
    out.x = idct_pack4(o[0], o[1], o[2], o[3]);
    out.y = idct_pack4(o[4], o[5], o[6], o[7]);
    return out;
}
// Dequantisation of a column held as eight signed bytes (raw.x = rows 0-3, raw.y = rows 4-7) straight into the four pairs: one SDWA
// multiply per value (sign-extended byte x 16-bit quantiser -> low 16 bits, written into its half of the pair).
#define IDCT_DEQ_PAIR(NAME, LO_SEL, HI_SEL)                                                                                                   \
    __device__ __forceinline__ s16x2 NAME(uint32_t wlo, uint32_t whi, uint32_t q)                                                             \
    {                                                                                                                                          \
        uint32_t d;                                                                                                                            \
        asm("v_mul_lo_u16_sdwa %0, sext(%1), %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:" LO_SEL " src1_sel:WORD_0\n\ts_nop 0\n\t"         \
            "v_mul_lo_u16_sdwa %0, sext(%2), %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:" HI_SEL " src1_sel:WORD_1\n\ts_nop 0"      \
            : "=&v"(d) : "v"(wlo), "v"(whi), "v"(q));                                                                                          \
        return __builtin_bit_cast(s16x2, d);                                                                                                   \
    }
IDCT_DEQ_PAIR(idct_deq04, "BYTE_0", "BYTE_0")   // (row 0 of raw.x, row 4 = byte 0 of raw.y)
IDCT_DEQ_PAIR(idct_deq26, "BYTE_2", "BYTE_2")   // (row 2, row 6)
IDCT_DEQ_PAIR(idct_deq71, "BYTE_3", "BYTE_1")   // (row 7 = byte 3 of raw.y, row 1 = byte 1 of raw.x): called with (raw.y, raw.x)
IDCT_DEQ_PAIR(idct_deq35, "BYTE_3", "BYTE_1")   // (row 3 = byte 3 of raw.x, row 5 = byte 1 of raw.y): called with (raw.x, raw.y)

__constant__ uint8_t c_nat2zigzag[64] = LP_NAT2ZIGZAG_INIT;

// Orders this wave's LDS writes before its following LDS reads of other lanes' data (and the other way round) for the compiler; the
// hardware keeps a wave's DS operations in issue order.
__device__ __forceinline__ void idct_wave_sync()
{
#ifdef LP_IDCT_WG_BARRIER
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

#ifndef IDCT_TPW
#define IDCT_TPW 16     // tiles of 32 blocks a workgroup walks along a block row (16 = a whole 4096-pixel luma row). The kernel is bound by
                        // how long its waves live against the rate they are dispatched at: 4 / 8 / 16 tiles -> 22.8 / 18.2 / 13.6 us per
                        // 4096 x 4096 image (scripts/archive/r03_run18.sh); taking several block rows per workgroup on top of that gains nothing
#endif
#ifndef IDCT_AHEAD
#define IDCT_AHEAD 4      // divides IDCT_TPW
#endif
// Grid: x = group of IDCT_TPW * 32 blocks along a block row, y = block row over the image's components stacked (all Y rows,
// then Cb, then Cr), z = image -- no integer division anywhere. A wave walks IDCT_TPW tiles of 8 blocks (its setup -- the
// descriptor loads, the quantisation column -- is paid once, and the next tile's coefficients are in flight while the
// current one is transformed; one-tile waves were bound by wave launch rate, not by HBM or VALU).
// PROG = false: the baseline images of the range (int8 blocks in decode order + wide slots + the DC array);
// PROG = true: the progressive ones (int16 blocks, raster order per component, see LpProgScan). Each skips the other kind.
#ifdef LP_IDCT_WPE // A/B: register budget for this many waves per SIMD
#define LP_IDCT_ATTR __attribute__((amdgpu_waves_per_eu(LP_IDCT_WPE)))
#else
#define LP_IDCT_ATTR
#endif
template <bool PROG>
__global__ __launch_bounds__(256) LP_IDCT_ATTR void k_idct(const LpJpeg* __restrict__ imgs, const LpJpegState* __restrict__ states,
                                              const int8_t* __restrict__ coef8_arena, const int16_t* __restrict__ wide_arena,
                                              const uint32_t* __restrict__ wide_id_arena, const int16_t* __restrict__ dc_arena,
                                              const int16_t* __restrict__ pcoef_arena, uint8_t* __restrict__ plane_arena)
{
    __shared__ __attribute__((aligned(16))) short s_w[4][8 * IDCT_WPITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_qt[64]; // transposed like the blocks: [column][row]
    __shared__ uint8_t s_n2z[64];
    const LpJpeg& img = imgs[blockIdx.z];
    if ((img.scan_path != 0) != PROG) return;
    uint32_t by = blockIdx.y, c = 0;
    while (c + 1 < img.ncomp && by >= img.bh[c]) { by -= img.bh[c]; c++; } // at most three steps (four components: CMYK / YCCK)
    if (c >= img.ncomp || by >= img.bh[c]) return;
    const uint32_t bw = img.bw[c];
    if (blockIdx.x * (32 * IDCT_TPW) >= bw) return;
    if (threadIdx.x < 64) s_qt[((threadIdx.x & 7) << 3) | (threadIdx.x >> 3)] = img.qt[c][threadIdx.x];
    if (PROG && threadIdx.x < 64) s_n2z[threadIdx.x] = c_nat2zigzag[threadIdx.x];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t j = lane >> 3, r = lane & 7;
    const uint32_t slot = idct_slot(r);
    // decode order: MCU (my, mx), then the component's blocks inside the MCU in scan order; sampling factors are 1 or 2
    const uint32_t hsh = img.hs[c] - 1u, vsh = img.vs[c] - 1u;
    __syncthreads();
    // this lane's column of the quantisation table as the four pairs of idct_1d_pk, unpacked once for all tiles
    uint32_t q04, q26, q71, q35;
    {
        const uint4 q = *reinterpret_cast<const uint4*>(&s_qt[r * 8]); // (q0, q1) (q2, q3) (q4, q5) (q6, q7)
        q04 = (q.x & 0xffffu) | (q.z << 16);
        q26 = (q.y & 0xffffu) | (q.w << 16);
        q71 = (q.w >> 16) | (q.x & 0xffff0000u);
        q35 = (q.y >> 16) | (q.z & 0xffff0000u);
    }
    // this lane's pixel row of the component (the descriptor is read here, once: behind the wave fences of the loop the compiler would
    // load it -- and redo the 64-bit address arithmetic -- for every tile)
    uint8_t* const dst_row = plane_arena + img.plane_off[c] + (size_t)(by * 8 + r) * img.plane_stride[c];
    // Baseline blocks: the coefficient column (8 bytes) and the DC of the tiles IDCT_AHEAD steps ahead are requested before the
    // current tile is transformed -- one 8-byte load per lane and tile in flight kept the kernel at a third of the HBM rate
    // (bytes in flight per CU, not VALU, bound it: 73 % of the wave cycles in s_waitcnt, profiles/r02_a_sq_counters.md).
    uint2 pre_raw[IDCT_AHEAD];
    int32_t pre_dc[IDCT_AHEAD];
    // the descriptor fields the block index needs, read once (a load inside the loop would make every step wait for the prefetches too)
    const uint32_t d_mcus_x = img.mcus_x, d_bpm = img.bpm, d_first = img.blk_first[c];
    const int8_t* const coef8 = coef8_arena + img.coef_off;
    const int16_t* const dc16 = dc_arena + img.coef_off / 64;
    const uint32_t row_blk = (by >> vsh) * d_mcus_x, row_in = d_first + ((by & vsh) << hsh);
    auto block_of = [&](uint32_t bx) { return (row_blk + (bx >> hsh)) * d_bpm + row_in + (bx & hsh); };
    // unconditional loads from a clamped block index (a load under a lane mask would be merged with a constant afterwards, and
    // that merge waits for the load): a tile past the end of the row re-reads the row's last block and is masked where it is used
    // the decode index of this lane's block advances by a constant from tile to tile (32 blocks = a whole number of MCUs further along
    // the row) and grows with bx, so "the block at min(bx, bw - 1)" is min(index, index of the row's last block): one add and one min per
    // tile instead of the shift / multiply / add chain of block_of
    const uint32_t blk_step = (32u >> hsh) * d_bpm, blk_last = block_of(bw - 1u);
    uint32_t blk_next = block_of(blockIdx.x * IDCT_TPW * 32 + wv * 8 + j); // tile 0; bx may lie past the row: the min below covers it
    auto fetch = [&](uint2& raw, int32_t& dcv) { // the next tile not yet requested
        if (PROG) { raw = make_uint2(0, 0); dcv = 0; return; }
        const uint32_t blk = blk_next < blk_last ? blk_next : blk_last;
        blk_next += blk_step;
        raw = *reinterpret_cast<const uint2*>(coef8 + (size_t)blk * 64 + r * 8);
        dcv = dc16[blk];
    };
#pragma unroll
    for (uint32_t a = 0; a < IDCT_AHEAD; a++) fetch(pre_raw[a], pre_dc[a]);
    // unrolled by IDCT_AHEAD: tile t lives in register slot t % IDCT_AHEAD, which is refilled with tile t + IDCT_AHEAD as soon as it
    // has been read (moving a pending load's register to another slot would wait for it)
    for (uint32_t t0 = 0; t0 < IDCT_TPW; t0 += IDCT_AHEAD) {
      bool out = false;
#pragma unroll
      for (uint32_t a = 0; a < IDCT_AHEAD; a++) {
        const uint32_t t = t0 + a;
        const uint32_t base = (blockIdx.x * IDCT_TPW + t) * 32;
        if (t >= IDCT_TPW || base >= bw) { out = true; break; } // workgroup-uniform
        const uint32_t bx = base + wv * 8 + j;
        const bool blk_ok = bx < bw;
        uint2 raw = pre_raw[a];
        raw.x = blk_ok ? raw.x : 0u;
        raw.y = blk_ok ? raw.y : 0u;
        const int32_t dc_now = pre_dc[a];
        fetch(pre_raw[a], pre_dc[a]);
        uint8_t* const dst = dst_row + bx * 8;
        s16x2 p04, p26, p71, p35;
        bool lane_rows17; // this lane's column has something in rows 1-7
        bool packed_bytes = false;
        if (!PROG) {
            // a byte equals 0x80 (the escape to the block's wide slot) <=> the byte of (x ^ 0x80808080) is zero; exact zero-byte test
            const uint32_t ex = raw.x ^ 0x80808080u, ey = raw.y ^ 0x80808080u;
            const uint32_t zx = ~(((ex & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ex | 0x7f7f7f7fu), zy = ~(((ey & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ey | 0x7f7f7f7fu);
            packed_bytes = __all((zx | zy) == 0); // the path almost every tile takes: the column is its eight bytes
        }
        if (packed_bytes) {
            p04 = idct_deq04(raw.x, raw.y, q04);
            p26 = idct_deq26(raw.x, raw.y, q26);
            p71 = idct_deq71(raw.y, raw.x, q71);
            p35 = idct_deq35(raw.x, raw.y, q35);
            if (r == 0 && blk_ok) p04.x = (short)((uint32_t)dc_now * (q04 & 0xffffu)); // the DC lives in its own 16-bit array (pmullw: low 16 bits)
            lane_rows17 = ((raw.x >> 8) | raw.y) != 0u;
        } else {
            int32_t cv[8];
            if (PROG) {
                // column r of the block: rows 0..7, gathered out of the zigzag order the scans store (the eight lanes of a block read
                // its 128 bytes between them)
                const uint32_t cbase = (c > 0 ? img.bw[0] * img.bh[0] : 0u) + (c > 1 ? img.bw[1] * img.bh[1] : 0u) + (c > 2 ? img.bw[2] * img.bh[2] : 0u);
                const int16_t* src = pcoef_arena + img.coef_off + ((size_t)cbase + (size_t)by * bw + bx) * 64;
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = blk_ok ? (int32_t)src[s_n2z[i * 8 + r]] : 0;
            } else {
                const uint32_t blk = block_of(bx);
                // column r of the block: 8 x int8 (see DevSink); -128 escapes to the block's wide slot
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = (int32_t)(int8_t)(((i < 4 ? raw.x : raw.y) >> (8 * (i & 3))) & 0xffu);
                const uint32_t ex = raw.x ^ 0x80808080u, ey = raw.y ^ 0x80808080u;
                const uint32_t zx = ~(((ex & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ex | 0x7f7f7f7fu), zy = ~(((ey & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ey | 0x7f7f7f7fu);
                if ((zx | zy) != 0) {
                    // a block no lane wrote (the stream ended early: the image is reported as failed) holds stale bytes, and a stale
                    // escape comes with a stale slot number: keep the read inside the image's own slots
                    uint32_t wid = wide_id_arena[img.coef_off / 64 + blk];
                    wid = wid < img.total_blocks ? wid : 0u;
                    const int16_t* w = wide_arena + img.coef_off + (size_t)wid * 64 + r * 8;
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (cv[i] == -128) cv[i] = w[i];
                }
                if (r == 0 && blk_ok) cv[0] = dc_now; // the DC lives in its own 16-bit array
            }
            auto pk = [](int32_t lo, int32_t hi) { return idct_pair(((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16)); };
            p04 = pk(cv[0], cv[4]) * idct_pair(q04);   // v_pk_mul_lo_u16 = pmullw
            p26 = pk(cv[2], cv[6]) * idct_pair(q26);
            p71 = pk(cv[7], cv[1]) * idct_pair(q71);
            p35 = pk(cv[3], cv[5]) * idct_pair(q35);
            lane_rows17 = false;
#pragma unroll
            for (int i = 1; i < 8; i++) lane_rows17 = lane_rows17 || cv[i] != 0;
        }
        // the routine's shortcut tests rows 1-7 of the whole block: the eight lanes of a block vote
        const uint64_t ac_mask = __builtin_amdgcn_ballot_w64(lane_rows17);
        const bool rows17_empty = ((ac_mask >> (j * 8u)) & 0xffull) == 0;
        // the transpose workspace is per wave (s_w[wv]) and a wave's LDS operations execute in order: no workgroup barrier between the
        // passes -- the first version had three per tile, which kept the four waves of a workgroup in lock step through every load wait
        idct_cols_pk(p04, p26, p71, p35, rows17_empty, s_w[wv], j, slot);
        idct_wave_sync();
        const uint2 px = idct_rows_pk(s_w[wv], j, r);
        idct_wave_sync();
        if (blk_ok) *reinterpret_cast<uint2*>(dst) = px;
      }
      if (out) break;
    }
}

// ------------------------------------------------------------------------------------------------
// Scans decoded by ONE LANE each (lp_prog_core.h), serially from the first bit to the last: the generic restatement of jdphuff.c / jdhuff.c.
// Since round 6 the progressive scans proper have a wave each (lp_kernels_prog.hip: ~15x faster per scan); this kernel keeps the
// sequential scan-path files (several scans, four components, table numbers 2 / 3: LpProgScan::sequential) and is the
// LILLIPUT_HIP_PROG_ENTROPY=lanes reference the wave decoder is tested against. The lanes
// of a launch share nothing (different streams, tables and blocks), so a workgroup carries only `lpw` of them: with few scans
// in flight every lane gets a wave -- and its SIMD's issue slots -- to itself.
struct DevProgMem {
    const uint32_t* words;
    uint32_t cap;
    const uint32_t* rst;
    const LpProgHuff* ht;
    int16_t* coef;  // the image's first block
    int16_t* stage; // this lane's 128 bytes of LDS: the block an AC refinement is working on
    uint64_t dirty; // elements changed since open(): only those go back -- other scans of the same level own the rest of the block
    uint32_t* err;  // the error word of the scan's pseudo stream (LpJpegState::error)
    // the scan stored outside its band (damaged data, lp_prog_core.h lp_prog_stray): what libjpeg makes of that depends on the order of the
    // scans in the file -- the image goes to the host route, like everything else the device decoders call irregular
    __device__ __forceinline__ void stray() { atomicOr(err, LP_PROG_IRREGULAR); }
    __device__ __forceinline__ uint32_t word(uint32_t w) const { return w < cap ? words[w] : 0u; }
    __device__ __forceinline__ uint32_t rst_bit(uint32_t k) const { return rst[k]; }
    __device__ __forceinline__ uint32_t lut8(uint32_t s, uint32_t i) const { return ht->lut8[s][i]; }
    __device__ __forceinline__ int32_t maxcode(uint32_t s, uint32_t l) const { return ht->maxcode[s][l]; }
    __device__ __forceinline__ int32_t valoff(uint32_t s, uint32_t l) const { return ht->valoff[s][l]; }
    __device__ __forceinline__ uint32_t val(uint32_t s, uint32_t i) const { return ht->vals[s][i]; }
    __device__ __forceinline__ void st(uint32_t blk, uint32_t e, int32_t v) { coef[(size_t)blk * 64 + e] = (int16_t)v; }
    __device__ __forceinline__ int32_t ld(uint32_t blk, uint32_t e) const { return coef[(size_t)blk * 64 + e]; }
    __device__ __forceinline__ uint64_t open(uint32_t blk) // eight independent 16-byte loads, the non-zero mask on the way into LDS
    {
        typedef uint4 __attribute__((may_alias)) uint4_a; // the same bytes are read and written as int16 elsewhere
        const uint4_a* src = reinterpret_cast<const uint4_a*>(coef + (size_t)blk * 64);
        uint4_a* dst = reinterpret_cast<uint4_a*>(stage);
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = src[i];
        uint64_t nz = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            dst[i] = v[i];
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                nz |= (uint64_t)(((w[q] & 0xffffu) ? 1u : 0u) | ((w[q] >> 16) ? 2u : 0u)) << (8 * i + 2 * q);
        }
        dirty = 0;
        return nz;
    }
    __device__ __forceinline__ int32_t get(uint32_t e) const { return stage[e]; }
    __device__ __forceinline__ void set(uint32_t e, int32_t v) { stage[e] = (int16_t)v; dirty |= 1ull << e; }
    __device__ __forceinline__ void close(uint32_t blk)
    {
        int16_t* dst = coef + (size_t)blk * 64;
        while (dirty) {
            const uint32_t e = (uint32_t)(__ffsll((unsigned long long)dirty) - 1);
            dirty &= dirty - 1ull;
            dst[e] = stage[e];
        }
    }
};

__global__ __launch_bounds__(64) void k_prog_scan(const LpProgScan* __restrict__ scans, uint32_t first, uint32_t n, uint32_t lpw, uint32_t only_sequential,
                                                  const LpJpeg* __restrict__ streams, LpJpegState* stream_states,
                                                  const LpProgHuff* __restrict__ huffs, const uint32_t* __restrict__ clean_arena,
                                                  const uint32_t* __restrict__ rst_arena, int16_t* __restrict__ pcoef)
{
    __shared__ __attribute__((aligned(16))) int16_t s_stage[64][64];
    if (threadIdx.x >= lpw) return;
    const uint32_t i = blockIdx.x * lpw + threadIdx.x;
    if (i >= n) return;
    const LpProgScan sc = scans[first + i];
    if (only_sequential && !sc.sequential) return; // k_prog_wave's
    const LpJpeg& stream = streams[sc.stream];
    LpJpegState& st = stream_states[sc.stream];
    DevProgMem m;
    m.err = &st.error;
    m.words = clean_arena + stream.clean_off;
    m.cap = stream.clean_cap_words;
    m.rst = rst_arena + stream.rst_off;
    m.ht = huffs + sc.huff;
    m.coef = pcoef + sc.coef_off;
    m.stage = s_stage[threadIdx.x];
    m.dirty = 0;
    lp_prog_scan(m, sc, st.clean_bytes * 8u, st.n_rst);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (plain C++ signatures; see lp_launch.h)
void lp_launch_unstuff(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_chunks, const uint8_t* d_raw, uint2* d_chunk_cnt,
                       LpJpegState* d_states, uint32_t* d_clean, uint32_t* d_rst)
{
    if (!nimg || !max_chunks) return;
    dim3 g((max_chunks + LP_UNSTUFF_CPW - 1) / LP_UNSTUFF_CPW, nimg);
    hipLaunchKernelGGL(k_unstuff_count, g, dim3(UNSTUFF_T), 0, s, d_imgs, d_raw, d_chunk_cnt, d_states);
    hipLaunchKernelGGL(k_unstuff_scan, dim3(nimg), dim3(256), 0, s, d_imgs, d_chunk_cnt, d_states, d_clean);
    hipLaunchKernelGGL(k_unstuff_scatter, g, dim3(UNSTUFF_T), 0, s, d_imgs, d_raw, (const uint2*)d_chunk_cnt, d_clean, d_rst, d_states);
}

void lp_launch_huff_spec(hipStream_t s, const LpHuffArgs& a)
{
    if (!a.nimg || !a.max_sub) return;
    dim3 g((a.max_sub + HUFF_T - 1) / HUFF_T, a.nimg);
    hipLaunchKernelGGL(k_huff_spec, g, dim3(HUFF_T), 0, s, a.imgs, a.states, a.huffs, a.clean, a.rst, a.ckpts, a.spec_exit, a.spec_total, a.cur_exit,
                       a.cur_total, a.entry_used, a.sched, a.tot_sub);
}

// Before the stages behind the verification run a second time (LpEngine::finish_decode): forget what their first run left in
// the per-image state -- the block-count verdict of k_sub_scan and the wide-slot counter of k_huff_write.
__global__ void k_reset_tail_state(LpJpegState* __restrict__ states, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    states[i].error &= ~2u;
    states[i].blocks_decoded = 0;
    states[i].n_wide = 0;
}
void lp_launch_reset_tail_state(hipStream_t s, LpJpegState* d_states, uint32_t n)
{
    if (n) hipLaunchKernelGGL(k_reset_tail_state, dim3((n + 63) / 64), dim3(64), 0, s, d_states, n);
}

// Small host -> device uploads (descriptors, op lists) as a kernel reading pinned host memory: a copy-engine transfer would queue
// behind the multi-hundred-megabyte H2D copies of the ingest pipeline and hold the compute stream up for milliseconds.
__global__ void k_copy_small(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
// Several of those in ONE launch (a one-image call is a chain of ~35 launches of ~5 us each, a third of them descriptor copies and small
// fills: profiles/r06_one_image.md): blockIdx.y = segment; a segment without a source is a zero fill of exactly its bytes.
__global__ void k_small_ops(LpSmallOps ops)
{
    const LpSmallSeg sg = ops.s[blockIdx.y];
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(sg.dst);
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(sg.src);
    if (src) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
    } else {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n16; i += gridDim.x * blockDim.x) dst[i] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0 && threadIdx.x < sg.tail) reinterpret_cast<uint8_t*>(dst + sg.n16)[threadIdx.x] = 0;
    }
}
void lp_launch_small_ops(hipStream_t s, const LpSmallOps& ops, uint32_t n)
{
    if (!n) return;
    uint32_t max16 = 1;
    for (uint32_t i = 0; i < n; i++) max16 = ops.s[i].n16 > max16 ? ops.s[i].n16 : max16;
    const uint32_t blocks = max16 < 256u * 2048u ? (max16 + 255u) / 256u : 2048u; // (a batch's encode clears tens of MB of bit buffers in here: the whole device's worth of workgroups)
    hipLaunchKernelGGL(k_small_ops, dim3(blocks, n), dim3(256), 0, s, ops);
}
void lp_launch_copy_small(hipStream_t s, void* dst, const void* src_pinned, size_t bytes)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (!n16) return;
    const uint32_t blocks = n16 < 256u * 64u ? (n16 + 255u) / 256u : 64u;
    hipLaunchKernelGGL(k_copy_small, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src_pinned), n16);
}

// Zero-copy ingest: the entropy-coded segments of a chunk that sit in pinned, device-mapped host memory (lp_hostmem.h) are fetched by
// THIS kernel -- 16-byte loads over the link, re-aligned on the way (a segment starts wherever its file's headers end; the arena wants
// 16-byte alignment) -- instead of one copy-engine transfer per segment: 32 transfers of ~4 MB on one queue reach 49.8 GB/s, spread
// over four queues next to the running decode kernels 40 - 45, one launch of this kernel with 64 workgroups 55 - 56 GB/s (one 135 MB
// transfer: 57.4; scripts/ingest_micro.hip, profiles/r03_a_ingest.md). A workgroup walks 4 KiB tiles grid-stride; tile_first[p] is the
// index of piece p's first tile, tile_first[n] the total.
__global__ __launch_bounds__(256) void k_gather_raw(const LpGatherPiece* __restrict__ pcs, const uint32_t* __restrict__ tile_first, uint32_t npieces, uint8_t* __restrict__ arena)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t total = tile_first[npieces];
    for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
        uint32_t lo = 0, hi = npieces; // the piece of tile t (workgroup-uniform binary search)
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (tile_first[mid] <= t) lo = mid; else hi = mid; }
        const LpGatherPiece pc = pcs[lo];
        const uint32_t off = (t - tile_first[lo]) * 4096u + threadIdx.x * 16u;
        if (off >= pc.len) continue;
        const uintptr_t src = (uintptr_t)pc.src + off;
        const uint32_t mis = __builtin_amdgcn_readfirstlane((uint32_t)(src & 15u)); // the same for every lane of the piece: offsets are multiples of 16
        const u32x4* a = reinterpret_cast<const u32x4*>(src - mis);
        const u32x4 q0 = __builtin_nontemporal_load(a);
        u32x4 out = q0;
        if (mis) {
            // the 16 bytes reach into the next aligned quad; when the segment ends inside this one the second load is skipped (it could
            // touch the page after the caller's buffer)
            u32x4 q1 = {0u, 0u, 0u, 0u};
            if (off + 16u - mis < pc.len) q1 = __builtin_nontemporal_load(a + 1);
            const uint32_t sh = (mis & 3u) * 8u; // v_alignbit: ({hi, lo} >> sh)[31:0], sh == 0 gives lo
#define LP_FUNNEL(A, B, C, D, E) out = (u32x4){__builtin_amdgcn_alignbit(B, A, sh), __builtin_amdgcn_alignbit(C, B, sh), __builtin_amdgcn_alignbit(D, C, sh), __builtin_amdgcn_alignbit(E, D, sh)}
            switch (mis >> 2) { // scalar branch
            case 0: LP_FUNNEL(q0.x, q0.y, q0.z, q0.w, q1.x); break;
            case 1: LP_FUNNEL(q0.y, q0.z, q0.w, q1.x, q1.y); break;
            case 2: LP_FUNNEL(q0.z, q0.w, q1.x, q1.y, q1.z); break;
            default: LP_FUNNEL(q0.w, q1.x, q1.y, q1.z, q1.w); break;
            }
#undef LP_FUNNEL
        }
        *reinterpret_cast<u32x4*>(arena + pc.dst_off + off) = out;
    }
}
void lp_launch_gather_raw(hipStream_t s, const LpGatherPiece* d_pcs, const uint32_t* d_tile_first, uint32_t npieces, uint8_t* d_arena, uint32_t workgroups)
{
    if (!npieces) return;
    hipLaunchKernelGGL(k_gather_raw, dim3(workgroups), dim3(256), 0, s, d_pcs, d_tile_first, npieces, d_arena);
}

void lp_launch_huff_verify(hipStream_t s, const LpHuffArgs& a, uint32_t round)
{
    if (!a.nimg || !a.max_sub) return;
    dim3 g((a.max_sub + HUFF_T - 1) / HUFF_T, a.nimg);
    hipLaunchKernelGGL(k_huff_verify, g, dim3(HUFF_T), 0, s, a.imgs, a.states, a.huffs, a.clean, a.rst, (const LpCkptPk*)a.ckpts,
                       (const LpSubState*)a.spec_exit, (const LpSubSum*)a.spec_total, a.cur_exit, a.cur_total, a.entry_used, a.changed,
                       round, a.sched.K, a.tot_sub);
}

uint32_t lp_dc_scan_max_ranges() { return DCSCAN_RANGES; }
void lp_launch_dc_scan(hipStream_t s, const LpJpeg* d_imgs, uint32_t nimg, uint32_t max_mcus, int16_t* d_dc, void* d_partials /* nimg * lp_dc_scan_max_ranges() * 16 bytes */)
{
    if (!nimg) return;
    uint32_t nr = (max_mcus + 255u) / 256u; // ~four steps of 64 MCUs per range
    nr = nr < 1u ? 1u : nr > DCSCAN_RANGES ? DCSCAN_RANGES : nr;
    hipLaunchKernelGGL(k_dc_sum, dim3(nr, nimg), dim3(64), 0, s, d_imgs, d_dc, reinterpret_cast<DcPartial*>(d_partials));
    hipLaunchKernelGGL(k_dc_apply, dim3(nr, nimg), dim3(64), 0, s, d_imgs, d_dc, reinterpret_cast<const DcPartial*>(d_partials));
}

void lp_launch_sub_scan(hipStream_t s, const LpHuffArgs& a)
{
    if (!a.nimg) return;
    hipLaunchKernelGGL(k_sub_scan, dim3(a.nimg), dim3(256), 0, s, a.imgs, a.states, (const LpSubSum*)a.cur_total, a.prefix);
}

// How many workgroups of the WRITE kernel the device holds at once (LDS-limited occupancy x CUs): a resident batch sizes its chunks
// so that a launch is one full round of them (a second, mostly empty round costs as much as the first).
// LILLIPUT_HIP_WRITE_LDS_PAD: bytes of (unused) dynamic LDS added to every WRITE workgroup -- an occupancy knob for measurements (4 KB
// more leaves three workgroups per CU, 16 KB more two); the resident chunk size follows through lp_huff_write_slots().
static size_t write_lds_pad()
{
    static const size_t pad = getenv("LILLIPUT_HIP_WRITE_LDS_PAD") ? (size_t)atol(getenv("LILLIPUT_HIP_WRITE_LDS_PAD")) : 0;
    return pad;
}
uint32_t lp_huff_write_slots()
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_huff_write<WriteMem>, HUFF_T, write_lds_pad()) != hipSuccess || per_cu <= 0) per_cu = 4;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)per_cu * (uint32_t)cus;
}

void lp_launch_huff_write(hipStream_t s, const LpHuffArgs& a)
{
    if (!a.nimg || !a.max_sub) return;
    dim3 g((a.max_sub + HUFF_T - 1) / HUFF_T, a.nimg);
    hipLaunchKernelGGL(k_huff_write<WriteMem>, g, dim3(HUFF_T), write_lds_pad(), s, a.imgs, a.states, a.huffs, a.clean, a.rst, (const LpSubState*)a.cur_exit,
                       (const LpSubSum*)a.prefix, a.coef8, a.wide, a.wide_id, a.dc16);
}

void lp_launch_idct(hipStream_t s, const LpJpeg* d_imgs, const LpJpegState* d_states, uint32_t nimg, uint32_t max_bw, uint32_t max_rows, const int8_t* d_coef8,
                    const int16_t* d_wide, const uint32_t* d_wide_id, const int16_t* d_dc, uint8_t* d_planes, uint32_t which, const int16_t* d_pcoef)
{
    if (!nimg || !max_bw || !max_rows) return;
    dim3 g((max_bw + 32 * IDCT_TPW - 1) / (32 * IDCT_TPW), max_rows, nimg); // max_rows = most block rows of an image, all components stacked
    if (which & 1u) hipLaunchKernelGGL(k_idct<false>, g, dim3(256), 0, s, d_imgs, d_states, d_coef8, d_wide, d_wide_id, d_dc, d_pcoef, d_planes);
    if (which & 2u) hipLaunchKernelGGL(k_idct<true>, g, dim3(256), 0, s, d_imgs, d_states, d_coef8, d_wide, d_wide_id, d_dc, d_pcoef, d_planes);
}

void lp_launch_prog_scans(hipStream_t s, const LpProgScan* d_scans, uint32_t first, uint32_t n, uint32_t lpw, bool only_sequential, const LpJpeg* d_streams,
                          LpJpegState* d_stream_states, const LpProgHuff* d_huffs, const uint32_t* d_clean, const uint32_t* d_rst, int16_t* d_pcoef)
{
    if (!n) return;
    if (lpw < 1) lpw = 1;
    if (lpw > 64) lpw = 64;
    hipLaunchKernelGGL(k_prog_scan, dim3((n + lpw - 1) / lpw), dim3(64), 0, s, d_scans, first, n, lpw, only_sequential ? 1u : 0u, d_streams, d_stream_states, d_huffs, d_clean,
                       d_rst, d_pcoef);
}
