// lp_prog_host.cpp -- see lp_prog_host.h.
#include "lp_prog_host.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "lp_hostmem.h"
#include "lp_prog_core.h"
#include "lp_huff_core.h" // LP_ZIGZAG_INIT

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

// bit k set <=> element k of the block is non-zero
static inline uint64_t lp_host_nonzero_mask(const int16_t* c)
{
#if defined(__SSE2__)
    const __m128i z = _mm_setzero_si128();
    uint64_t zero = 0;
    for (int q = 0; q < 4; q++) {
        const __m128i a = _mm_cmpeq_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(c + 16 * q)), z);
        const __m128i b = _mm_cmpeq_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(c + 16 * q + 8)), z);
        zero |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_packs_epi16(a, b)) << (16 * q);
    }
    return ~zero;
#else
    uint64_t nz = 0;
    for (int k = 0; k < 64; k++) nz |= (uint64_t)(c[k] != 0) << k;
    return nz;
#endif
}

namespace {
struct HostProgMem {
    const uint32_t* words;
    size_t nwords;
    const uint32_t* rst;
    const LpProgHuff* ht;
    int16_t* coef;
    int16_t* cur;
    uint32_t word(uint32_t w) const { return w < nwords ? words[w] : 0u; }
    uint32_t rst_bit(uint32_t k) const { return rst[k]; }
    uint32_t lut8(uint32_t s, uint32_t i) const { return ht->lut8[s][i]; }
    int32_t maxcode(uint32_t s, uint32_t l) const { return ht->maxcode[s][l]; }
    int32_t valoff(uint32_t s, uint32_t l) const { return ht->valoff[s][l]; }
    uint32_t val(uint32_t s, uint32_t i) const { return ht->vals[s][i]; }
    void st(uint32_t blk, uint32_t e, int32_t v) { coef[(size_t)blk * 64 + e] = (int16_t)v; }
    int32_t ld(uint32_t blk, uint32_t e) const { return coef[(size_t)blk * 64 + e]; }
    uint64_t open(uint32_t blk)
    {
        cur = coef + (size_t)blk * 64;
        return lp_host_nonzero_mask(cur);
    }
    int32_t get(uint32_t e) const { return cur[e]; }
    void set(uint32_t e, int32_t v) { cur[e] = (int16_t)v; }
    void close(uint32_t) {}
};

// What k_unstuff_* produce on the device: stuffed zero bytes and restart markers removed, the bit position of every restart
// boundary recorded, the stream packed into big-endian words. (Fill bytes before a marker -- FF FF Dn -- are dropped as well.)
void unstuff(const uint8_t* raw, size_t n, std::vector<uint8_t>& clean, std::vector<uint32_t>& rst)
{
    clean.clear();
    rst.clear();
    clean.reserve(n + 8);
    size_t q = 0;
    while (q < n) {
        const uint8_t* ff = static_cast<const uint8_t*>(memchr(raw + q, 0xFF, n - q));
        const size_t run = ff ? (size_t)(ff - (raw + q)) : n - q;
        clean.insert(clean.end(), raw + q, raw + q + run);
        q += run;
        if (q >= n) break;
        // raw[q] == 0xFF
        const uint8_t next = q + 1 < n ? raw[q + 1] : 0xD9;
        if (next == 0x00) { clean.push_back(0xFF); q += 2; }
        else if (next == 0xFF) q += 1;                                  // fill byte
        else if (next >= 0xD0 && next <= 0xD7) { rst.push_back((uint32_t)clean.size() * 8u); q += 2; }
        else q += 2;                                                    // cannot happen: the parser ends the scan at the first other marker
    }
}

void run_task(const LpProgHostTask& t, std::vector<uint8_t>& clean, std::vector<uint32_t>& rst, std::vector<uint32_t>& words)
{
    const LpProgScanHost& sh = *t.scan;
    if (sh.arith) { // a QM-coded scan reads its raw bytes itself (lp_arith_host.h); an impossible code leaves the rest of the scan alone, like libjpeg's warning
        if (lp_arith_scan(t.data + sh.ecs_off, sh.ecs_len, sh.s, sh.ar, t.coef) == 2) __atomic_or_fetch(t.error, 4u, __ATOMIC_RELAXED); // the scan ended on an unknown marker
        return;
    }
    unstuff(t.data + sh.ecs_off, sh.ecs_len, clean, rst);
    const LpProgScan& sc = sh.s;
    const uint32_t rst_cap = sc.dri ? (sc.mcux * sc.mcuy + sc.dri - 1) / sc.dri + 2 : 2;
    uint32_t n_rst = (uint32_t)rst.size();
    if (n_rst > rst_cap) { __atomic_or_fetch(t.error, 4u, __ATOMIC_RELAXED); n_rst = rst_cap; }
    words.assign((clean.size() + 3) / 4 + 4, 0u);
    for (size_t q = 0; q < clean.size(); q++) words[q >> 2] |= (uint32_t)clean[q] << (24 - 8 * (q & 3));
    rst.push_back(0);
    HostProgMem m{words.data(), words.size(), rst.data(), &sh.tables, t.coef, t.coef};
    lp_prog_scan(m, sc, (uint32_t)clean.size() * 8u, n_rst);
}

std::atomic<int> g_mode{-1};
} // namespace

// Where the scans of a progressive file are entropy-decoded. Host threads always exist (the default: a scan is serial by construction and
// a host core walks it ~30x faster than one GPU lane; 48 images/s against 1.9 at 4096 x 4096). The device-lane decoder (k_prog_scan)
// only pays with thousands of images in flight and is a BUILD option since round 3 (make DEFS=-DLP_PROG_DEVICE_LANES); without it the
// switch below is inert.
bool lp_prog_entropy_on_device()
{
#ifdef LP_PROG_DEVICE_LANES
    int m = g_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("LILLIPUT_HIP_PROG_ENTROPY");
        m = e && !strcmp(e, "device") ? 1 : 0;
        g_mode.store(m, std::memory_order_relaxed);
    }
    return m != 0;
#else
    return false;
#endif
}
extern "C" void lilliput_hip_set_progressive_entropy(int on_device) { g_mode.store(on_device ? 1 : 0, std::memory_order_relaxed); }
extern "C" int lilliput_hip_progressive_device_lanes_built(void)
{
#ifdef LP_PROG_DEVICE_LANES
    return 1;
#else
    return 0;
#endif
}

void lp_prog_levels(const std::vector<LpProgScanHost>& scans, std::vector<uint32_t>& level)
{
    level.assign(scans.size(), 0);
    for (size_t a = 0; a < scans.size(); a++) {
        const LpProgScan& sa = scans[a].s;
        for (size_t b = 0; b < a; b++) {
            const LpProgScan& sb = scans[b].s;
            bool share = false;
            for (uint32_t x = 0; x < sa.ns; x++)
                for (uint32_t y = 0; y < sb.ns; y++) share = share || sa.comp[x] == sb.comp[y];
            // a sequential scan writes whole blocks whatever band its header declares (libjpeg only warns, JWRN_NOT_SEQUENTIAL):
            // it depends on, and is depended on by, every other scan of its components, in file order
            const bool overlap = sa.sequential || sb.sequential || (sa.Ss <= sb.Se && sb.Ss <= sa.Se);
            if (share && overlap) level[a] = std::max(level[a], level[b] + 1);
        }
    }
}

void lp_prog_host_run(std::vector<LpProgHostTask>& tasks, int nthreads)
{
    if (tasks.empty()) return;
    if (nthreads <= 0) {
        const char* e = getenv("LILLIPUT_HIP_PROG_THREADS");
        nthreads = e ? atoi(e) : 0;
        if (nthreads <= 0) { // a quarter of the cores (the batch front end runs up to four uploads side by side), at least min(16, cores)
            const unsigned hc = std::max(1u, std::min(std::thread::hardware_concurrency(), 4u * lp_usable_cpus_per_device())); // (the formula below takes a quarter)
            nthreads = (int)std::min(64u, std::max(std::min(16u, hc), hc / 4));
        }
    }
    // lowest level first; inside a level the longest scans first (they bound the level's finishing time)
    std::stable_sort(tasks.begin(), tasks.end(), [](const LpProgHostTask& x, const LpProgHostTask& y) {
        return x.level != y.level ? x.level < y.level : x.scan->ecs_len > y.scan->ecs_len;
    });
    size_t lo = 0;
    while (lo < tasks.size()) {
        size_t hi = lo;
        while (hi < tasks.size() && tasks[hi].level == tasks[lo].level) hi++;
        std::atomic<size_t> next{lo};
        auto worker = [&]() {
            std::vector<uint8_t> clean;
            std::vector<uint32_t> rst, words;
            for (;;) {
                const size_t k = next.fetch_add(1, std::memory_order_relaxed);
                if (k >= hi) break;
                run_task(tasks[k], clean, rst, words);
            }
        };
        // a level with little data is decoded faster than threads start: one thread per ~8 KiB of entropy-coded data (≈0.4 ms of work)
        size_t level_bytes = 0;
        for (size_t k = lo; k < hi; k++) level_bytes += tasks[k].scan->ecs_len;
        const int nt = (int)std::min<size_t>(std::min<size_t>((size_t)nthreads, hi - lo), level_bytes / 8192 + 1);
        if (nt <= 1) worker();
        else {
            std::vector<std::thread> th;
            for (int i = 1; i < nt; i++) th.emplace_back(worker);
            worker();
            for (auto& x : th) x.join();
        }
        lo = hi;
    }
}

// Test access (no device work): the coefficients of component `comp` as the hybrid mode's host threads decode them,
// [block row][block column][64 natural-order values] over the MCU-padded grid. Returns 0, or -1 (not a progressive JPEG the
// parser accepts) / -3 (dst too small).
extern "C" int lilliput_hip_progressive_coefs_host(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh, int nthreads)
{
    LpJpegHeader h;
    if (lp_jpeg_parse(static_cast<const uint8_t*>(data), len, &h) != LP_PARSE_OK || !h.scan_path || comp < 0 || comp >= h.j.ncomp) return -1;
    size_t total = 0, base = 0;
    for (int c = 0; c < h.j.ncomp; c++) {
        if (c == comp) base = total;
        total += (size_t)h.j.bw[c] * h.j.bh[c] * 64;
    }
    const size_t ne = (size_t)h.j.bw[comp] * h.j.bh[comp] * 64;
    if (ne > cap_elems) return -3;
    std::vector<int16_t> coef(total, 0);
    std::vector<uint32_t> lev;
    lp_prog_levels(h.scans, lev);
    uint32_t err = 0;
    std::vector<LpProgHostTask> tasks;
    for (size_t q = 0; q < h.scans.size(); q++) tasks.push_back(LpProgHostTask{static_cast<const uint8_t*>(data), &h.scans[q], coef.data(), lev[q], &err});
    lp_prog_host_run(tasks, nthreads);
    static const uint8_t zz[80] = LP_ZIGZAG_INIT;
    for (size_t q = 0; q < ne; q++) dst[(q & ~(size_t)63) | zz[q & 63]] = coef[base + q]; // stored in zigzag order
    *bw = (int)h.j.bw[comp];
    *bh = (int)h.j.bh[comp];
    return err ? -2 : 0;
}
