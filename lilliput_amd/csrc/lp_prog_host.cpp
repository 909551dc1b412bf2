// lp_prog_host.cpp -- see lp_prog_host.h.
#include "lp_prog_host.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include "lp_hostmem.h"
#include "lp_jbits.h"
#include "lp_prog_core.h"
#include "lp_huff_core.h" // LP_ZIGZAG_INIT

#if defined(__SSE2__)
#include <emmintrin.h>
#include "lp_abi_guard.h"
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
struct HostProgMem { // the coefficient side of lp_prog_core.h's memory policy (the bits come from LpJBits)
    int16_t* coef;
    int16_t* cur;
    bool strayed = false; // the scan stored outside its band (lp_prog_core.h lp_prog_stray): the image's result depends on the order of its scans
    void stray() { strayed = true; }
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
    // The refinement scans are two thirds of a progressive file's decode time (round 5: 34 of 49 ms for a 2048 x 2048 q90 file), most of it
    // in two walks over bit masks: "one correction bit for every non-zero coefficient of this stretch" and "the (r + 1)-th still-zero
    // coefficient from here". With BMI2 both are a PDEP: the correction bits of up to 32 coefficients are read in one piece (LpJBits::
    // get_each takes them in one shift when the register holds them), deposited onto the coefficients' positions, and only the set ones
    // are visited. Same bits read, same coefficients written as lp_prog_core.h's loops (which CPUs without BMI2 run).
    static constexpr bool kBulkCorrect = true;
    template <class B>
    void correct_bulk(B& b, uint64_t bits, int32_t p1, int32_t m1)
    {
        static const bool bmi2 = __builtin_cpu_supports("bmi2");
        if (bmi2) { correct_bmi2(b, bits, p1, m1); return; }
        while (bits) {
            uint32_t n = (uint32_t)__builtin_popcountll(bits);
            if (n > 32u) n = 32u;
            uint32_t v = b.get_each(n) << (32u - n); // first coefficient's bit on top
            for (; n; n--, v <<= 1) {
                const uint32_t e = (uint32_t)__builtin_ctzll(bits);
                bits &= bits - 1ull;
                if (v & 0x80000000u) {
                    const int32_t co = cur[e];
                    if ((co & p1) == 0) cur[e] = (int16_t)(co >= 0 ? co + p1 : co + m1);
                }
            }
        }
    }
    template <class B>
    __attribute__((target("bmi2"))) void correct_bmi2(B& b, uint64_t bits, int32_t p1, int32_t m1)
    {
        while (bits) {
            uint32_t n = (uint32_t)__builtin_popcountll(bits);
            uint64_t grp = bits;
            if (n > 32u) { n = 32u; grp = _pdep_u64(0xffffffffull, bits); } // the first 32 coefficients of the stretch
            const uint32_t v = b.get_each(n);                              // their bits, the first coefficient's at bit n - 1
            uint64_t apply = _pdep_u64((uint64_t)(__builtin_bitreverse32(v) >> (32u - n)), grp); // ... at the first coefficient's position
            bits &= ~grp;
            while (apply) {
                const uint32_t e = (uint32_t)__builtin_ctzll(apply);
                apply &= apply - 1ull;
                const int32_t co = cur[e];
                if ((co & p1) == 0) cur[e] = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
        }
    }
    // zeros with its r lowest set bits cleared (0 when it has no more than r)
    uint64_t drop_lowest(uint64_t zeros, uint32_t r) const
    {
        static const bool bmi2 = __builtin_cpu_supports("bmi2");
        if (bmi2) return drop_lowest_bmi2(zeros, r);
        for (; r && zeros; r--) zeros &= zeros - 1ull;
        return zeros;
    }
    __attribute__((target("bmi2"))) static uint64_t drop_lowest_bmi2(uint64_t zeros, uint32_t r)
    {
        return zeros & ~_pdep_u64((1ull << r) - 1ull, zeros); // r <= 15
    }
};

// One scan, the way libjpeg reads it under cv::JpegDecoder (lp_jbits.h): from the raw bytes -- stuffed zeros, fill bytes, restart
// markers with whatever numbers they carry and byte pairs that only look like markers are the reader's business -- to the marker
// that ends the scan or the end of the file, whichever the decoder meets.
void run_task(const LpProgHostTask& t)
{
    const LpProgScanHost& sh = *t.scan;
    int rc;
    static const bool timing = getenv("LILLIPUT_HIP_PROG_TIMING") != nullptr; // per-scan decode times on stderr (measurements)
    const auto t0 = std::chrono::steady_clock::now();
    struct Tm { const LpProgScanHost& sh; std::chrono::steady_clock::time_point t0; bool on;
                ~Tm() { if (on) fprintf(stderr, "[lilliput_hip] scan comps %u Ss %u Se %u Ah %u Al %u: %zu bytes %.2f ms\n", sh.s.ns, sh.s.Ss, sh.s.Se, sh.s.Ah, sh.s.Al, sh.ecs_len,
                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } tm{sh, t0, timing};
    if (sh.arith) rc = lp_arith_scan(t.data + sh.ecs_off, t.data + t.len, sh.s, sh.ar, t.coef, t.whole_file); // a QM-coded scan (lp_arith_host.h)
    else {
        LpJBits b(t.data + sh.ecs_off, t.data + t.len, &sh.tables);
        HostProgMem m{t.coef, t.coef};
        rc = lp_prog_scan_with(m, b, sh.s) ? LP_SCAN_OK : LP_SCAN_OUT_OF_DATA;
        if (m.strayed) __atomic_or_fetch(t.error, LP_PROG_HOST_STRAY, __ATOMIC_RELAXED);
        if (rc == LP_SCAN_OK && t.whole_file) rc = b.src.after_scan(sh.s.dri != 0);
    }
    if (rc == LP_SCAN_OUT_OF_DATA) __atomic_or_fetch(t.error, 8u, __ATOMIC_RELAXED);
    else if (rc == LP_SCAN_BAD_MARKER) __atomic_or_fetch(t.error, 16u, __ATOMIC_RELAXED);
}

std::atomic<int> g_mode{-2}; // -2: not read from the environment yet
} // namespace

// Where the scans of a progressive file are entropy-decoded (lp_prog_host.h): -1 auto (default), 0 host threads, 1 device, 2 device lanes.
int lp_prog_entropy_mode()
{
    int m = g_mode.load(std::memory_order_relaxed);
    if (m == -2) {
        const char* e = getenv("LILLIPUT_HIP_PROG_ENTROPY");
        m = !e ? -1 : !strcmp(e, "host") ? 0 : !strcmp(e, "device") ? 1 : !strcmp(e, "lanes") ? 2 : -1;
        g_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
uint32_t lp_prog_device_min_images()
{
    // Measured on the 16-CPU box (profiles/r06_progressive.md): the device's time for a set is its longest scan chain whatever the count
    // (29 / 115 ms at 1024 / 2048 pixels a side), the host threads' time grows with the count; they cross between 64 and 96 files at
    // every size -- about five files per host thread.
    static const uint32_t v = [] {
        const char* e = getenv("LILLIPUT_HIP_PROG_DEVICE_MIN");
        if (e && atoi(e) > 0) return (uint32_t)atoi(e);
        return std::max(16u, std::min(512u, 5u * lp_usable_cpus_per_device()));
    }();
    return v;
}
extern "C" void lilliput_hip_set_progressive_entropy(int mode) { g_mode.store(mode < -1 || mode > 2 ? -1 : mode, std::memory_order_relaxed); }
static std::atomic<uint64_t> g_prog_stats[3];
void lp_prog_count(uint64_t device_images, uint64_t gave_up, uint64_t device_scans)
{
    g_prog_stats[0] += device_images; g_prog_stats[1] += gave_up; g_prog_stats[2] += device_scans;
}
extern "C" void lilliput_hip_progressive_stats(uint64_t out[3]) { for (int i = 0; i < 3; i++) out[i] = g_prog_stats[i].load(); }
extern "C" int lilliput_hip_progressive_device_lanes_built(void) { return 1; } // (a build option in rounds 3-5; both device decoders are in every library now)

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
    // what the caller had put into the images' error words (a file without its EOI: |= 8 before any scan ran)
    std::vector<std::pair<uint32_t*, uint32_t>> preset;
    for (const LpProgHostTask& t : tasks)
        if (preset.empty() || preset.back().first != t.error) preset.emplace_back(t.error, __atomic_load_n(t.error, __ATOMIC_RELAXED));
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
            for (;;) {
                const size_t k = next.fetch_add(1, std::memory_order_relaxed);
                if (k >= hi) break;
                run_task(tasks[k]);
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
    // A scan that stored outside its band (damaged data, lp_prog_core.h lp_prog_stray): libjpeg's answer is the one of the scans taken in
    // FILE order -- a later scan that owns the coefficient overwrites or refines the stray value -- and the levels above ran them longest
    // first and side by side. Such an image is decoded again from zeroed coefficients, its scans one after the other as the file lists them
    // (found by tests/test_progressive.py on fresh seeds, round 6: one coefficient of 4 800 damaged files).
    for (const auto& pe : preset) {
        if (!(__atomic_load_n(pe.first, __ATOMIC_RELAXED) & LP_PROG_HOST_STRAY)) continue;
        std::vector<const LpProgHostTask*> mine;
        for (const LpProgHostTask& t : tasks)
            if (t.error == pe.first) mine.push_back(&t);
        std::sort(mine.begin(), mine.end(), [](const LpProgHostTask* x, const LpProgHostTask* y) { return x->scan < y->scan; }); // the scans of one header: one vector
        if (mine.front()->coef_elems) memset(mine.front()->coef, 0, mine.front()->coef_elems * sizeof(int16_t));
        __atomic_store_n(pe.first, pe.second, __ATOMIC_RELAXED);
        for (const LpProgHostTask* t : mine) run_task(*t);
        __atomic_and_fetch(pe.first, ~LP_PROG_HOST_STRAY, __ATOMIC_RELAXED); // the bit is the runner's own: the caller reads "non-zero = the image fails"
    }
}

void lp_prog_smooth(const LpJpegHeader& h, int16_t* coef)
{
    if (!h.ref_smooths) return;
    const LpJpeg& j = h.j;
    static const int nat[10] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24}; // natural positions of zigzag 0..9 (the quantisation tables are in natural order)
    size_t base = 0;
    std::vector<int32_t> dc;
    for (int c = 0; c < j.ncomp && c < 4; base += (size_t)j.bw[c] * j.bh[c] * 64, c++) {
        const int8_t* cb = h.coef_bits[c];
        bool useful = false, change_dc = true;
        for (int k = 1; k < 10; k++) { useful = useful || cb[k] != 0; change_dc = change_dc && cb[k] == -1; }
        if (!useful) continue; // (libjpeg runs the filter over every component once smoothing_ok said yes; one whose nine precisions are final changes nothing)
        const int bw = (int)j.bw[c], bh = (int)j.bh[c], v = j.ncomp == 1 ? 1 : (int)j.vs[c];
        // the component's real size in blocks (jdinput.c: width_in_blocks / height_in_blocks), without the MCU padding
        const int cw = j.ncomp == 1 ? (int)j.width : (int)(((uint64_t)j.width * j.hs[c] + j.hmax - 1) / j.hmax);
        const int ch = j.ncomp == 1 ? (int)j.height : (int)(((uint64_t)j.height * j.vs[c] + j.vmax - 1) / j.vmax);
        const int wib = (cw + 7) / 8, hib = (ch + 7) / 8, last_imcu = bh / v - 1;
        int16_t* co = coef + base;
        dc.resize((size_t)bw * bh);
        for (size_t b = 0; b < dc.size(); b++) dc[b] = co[b * 64];
        int64_t Q[10];
        for (int k = 0; k < 10; k++) Q[k] = j.qt[c][nat[k]];
        for (int y = 0; y < hib; y++) {
            // rows below: the last iMCU row only sees its real rows (the bottom one is repeated), the rows above it see the whole of the
            // next iMCU rows -- the dummy rows of the padding included (decompress_smooth_data's access_rows)
            const int ylim = (y / v == last_imcu ? hib : bh) - 1;
            const int ry[5] = {std::max(y - 2, 0), std::max(y - 1, 0), y, std::min(y + 1, ylim), std::min(y + 2, ylim)};
            for (int x = 0; x < wib; x++) {
                const int rx[5] = {std::max(x - 2, 0), std::max(x - 1, 0), x, std::min(x + 1, wib - 1), std::min(x + 2, wib - 1)};
                int64_t D[26];
                for (int a = 0; a < 5; a++)
                    for (int b = 0; b < 5; b++) D[1 + 5 * a + b] = dc[(size_t)ry[a] * bw + rx[b]];
                int16_t* ws = co + ((size_t)y * bw + x) * 64;
                auto est = [&](int k, int64_t sum) {
                    const int Al = cb[k];
                    if (Al == 0 || ws[k] != 0) return;
                    const int64_t num = Q[0] * sum, q = Q[k];
                    int64_t pred = ((q << 7) + (num >= 0 ? num : -num)) / (q << 8);
                    if (Al > 0 && pred >= (1 << Al)) pred = (1 << Al) - 1;
                    ws[k] = (int16_t)(num >= 0 ? pred : -pred);
                };
                if (change_dc) {
                    est(1, -D[1] - D[2] + D[4] + D[5] - 3 * D[6] + 13 * D[7] - 13 * D[9] + 3 * D[10] - 3 * D[11] + 38 * D[12] - 38 * D[14] + 3 * D[15] - 3 * D[16] +
                               13 * D[17] - 13 * D[19] + 3 * D[20] - D[21] - D[22] + D[24] + D[25]);
                    est(2, -D[1] - 3 * D[2] - 3 * D[3] - 3 * D[4] - D[5] - D[6] + 13 * D[7] + 38 * D[8] + 13 * D[9] - D[10] + D[16] - 13 * D[17] - 38 * D[18] -
                               13 * D[19] + D[20] + D[21] + 3 * D[22] + 3 * D[23] + 3 * D[24] + D[25]);
                    est(3, D[3] + 2 * D[7] + 7 * D[8] + 2 * D[9] - 5 * D[12] - 14 * D[13] - 5 * D[14] + 2 * D[17] + 7 * D[18] + 2 * D[19] + D[23]);
                    est(4, -D[1] + D[5] + 9 * D[7] - 9 * D[9] - 9 * D[17] + 9 * D[19] + D[21] - D[25]);
                    est(5, 2 * D[7] - 5 * D[8] + 2 * D[9] + D[11] + 7 * D[12] - 14 * D[13] + 7 * D[14] + D[15] + 2 * D[17] - 5 * D[18] + 2 * D[19]);
                    est(6, D[7] - D[9] + 2 * D[12] - 2 * D[14] + D[17] - D[19]);
                    est(7, D[7] - 3 * D[8] + D[9] - D[17] + 3 * D[18] - D[19]);
                    est(8, D[7] - D[9] - 3 * D[12] + 3 * D[14] + D[17] - D[19]);
                    est(9, D[7] + 2 * D[8] + D[9] - D[17] - 2 * D[18] - D[19]);
                    // the DC itself: always replaced, no clamp
                    const int64_t num = Q[0] * (-2 * D[1] - 6 * D[2] - 8 * D[3] - 6 * D[4] - 2 * D[5] - 6 * D[6] + 6 * D[7] + 42 * D[8] + 6 * D[9] - 6 * D[10] - 8 * D[11] +
                                                42 * D[12] + 152 * D[13] + 42 * D[14] - 8 * D[15] - 6 * D[16] + 6 * D[17] + 42 * D[18] + 6 * D[19] - 6 * D[20] - 2 * D[21] -
                                                6 * D[22] - 8 * D[23] - 6 * D[24] - 2 * D[25]);
                    const int64_t pred = ((Q[0] << 7) + (num >= 0 ? num : -num)) / (Q[0] << 8);
                    ws[0] = (int16_t)(num >= 0 ? pred : -pred);
                } else { // T.81 K.8's estimates on a 5 x 5 window
                    est(1, -7 * D[11] + 50 * D[12] - 50 * D[14] + 7 * D[15]);
                    est(2, -7 * D[3] + 50 * D[8] - 50 * D[18] + 7 * D[23]);
                    est(3, -D[3] + 13 * D[8] - 24 * D[13] + 13 * D[18] - D[23]);
                    est(4, D[10] + D[16] - 10 * D[17] + 10 * D[19] - D[2] - D[20] + D[22] - D[24] + D[4] - D[6] + 10 * D[7] - 10 * D[9]);
                    est(5, -D[11] + 13 * D[12] - 24 * D[13] + 13 * D[14] - D[15]);
                }
            }
        }
    }
}

// Test access (no device work): the coefficients of component `comp` as the hybrid mode's host threads decode them,
// [block row][block column][64 natural-order values] over the MCU-padded grid. Returns 0, or -1 (not a progressive JPEG the
// parser accepts) / -3 (dst too small).
// nthreads < 0: the serial route of ANY sequential file (lp_jpeg_parse_opts force_scans: what a baseline stream the device decoder
// flagged as irregular is decoded by), on -nthreads threads. Returns -2 when the reference's decoder fails on the file (out of data,
// unknown marker behind a scan of a multi-scan file), with the coefficients as far as they were decoded.
static int progressive_coefs(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh, int nthreads, bool smooth)
{
    LpJpegHeader h;
    const bool force = nthreads < 0;
    if (force) nthreads = -nthreads;
    if (lp_jpeg_parse_opts(static_cast<const uint8_t*>(data), len, &h, force) != LP_PARSE_OK || !h.scan_path || comp < 0 || comp >= h.j.ncomp) return -1;
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
    uint32_t err = h.decode_fails ? 8u : 0u;
    std::vector<LpProgHostTask> tasks;
    for (size_t q = 0; q < h.scans.size(); q++) tasks.push_back(LpProgHostTask{static_cast<const uint8_t*>(data), len, &h.scans[q], coef.data(), lev[q], &err, !h.one_pass, total});
    lp_prog_host_run(tasks, nthreads);
    if (smooth) lp_prog_smooth(h, coef.data());
    static const uint8_t zz[80] = LP_ZIGZAG_INIT;
    for (size_t q = 0; q < ne; q++) dst[(q & ~(size_t)63) | zz[q & 63]] = coef[base + q]; // stored in zigzag order
    *bw = (int)h.j.bw[comp];
    *bh = (int)h.j.bh[comp];
    return err ? -2 : 0;
}
extern "C" int lilliput_hip_progressive_coefs_host(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh, int nthreads)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    return progressive_coefs(data, len, comp, dst, cap_elems, bw, bh, nthreads, false);
}
LP_ABI_CATCH("lilliput_hip_progressive_coefs_host", return -1)
// ... and as they reach the IDCT: behind libjpeg's interblock smoothing where the file calls for it (lp_prog_smooth; a no-op for the others)
extern "C" int lilliput_hip_progressive_coefs_smoothed(const void* data, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh)
try {
    return progressive_coefs(data, len, comp, dst, cap_elems, bw, bh, 1, true);
}
LP_ABI_CATCH("lilliput_hip_progressive_coefs_smoothed", return -1)

// Test access: does libjpeg's interblock smoothing change the pixels the reference returns for this file (LpJpegHeader::ref_smooths)?
// 1 / 0; -1: not a JPEG the parser takes. The product does not restate that filter.
extern "C" int lilliput_hip_jpeg_reference_smooths(const void* data, size_t len)
try {
    LpJpegHeader h;
    if (lp_jpeg_parse(static_cast<const uint8_t*>(data), len, &h) != LP_PARSE_OK) return -1;
    return h.ref_smooths ? 1 : 0;
}
LP_ABI_CATCH("lilliput_hip_jpeg_reference_smooths", return -1)
