// lp_batch.cpp -- Part B of include/lilliput_hip.h: the batched JPEG -> (orientation, Fit/Resize) -> JPEG
// entry point. Semantics per item are those of ImageOps.Transform for a static JPEG source
// (/root/reference/ops.go:352-479, opencv.go:326-374, 816-900); the images of a batch are independent,
// so a multi-GPU caller simply gives each device's batch object its own shard of the items.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include <algorithm>
#include <map>
#include <vector>

#include "lp_abi.h"
#include "lp_coalesce.h"
#include "lp_ops_logic.h"
#include "lp_abi_guard.h"
#include "lp_jpeg_parse.h"
#include "lp_pxm.h"
#include "lp_prog_host.h"

// One engine (= one compute stream + one copy stream + its arenas) per worker; a batch is split into contiguous parts, one per
// worker, and the workers run concurrently on host threads so that one part's HBM-bound stages (IDCT, resample, unstuff) overlap
// another part's latency-bound Huffman stages on the same GPU.
struct LpBatchPart {
    std::unique_ptr<LpEngine> eng;
    std::vector<LpJpegHeader> hdrs;     // parsed headers of this part's items (upload order) -- resident (staged) API
    std::vector<int> items;             // their indices in the item array
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rounds = 0;
    double tw[6] = {0, 0, 0, 0, 0, 0};
    double stage_ms = 0, stall_ms = 0;  // pipelined transform: host time spent staging (parse + memcpy + enqueue), compute thread waiting for a staged chunk
    size_t staged_bytes = 0;            // entropy-coded bytes that reached the device in the last transform ...
    size_t copied_bytes = 0;            // ... of them: through the engine's pinned slots (a host memcpy each)
    size_t direct_bytes = 0;            // ... of them: read by the DMA engine from the caller's own (pinned or registered) pages
    double register_ms = 0;             // host ms inside hipHostRegister for this part's chunks
    std::string err;                    // written by the compute thread only (the stager reports through its job)
    int rc = 0;                         // resident API: the part's run failed
    size_t failed_chunks = 0;           // pipelined transform: chunks whose items were failed as a group
};

struct LpOtherItem { int item; const uint8_t* data; size_t len; size_t dst_cap; std::vector<uint8_t> copy; };

// Long-lived helper threads of a batch for the items the one-image path serves (PNG, GIF, WebP, handed-over pixels): a worker keeps
// its per-thread engine (stream + arenas) from one batch to the next instead of building and tearing one down per call.
struct LpWorkerPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable wake, idle;
    std::function<void(size_t)> job;
    size_t gen = 0, want = 0, running = 0;
    bool stop = false;
    void body(size_t wi)
    {
        size_t seen = 0;
        for (;;) {
            std::function<void(size_t)> f;
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return stop || (gen != seen && wi < want); });
                if (stop) return;
                seen = gen;
                f = job;
            }
            f(wi + 1); // worker 0 is the calling thread
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) idle.notify_all();
            }
        }
    }
    // runs f(1 .. n) on the pool and f(0) on the caller; returns when all are done
    void run(size_t n, const std::function<void(size_t)>& f)
    {
        if (n) {
            std::lock_guard<std::mutex> lk(m);
            while (th.size() < n) { const size_t wi = th.size(); th.emplace_back([this, wi] { body(wi); }); }
            job = f; want = n; running = n; gen++;
        }
        if (n) wake.notify_all();
        f(0);
        if (n) {
            std::unique_lock<std::mutex> lk(m);
            idle.wait(lk, [&] { return running == 0; });
        }
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(m); if (stop) return; stop = true; }
        wake.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
    LpWorkerPool() { registry(this, true); }
    ~LpWorkerPool() { shutdown(); registry(this, false); }
    // Pools that are still alive when the process exits are stopped from an atexit handler registered after the HIP runtime's own, so
    // it runs before it: a worker's per-thread engine frees its stream and arenas while the runtime is still there (a thread that is
    // torn down during or after the runtime's shutdown was seen to abort the process once in a handful of runs).
    static void registry(LpWorkerPool* p, bool add)
    {
        static std::mutex rm;
        static std::vector<LpWorkerPool*>* live = new std::vector<LpWorkerPool*>(); // never destroyed: used from atexit
        static bool hooked = false;
        std::lock_guard<std::mutex> lk(rm);
        if (add) {
            live->push_back(p);
            if (!hooked) {
                hooked = true;
                atexit([] { registry(nullptr, false); });
            }
        } else if (p) {
            live->erase(std::remove(live->begin(), live->end(), p), live->end());
        } else {
            for (LpWorkerPool* q : *live) q->shutdown();
        }
    }
};

struct LpBatch {
    int device = 0;
    std::vector<LpBatchPart> parts;
    std::vector<int> parse_status;      // LILLIPUT_* per uploaded item
    // results of the last run, indexed like the item array
    std::vector<int> status, out_w, out_h;
    std::vector<uint32_t> out_len;
    std::vector<std::vector<uint8_t>> out_bytes;   // host copies fetched during run (download hands them to the caller)
    size_t n_items = 0;
    uint32_t S = 0, C = 0;
    LpTimings tm = {};
    double last_stage_ms = 0, last_stall_ms = 0, last_wall_ms = 0, last_register_ms = 0;
    size_t last_staged_bytes = 0, last_copied_bytes = 0, last_direct_bytes = 0;
    int last_numa_node = -1;
    // sources other than JPEG that the one-image path can serve (GIF: first frame through the animated composite path; PNG):
    // transformed one by one on a few host workers while the JPEG parts run
    std::vector<LpOtherItem> other;
    std::vector<lilliput_image_ops> other_ops; // one per worker
    std::vector<std::unique_ptr<LpEngine>> other_eng; // ... and its engine: a worker keeps stream and arenas from item to item and call to call
                                                      // (the pool of the one-image ABI keeps only a few idle engines: 32 workers cycling through
                                                      // it built and tore down an engine per item)
    LpWorkerPool pool;
    std::mutex retry_mu;
    std::vector<int> retry;                     // baseline items the device decoder gave up on (too few blocks): decoded again libjpeg's way
    hipStream_t shared_copy = nullptr;         // the pipelined transform's H2D copies: one queue, so chunks arrive in the order they were claimed
    hipStream_t extra_copy[3] = {nullptr, nullptr, nullptr}; // more queues for the many per-source copies of zero-copy ingest (see LpEngine::upload_commit)
    int n_extra_copy = 0;
    bool stage_timing = true;                  // the resident run records the engines' stage events (lilliput_hip_batch_timings); off for the batch of one a lone Part A chain runs as
    int node_index = 0;                        // position among the devices of a lilliput_hip_node (trace output)
    size_t last_images = 0;                    // images this device's engines served in the last transform
    size_t last_chunks = 0, last_stolen = 0;   // (devs[0] of a call) chunks of the last transform, and how many a device took from another device's share
    ~LpBatch() { for (auto o : other_ops) if (o) lilliput_image_ops_close(o); if (shared_copy) { (void)hipStreamSynchronize(shared_copy); (void)hipStreamDestroy(shared_copy); } for (auto q : extra_copy) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); } }
    LpEngine& eng0() { return *parts[0].eng; }
    bool ensure_parts(size_t n)
    {
        while (parts.size() < n) {
            LpBatchPart p;
            p.eng.reset(new LpEngine(device));
            if (!p.eng->ok()) { lp_set_error(p.eng->last_error()); return false; }
            p.eng->set_subsequence(S, C);
            parts.push_back(std::move(p));
        }
        return true;
    }
};

static int map_parse(int rc)
{
    switch (rc) {
    case LP_PARSE_OK: return LILLIPUT_OK;
    case LP_PARSE_UNSUPPORTED: return LILLIPUT_ERR_UNSUPPORTED;
    default: return LILLIPUT_ERR_INVALID_IMAGE;
    }
}

static int map_status(int st)
{
    switch (st) {
    case LP_OK: return LILLIPUT_OK;
    case LP_ERR_INVALID_IMAGE: return LILLIPUT_ERR_INVALID_IMAGE;
    case LP_ERR_DECODE_FAILED: return LILLIPUT_ERR_DECODING_FAILED;
    case LP_ERR_BUF_TOO_SMALL: return LILLIPUT_ERR_BUF_TOO_SMALL;
    case LP_ERR_UNSUPPORTED: return LILLIPUT_ERR_UNSUPPORTED;
    default: return LILLIPUT_ERR_DEVICE;
    }
}

// The reference sizes its frame buffers with NewImageOps(maxSize) and answers ErrBufTooSmall for anything larger
// (opencv.go:250-267 resizeMat); the batch has the same bound -- 8192 x 8192 pixels unless LILLIPUT_HIP_BATCH_MAX_PIXELS says
// otherwise -- so that one file with an absurd frame header cannot take the other images' arenas down with it.
static uint64_t batch_max_pixels()
{
    static const uint64_t v = getenv("LILLIPUT_HIP_BATCH_MAX_PIXELS") ? strtoull(getenv("LILLIPUT_HIP_BATCH_MAX_PIXELS"), nullptr, 10) : 8192ull * 8192ull;
    return v;
}

static bool is_other_format(const uint8_t* sp, size_t len)
{
    static const uint8_t png_sig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
    return sp && ((len >= 6 && (memcmp(sp, "GIF87a", 6) == 0 || memcmp(sp, "GIF89a", 6) == 0)) || // lilliput.go:100-102 isGIF
                  (len >= 8 && memcmp(sp, png_sig, 8) == 0) ||
                  (len >= 2 && sp[0] == 'B' && sp[1] == 'M') || // what cv::findDecoder takes for a BMP: the OpenCV decoder's path (lp_bmp.h)
                  lp_pxm_signature(sp, len) ||                   // ... and for PBM / PGM / PPM (lp_pxm.h)
                  (len >= 12 && memcmp(sp, "RIFF", 4) == 0 && memcmp(sp + 8, "WEBP", 4) == 0) || // isWebp, lilliput.go:104-115
                  (len >= sizeof(lilliput_hip_pixels_header) && memcmp(sp, LILLIPUT_HIP_PIXELS_MAGIC, 8) == 0)); // frames a host decoder handed over
}

// Scan-path sources (progressive, multi-scan ...) are entropy-decoded by host threads into a pinned int16 buffer whose size follows
// from the frame header alone -- a tiny file claiming 8192 x 8192 4:4:4 asks for 400 MB. One upload set never pins more than
// LILLIPUT_HIP_PROG_PINNED_MAX bytes (default 4 GiB) of it: the items that would go beyond answer ErrBufTooSmall, the others are untouched.
static size_t prog_pinned_max()
{
    static const size_t v = getenv("LILLIPUT_HIP_PROG_PINNED_MAX") ? (size_t)strtoull(getenv("LILLIPUT_HIP_PROG_PINNED_MAX"), nullptr, 10) : (size_t)4 << 30;
    return v;
}
// ... and when the set's scans are decoded on the device (a call with many progressive files, lp_prog_host.h) nothing is pinned: the bound is the
// device's coefficient arena, LILLIPUT_HIP_PROG_DEVICE_MAX bytes per set (default 16 GiB of the 288: 340 files of 4096 x 4096 -- the wave decoder's
// launches last as long as their longest chain of scans whatever they hold, so the more files a set holds the better).
static size_t prog_device_max()
{
    static const size_t v = getenv("LILLIPUT_HIP_PROG_DEVICE_MAX") ? (size_t)strtoull(getenv("LILLIPUT_HIP_PROG_DEVICE_MAX"), nullptr, 10) : (size_t)16 << 30;
    return std::max(v, prog_pinned_max());
}

// Header walk of one JPEG item + the batch's size bounds. Returns LILLIPUT_OK when the item goes to the device. *pinned accumulates the
// host coefficient bytes of the set the item joins.
static int parse_item(const void* src, size_t len, LpJpegHeader* h, size_t pinned[2], int parsed_rc = -1000, size_t scan_path_bound = 0)
{
    const int rc = parsed_rc != -1000 ? parsed_rc : (src && len) ? lp_jpeg_parse((const uint8_t*)src, len, h) : LP_PARSE_NOT_JPEG; // (parsed_rc: the walk was done already, see pipe_stager)
    if (rc != LP_PARSE_OK) return map_parse(rc);
    if ((uint64_t)h->j.width * h->j.height > batch_max_pixels()) return LILLIPUT_ERR_BUF_TOO_SMALL;
    if (h->scan_path) {
        size_t need = 0;
        for (int c = 0; c < h->j.ncomp; c++) need += (size_t)h->j.bw[c] * h->j.bh[c] * 128;
        // (the larger bound of a set whose progressive files go to the device is for those files: QM-coded and sequential multi-scan ones stay with the host threads)
        const bool wave_file = scan_path_bound && !h->arith && lp_jpeg_sniff_progressive((const uint8_t*)src, len);
        size_t& held = pinned[wave_file ? 1 : 0];  // [0] what the host threads will decode into pinned memory, [1] the wave decoder's files
        if (held + need > (wave_file ? scan_path_bound : prog_pinned_max())) return LILLIPUT_ERR_BUF_TOO_SMALL;
        held += need;
    }
    return LILLIPUT_OK;
}

extern "C" {

lilliput_hip_batch lilliput_hip_batch_create(int device)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = new LpBatch();
    b->device = device;
    if (!b->ensure_parts(1)) { delete b; return nullptr; }
    return b;
}
LP_ABI_CATCH("lilliput_hip_batch_create", return nullptr)

void lilliput_hip_batch_destroy(lilliput_hip_batch b) { delete static_cast<LpBatch*>(b); }

void lp_batch_set_stage_timing(lilliput_hip_batch bb, bool on) { if (bb) static_cast<LpBatch*>(bb)->stage_timing = on; }

void lilliput_hip_batch_set_subsequence(lilliput_hip_batch bb, unsigned S, unsigned C)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    b->S = S; b->C = C;
    for (auto& p : b->parts) p.eng->set_subsequence(S, C);
}
LP_ABI_CATCH("lilliput_hip_batch_set_subsequence", return)

int lilliput_hip_batch_resident_round(lilliput_hip_batch bb, size_t max_src_len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    static const int round_env = getenv("LILLIPUT_HIP_RESIDENT_CHUNK") ? atoi(getenv("LILLIPUT_HIP_RESIDENT_CHUNK")) : 0;
    if (round_env > 0) return round_env;
    if (!b || b->parts.empty() || !b->parts[0].eng) return 112;
    return (int)b->parts[0].eng->resident_round(max_src_len);
}
LP_ABI_CATCH("lilliput_hip_batch_resident_round", return 0)

void lilliput_hip_batch_timings(lilliput_hip_batch bb, float out_ms[10], int* verify_rounds)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    const LpTimings& t = static_cast<LpBatch*>(bb)->tm;
    out_ms[6] = t.huff_spec_ms; out_ms[7] = t.huff_verify_ms; out_ms[8] = t.huff_scan_ms; out_ms[9] = t.huff_write_ms;
    out_ms[0] = t.unstuff_ms; out_ms[1] = t.huff_ms; out_ms[2] = t.idct_ms; out_ms[3] = t.color_ms; out_ms[4] = t.resize_ms; out_ms[5] = t.encode_ms;
    if (verify_rounds) *verify_rounds = (int)t.verify_rounds;
}
LP_ABI_CATCH("lilliput_hip_batch_timings", return)

void lilliput_hip_batch_ingest_stats(lilliput_hip_batch bb, double out[4])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    out[0] = (double)b->last_staged_bytes; out[1] = b->last_stage_ms; out[2] = b->last_stall_ms; out[3] = b->last_wall_ms;
}
LP_ABI_CATCH("lilliput_hip_batch_ingest_stats", return)

void lilliput_hip_batch_ingest_stats2(lilliput_hip_batch bb, double out[8])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    out[0] = (double)b->last_staged_bytes; out[1] = b->last_stage_ms; out[2] = b->last_stall_ms; out[3] = b->last_wall_ms;
    out[4] = (double)b->last_copied_bytes; out[5] = (double)b->last_direct_bytes; out[6] = b->last_register_ms; out[7] = (double)b->last_numa_node;
}
LP_ABI_CATCH("lilliput_hip_batch_ingest_stats2", return)

} // extern "C"

// Engines a call is split over. resident_items > 0: the resident form (upload / run / download) with that many items -- a large set takes
// eight engines, one launch each: 1 024 sources of 4096 x 4096 run at 21.1 k images/s as 8 x 128 against 19.5 k as 4 x (113 + 113 + 30)
// (profiles/r06_resident.md: a launch pays for its slowest verify lane and its ramps once, whatever it holds, and the other engines'
// kernels fill what a launch beyond one round of WRITE workgroups leaves idle). The pipelined transform keeps four: it is bound by the
// link, and more engines cost it staging slots (12.1 k -> 10.7 k with eight).
static int batch_streams(int requested, size_t resident_items = 0)
{
    int n = requested;
    if (n <= 0) { const char* e = getenv("LILLIPUT_HIP_STREAMS"); n = e ? atoi(e) : (resident_items >= 512 ? 8 : 4); }
    return std::max(1, std::min(8, n));
}

extern "C" int lilliput_hip_batch_upload2(lilliput_hip_batch bb, const lilliput_batch_item* items, size_t n, int streams)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    if (!b) return LILLIPUT_ERR_DEVICE;
    b->n_items = n;
    b->parse_status.assign(n, LILLIPUT_OK);
    b->other.clear();
    std::vector<int> valid;
    std::vector<LpJpegHeader> hv;
    size_t pinned[2] = {0, 0}; // the whole upload shares the bound (its parts are cut afterwards)
    // ... the bound of what host threads decode into pinned memory at upload time. A set whose progressive files go to the device's wave decoder
    // (lp_prog_host.h: forced, or enough of them in the upload) pins nothing and keeps only compressed bytes resident -- the run's launches bound
    // their own arenas (auto_chunk) -- so it may hold any number of them.
    uint32_t nprog = 0;
    const int prog_mode = lp_prog_entropy_mode();
    if (prog_mode != 0)
        for (size_t i = 0; i < n; i++) nprog += lp_jpeg_sniff_progressive((const uint8_t*)items[i].src, items[i].src_len) ? 1u : 0u;
    const bool prog_on_device = prog_mode > 0 || (prog_mode < 0 && nprog >= lp_prog_device_min_images());
    const size_t scan_path_bound = prog_on_device ? (size_t)1 << 60 : 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* sp = (const uint8_t*)items[i].src;
        if (is_other_format(sp, items[i].src_len)) {
            LpOtherItem o{(int)i, nullptr, items[i].src_len, items[i].dst_cap, std::vector<uint8_t>(sp, sp + items[i].src_len)};
            b->other.push_back(std::move(o));
            continue;
        }
        LpJpegHeader h;
        b->parse_status[i] = parse_item(items[i].src, items[i].src_len, &h, pinned, -1000, scan_path_bound);
        if (b->parse_status[i] == LILLIPUT_OK) { valid.push_back((int)i); hv.push_back(h); }
    }
    for (auto& o : b->other) o.data = o.copy.data();
    // contiguous parts, one per worker; small batches stay on one stream
    size_t np = (size_t)batch_streams(streams, valid.size());
    if (valid.size() < 2 * np) np = 1;
    if (!b->ensure_parts(np)) return LILLIPUT_ERR_DEVICE;
    for (auto& p : b->parts) { p.hdrs.clear(); p.items.clear(); }
    for (size_t q = 0; q < valid.size(); q++) {
        LpBatchPart& p = b->parts[q * np / valid.size()];
        p.items.push_back(valid[q]);
        p.hdrs.push_back(hv[q]);
    }
    for (size_t k = 0; k < np; k++) {
        LpBatchPart& p = b->parts[k];
        if (p.items.empty()) continue;
        std::vector<LpJpegSrc> srcs;
        for (int it : p.items) srcs.push_back(LpJpegSrc{(const uint8_t*)items[it].src, items[it].src_len});
        p.eng->set_progressive_in_call(nprog); // every part takes the route the whole upload was admitted for
        int rc = p.eng->upload_jpegs(srcs.data(), (int)srcs.size(), p.hdrs.data());
        if (rc) { lp_set_error(p.eng->last_error()); return map_status(rc); }
    }
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_hip_batch_upload2", return LILLIPUT_ERR_DEVICE)

extern "C" int lilliput_hip_batch_upload(lilliput_hip_batch bb, const lilliput_batch_item* items, size_t n) { return lilliput_hip_batch_upload2(bb, items, n, 0); }

// Where the bytes of a finished item go: the resident API keeps them for download(), the pipelined transform writes the caller's buffer.
struct LpSink {
    LpBatch* b;
    lilliput_batch_item* items;     // nullptr: resident API
    void put(size_t item, const uint8_t* p, uint32_t len, int w, int h) const
    {
        b->out_w[item] = w; b->out_h[item] = h; b->out_len[item] = len;
        if (!items) { b->out_bytes[item].assign(p, p + len); return; }
        if (len > items[item].dst_cap || !items[item].dst) { b->status[item] = LILLIPUT_ERR_BUF_TOO_SMALL; return; }
        memcpy(items[item].dst, p, len);
    }
};

// The integer-scale op of one image: where the box of destination (0, 0) lies in the un-oriented decoded image and how it moves with
// the destination coordinates (cv::ExifTransform inverse folded into the addressing). Everything but the image index and dst.off.
static void fused_op_of(const LpJpeg& j, const LpOpsPlan& plan, int ix, int iy, LpFusedOp* out)
{
    const bool swap = j.orientation >= 5;
    const int OW = swap ? (int)j.height : (int)j.width, OH = swap ? (int)j.width : (int)j.height;
    // oriented-frame rectangle of destination (dx, dy) -> rectangle of the un-oriented decoded image
    auto map = [&](int dx, int dy, int* fx, int* fy) {
        const int ox0 = plan.crop_x + dx * ix, oy0 = plan.crop_y + dy * iy;
        switch (j.orientation) {
        case 2: *fx = OW - ox0 - ix; *fy = oy0; break;
        case 3: *fx = OW - ox0 - ix; *fy = OH - oy0 - iy; break;
        case 4: *fx = ox0; *fy = OH - oy0 - iy; break;
        case 5: *fx = oy0; *fy = ox0; break;
        case 6: *fx = oy0; *fy = OW - ox0 - ix; break;
        case 7: *fx = OH - oy0 - iy; *fy = OW - ox0 - ix; break;
        case 8: *fx = OH - oy0 - iy; *fy = ox0; break;
        default: *fx = ox0; *fy = oy0; break;
        }
    };
    LpFusedOp& op = *out;
    memset(&op, 0, sizeof(op));
    op.rw = (uint32_t)(swap ? iy : ix);
    op.rh = (uint32_t)(swap ? ix : iy);
    int ax, ay, bx, by;
    map(0, 0, &op.x0, &op.y0);
    map(1, 0, &ax, &ay);
    map(0, 1, &bx, &by);
    op.dxx = ax - op.x0; op.dxy = ay - op.y0; op.dyx = bx - op.x0; op.dyy = by - op.y0;
    op.inv_area = 1.f / (float)(ix * iy);
    op.round_2x2 = (ix == 2 && iy == 2) ? 1 : 0;
    op.dst.w = (uint32_t)plan.out_w; op.dst.h = (uint32_t)plan.out_h; op.dst.cn = j.ncomp == 1 ? 1 : 3;
    op.dst.stride = op.dst.w * op.dst.cn;
}

// Test access (no device work): which kernel the batch path hands an image of this shape to on its way from decoded planes to the
// output size. 0 = through a materialised frame (no resize, CMYK, unusual sampling, grey at a fractional scale ...), 1 = k_resample_420
// (8 / 16 / 32-pixel boxes), 2 = k_resample_420_small (2 / 4), 3 = k_resample_hv1 (4:4:4 / 4:2:2), 4 = k_resample_gray, 5 = the area walk
// with cv::resize's float taps (fractional scales), 6 = the area walk with unit taps (the other integer scales), 7 = k_resample_fused's
// wave per destination pixel (what is left). tests/test_host_logic.py holds the shapes a service sees to routes 1 - 6: round 5 found
// every integer scale but 8 / 16 / 32 on route 7 at 30 - 58 us per image.
extern "C" int lilliput_hip_resample_route(int width, int height, int orientation, int ncomp, int hs, int vs, int out_w, int out_h, int resize_method,
                                           int normalize_orientation)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    LpJpeg j;
    memset(&j, 0, sizeof(j));
    j.width = (uint32_t)width; j.height = (uint32_t)height; j.orientation = (uint8_t)orientation; j.ncomp = (uint8_t)ncomp;
    j.colorspace = ncomp == 1 ? 1 : 2;
    j.hs[0] = (uint8_t)hs; j.vs[0] = (uint8_t)vs; j.hmax = (uint8_t)hs; j.vmax = (uint8_t)vs;
    for (int c = 1; c < ncomp && c < LP_GEOM_COMP; c++) { j.hs[c] = 1; j.vs[c] = 1; }
    const bool swap = j.orientation >= 5;
    const int OW = swap ? height : width, OH = swap ? width : height;
    const LpOpsPlan plan = lp_plan_static_transform(width, height, orientation, out_w, out_h, resize_method, normalize_orientation != 0, OW, OH);
    if (!plan.resize || j.ncomp == 4) return 0;
    int ix = 1, iy = 1;
    const int mode = lp_resize_mode(plan.crop_w, plan.crop_h, plan.out_w, plan.out_h, &ix, &iy);
    if (mode == 2) {
        if (lp_area_sampling(j) < 0) return 0;
        return (swap ? lp_area420_bucket(plan.crop_h, plan.out_h, true) : lp_area420_bucket(plan.crop_w, plan.out_w)) ? 5 : 0;
    }
    if (mode != 1) return 0;
    LpFusedOp op;
    fused_op_of(j, plan, ix, iy, &op);
    uint32_t fast = 0;
    if (lp_fused_op_is_fast(op, j, &fast)) return fast == 0x1000u ? 4 : fast >= 0x100u ? 3 : (op.rw == 2 || op.rw == 4) ? 2 : 1;
    if (lp_area_sampling(j) >= 0 && lp_area420_bucket_int(swap ? iy : ix, swap) != 0) return 6;
    return 7;
}
LP_ABI_CATCH("lilliput_hip_resample_route", return -1)

static size_t auto_chunk(const lilliput_batch_options* opt, const LpJpegHeader* hdrs, size_t n, size_t cap)
{
    // chunk size: bound the working set (coefficients + planes + BGR frame ~ 7.5 B/pixel + oriented copy)
    size_t max_px = 1;
    for (size_t i = 0; i < n; i++) max_px = std::max(max_px, (size_t)hdrs[i].j.mcus_x * hdrs[i].j.hmax * 8 * hdrs[i].j.mcus_y * hdrs[i].j.vmax * 8);
    return opt->chunk > 0 ? (size_t)opt->chunk : std::max<size_t>(1, std::min<size_t>(cap, (size_t)(24ull << 30) / (max_px * 12)));
}

// Every device stage of ImageOps.Transform for images [first, first + cnt) of the engine's selected upload set: hdrs / item_of are
// indexed from `first` (hdrs[k] belongs to image first + k). Runs on the part's own host thread.
// deferred: the decode is only enqueued and collected after the encode, so the chunk runs without a host round trip between its
// stages (the verify rounds, the resample and the encode are queued behind each other on the engine's stream); the rare chunk whose
// entropy streams need more verify rounds than were enqueued is run again the plain way.
static int run_chunk(LpBatch* b, LpBatchPart& part, int first, int cnt, const LpJpegHeader* hdrs, const int* item_of, const lilliput_batch_options* opt,
                     const LpSink& sink, bool deferred = true)
{
    LpEngine& eng = *part.eng;
    float* acc = part.acc;
    auto fail = [&](const std::string& m) { part.err = m; return LILLIPUT_ERR_DEVICE; };
    const int quality = opt->jpeg_quality > 0 ? opt->jpeg_quality : 95;
    uint32_t& rounds = part.rounds;
    double* tw = part.tw;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](int k) { const double t = now(); tw[k] += t - t_prev; t_prev = t; };
    {
        // How an image gets from planes to its output size: 1 = integer-scale area resize, fused (k_resample_*); 2 = fractional area
        // resize of a YCbCr 4:2:0 / 4:2:2 / 4:4:4 image, fused (k_area_420 / k_area_420t; LILLIPUT_HIP_AREA_FUSED=0 sends these through the frame as
        // before round 3); 0 = through a materialised BGR frame, exactly like the one-image ABI.
        static const bool area_on = !(getenv("LILLIPUT_HIP_AREA_FUSED") && atoi(getenv("LILLIPUT_HIP_AREA_FUSED")) == 0);
        std::map<std::pair<int, int>, uint32_t> bucket_cache;
        auto route_of = [&](const LpJpeg& j, const LpOpsPlan& plan) -> int {
            if (!plan.resize || j.ncomp == 4 || j.generic_sampling) return 0; // CMYK / YCCK, unusual sampling factors: the fused kernels read grey / YCbCr / RGB planes
            int ix, iy;
            const int mode = lp_resize_mode(plan.crop_w, plan.crop_h, plan.out_w, plan.out_h, &ix, &iy);
            if (mode == 1) return 1;
            if (mode != 2 || !area_on || lp_area_sampling(j) < 0) return 0;
            // the axis of the oriented crop that runs along source x picks the instantiation: x for orientations 1-4, y for 5-8
            const bool swapped = j.orientation >= 5;
            const auto key = swapped ? std::make_pair(-plan.crop_h, plan.out_h) : std::make_pair(plan.crop_w, plan.out_w);
            auto it = bucket_cache.find(key);
            if (it == bucket_cache.end())
                it = bucket_cache.emplace(key, swapped ? lp_area420_bucket(plan.crop_h, plan.out_h, true) : lp_area420_bucket(plan.crop_w, plan.out_w)).first;
            return it->second ? 2 : 0;
        };
        // frame heap: thumbnails for the fused images; decoded frame (+ oriented copy) + resized frame for the others
        size_t need = 0;
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = hdrs[k].j;
            const bool swap = j.orientation >= 5;
            const int OW = swap ? (int)j.height : (int)j.width, OH = swap ? (int)j.width : (int)j.height;
            LpOpsPlan plan = lp_plan_static_transform((int)j.width, (int)j.height, (int)j.orientation, opt->width, opt->height, opt->resize_method,
                                                      opt->normalize_orientation != 0, OW, OH);
            const bool fused = route_of(j, plan) != 0;
            const size_t cn = j.ncomp == 1 ? 1 : 3, fb = (size_t)j.width * j.height * cn;
            if (!fused) need += fb + 512 + (j.orientation != 1 ? fb + 512 : 0);
            if (plan.resize) need += (size_t)plan.out_w * plan.out_h * cn + 512;
        }
        if (!eng.heap_reserve(need + 4096)) return fail("frame heap allocation failed");
        eng.heap_reset();
        // Plan every image first (ops.go:449-479 + opencv.go:294-374). Integer-scale area resizes take the fused
        // path (planes -> thumbnail, orientation and crop folded into the addressing); everything else goes through
        // a materialised BGR frame exactly like the one-image ABI does.
        std::vector<LpFrame> frames((size_t)cnt);
        std::vector<int> st((size_t)cnt, 0);
        std::vector<uint8_t> want((size_t)cnt, 1);
        std::vector<LpOpsPlan> plans((size_t)cnt);
        std::vector<LpFusedOp> fops;
        std::vector<int> fidx;
        std::vector<LpAreaReq> areqs;
        std::vector<int> aidx;
        memset(frames.data(), 0, sizeof(LpFrame) * (size_t)cnt);
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = hdrs[k].j;
            const bool swap = j.orientation >= 5;
            const int OW = swap ? (int)j.height : (int)j.width, OH = swap ? (int)j.width : (int)j.height;
            LpOpsPlan plan = lp_plan_static_transform((int)j.width, (int)j.height, (int)j.orientation, opt->width, opt->height, opt->resize_method,
                                                      opt->normalize_orientation != 0, OW, OH);
            plans[(size_t)k] = plan;
            int ix = 1, iy = 1;
            const int route = route_of(j, plan);
            if (route == 0) continue;
            if (route == 2) {
                LpAreaReq rq;
                memset(&rq, 0, sizeof(rq));
                rq.img = (uint32_t)k;
                lp_area420_place((int)j.orientation, (int)j.width, (int)j.height, plan.crop_x, plan.crop_y, &rq);
                rq.crop_w = (uint32_t)plan.crop_w; rq.crop_h = (uint32_t)plan.crop_h;
                rq.dst.w = (uint32_t)plan.out_w; rq.dst.h = (uint32_t)plan.out_h; rq.dst.cn = 3;
                rq.dst.stride = rq.dst.w * 3;
                uint8_t* p = eng.heap_alloc((size_t)rq.dst.stride * rq.dst.h);
                if (!p) return fail("frame heap exhausted");
                rq.dst.off = (uint64_t)(uintptr_t)p;
                areqs.push_back(rq);
                aidx.push_back(k);
                want[(size_t)k] = 0;
                continue;
            }
            lp_resize_mode(plan.crop_w, plan.crop_h, plan.out_w, plan.out_h, &ix, &iy);
            LpFusedOp op;
            fused_op_of(j, plan, ix, iy, &op);
            op.img = (uint32_t)k;
            op.dst.w = (uint32_t)plan.out_w; op.dst.h = (uint32_t)plan.out_h; op.dst.cn = j.ncomp == 1 ? 1 : 3;
            op.dst.stride = op.dst.w * op.dst.cn;
            uint8_t* p = eng.heap_alloc((size_t)op.dst.stride * op.dst.h);
            if (!p) return fail("frame heap exhausted");
            op.dst.off = (uint64_t)(uintptr_t)p;
            // A box none of the thread-per-box kernels takes (3, 5, 6, 12 ... pixels wide, a crop off their grid, 2- and 4-pixel boxes of
            // 4:2:2 / 4:4:4 sources) would cost a whole wave per destination pixel in the general kernel (33 - 58 us per image against
            // 7.4 for the 16 x 16 boxes of the headline); the area walk does it as unit taps instead (LpArea420Op::post: same integers)
            const bool swapped_src = j.orientation >= 5;
            if (area_on && !lp_fused_op_is_fast(op, j) && lp_area_sampling(j) >= 0 &&
                lp_area420_bucket_int(swapped_src ? iy : ix, swapped_src) != 0) {
                LpAreaReq rq;
                memset(&rq, 0, sizeof(rq));
                rq.img = (uint32_t)k;
                lp_area420_place((int)j.orientation, (int)j.width, (int)j.height, plan.crop_x, plan.crop_y, &rq);
                rq.crop_w = (uint32_t)plan.crop_w; rq.crop_h = (uint32_t)plan.crop_h;
                rq.int_x = (uint32_t)ix; rq.int_y = (uint32_t)iy;
                rq.dst = op.dst;
                areqs.push_back(rq);
                aidx.push_back(k);
                want[(size_t)k] = 0;
                continue;
            }
            fops.push_back(op);
            fidx.push_back(k);
            want[(size_t)k] = 0;
        }
        lap(0);
        auto add_decode_timings = [&] { const LpTimings& t = eng.timings(); acc[0] += t.unstuff_ms; acc[1] += t.huff_ms; acc[2] += t.idct_ms; acc[3] += t.color_ms; acc[6] += t.huff_spec_ms; acc[7] += t.huff_verify_ms; acc[8] += t.huff_scan_ms; acc[9] += t.huff_write_ms; rounds = std::max(rounds, t.verify_rounds); };
        int rc = deferred ? eng.decode_begin(first, cnt, frames.data(), want.data()) : eng.decode_uploaded(first, cnt, frames.data(), st.data(), want.data());
        lap(1);
        if (rc == LP_ERR_DEVICE || (deferred && rc)) return fail(eng.last_error());
        if (!deferred) add_decode_timings();
        std::vector<LpFrame> final_frames = frames;
        if (!fops.empty()) {
            if (eng.fused_resample(fops.data(), (int)fops.size())) return fail(eng.last_error());
            for (size_t q = 0; q < fops.size(); q++) final_frames[(size_t)fidx[q]] = fops[q].dst;
        }
        if (!areqs.empty()) {
            if (eng.area_resample(areqs.data(), (int)areqs.size(), !fops.empty())) return fail(eng.last_error());
            for (size_t q = 0; q < areqs.size(); q++) final_frames[(size_t)aidx[q]] = areqs[q].dst;
        }
        lap(2);
        // orientation (ops.go:392: unconditional) for the images that kept a frame
        std::vector<LpOrientOp> oops;
        std::vector<int> oidx;
        for (int k = 0; k < cnt; k++) {
            const LpJpeg& j = hdrs[k].j;
            if (st[(size_t)k] || j.orientation == 1 || !want[(size_t)k]) continue;
            LpOrientOp op;
            memset(&op, 0, sizeof(op));
            op.src = frames[(size_t)k];
            op.orientation = j.orientation;
            const bool swap = j.orientation >= 5;
            op.dst = op.src;
            op.dst.w = swap ? op.src.h : op.src.w;
            op.dst.h = swap ? op.src.w : op.src.h;
            op.dst.stride = op.dst.w * op.src.cn;
            uint8_t* p = eng.heap_alloc((size_t)op.dst.stride * op.dst.h);
            if (!p) return fail("frame heap exhausted");
            op.dst.off = (uint64_t)(uintptr_t)p;
            oops.push_back(op);
            oidx.push_back(k);
        }
        if (!oops.empty()) {
            if (eng.orient(oops.data(), (int)oops.size())) return fail(eng.last_error());
            for (size_t q = 0; q < oops.size(); q++) { frames[(size_t)oidx[q]] = oops[q].dst; final_frames[(size_t)oidx[q]] = oops[q].dst; }
        }
        // fit / resize through frames
        std::vector<LpResizeReq> rqs;
        std::vector<LpFrame> rdst;
        std::vector<int> ridx;
        for (int k = 0; k < cnt; k++) {
            if (st[(size_t)k] || !want[(size_t)k]) continue;
            const LpFrame& f = frames[(size_t)k];
            const LpOpsPlan& plan = plans[(size_t)k];
            if (!plan.resize) continue;
            LpResizeReq rq;
            rq.src = f;
            rq.crop_x = (uint32_t)plan.crop_x; rq.crop_y = (uint32_t)plan.crop_y; rq.crop_w = (uint32_t)plan.crop_w; rq.crop_h = (uint32_t)plan.crop_h;
            rq.dst_w = (uint32_t)plan.out_w; rq.dst_h = (uint32_t)plan.out_h;
            LpFrame d;
            memset(&d, 0, sizeof(d));
            uint8_t* p = eng.heap_alloc((size_t)plan.out_w * plan.out_h * f.cn);
            if (!p) return fail("frame heap exhausted");
            d.off = (uint64_t)(uintptr_t)p;
            rqs.push_back(rq);
            rdst.push_back(d);
            ridx.push_back(k);
        }
        if (!rqs.empty()) {
            std::vector<int> rst(rqs.size(), 0);
            if (eng.resize(rqs.data(), (int)rqs.size(), rdst.data(), rst.data()) == LP_ERR_DEVICE) return fail(eng.last_error());
            acc[4] += eng.timings().resize_ms;
            for (size_t q = 0; q < rqs.size(); q++) {
                if (rst[q]) st[(size_t)ridx[q]] = rst[q];
                else final_frames[(size_t)ridx[q]] = rdst[q];
            }
        }
        // encode (opencv.go:872-900)
        std::vector<LpEncodeReq> erq;
        std::vector<int> eidx;
        for (int k = 0; k < cnt; k++) {
            if (st[(size_t)k]) continue;
            LpEncodeReq e;
            e.src = final_frames[(size_t)k];
            e.quality = quality;
            e.out_cap = (size_t)e.src.w * e.src.h * 4 + 4096;
            erq.push_back(e);
            eidx.push_back(k);
        }
        if (!erq.empty()) {
            std::vector<int> est(erq.size(), 0);
            std::vector<uint32_t> elen(erq.size(), 0);
            std::vector<std::vector<uint8_t>> prog;
            if (opt->jpeg_progressive) { // EncodeOptions[JpegProgressive]: FDCT on the device, the scans on host threads
                if (eng.encode_jpegs_progressive(erq.data(), (int)erq.size(), est.data(), prog) == LP_ERR_DEVICE) return fail(eng.last_error());
                lap(3);
                lap(4);
            } else {
                if (eng.encode_jpegs(erq.data(), (int)erq.size(), est.data(), elen.data()) == LP_ERR_DEVICE) return fail(eng.last_error());
                acc[5] += eng.timings().encode_ms;
                lap(3);
                if (eng.encoded_fetch_all()) return fail(eng.last_error());
                lap(4);
            }
            if (deferred) { // collect the decode: by now its stream has drained
                std::vector<int> dst((size_t)cnt, 0);
                rc = eng.finish_decode(dst.data());
                if (rc == LP_RETRY) return run_chunk(b, part, first, cnt, hdrs, item_of, opt, sink, false);
                if (rc == LP_ERR_DEVICE) return fail(eng.last_error());
                add_decode_timings();
                for (int k = 0; k < cnt; k++) if (dst[(size_t)k]) st[(size_t)k] = dst[(size_t)k];
                deferred = false;
            }
            if (!fops.empty() || !areqs.empty()) acc[4] += eng.fused_resample_ms();
            for (size_t q = 0; q < erq.size(); q++) {
                const int k = eidx[q];
                const size_t item = (size_t)item_of[k];
                if (st[(size_t)k]) continue;
                if (est[q]) { st[(size_t)k] = est[q]; continue; }
                if (opt->jpeg_progressive) sink.put(item, prog[q].data(), (uint32_t)prog[q].size(), (int)erq[q].src.w, (int)erq[q].src.h);
                else sink.put(item, eng.encoded_host((int)q), elen[q], (int)erq[q].src.w, (int)erq[q].src.h);
            }
        }
        if (deferred) { // nothing was encoded (every image failed earlier): the decode still has to be collected
            std::vector<int> dst((size_t)cnt, 0);
            rc = eng.finish_decode(dst.data());
            if (rc == LP_RETRY) return run_chunk(b, part, first, cnt, hdrs, item_of, opt, sink, false);
            if (rc == LP_ERR_DEVICE) return fail(eng.last_error());
            add_decode_timings();
            for (int k = 0; k < cnt; k++) if (dst[(size_t)k]) st[(size_t)k] = dst[(size_t)k];
        }
        for (int k = 0; k < cnt; k++) {
            const size_t item = (size_t)item_of[k];
            if (st[(size_t)k]) b->status[item] = map_status(st[(size_t)k]);
            if (st[(size_t)k] == LP_ERR_DECODE_FAILED && sink.items && (!hdrs[k].scan_path || eng.scan_gave_up(k))) { // see LpEngine::decode_jpegs
                std::lock_guard<std::mutex> lk(b->retry_mu);
                b->retry.push_back((int)item);
            }
        }
    }
    lap(5);
    return LILLIPUT_OK;
}

// Resident (staged) API: the part's whole upload set sits in slot 0; chunks of it are decoded one after the other.
static int run_part(LpBatch* b, LpBatchPart& part, const lilliput_batch_options* opt)
{
    for (int i = 0; i < 10; i++) part.acc[i] = 0;
    for (int i = 0; i < 6; i++) part.tw[i] = 0;
    part.rounds = 0;
    const size_t nv = part.items.size();
    if (!nv) return LILLIPUT_OK;
    LpEngine& eng = *part.eng;
    eng.select_upload(0);
    // One full round of the WRITE kernel's workgroups per launch: a 4096 x 4096 image is 9 of them and the CUs hold five each (LDS), so
    // 142 images fill the device exactly once; a few more would leave a second, nearly empty round that costs as much as the first
    // (measured with four per CU: 112 images 28 us per image, 128 images 36)
    size_t max_raw = 0;
    for (const LpJpegHeader& h : part.hdrs) max_raw = std::max(max_raw, (size_t)h.ecs_len);
    static const size_t round_env = getenv("LILLIPUT_HIP_RESIDENT_CHUNK") ? (size_t)atoi(getenv("LILLIPUT_HIP_RESIDENT_CHUNK")) : 0;
    size_t chunk = auto_chunk(opt, part.hdrs.data(), nv, round_env ? round_env : eng.resident_round(max_raw));
    if (opt->chunk <= 0 && !round_env) {
        // equal launches instead of full ones and a remainder; a part up to a quarter beyond the bound is one launch (round 6: the fixed
        // cost of a launch -- the verify tail, the ramps -- outweighs a partly filled second round of WRITE workgroups)
        if (nv <= chunk + chunk / 4) chunk = nv;
        else { const size_t launches = (nv + chunk - 1) / chunk; chunk = (nv + launches - 1) / launches; }
    }
    const LpSink sink{b, nullptr};
    eng.enable_timing(b->stage_timing);
    int rc = LILLIPUT_OK;
    for (size_t first = 0; first < nv && rc == LILLIPUT_OK; first += chunk)
        rc = run_chunk(b, part, (int)first, (int)std::min(chunk, nv - first), part.hdrs.data() + first, part.items.data() + first, opt, sink);
    eng.enable_timing(false);
    return rc;
}

// GIF, PNG and WebP items: Decoder + ImageOps.Transform of the Go-API mirror (for animated sources the JPEG writer returns after the first composited frame)
static void run_other(LpBatch* b, const lilliput_batch_options* opt, lilliput_batch_item* items)
{
    if (b->other.empty()) return;
    const int enc_opts[4] = {CV_IMWRITE_JPEG_PROGRESSIVE, opt->jpeg_progressive ? 1 : 0, CV_IMWRITE_JPEG_QUALITY, opt->jpeg_quality};
    lilliput_image_options io;
    memset(&io, 0, sizeof(io));
    io.file_type = ".jpeg";
    io.width = opt->width; io.height = opt->height;
    io.resize_method = opt->resize_method;
    io.normalize_orientation = opt->normalize_orientation;
    io.encode_options = enc_opts;
    io.encode_options_len = opt->jpeg_quality ? 4 : 2;
    io.encode_timeout_ns = 30ll * 1000000000ll;
    // The host side of these sources is serial per image (inflate, LZW), so the items are spread over a few workers, each with its
    // own ImageOps (ops.go: "one ImageOps per goroutine") and its own per-thread engine on the batch's device.
    // workers: the CPUs this device's share of the host really has (lp_usable_cpus_per_device: affinity, cgroup quota, ranks of the node), at
    // most 32 (LILLIPUT_HIP_OTHER_WORKERS overrides) -- a mixed-format firehose is bound by the serial host codecs (inflate, LZW, VP8),
    // each worker keeps one engine's worth of device arenas
    static const int other_workers = getenv("LILLIPUT_HIP_OTHER_WORKERS") ? std::max(1, atoi(getenv("LILLIPUT_HIP_OTHER_WORKERS")))
                                                                           : std::max(2, std::min(32, (int)(lp_usable_cpus_per_device() * (lp_cpu_quota_limited() ? 1.5 : 1.0))));
    // (inside a CPU quota the waits of these workers sleep -- LpEngine's blocking-sync default -- so one and a half workers per granted CPU keep
    // the CPUs busy while half a worker's worth of them waits for the device: +10 % on the mixed stream, profiles/r04_j_firehose_host.md)
    const size_t nw = std::min<size_t>(b->other.size(), (size_t)other_workers);
    while (b->other_ops.size() < nw) b->other_ops.push_back(lilliput_new_image_ops(8192));
    while (b->other_eng.size() < nw) {
        b->other_eng.emplace_back(new LpEngine(b->device));
        if (!b->other_eng.back()->ok()) b->other_eng.back().reset(); // the worker then leases from the pool like any caller
    }
    std::atomic<size_t> next{0};
    static const bool trace_items = getenv("LILLIPUT_HIP_TRACE") && atoi(getenv("LILLIPUT_HIP_TRACE")) >= 2;
    auto one = [&](lilliput_image_ops ops, LpOtherItem& it) -> int {
        const size_t i = (size_t)it.item;
        if (!ops) return LILLIPUT_ERR_DEVICE;
        lilliput_decoder d = nullptr;
        int rc = lilliput_new_decoder(it.data, it.len, &d);
        if (rc) return rc;
        int w = 0, h = 0;
        rc = lilliput_decoder_header(d, &w, &h, nullptr, nullptr, nullptr, nullptr);
        // the same frame bound as the JPEG items: the header is untrusted (a 60-byte PNG may claim 10^6 x 10^6 pixels)
        if (!rc && (w <= 0 || h <= 0)) rc = LILLIPUT_ERR_INVALID_IMAGE;
        if (!rc && (uint64_t)w * (uint64_t)h > batch_max_pixels()) rc = LILLIPUT_ERR_BUF_TOO_SMALL;
        if (rc) { lilliput_decoder_close(d); return rc; }
        int ow = w, oh = h;
        if (opt->resize_method == LILLIPUT_OPS_FIT) lilliput_calculate_expected_size(w, h, opt->width, opt->height, &ow, &oh);
        else if (opt->resize_method == LILLIPUT_OPS_RESIZE) { ow = std::max(opt->width, 1); oh = std::max(opt->height, 1); }
        // output buffer: the caller's (pipelined transform) or one sized from the OUTPUT dimensions (never from the source header alone)
        uint8_t* out = nullptr;
        size_t cap = 0;
        if (items) { out = (uint8_t*)items[i].dst; cap = items[i].dst_cap; }
        else {
            cap = std::min<size_t>(it.dst_cap ? it.dst_cap : SIZE_MAX, (size_t)ow * (size_t)oh * 3 + 65536);
            b->out_bytes[i].resize(cap);
            out = b->out_bytes[i].data();
        }
        size_t n = 0;
        rc = out ? lilliput_image_ops_transform(ops, d, &io, out, cap, &n) : LILLIPUT_ERR_BUF_TOO_SMALL;
        if (!rc) {
            if (!items) b->out_bytes[i].resize(n);
            b->out_len[i] = (uint32_t)n;
            b->out_w[i] = ow; b->out_h[i] = oh;
        }
        lilliput_decoder_close(d);
        return rc;
    };
    auto worker = [&](size_t wi) {
        LpCoalesceSuppress no_coalesce; // these Transform calls belong to this batch: they must not queue behind it (lp_coalesce.h)
        const int prev_dev = lp_thread_device(b->device);
        lilliput_image_ops ops = b->other_ops[wi];
        LpEngineLease own(b->other_eng[wi].get()); // every ABI call of this worker nests inside: its own engine, no pool traffic
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= b->other.size()) break;
            int rc;
            // nothing may unwind through a std::thread or the C ABI
            const auto t_item = std::chrono::steady_clock::now();
            try { rc = one(ops, b->other[k]); }
            catch (const std::bad_alloc&) { rc = LILLIPUT_ERR_BUF_TOO_SMALL; }
            catch (...) { rc = LILLIPUT_ERR_DEVICE; }
            if (trace_items)
                fprintf(stderr, "[lilliput_hip] item %d (%.4s, %zu bytes) on worker %zu: %.2f ms, status %d\n", b->other[k].item,
                        b->other[k].len >= 12 && !memcmp(b->other[k].data, "RIFF", 4) ? "webp" : b->other[k].len >= 8 && !memcmp(b->other[k].data, "LPPIXELS", 8) ? "pix " : (const char*)b->other[k].data + 1,
                        b->other[k].len, wi, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_item).count(), rc);
            b->status[(size_t)b->other[k].item] = rc;
        }
        (void)lp_thread_device(prev_dev);
    };
    b->pool.run(nw - 1, worker);
}

static void begin_run(LpBatch* b, size_t n)
{
    b->status = b->parse_status;
    b->out_w.assign(n, 0);
    b->out_h.assign(n, 0);
    b->out_len.assign(n, 0);
    b->out_bytes.assign(n, std::vector<uint8_t>());
}

static int end_run(LpBatch* b, size_t n, bool trace, std::chrono::steady_clock::time_point t0)
{
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rounds = 0;
    int rc = LILLIPUT_OK;
    b->last_stage_ms = b->last_stall_ms = b->last_register_ms = 0;
    b->last_staged_bytes = b->last_copied_bytes = b->last_direct_bytes = 0;
    b->last_numa_node = lp_device_numa_node(b->device);
    for (auto& p : b->parts) {
        if (p.rc) { rc = p.rc; lp_set_error(p.err); }
        else if (p.failed_chunks) lp_set_error(p.err); // the items of those chunks carry the status; the call itself went through
        for (int i = 0; i < 10; i++) acc[i] += p.acc[i];
        rounds = std::max(rounds, p.rounds);
        b->last_stage_ms += p.stage_ms; b->last_stall_ms += p.stall_ms; b->last_staged_bytes += p.staged_bytes;
        b->last_copied_bytes += p.copied_bytes; b->last_direct_bytes += p.direct_bytes; b->last_register_ms += p.register_ms;
        if (trace)
            fprintf(stderr, "[lilliput_hip] part: plan %.2f ms, decode %.2f, resample %.2f, orient+resize+encode %.2f, fetch %.2f, copy-out %.2f | kernels: unstuff %.2f huff %.2f idct %.2f colour %.2f resize %.2f encode %.2f | ingest %.2f ms for %.1f MB (%.1f MB copied through pinned slots, %.1f MB read in place, %.2f ms registering), waited %.2f ms for chunks\n",
                    p.tw[0], p.tw[1], p.tw[2], p.tw[3], p.tw[4], p.tw[5], p.acc[0], p.acc[1], p.acc[2], p.acc[3], p.acc[4], p.acc[5], p.stage_ms, p.staged_bytes / 1e6, p.copied_bytes / 1e6,
                    p.direct_bytes / 1e6, p.register_ms, p.stall_ms);
    }
    b->last_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (trace) fprintf(stderr, "[lilliput_hip] run of %zu items: %.2f ms wall\n", n, b->last_wall_ms);
    b->tm = LpTimings{acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], rounds, acc[6], acc[7], acc[8], acc[9]}; // read by lilliput_hip_batch_timings
    return rc;
}

extern "C" int lilliput_hip_batch_run(lilliput_hip_batch bb, const lilliput_batch_options* opt)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    if (!b || !opt) return LILLIPUT_ERR_DEVICE;
    const size_t n = b->n_items;
    begin_run(b, n);
    // LILLIPUT_HIP_TRACE=1: host wall-clock per phase of a run (plan / decode / resample / encode / fetch / copy-out)
    const bool trace = getenv("LILLIPUT_HIP_TRACE") != nullptr;
    const auto t_run0 = std::chrono::steady_clock::now();
    size_t active = 0;
    for (auto& p : b->parts) { active += p.items.empty() ? 0 : 1; p.stage_ms = p.stall_ms = 0; p.staged_bytes = 0; }
    if (active <= 1 && b->other.empty()) {
        for (auto& p : b->parts) p.rc = p.items.empty() ? LILLIPUT_OK : run_part(b, p, opt);
    } else {
        std::vector<std::thread> th;
        for (auto& p : b->parts)
            if (!p.items.empty()) th.emplace_back([b, &p, opt] { p.rc = run_part(b, p, opt); });
            else p.rc = LILLIPUT_OK;
        run_other(b, opt, nullptr);
        for (auto& t : th) t.join();
    }
    lp_retired_collect();
    return end_run(b, n, trace, t_run0);
}
LP_ABI_CATCH("lilliput_hip_batch_run", return LILLIPUT_ERR_DEVICE)

extern "C" int lilliput_hip_batch_download(lilliput_hip_batch bb, lilliput_batch_item* items, size_t n)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    if (!b || n != b->n_items) return -1;
    int failed = 0;
    for (size_t i = 0; i < n; i++) {
        items[i].status = b->status[i];
        items[i].dst_len = 0;
        items[i].out_width = b->out_w[i];
        items[i].out_height = b->out_h[i];
        if (b->status[i] == LILLIPUT_OK) {
            if (b->out_len[i] > items[i].dst_cap || !items[i].dst) items[i].status = LILLIPUT_ERR_BUF_TOO_SMALL;
            else {
                memcpy(items[i].dst, b->out_bytes[i].data(), b->out_len[i]);
                items[i].dst_len = b->out_len[i];
            }
        }
        if (items[i].status) failed++;
    }
    return failed;
}
LP_ABI_CATCH("lilliput_hip_batch_download", return LILLIPUT_ERR_DEVICE)

// ------------------------------------------------------------------------------------------------
// Host bytes in -> host bytes out, pipelined: what n calls of ImageOps.Transform do in the reference (each starts from the caller's
// []byte, opencv.cpp:99-171, and ends with the encoded bytes in the caller's dst, opencv.go:872-900), as one call.
// Every part (engine) has a stager thread and a compute thread. The stager walks the part's chunks: header walk of the chunk's
// items, layout, memcpy of the entropy-coded segments into the slot's pinned buffer, asynchronous H2D on the engine's copy stream
// (one PCIe Gen5 x16 link moves ~55 GB/s -- scripts/microbench.hip -- and one host thread stages ~30 GB/s, so one stager per part
// keeps the link busy); the compute thread decodes chunk k while chunks k + 1 and k + 2 are being staged and copied. The compute
// stream waits for a chunk's copies through an event, never the host.
struct LpPipeJob {
    size_t i0 = 0, i1 = 0;              // item range
    std::vector<LpJpegHeader> hdrs;     // the items the device takes
    std::vector<int> items;
    std::vector<LpJpegSrc> srcs;
    int rc = 0;
    std::string err;                    // why staging failed (the stager's own string: the engine's is the compute thread's)
    double t_claim = 0, t_staged = 0, t_begin = 0, t_done = 0; // ms since the call started (LILLIPUT_HIP_TRACE)
    int part = -1;
};
// The chunks of a batch form one queue; every part's stager claims the next chunk when one of its slots is free, so the parts
// finish within a chunk of each other whatever the mix of image sizes (and whichever copy stream the link served first).
// The chunk queue of one call. With several devices (lilliput_hip_node_*) the chunks are dealt out as SURVEY.md 8(e) asks: device k
// owns the k-th contiguous share of the chunk list (static block assignment: a caller that keeps the k-th share of its sources in
// pinned memory next to device k gets its DMA reads from the near NUMA node) and claims from it first; a device that has run its
// share dry steals from the share with the most chunks left, so heterogeneous streams and unequal devices still finish together.
// The cursors are plain host atomics: one process drives every device of the node, no message crosses between them.
struct LpPipeShared {
    std::vector<LpPipeJob> jobs;
    size_t scan_path_bound = 0;         // int16 coefficient bytes of scan-path files one chunk may hold (set before the chunks are cut; 0 until then)
    struct Share { std::atomic<size_t> next{0}; size_t end = 0; };
    std::unique_ptr<Share[]> share;     // [n_share]: share k = jobs [share[k].next, share[k].end)
    size_t n_share = 1;
    std::atomic<size_t> stolen{0};
    // Fair first round: a stager claims its SECOND chunk only after every pipe of the call has claimed its first. Stagers run ahead of
    // their compute threads, and the pipes' threads start tens of microseconds apart -- with as many chunks as engines (a call of
    // latency-bound chunks: 128 progressive files of 4096 x 4096 as four chunks of 0.48 s each) one engine now and then took two chunks
    // and another none: 0.98 s instead of 0.49 (profiles/r06_progressive.md).
    std::atomic<size_t> first_claims{0};
    size_t n_pipes = 0;
    double t0 = 0;
    void deal(size_t devices)
    {
        n_share = std::max<size_t>(1, devices);
        share.reset(new Share[n_share]);
        for (size_t k = 0; k < n_share; k++) {
            share[k].next.store(jobs.size() * k / n_share);
            share[k].end = jobs.size() * (k + 1) / n_share;
        }
    }
    // the next chunk for an engine of device `home` (its index in the node), or jobs.size() when the call has none left
    size_t claim(size_t home)
    {
        Share& own = share[home % n_share];
        size_t j = own.next.fetch_add(1);
        if (j < own.end) return j;
        static const bool no_steal = getenv("LILLIPUT_HIP_NODE_STEAL") && atoi(getenv("LILLIPUT_HIP_NODE_STEAL")) == 0; // tests / A-B: static shares only
        if (no_steal) return jobs.size();
        for (;;) { // steal from the fullest share
            size_t best = n_share, left = 0;
            for (size_t k = 0; k < n_share; k++) {
                const size_t nx = share[k].next.load();
                if (nx < share[k].end && share[k].end - nx > left) { left = share[k].end - nx; best = k; }
            }
            if (best == n_share) return jobs.size();
            j = share[best].next.fetch_add(1);
            if (j < share[best].end) { stolen.fetch_add(1); return j; }
        }
    }
};
struct LpPipe {
    std::vector<size_t> mine;           // claimed jobs, in order (job k of this part uses upload slot k % LP_UPLOAD_SLOTS)
    std::mutex mu;
    std::condition_variable cv;
    size_t staged = 0, done = 0;
    bool no_more = false, abort = false;
};

// A chunk that fails -- its staging (allocation, copy enqueue) or its decode -- fails ITS items only: the other chunks of the call go on
// (one absurd file must not take the batch down). A lost device shows the same way, chunk after chunk.
static void fail_job(LpBatch* res, const LpPipeJob& job, int rc, const lilliput_batch_item* items)
{
    for (size_t i = job.i0; i < job.i1; i++)
        if (res->status[i] == LILLIPUT_OK && !is_other_format((const uint8_t*)items[i].src, items[i].src_len)) res->status[i] = rc; // the others are run_other's
}

static void pipe_stager(LpBatch* b, LpBatch* res, LpBatchPart& part, LpPipe& pp, LpPipeShared& sh, const lilliput_batch_item* items)
{
    LpEngine& eng = *part.eng;
    (void)lp_bind_thread_near(b->device); // the pinned slots this thread fills (first touch) and the copies it enqueues belong next to the GPU
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t k = 0;; k++) {
        {
            std::unique_lock<std::mutex> lk(pp.mu);
            pp.cv.wait(lk, [&] { return pp.abort || k < pp.done + LP_UPLOAD_SLOTS; }); // slot k % SLOTS: its previous chunk has been decoded
            if (pp.abort) break;
        }
        if (k == 1) // (bounded: a pipe whose thread never starts must not hold the others)
            for (int spin = 0; sh.first_claims.load() < sh.n_pipes && spin < 20000; spin++) std::this_thread::sleep_for(std::chrono::microseconds(5));
        const size_t ji = sh.claim((size_t)b->node_index);
        if (k == 0) sh.first_claims.fetch_add(1);
        if (ji >= sh.jobs.size()) break;
        const double t0 = now();
        LpPipeJob& job = sh.jobs[ji];
        job.t_claim = t0 - sh.t0;
        job.part = (int)(&part - b->parts.data()) + 16 * b->node_index;
        const int slot = (int)(k % LP_UPLOAD_SLOTS);
        try {
            size_t pinned[2] = {0, 0};
            // The header walks of a chunk whose files are expensive to walk (a progressive file's scans are found by reading through its
            // entropy-coded bytes: ~0.1 ms per 1024 x 1024 file, 7 ms of a 64-file chunk's ingest) are spread over a few helpers; the walk of
            // a baseline file ends at its first scan and is not worth a thread.
            const size_t cnt = job.i1 - job.i0;
            std::vector<LpJpegHeader> walked;
            std::vector<int> walked_rc;
            if (cnt >= 16 && lp_jpeg_sniff_progressive((const uint8_t*)items[job.i0].src, items[job.i0].src_len)) {
                walked.resize(cnt);
                walked_rc.assign(cnt, -1000);
                static const size_t team = getenv("LILLIPUT_HIP_PARSE_THREADS") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_PARSE_THREADS"))) : 4;
                const size_t nt = std::min(team, cnt / 8);
                auto walk = [&](size_t t) {
                    for (size_t q = cnt * t / nt; q < cnt * (t + 1) / nt; q++) {
                        const lilliput_batch_item& it = items[job.i0 + q];
                        if (is_other_format((const uint8_t*)it.src, it.src_len)) continue;
                        walked_rc[q] = (it.src && it.src_len) ? lp_jpeg_parse((const uint8_t*)it.src, it.src_len, &walked[q]) : LP_PARSE_NOT_JPEG;
                    }
                };
                std::vector<std::thread> helpers;
                for (size_t t = 1; t < nt; t++) helpers.emplace_back(walk, t);
                walk(0);
                for (auto& h : helpers) h.join();
            }
            for (size_t i = job.i0; i < job.i1; i++) {
                if (is_other_format((const uint8_t*)items[i].src, items[i].src_len)) continue; // run_other's
                const bool pre = !walked_rc.empty() && walked_rc[i - job.i0] != -1000;
                if (pre) job.hdrs.emplace_back(std::move(walked[i - job.i0]));
                else job.hdrs.emplace_back();
                const int st = parse_item(items[i].src, items[i].src_len, &job.hdrs.back(), pinned, pre ? walked_rc[i - job.i0] : -1000, sh.scan_path_bound);
                res->status[i] = st;
                if (st != LILLIPUT_OK) { job.hdrs.pop_back(); continue; }
                job.items.push_back((int)i);
                job.srcs.push_back(LpJpegSrc{(const uint8_t*)items[i].src, items[i].src_len});
            }
            if (!job.items.empty()) {
                int rc = eng.upload_layout(slot, job.srcs.data(), (int)job.srcs.size(), job.hdrs.data());
                if (!rc) {
                    // What is not read from the caller's pages directly goes through the slot's pinned buffer. That memcpy has to outrun the
                    // link (~55 GB/s for all engines together); one thread moves 20-30 GB/s when the DMA engine reads the same memory, so
                    // the stager brings a few helpers (LILLIPUT_HIP_STAGE_THREADS, default 3 in all) when there is enough to copy.
                    static const size_t team = getenv("LILLIPUT_HIP_STAGE_THREADS") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_STAGE_THREADS"))) : 3;
                    const size_t np = eng.upload_pieces(slot), copied = eng.upload_staged_bytes(slot);
                    const size_t nt = copied < (8u << 20) ? 1 : std::min(team, std::max<size_t>(1, np / 4));
                    std::vector<std::thread> helpers;
                    const int dev = b->device;
                    for (size_t t = 1; t < nt; t++) helpers.emplace_back([&eng, slot, np, nt, t, dev] { (void)lp_bind_thread_near(dev); eng.upload_copy(slot, np * t / nt, np * (t + 1) / nt); });
                    if (copied) eng.upload_copy(slot, 0, np / nt);
                    for (auto& h : helpers) h.join();
                    rc = eng.upload_commit(slot, b->shared_copy, b->extra_copy, b->n_extra_copy);
                    part.staged_bytes += eng.upload_bytes(slot);
                    part.copied_bytes += copied;
                    part.direct_bytes += eng.upload_direct_bytes(slot);
                    part.register_ms += eng.upload_register_ms(slot);
                }
                if (rc) { job.rc = map_status(rc); job.err = eng.last_error(); }
            }
        } catch (...) { job.rc = LILLIPUT_ERR_DEVICE; job.err = "staging failed (out of host memory?)"; }
        part.stage_ms += now() - t0;
        job.t_staged = now() - sh.t0;
        {
            std::lock_guard<std::mutex> lk(pp.mu);
            pp.mine.push_back(ji);
            pp.staged = k + 1;
        }
        pp.cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(pp.mu);
        pp.no_more = true;
    }
    pp.cv.notify_all();
}

static void pipe_compute(LpBatch* res, LpBatchPart& part, LpPipe& pp, LpPipeShared& sh, const lilliput_batch_options* opt, const LpSink& sink)
{
    LpEngine& eng = *part.eng;
    (void)lp_bind_thread_near(eng.device());
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    eng.enable_timing(true);
    for (size_t k = 0;; k++) {
        const double t0 = now();
        size_t ji;
        {
            std::unique_lock<std::mutex> lk(pp.mu);
            pp.cv.wait(lk, [&] { return pp.staged > k || pp.no_more; });
            if (pp.staged <= k) break; // nothing more for this part
            ji = pp.mine[k];
        }
        part.stall_ms += now() - t0;
        LpPipeJob& job = sh.jobs[ji];
        int rc = job.rc;
        if (rc) part.err = job.err;
        job.t_begin = now() - sh.t0;
        if (!rc && !job.items.empty()) {
            eng.select_upload((int)(k % LP_UPLOAD_SLOTS));
            try { rc = run_chunk(res, part, 0, (int)job.items.size(), job.hdrs.data(), job.items.data(), opt, sink); }
            catch (...) { rc = LILLIPUT_ERR_DEVICE; part.err = "chunk failed (out of host memory?)"; }
            if (rc) (void)eng.sync(); // whatever of the chunk is still in flight reads the slot the stager is about to reuse
        }
        job.t_done = now() - sh.t0;
        if (rc) {
            fail_job(res, job, rc, sink.items);
            part.failed_chunks++;
            if (getenv("LILLIPUT_HIP_TRACE")) fprintf(stderr, "[lilliput_hip] chunk %zu failed (%d): %s\n", ji, rc, part.err.c_str());
        }
        {
            std::lock_guard<std::mutex> lk(pp.mu);
            pp.done = k + 1;
        }
        pp.cv.notify_all();
    }
    eng.enable_timing(false);
    eng.select_upload(0);
}

// devs[0] owns the result arrays and serves the PNG / GIF items; every device contributes its engines to one chunk queue.
static int transform_on(const std::vector<LpBatch*>& devs, lilliput_batch_item* items, size_t n, const lilliput_batch_options* opt)
{
    LpBatch* b = devs[0];
    const bool trace = getenv("LILLIPUT_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    b->n_items = n;
    b->parse_status.assign(n, LILLIPUT_OK);
    begin_run(b, n);
    b->other.clear();
    b->retry.clear();
    size_t njpeg = 0;
    for (size_t i = 0; i < n; i++) {
        if (is_other_format((const uint8_t*)items[i].src, items[i].src_len)) b->other.push_back(LpOtherItem{(int)i, (const uint8_t*)items[i].src, items[i].src_len, items[i].dst_cap, {}});
        else njpeg++;
    }
    size_t np = (size_t)batch_streams(0);
    if (njpeg < 2 * np * devs.size()) np = 1;
    int rc = LILLIPUT_OK;
    static const bool one_copy_queue = !getenv("LILLIPUT_HIP_COPY_QUEUES") || atoi(getenv("LILLIPUT_HIP_COPY_QUEUES")) <= 1;
    for (LpBatch* d : devs) {
        if (!d->ensure_parts(np)) rc = LILLIPUT_ERR_DEVICE;
        if (!rc && one_copy_queue && !d->shared_copy) {
            (void)hipSetDevice(d->device);
            if (hipStreamCreateWithFlags(&d->shared_copy, hipStreamNonBlocking) != hipSuccess) d->shared_copy = nullptr; // the engines' own streams then
            static const int xq = getenv("LILLIPUT_HIP_DIRECT_COPY_QUEUES") ? std::max(1, std::min(4, atoi(getenv("LILLIPUT_HIP_DIRECT_COPY_QUEUES")))) : 1; // more queues LOSE next to the decode kernels (r03_a_ingest.md)
            while (d->shared_copy && d->n_extra_copy < xq - 1 && hipStreamCreateWithFlags(&d->extra_copy[d->n_extra_copy], hipStreamNonBlocking) == hipSuccess) d->n_extra_copy++;
        }
    }
    if (!rc) {
        // chunk: LILLIPUT_HIP_PIPE_CHUNK images (default 32) -- small enough that the first chunk's copy is short, large enough to fill the device
        static const size_t pipe_chunk = getenv("LILLIPUT_HIP_PIPE_CHUNK") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_PIPE_CHUNK"))) : 32;
        const size_t chunk = opt->chunk > 0 ? (size_t)opt->chunk : pipe_chunk;
        // Small sources (round 5): a chunk of 32 files of 100 KB is 3 MB -- twenty launches and their waits for 0.2 ms of kernels
        // (512 x 512 sources ran at 66 k images/s end to end against 184 k resident). Beyond its first `chunk` items a chunk goes on while it
        // holds less than LILLIPUT_HIP_PIPE_CHUNK_MB (default 128: what 32 of the 4 MB headline sources weigh) and fewer than its share of a
        // queue of ONE chunk per engine -- a chunk of small files is ~2.5 ms of launch and walk latencies around 0.4 ms of kernels, so the
        // fewer chunks the better (2 048 sources of 256 x 256: 127 k images/s with four chunks per engine, 182 k with one); files big enough
        // for the copy to matter run into the byte bound long before. An explicit chunk size (option or environment) is taken as it is.
        static const size_t pipe_chunk_bytes = (getenv("LILLIPUT_HIP_PIPE_CHUNK_MB") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_PIPE_CHUNK_MB"))) : 128) << 20;
        const bool grow = opt->chunk <= 0 && !getenv("LILLIPUT_HIP_PIPE_CHUNK");
        const size_t engines = std::max<size_t>(1, np * devs.size());
        const size_t chunk_max = grow ? std::max(chunk, std::min<size_t>(1024, (n + engines - 1) / engines)) : chunk;
        std::vector<std::unique_ptr<LpPipe>> pipes;
        LpPipeShared sh;
        sh.scan_path_bound = prog_pinned_max();
        const bool prog_on_device_possible = lp_prog_entropy_mode() != 0;
        if (prog_on_device_possible) { // how many progressive files the call holds: the engines' host / device choice (lp_prog_host.h)
            uint32_t nprog = 0;
            for (size_t i = 0; i < n; i++) nprog += lp_jpeg_sniff_progressive((const uint8_t*)items[i].src, items[i].src_len) ? 1u : 0u;
            for (LpBatch* d : devs)
                for (auto& part : d->parts) part.eng->set_progressive_in_call(nprog / (uint32_t)devs.size());
            if (lp_prog_entropy_mode() > 0 || nprog / (uint32_t)devs.size() >= lp_prog_device_min_images()) sh.scan_path_bound = prog_device_max(); // the engines will take the device route
        }
        for (size_t i = 0; i < n;) { // chunks of at most `chunk` items and 1 GiB of encoded bytes (the frame sizes are only known after the header walk)
            LpPipeJob job;
            job.i0 = i;
            size_t bytes = 0, cnt = 0;
            // (a progressive file counts a sixteenth of its bytes: the device walks its scans one wave each, in a time that does not depend
            // on how many files the chunk holds -- the more of them are in flight, the better; lp_kernels_prog.hip)
            // ... and a chunk never holds more progressive files than the set's coefficient bound lets through (parse_item): the files beyond it
            // open the next chunk instead of answering ErrBufTooSmall (1 024 progressive files of 4096 x 4096 in one call: 340 served, round 6)
            uint64_t coef = 0, need = 0;
            auto weight = [&](size_t k) { return lp_jpeg_sniff_progressive((const uint8_t*)items[k].src, items[k].src_len, &need) && prog_on_device_possible ? items[k].src_len / 16 + 1 : items[k].src_len; };
            while (i < n) {
                const size_t w = weight(i);
                if (!(cnt < chunk ? (cnt == 0 || bytes + w <= (1ull << 30)) : (cnt < chunk_max && bytes + w <= pipe_chunk_bytes))) break;
                if (cnt && need && coef + need > sh.scan_path_bound) break;
                bytes += w; cnt++; i++; coef += need;
            }
            job.i1 = i;
            sh.jobs.push_back(std::move(job));
        }
        for (LpBatch* d : devs)
            for (auto& part : d->parts) {
                for (int i = 0; i < 10; i++) part.acc[i] = 0;
                for (int i = 0; i < 6; i++) part.tw[i] = 0;
                part.rounds = 0; part.rc = 0; part.stage_ms = part.stall_ms = part.register_ms = 0; part.staged_bytes = part.copied_bytes = part.direct_bytes = 0;
                part.failed_chunks = 0; part.err.clear();
            }
        sh.deal(devs.size());
        const LpSink sink{b, items};
        sh.t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        std::vector<std::thread> th;
        sh.n_pipes = std::min(np * devs.size(), sh.jobs.size());
        // engines are started device by device in turn (engine 0 of every device, then engine 1 ...) so that a short queue spreads over the devices
        for (size_t p = 0; p < np; p++)
            for (LpBatch* d : devs) {
                if (pipes.size() >= sh.jobs.size()) break;
                pipes.emplace_back(new LpPipe());
                LpPipe& pp = *pipes.back();
                th.emplace_back(pipe_stager, d, b, std::ref(d->parts[p]), std::ref(pp), std::ref(sh), items);
                th.emplace_back(pipe_compute, b, std::ref(d->parts[p]), std::ref(pp), std::ref(sh), opt, std::cref(sink));
            }
        run_other(b, opt, items);
        for (auto& t : th) t.join();
        for (LpBatch* d : devs) // every chunk has been decoded: the caller's pages that were registered for this call are released
            for (auto& part : d->parts) part.eng->upload_release_pins();
        lp_retired_collect(); // arenas that grew during the call: their old blocks go now, while nothing is in flight
        if (!b->retry.empty()) { // short baseline streams: once more through the one-image path, whose decoder falls back to libjpeg's serial rule
            b->other.clear();
            std::sort(b->retry.begin(), b->retry.end());
            for (int idx : b->retry) b->other.push_back(LpOtherItem{idx, (const uint8_t*)items[idx].src, items[idx].src_len, items[idx].dst_cap, {}});
            b->retry.clear();
            run_other(b, opt, items);
            b->other.clear();
        }
        for (LpBatch* d : devs) d->last_images = 0;
        b->last_chunks = sh.jobs.size();
        b->last_stolen = sh.stolen.load();
        for (size_t j = 0; j < sh.jobs.size(); j++) {
            if (sh.jobs[j].part >= 0) devs[(size_t)(sh.jobs[j].part / 16)]->last_images += sh.jobs[j].items.size();
            if (trace)
                fprintf(stderr, "[lilliput_hip] chunk %2zu device %d engine %d: claimed %6.1f staged %6.1f compute %6.1f .. %6.1f ms (%zu images)\n", j, sh.jobs[j].part / 16,
                        sh.jobs[j].part % 16, sh.jobs[j].t_claim, sh.jobs[j].t_staged, sh.jobs[j].t_begin, sh.jobs[j].t_done, sh.jobs[j].items.size());
        }
        for (size_t k = devs.size(); k-- > 0;) { // devs[0] last: its totals are what lilliput_hip_batch_timings reports for a one-device call
            const int r = end_run(devs[k], n, trace && k == 0, t0);
            if (r) rc = r;
        }
    }
    int failed = 0;
    for (size_t i = 0; i < n; i++) {
        if (rc) { items[i].status = rc; items[i].dst_len = 0; failed++; continue; }
        items[i].status = b->status[i];
        items[i].dst_len = b->status[i] == LILLIPUT_OK ? b->out_len[i] : 0;
        items[i].out_width = b->out_w[i];
        items[i].out_height = b->out_h[i];
        if (items[i].status) failed++;
    }
    return failed;
}

extern "C" int lilliput_hip_batch_transform(lilliput_hip_batch bb, lilliput_batch_item* items, size_t n, const lilliput_batch_options* opt)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    if (!b || !opt || (!items && n)) return (int)n;
    b->node_index = 0;
    return transform_on(std::vector<LpBatch*>{b}, items, n, opt);
}
LP_ABI_CATCH("lilliput_hip_batch_transform", return (int)n)

// ------------------------------------------------------------------------------------------------
// One process, every GPU of the node: what a Go service links (cgo cannot run one process per GPU under torchrun). The devices
// share ONE chunk queue in host memory -- an atomic counter, claimed chunk by chunk by every engine of every device -- so a device
// that finishes early (smaller images, a faster link) simply takes more chunks: work stealing without any device-to-device traffic.
// (lilliput_amd/dist.py does the same across PROCESSES, where the queue state travels through one tiny RCCL all-gather per epoch.)
struct LpNode {
    std::vector<std::unique_ptr<LpBatch>> devs;
};

extern "C" lilliput_hip_node lilliput_hip_node_create(const int* devices, int n_devices)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { lp_set_error("no HIP device visible"); fprintf(stderr, "lilliput_hip: no HIP device visible\n"); return nullptr; }
    std::vector<int> ids;
    if (devices && n_devices > 0) ids.assign(devices, devices + n_devices);
    else for (int i = 0; i < visible; i++) ids.push_back(i);
    auto node = new LpNode();
    for (size_t k = 0; k < ids.size(); k++) {
        if (ids[k] < 0 || ids[k] >= visible) { lp_set_error("HIP device index out of range"); delete node; return nullptr; }
        std::unique_ptr<LpBatch> b(new LpBatch());
        b->device = ids[k];
        b->node_index = (int)k;
        if (!b->ensure_parts(1)) { delete node; return nullptr; }
        node->devs.push_back(std::move(b));
    }
    return node;
}
LP_ABI_CATCH("lilliput_hip_node_create", return nullptr)

extern "C" void lilliput_hip_node_destroy(lilliput_hip_node n) { delete static_cast<LpNode*>(n); }

extern "C" int lilliput_hip_node_device_count(lilliput_hip_node n) { return n ? (int)static_cast<LpNode*>(n)->devs.size() : 0; }

extern "C" int lilliput_hip_node_transform(lilliput_hip_node nn, lilliput_batch_item* items, size_t n, const lilliput_batch_options* opt)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto node = static_cast<LpNode*>(nn);
    if (!node || node->devs.empty() || !opt || (!items && n)) return (int)n;
    std::vector<LpBatch*> devs;
    for (auto& d : node->devs) devs.push_back(d.get());
    return transform_on(devs, items, n, opt);
}
LP_ABI_CATCH("lilliput_hip_node_transform", return (int)n)

// chunks of the last node transform and how many of them a device claimed out of another device's share (work stealing)
extern "C" void lilliput_hip_node_queue_stats(lilliput_hip_node nn, double out[2])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto node = static_cast<LpNode*>(nn);
    out[0] = out[1] = 0;
    if (!node || node->devs.empty()) return;
    out[0] = (double)node->devs[0]->last_chunks;
    out[1] = (double)node->devs[0]->last_stolen;
}
LP_ABI_CATCH("lilliput_hip_node_queue_stats", return)

// images served and bytes staged by device k in the last node transform
extern "C" void lilliput_hip_node_device_stats(lilliput_hip_node nn, int k, double out[2])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto node = static_cast<LpNode*>(nn);
    out[0] = out[1] = 0;
    if (!node || k < 0 || (size_t)k >= node->devs.size()) return;
    out[0] = (double)node->devs[(size_t)k]->last_images;
    out[1] = (double)node->devs[(size_t)k]->last_staged_bytes;
}
LP_ABI_CATCH("lilliput_hip_node_device_stats", return)

extern "C" {

// ---- stage-level access for parity tests
static int decode_one(LpBatch* b, const void* src, size_t len, LpJpegHeader* h, LpFrame* f)
{
    int rc = lp_jpeg_parse((const uint8_t*)src, len, h);
    if (rc) return map_parse(rc);
    const LpJpeg& j = h->j;
    size_t fb = (size_t)j.width * j.height * (j.ncomp == 1 ? 1 : 3);
    if (!b->eng0().heap_reserve(fb + 4096)) return LILLIPUT_ERR_DEVICE;
    b->eng0().heap_reset();
    LpJpegSrc s{(const uint8_t*)src, len};
    memset(f, 0, sizeof(*f));
    int st = 0;
    rc = b->eng0().decode_jpegs(&s, 1, h, f, &st);
    if (rc || st) { lp_set_error(b->eng0().last_error()); return map_status(rc ? rc : st); }
    return LILLIPUT_OK;
}

int lilliput_hip_decode_jpeg(lilliput_hip_batch bb, const void* src, size_t len, void* dst, size_t cap, int* w, int* h, int* channels, int* orientation)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    *w = (int)f.w; *h = (int)f.h; *channels = (int)f.cn; *orientation = hd.j.orientation;
    size_t nb = (size_t)f.stride * f.h;
    if (nb > cap) return LILLIPUT_ERR_BUF_TOO_SMALL;
    if (hipMemcpyAsync(dst, (const void*)(uintptr_t)f.off, nb, hipMemcpyDeviceToHost, b->eng0().stream()) != hipSuccess) return LILLIPUT_ERR_DEVICE;
    return b->eng0().sync() ? LILLIPUT_ERR_DEVICE : LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_hip_decode_jpeg", return LILLIPUT_ERR_DEVICE)

int lilliput_hip_decode_jpeg_coefs(lilliput_hip_batch bb, const void* src, size_t len, int comp, int16_t* dst, size_t cap_elems, int* bw, int* bh)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    if (comp < 0 || comp >= hd.j.ncomp) return LILLIPUT_ERR_INVALID_IMAGE;
    *bw = (int)hd.j.bw[comp]; *bh = (int)hd.j.bh[comp];
    return map_status(b->eng0().copy_coefs(0, comp, dst, cap_elems));
}
LP_ABI_CATCH("lilliput_hip_decode_jpeg_coefs", return LILLIPUT_ERR_DEVICE)

int lilliput_hip_decode_jpeg_plane(lilliput_hip_batch bb, const void* src, size_t len, int comp, uint8_t* dst, size_t cap, int* pw, int* ph)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto b = static_cast<LpBatch*>(bb);
    LpJpegHeader hd;
    LpFrame f;
    int rc = decode_one(b, src, len, &hd, &f);
    if (rc) return rc;
    if (comp < 0 || comp >= hd.j.ncomp) return LILLIPUT_ERR_INVALID_IMAGE;
    *pw = (int)hd.j.bw[comp] * 8; *ph = (int)hd.j.bh[comp] * 8;
    return map_status(b->eng0().copy_plane(0, comp, dst, cap));
}
LP_ABI_CATCH("lilliput_hip_decode_jpeg_plane", return LILLIPUT_ERR_DEVICE)

} // extern "C"
