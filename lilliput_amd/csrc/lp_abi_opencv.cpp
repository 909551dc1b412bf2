// lp_abi_opencv.cpp -- Part A of include/lilliput_hip.h: the reference's opencv.hpp C ABI
// (/root/reference/opencv.hpp:57-145, implemented there by opencv.cpp on top of cv::Mat) re-implemented
// over an own Mat header with a device mirror. Go keeps owning every big host buffer
// (/root/reference/opencv.go:207-267, 443, 848-849); by default results are written back into those buffers
// before a call returns, so Go code that reads Framebuffer.buf directly keeps working. With lazy host
// write-back (lilliput_hip_set_lazy_host / LILLIPUT_HIP_LAZY_HOST=1) pixels stay on the device until somebody
// asks for them (opencv_mat_get_data, lilliput_hip_mat_sync_host): ImageOps.Transform never reads decoded or
// resized pixels on the host (ops.go:331-446 only passes Mats back into this ABI), so the JPEG->JPEG path
// loses two device->host copies per image.
#include "lp_abi.h"
#include "lp_coalesce.h"
#include "lp_inflate.h"
#include "lp_ops_logic.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include "lp_abi_guard.h"

extern "C" {
const int CV_INTER_AREA = 3;   // cv::INTER_AREA
const int CV_INTER_LINEAR = 1; // cv::INTER_LINEAR
const int CV_INTER_CUBIC = 2;  // cv::INTER_CUBIC
}

static thread_local std::string g_last_error;
void lp_set_error(const std::string& s) { g_last_error = s; }
extern "C" const char* lilliput_hip_last_error(void) { return g_last_error.c_str(); }
// lp_abi_guard.h's test hook: the next n calls of lp_abi_test_fault() throw (what an allocation that fails looks like to the guard)
static std::atomic<int> g_test_faults{0};
extern "C" void lilliput_hip_test_fault(int n) { g_test_faults.store(n); }
void lp_abi_test_fault()
{
    int v = g_test_faults.load(std::memory_order_relaxed);
    while (v > 0 && !g_test_faults.compare_exchange_weak(v, v - 1)) {}
    if (v > 0) throw std::bad_alloc();
}
extern "C" int lilliput_hip_device_count(void)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
LP_ABI_CATCH("lilliput_hip_device_count", return 0)

static thread_local int t_device = -1; // lp_thread_device: which GPU this thread's one-image calls run on (-1: LILLIPUT_HIP_DEVICE, else 0)
int lp_thread_device(int device) { int prev = t_device; t_device = device; return prev; }
int lp_current_device() // the device a one-image call of this thread runs on
{
    if (t_device >= 0) return t_device;
    static const int env = getenv("LILLIPUT_HIP_DEVICE") ? atoi(getenv("LILLIPUT_HIP_DEVICE")) : 0;
    return env;
}

// ---- engine pool (see LpEngineLease in lp_abi.h)
namespace {
struct EnginePool {
    std::mutex mu;
    std::vector<std::pair<int, LpEngine*>> idle;    // most recently used last
    size_t live = 0, created = 0, trimmed = 0;
};
EnginePool& engine_pool()
{
    static EnginePool* p = new EnginePool(); // never destroyed: at process exit the HIP runtime may already be gone
    return *p;
}
// Idle engines kept per process: at most LILLIPUT_HIP_ENGINE_POOL of them (default 64) holding at most LILLIPUT_HIP_ENGINE_POOL_MB of
// device arenas between them (default 16 384 of the 288 GB; 4 096 until round 6, when a few engines that had served a 4096 x 4096 decode each
// on their caller's thread pushed the pool over it and every trim's hipFree stalled the dispatchers' launches). Round 3 kept 8: a service with more callers than that in flight on sources the call
// coalescer does not take (PNG, WebP, GIF) built and tore down an engine -- two streams, sixteen events, some forty arenas whose
// hipFree synchronises the device -- for a third of its requests (bench.py --workload abi, 64 callers, direct route: 314 engines
// created for 1 024 requests, 367 images/s against 2 366 with 8 callers; profiles/r04_a_service.md).
size_t pool_keep()
{
    static const size_t v = getenv("LILLIPUT_HIP_ENGINE_POOL") ? (size_t)std::max(0, atoi(getenv("LILLIPUT_HIP_ENGINE_POOL"))) : 64;
    return v;
}
size_t pool_keep_bytes()
{
    static const size_t v = (getenv("LILLIPUT_HIP_ENGINE_POOL_MB") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_ENGINE_POOL_MB"))) : 16384) << 20;
    return v;
}
size_t pool_trim_bytes() // an engine whose arenas grew beyond this is not kept (LILLIPUT_HIP_ENGINE_TRIM_MB, default 1 GiB: a 8192 x 8192 decode is ~0.5 GiB)
{
    static const size_t v = (getenv("LILLIPUT_HIP_ENGINE_TRIM_MB") ? (size_t)std::max(1, atoi(getenv("LILLIPUT_HIP_ENGINE_TRIM_MB"))) : 1024) << 20;
    return v;
}
thread_local LpEngine* t_lease_eng = nullptr;
thread_local int t_lease_dev = -1;
}

LpEngineLease::LpEngineLease() { acquire(); }

void LpEngineLease::acquire()
{
    int dev = lp_current_device();
    dev_ = dev;
    if (t_lease_eng && t_lease_dev == dev) { eng_ = t_lease_eng; return; } // nested: the outer lease's engine, no hand-over
    EnginePool& P = engine_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (size_t i = P.idle.size(); i-- > 0;)
            if (P.idle[i].first == dev) { eng_ = P.idle[i].second; P.idle.erase(P.idle.begin() + (long)i); break; }
        if (eng_) P.live++;
    }
    if (!eng_) {
        LpEngine* e = new LpEngine(dev);
        if (!e->ok()) {
            lp_set_error(e->last_error());
            fprintf(stderr, "lilliput_hip: no usable MI355X device (%s); the HIP path has no CPU fallback\n", e->last_error().c_str());
            delete e;
            return;
        }
        eng_ = e;
        std::lock_guard<std::mutex> lk(P.mu);
        P.live++; P.created++;
    }
    owner_ = true;
    if (!t_lease_eng) { t_lease_eng = eng_; t_lease_dev = dev; tls_ = true; }
}

LpEngineLease::LpEngineLease(LpEngine* own)
{
    if (!own) { acquire(); return; }
    eng_ = own;
    dev_ = own->device();
    adopted_ = true;
    if (!t_lease_eng) { t_lease_eng = own; t_lease_dev = dev_; tls_ = true; }
}

LpEngineLease::~LpEngineLease()
{
    if (adopted_) { if (tls_) { t_lease_eng = nullptr; t_lease_dev = -1; } (void)eng_->sync(); return; }
    if (!owner_ || !eng_) return;
    if (tls_) { t_lease_eng = nullptr; t_lease_dev = -1; }
    // what the call left in flight (lazy write-back) must be visible to whichever engine serves the handle's next call
    (void)eng_->sync();
    EnginePool& P = engine_pool();
    const size_t mine = eng_->device_bytes();
    bool keep = mine <= pool_trim_bytes() && pool_keep() > 0;
    std::vector<LpEngine*> drop;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.live--;
        if (keep) {
            P.idle.emplace_back(dev_, eng_);
            // over either bound ON THIS DEVICE (the bounds are per GPU: with several ranks or processes per node every device has its own
            // HBM to answer for): its least recently used ones go (the one just returned is the most recently used and stays)
            size_t total = 0, count = 0;
            for (auto& e : P.idle)
                if (e.first == dev_) { total += e.second->device_bytes(); count++; }
            while (count > 1 && (count > pool_keep() || total > pool_keep_bytes())) {
                auto it = P.idle.begin();
                while (it != P.idle.end() && it->first != dev_) ++it;
                if (it == P.idle.end() || it->second == eng_) break;
                total -= it->second->device_bytes();
                count--;
                drop.push_back(it->second);
                P.idle.erase(it);
            }
        } else
            drop.push_back(eng_);
        P.trimmed += drop.size();
    }
    for (LpEngine* e : drop) delete e;
}

// engines checked out now, idle in the pool, created so far, destroyed by the pool's bounds
extern "C" void lilliput_hip_engine_pool_stats(size_t out[4])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    EnginePool& P = engine_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    out[0] = P.live; out[1] = P.idle.size(); out[2] = P.created; out[3] = P.trimmed;
}
LP_ABI_CATCH("lilliput_hip_engine_pool_stats", return)

extern "C" int lilliput_hip_mem_info(int device, size_t* free_bytes, size_t* total_bytes)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return LILLIPUT_ERR_DEVICE; }
    size_t f = 0, t = 0;
    const hipError_t e = hipMemGetInfo(&f, &t);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) { (void)hipGetLastError(); return LILLIPUT_ERR_DEVICE; }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_hip_mem_info", return LILLIPUT_ERR_DEVICE)

// ---- device block pool (size-bucketed free lists; hipMalloc is too slow to call per Mat)
namespace {
struct Pool {
    std::mutex mu;
    std::vector<std::pair<size_t, void*>> free_list;
    ~Pool() { for (auto& b : free_list) lp_dev_free(b.second); }
} g_pool;
size_t bucket(size_t n) { size_t b = 4096; while (b < n) b <<= 1; return b; }
}

LpDevBlock::~LpDevBlock()
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (g_pool.free_list.size() < 64 && !lp_guard_on()) g_pool.free_list.emplace_back(cap, p);
    else lp_dev_free(p);
}

std::shared_ptr<LpDevBlock> lp_dev_alloc(size_t bytes)
{
    // kernels may read up to a few vector widths past the last pixel of a row (guard mode, lp_guard.h: that slack and not a byte more)
    size_t want = lp_guard_on() ? bytes + 256 : bucket(bytes + 256);
    auto blk = std::make_shared<LpDevBlock>();
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        for (size_t i = 0; i < g_pool.free_list.size(); i++)
            if (g_pool.free_list[i].first == want) {
                blk->p = g_pool.free_list[i].second;
                blk->cap = want;
                g_pool.free_list.erase(g_pool.free_list.begin() + (long)i);
                return blk;
            }
    }
    if (lp_dev_malloc(&blk->p, want, "mat")) { lp_set_error("hipMalloc failed"); return nullptr; }
    blk->cap = want;
    return blk;
}

// CV_ELEM_SIZE for the 8-bit / 16-bit depths that reach this ABI
static inline int cv_channels(int type) { return (type >> 3) + 1; }
static inline int cv_depth_bytes(int type)
{
    switch (type & 7) { case 0: case 1: return 1; case 2: case 3: return 2; case 4: case 5: return 4; case 6: return 8; default: return 2; }
}
static inline size_t cv_elem_size(int type) { return (size_t)cv_channels(type) * cv_depth_bytes(type); }

bool lp_mat_to_device(LpMat* m, LpEngine* eng)
{
    if (m->lazy && !lp_mat_materialize(m)) return false; // a deferred chain: its pixels are needed now
    if (m->dev && m->dev_valid) return true;
    const size_t rowb = (size_t)m->cols * cv_elem_size(m->type);
    const size_t need = rowb * (size_t)m->rows;
    if (!need) return false;
    if (!m->dev || m->dev->cap < need || m->dev_shared) {
        m->dev = lp_dev_alloc(need);
        m->dev_off = 0;
        m->dev_shared = false;
        if (!m->dev) return false;
    }
    m->dev_step = rowb;
    // rows that follow each other without a gap on both sides travel as ONE copy: the runtime executes a 2-D copy between pageable host
    // memory and the device row by row (HIP API trace, profiles/r04_a_service.md: 300 transfers of 5 us for a 297-row frame, 16 ms per
    // call with eight callers queueing behind each other)
    if (m->step == rowb && need <= ((size_t)4 << 20)) { // small frames: through the engine's pinned buffers
        if (!eng->upload_any((uint8_t*)m->dev->p + m->dev_off, m->data, need) || eng->sync()) return false;
        m->dev_valid = true;
        return true;
    }
    const hipError_t ce = m->step == rowb ? hipMemcpyAsync((uint8_t*)m->dev->p + m->dev_off, m->data, need, hipMemcpyHostToDevice, eng->stream())
                                          : hipMemcpy2DAsync((uint8_t*)m->dev->p + m->dev_off, rowb, m->data, m->step, rowb, (size_t)m->rows, hipMemcpyHostToDevice, eng->stream());
    if (ce != hipSuccess)
        return false;
    if (eng->sync()) return false;
    m->dev_valid = true;
    return true;
}

// ---- lazy host write-back
static int g_lazy_host = -1;                 // process default; -1 = read LILLIPUT_HIP_LAZY_HOST on first use
static thread_local int t_lazy_host = -1;    // scoped per-thread override (lp_lazy_host_scope)
static bool lazy_host()
{
    if (t_lazy_host >= 0) return t_lazy_host != 0;
    int g = __atomic_load_n(&g_lazy_host, __ATOMIC_RELAXED);
    if (g < 0) {
        const char* e = getenv("LILLIPUT_HIP_LAZY_HOST");
        g = e && atoi(e) != 0;
        __atomic_store_n(&g_lazy_host, g, __ATOMIC_RELAXED);
    }
    return g != 0;
}
int lp_lazy_host_scope(int on) { int prev = t_lazy_host; t_lazy_host = on; return prev; }
extern "C" void lilliput_hip_set_lazy_host(int on) { __atomic_store_n(&g_lazy_host, on ? 1 : 0, __ATOMIC_RELAXED); }

static bool mat_copy_to_host(LpMat* m, LpEngine* eng)
{
    const size_t rowb = (size_t)m->cols * cv_elem_size(m->type);
    if (!m->dev || !rowb || !m->rows) return false;
    const bool flat = m->step == rowb && m->dev_step == rowb; // contiguous on both sides: one copy (see lp_mat_to_device)
    const size_t all = rowb * (size_t)m->rows;
    if (flat && all <= ((size_t)4 << 20)) { // small frames: through the engine's pinned buffer, not the runtime's pageable path
        const uint8_t* got = eng->download_begin((uint8_t*)m->dev->p + m->dev_off, all);
        if (!got || eng->sync() != LP_OK) return false;
        memcpy(m->data, got, all);
        m->host_stale = false;
        return true;
    }
    const hipError_t ce = flat ? hipMemcpyAsync(m->data, (uint8_t*)m->dev->p + m->dev_off, all, hipMemcpyDeviceToHost, eng->stream())
                               : hipMemcpy2DAsync(m->data, m->step, (uint8_t*)m->dev->p + m->dev_off, m->dev_step, rowb, (size_t)m->rows, hipMemcpyDeviceToHost, eng->stream());
    if (ce != hipSuccess)
        return false;
    m->host_stale = false;
    return eng->sync() == LP_OK;
}

// The device mirror of `m` has just been (re)written: bring the caller's buffer up to date, now or on demand.
bool lp_mat_to_host(LpMat* m, LpEngine* eng)
{
    if (lazy_host()) { m->host_stale = true; return true; }
    return mat_copy_to_host(m, eng);
}

// Make the host pixels current before something reads or partially overwrites them.
bool lp_mat_host_current(LpMat* m)
{
    if (m->lazy && !lp_mat_materialize(m)) return false; // a deferred chain: somebody is about to look at the pixels
    if (!m->host_stale) return true;
    if (!m->dev || !m->dev_valid) { m->host_stale = false; return true; }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    return eng && mat_copy_to_host(m, eng);
}

extern "C" int lilliput_hip_mat_sync_host(opencv_mat mat)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    return m && lp_mat_host_current(m) ? 0 : -1;
}
LP_ABI_CATCH("lilliput_hip_mat_sync_host", return -1)

LpFrame lp_mat_frame(const LpMat* m)
{
    LpFrame f;
    f.off = (uint64_t)(uintptr_t)((uint8_t*)m->dev->p + m->dev_off);
    f.w = (uint32_t)m->cols; f.h = (uint32_t)m->rows; f.stride = (uint32_t)m->dev_step; f.cn = (uint32_t)cv_channels(m->type);
    return f;
}

// Give `m` a fresh, exclusively owned device block for rows x cols of its type.
static bool mat_new_dev(LpMat* m)
{
    const size_t rowb = (size_t)m->cols * cv_elem_size(m->type);
    m->dev = lp_dev_alloc(rowb * (size_t)m->rows);
    m->dev_off = 0;
    m->dev_step = rowb;
    m->dev_shared = false;
    m->dev_valid = false;
    m->host_stale = false; // every pixel is about to be produced afresh
    return (bool)m->dev;
}

// cv::Mat::create semantics for an output Mat: keep the external buffer when the new shape fits.
bool lp_mat_reshape(LpMat* m, int rows, int cols, int type)
{
    const size_t need = (size_t)rows * cols * cv_elem_size(type);
    if (m->rows == rows && m->cols == cols && m->type == type && m->data) return true;
    if (m->data && m->datastart && (size_t)(m->datalimit - m->datastart) >= need && m->data == m->datastart) {
        m->rows = rows; m->cols = cols; m->type = type; m->step = (size_t)cols * cv_elem_size(type);
        return true;
    }
    m->own.assign(need, 0); // reallocation: the Mat no longer aliases the caller's buffer (cv::Mat::create)
    m->data = m->datastart = m->own.data();
    m->datalimit = m->data + need;
    m->rows = rows; m->cols = cols; m->type = type; m->step = (size_t)cols * cv_elem_size(type);
    return true;
}

// ---- deferred chains (LpLazy, lp_abi.h): unchanged ops.go through Part A
// ops.go's Transform is a dozen opencv_* calls per image (opencv.go:816-839 DecodeTo, :271-279 OrientationTransform, :326-374 Fit, :872-900
// Encode). Executed one by one each pays its own launches and synchronisations, the decode of ONE image cannot fill the device, and the
// decoded frame (48 MB for 4096 x 4096) crosses PCIe for nothing: 452 images/s for one caller, 2.3 k at eight, collapsing beyond
// (profiles/r04_a_service.md). Instead the Mats the library produces stay UNCOMPUTED: read_data of a baseline JPEG records the
// source, orientation_transform / crop / resize append to the record (every dimension is known from the header), and
// opencv_encoder_write(".jpeg") -- the first point where bytes must exist -- hands {source, orientation, crop, size, quality} to the
// batched path through the call coalescer (lp_coalesce.h), where it shares launches with whatever other goroutines' calls are in
// flight. Whatever else touches such a Mat (opencv_mat_get_data, a PNG / WebP / GIF / ThumbHash encoder, a composite, a second
// resize) runs the chain the old way first (lp_mat_materialize). The contract this relies on is the one lazy host write-back already
// states: the caller must not read Framebuffer.buf behind the library's back, and must leave the encoded source
// bytes alone until the decoder is closed (Close copies them if a chain still needs them). LILLIPUT_HIP_DEFER=0 or
// lilliput_hip_set_deferred(0): every call eager again.
// Which sources: baseline JPEGs the device decoder takes whole (not scan-path: those are decoded on host threads and may FAIL in
// read_data -- a stream that runs out of bytes -- which must surface there, as ErrDecodingFailed; a closed baseline stream cannot fail).
static int g_defer = -1;
static thread_local int t_eager = 0;
LpEagerScope::LpEagerScope() { prev = t_eager; t_eager = 1; }
LpEagerScope::~LpEagerScope() { t_eager = prev; }
static bool defer_on()
{
    if (t_eager) return false;
    int g = __atomic_load_n(&g_defer, __ATOMIC_RELAXED);
    if (g < 0) {
        const char* e = getenv("LILLIPUT_HIP_DEFER");
        g = !(e && atoi(e) == 0 && e[0] != '\0');
        __atomic_store_n(&g_defer, g, __ATOMIC_RELAXED);
    }
    return g != 0;
}
extern "C" void lilliput_hip_set_deferred(int on) { __atomic_store_n(&g_defer, on ? 1 : 0, __ATOMIC_RELAXED); }
// A recorded chain that finds no other being served runs on the caller's thread (opencv_encoder_write); 0 = always the batched path
static int g_defer_inline = -1;
static bool defer_inline_on()
{
    int g = __atomic_load_n(&g_defer_inline, __ATOMIC_RELAXED);
    if (g < 0) {
        const char* e = getenv("LILLIPUT_HIP_DEFER_INLINE");
        g = !(e && atoi(e) == 0 && e[0] != '\0');
        __atomic_store_n(&g_defer_inline, g, __ATOMIC_RELAXED);
    }
    return g != 0;
}
extern "C" int lilliput_hip_set_deferred_inline(int on) { const int prev = defer_inline_on() ? 1 : 0; __atomic_store_n(&g_defer_inline, on ? 1 : 0, __ATOMIC_RELAXED); return prev; }
static std::atomic<int> g_part_a_in_flight{0};
void LpLazySrc::enter() { if (!in_flight.exchange(true)) g_part_a_in_flight.fetch_add(1, std::memory_order_relaxed); }
void LpLazySrc::leave() { if (in_flight.exchange(false)) g_part_a_in_flight.fetch_sub(1, std::memory_order_relaxed); }
int lp_part_a_in_flight() { return g_part_a_in_flight.load(std::memory_order_relaxed); }
// A chain served on its caller's thread runs as a resident batch of one (upload / run / download on this thread: the fused planes -> thumbnail
// kernels, one deferred status fetch, no frame in between) on a batch object from a small pool -- at most LILLIPUT_HIP_DEFER_INLINE_MAX are in
// use at once, so that many idle ones are kept (engine, streams and one image's arenas each); never destroyed, like the engine pool.
namespace {
struct LoneBatch { int dev; unsigned index; lilliput_hip_batch b; };
struct LoneBatchPool {
    std::mutex mu;
    std::vector<LoneBatch> idle;
    unsigned made[64] = {0};    // per device: batches created so far (their index)
};
LoneBatchPool& lone_pool() { static LoneBatchPool* p = new LoneBatchPool(); return *p; }
// The runtime maps a process's streams onto FOUR hardware queues per stream priority; two one-image chains on one queue run one after the other
// (eight lone callers: 2.3 k images/s where 4.3 k are possible, profiles/r06_part_a.md section 6). The pool therefore builds its first four batches of a
// device on streams of the default priority and the next four on the highest -- the second pool of queues -- and hands out the idle batch with the
// lowest index, so that up to four callers never meet the other priority (two callers, one on each: 1.2 k against 1.7 k). LILLIPUT_HIP_LONE_PRIORITY=0: all default.
struct LoneBatchLease {
    int dev;
    unsigned index = 0;
    lilliput_hip_batch b = nullptr;
    explicit LoneBatchLease(int device) : dev(device)
    {
        LoneBatchPool& P = lone_pool();
        {
            std::lock_guard<std::mutex> lk(P.mu);
            size_t best = P.idle.size();
            for (size_t i = 0; i < P.idle.size(); i++)
                if (P.idle[i].dev == dev && (best == P.idle.size() || P.idle[i].index < P.idle[best].index)) best = i;
            if (best < P.idle.size()) { b = P.idle[best].b; index = P.idle[best].index; P.idle.erase(P.idle.begin() + (long)best); }
            else index = P.made[dev & 63]++;
        }
        if (!b) {
            static const bool second_pool = !(getenv("LILLIPUT_HIP_LONE_PRIORITY") && atoi(getenv("LILLIPUT_HIP_LONE_PRIORITY")) == 0);
            const int prev = lp_engine_stream_priority_hint(second_pool && (index & 4u) ? 1 : 0);
            b = lilliput_hip_batch_create(dev);
            lp_engine_stream_priority_hint(prev);
            lp_batch_set_stage_timing(b, false);
        }
    }
    ~LoneBatchLease()
    {
        if (!b) return;
        LoneBatchPool& P = lone_pool();
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (P.idle.size() < 16) { P.idle.push_back(LoneBatch{dev, index, b}); b = nullptr; }
        }
        if (b) lilliput_hip_batch_destroy(b);
    }
};
}
int lp_lone_inline_max()
{
    static const int v = getenv("LILLIPUT_HIP_DEFER_INLINE_MAX") ? std::max(1, atoi(getenv("LILLIPUT_HIP_DEFER_INLINE_MAX"))) : 8;
    return v;
}
bool lp_lone_batch_enabled()
{
    // LILLIPUT_HIP_DEFER_INLINE_FUSED=0: a chain served on its caller's thread is materialised call by call instead (decode to a frame, resize,
    // encode: the round-6 route before the batch of one); Part C's direct route likewise runs its own calls
    static const bool on = !(getenv("LILLIPUT_HIP_DEFER_INLINE_FUSED") && atoi(getenv("LILLIPUT_HIP_DEFER_INLINE_FUSED")) == 0);
    return on;
}
static std::atomic<uint64_t> g_lone_batches{0};
extern "C" uint64_t lilliput_hip_lone_batch_count() { return g_lone_batches.load(std::memory_order_relaxed); }
int lp_lone_batch_transform(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& bo, size_t* out_len)
{
    g_lone_batches.fetch_add(1, std::memory_order_relaxed);
    *out_len = 0;
    if (!src || !len || !dst || !cap) return LILLIPUT_ERR_INVALID_IMAGE;
    LoneBatchLease lb(device);
    if (!lb.b) return LILLIPUT_ERR_DEVICE;
    lilliput_batch_item it;
    memset(&it, 0, sizeof(it));
    it.src = src; it.src_len = len; it.dst = dst; it.dst_cap = cap; it.status = LILLIPUT_ERR_DEVICE;
    int rc = lilliput_hip_batch_upload2(lb.b, &it, 1, 1);
    if (rc != LILLIPUT_OK) return rc;
    if (lilliput_hip_batch_run(lb.b, &bo) != LILLIPUT_OK) return LILLIPUT_ERR_DEVICE;
    (void)lilliput_hip_batch_download(lb.b, &it, 1);
    if (it.status == LILLIPUT_OK && (it.dst_len == 0 || it.dst_len > cap)) return LILLIPUT_ERR_DEVICE;
    if (it.status == LILLIPUT_OK) *out_len = it.dst_len;
    return it.status;
}
static std::atomic<uint64_t> g_defer_stats[4]; // chains recorded, served by the batched path, materialised, sources copied at decoder release
extern "C" void lilliput_hip_deferred_stats(uint64_t out[4]) { for (int i = 0; i < 4; i++) out[i] = g_defer_stats[i].load(); }

// device-only forms of the two pixel operations (no host buffer involved): cv::ExifTransform and cv::resize(INTER_AREA)
static bool dev_orient(LpMat* m, int o, LpEngine* eng)
{
    const bool swap = o >= 5;
    LpOrientOp op;
    op.src = lp_mat_frame(m);
    op.orientation = (uint32_t)o;
    op.pad = 0;
    auto src_blk = m->dev; // keep the source alive until the kernel has run
    const int nr = swap ? m->cols : m->rows, nc = swap ? m->rows : m->cols;
    LpMat tmp;
    tmp.rows = nr; tmp.cols = nc; tmp.type = m->type;
    if (!mat_new_dev(&tmp)) return false;
    op.dst = lp_mat_frame(&tmp);
    if (eng->orient(&op, 1)) return false;
    m->rows = nr; m->cols = nc;
    m->dev = tmp.dev; m->dev_off = 0; m->dev_step = tmp.dev_step; m->dev_shared = false; m->dev_valid = true;
    return true;
}
static bool dev_resize(LpMat* s, LpMat* d, int width, int height, LpEngine* eng) // d: rows / cols / type set, gets a fresh device block
{
    if (!mat_new_dev(d)) return false;
    LpResizeReq rq;
    rq.src = lp_mat_frame(s);
    rq.crop_x = rq.crop_y = 0; rq.crop_w = (uint32_t)s->cols; rq.crop_h = (uint32_t)s->rows;
    rq.dst_w = (uint32_t)width; rq.dst_h = (uint32_t)height;
    LpFrame df = lp_mat_frame(d);
    int st = 0;
    if (eng->resize(&rq, 1, &df, &st) || st) { fprintf(stderr, "lilliput_hip: resize failed: %s\n", eng->last_error().c_str()); return false; }
    d->dev_valid = true;
    return true;
}

bool lp_mat_materialize(LpMat* m)
{
    if (!m->lazy) return true;
    const std::shared_ptr<LpLazy> z = std::move(m->lazy);
    m->lazy.reset();
    g_defer_stats[2]++;
    struct Leave { LpLazySrc* s; ~Leave() { s->leave(); } } leave_when_done{z->src.get()};
    LpEagerScope eager;
    if (!z->src->p) {
        lp_set_error("deferred chain: its decoder was closed after the chain had been encoded once; keep the decoder open until the framebuffer's last use (or LILLIPUT_HIP_DEFER=0)");
        fprintf(stderr, "lilliput_hip: a framebuffer was asked for pixels after its decoder had been closed and its recorded chain already encoded once; "
                        "keep the decoder open until the framebuffer's last use, or set LILLIPUT_HIP_DEFER=0\n");
        return false;
    }
    std::unique_ptr<LpJpegHeader> hdr(new LpJpegHeader());
    if (lp_jpeg_parse(z->src->p, z->src->len, hdr.get()) != LP_PARSE_OK) { lp_set_error("deferred decode: the source no longer parses (was the buffer modified before the decoder was closed?)"); return false; }
    const LpJpeg& j = hdr->j;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) return false;
    LpMat cur; // device-only intermediate
    cur.rows = (int)j.height; cur.cols = (int)j.width; cur.type = j.ncomp == 1 ? CV_8UC1 : CV_8UC3;
    if (!mat_new_dev(&cur)) return false;
    {
        LpFrame f = lp_mat_frame(&cur);
        LpJpegSrc src{z->src->p, z->src->len};
        int st = 0;
        const int rc = eng->decode_jpegs(&src, 1, hdr.get(), &f, &st);
        if (rc || st) { lp_set_error(eng->last_error()); return false; }
        cur.dev_valid = true;
    }
    if (z->orientation > 1 && !dev_orient(&cur, z->orientation, eng)) return false;
    if (z->has_crop) { // a view, like cv::Mat(Rect)
        const size_t es = cv_elem_size(cur.type);
        cur.dev_off += (size_t)z->cy * cur.dev_step + (size_t)z->cx * es;
        cur.rows = z->ch; cur.cols = z->cw;
        cur.dev_shared = true;
    }
    if (z->has_resize) {
        LpMat out;
        out.rows = z->rh; out.cols = z->rw; out.type = cur.type;
        if (!dev_resize(&cur, &out, z->rw, z->rh, eng)) return false;
        cur = out;
    }
    if (eng->sync()) return false;
    m->dev = cur.dev; m->dev_off = cur.dev_off; m->dev_step = cur.dev_step; m->dev_shared = cur.dev_shared; m->dev_valid = true;
    m->host_stale = true; // the caller's buffer is brought up to date when somebody asks for it (lp_mat_host_current)
    return true;
}

// The batched path's options for a recorded chain, or false when the chain is not what ops.go's Fit / ResizeTo produce for some request
// (a crop of the caller's own, an orientation other than the header's): such a chain is materialised instead.
static bool lazy_plan_options(const LpLazy& z, int quality, bool progressive, lilliput_batch_options* o)
{
    if (z.orientation != z.hdr_orientation && !(z.orientation == 1 && (z.hdr_orientation < 1 || z.hdr_orientation > 8))) return false; // ops.go:392 applies the header's
    const int ori = z.hdr_orientation >= 1 && z.hdr_orientation <= 8 ? z.hdr_orientation : 1;
    const bool swap = lp_swaps_axes(ori);
    const int ow = swap ? z.hdr_h : z.hdr_w, oh = swap ? z.hdr_w : z.hdr_h; // the oriented frame
    memset(o, 0, sizeof(*o));
    o->jpeg_quality = quality;
    o->jpeg_progressive = progressive ? 1 : 0;
    if (!z.has_resize) { // the oriented frame (or nothing at all) encoded as it is: ImageOpsNoResize
        if (z.has_crop) return false;
        o->width = ow; o->height = oh; o->resize_method = LILLIPUT_OPS_NO_RESIZE;
        return true;
    }
    for (int method = 1; method <= 2; method++)
        for (int norm = 1; norm >= 0; norm--) {
            const LpOpsPlan p = lp_plan_static_transform(z.hdr_w, z.hdr_h, ori, z.rw, z.rh, method, norm != 0, ow, oh);
            const bool crop_same = z.has_crop ? (p.crop_x == z.cx && p.crop_y == z.cy && p.crop_w == z.cw && p.crop_h == z.ch) : (p.crop_x == 0 && p.crop_y == 0 && p.crop_w == ow && p.crop_h == oh);
            if (p.resize && crop_same && p.out_w == z.rw && p.out_h == z.rh) {
                o->width = z.rw; o->height = z.rh; o->resize_method = method == 1 ? LILLIPUT_OPS_FIT : LILLIPUT_OPS_RESIZE; o->normalize_orientation = norm;
                return true;
            }
        }
    return false;
}

// ---- PNG output: cv::PngEncoder::write (OpenCV 4.11 grfmt_png.cpp; source not in the reference tree) over libpng 1.6.47.
// Without IMWRITE_PNG_COMPRESSION in the options OpenCV asks for the SUB filter only, Z_BEST_SPEED and Z_RLE; with it, for libpng's
// adaptive filtering at that level with the default strategy. Filter choice and filtering run on the device (k_png_filter, the
// inflated stream equals libpng's byte for byte -- tests/test_png_output.py); deflate runs on the host like inflate does for PNG
// sources. The reference links zlib-ng, this library the system zlib: the compressed bytes differ, the pixels and the rows do not.
static void png_put_chunk(std::vector<uint8_t>& o, const char* type, const uint8_t* data, size_t n)
{
    const uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
    o.insert(o.end(), len, len + 4);
    const size_t at = o.size();
    o.insert(o.end(), type, type + 4);
    if (n) o.insert(o.end(), data, data + n);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), o.data() + at, (uInt)(n + 4));
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    o.insert(o.end(), c, c + 4);
}

static bool lp_png_encoder_write(LpEncoder* e, LpMat* s, const int* opt, size_t opt_len)
{
    const int cn = opencv_type_channels(s->type);
    if (opencv_type_depth(s->type) != 8 || (cn != 1 && cn != 3 && cn != 4)) return false;
    int level = -1, strategy = 3; // IMWRITE_PNG_STRATEGY_RLE
    for (size_t i = 0; i + 1 < opt_len; i += 2)
        if (opt[i] == CV_IMWRITE_PNG_COMPRESSION) {
            strategy = 0; // IMWRITE_PNG_STRATEGY_DEFAULT
            level = opt[i + 1] < 0 ? 0 : opt[i + 1] > 9 ? 9 : opt[i + 1];
        }
    // png_set_filter(SUB) or libpng's default for 8-bit non-palette images (all five), pruned by png_write_start_row
    uint32_t filters = level < 0 ? 0x02u : 0x1fu;
    if (s->rows == 1) filters &= ~0x1cu; // no UP / AVG / PAETH on a single row
    if (s->cols == 1) filters &= ~0x1au; // no SUB / AVG / PAETH on a single column
    if (!filters) filters = 0x01u;
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(s, eng)) return false;
    const size_t row = (size_t)s->cols * cn + 1, raw_len = row * (size_t)s->rows;
    std::vector<uint8_t> raw(raw_len);
    if (eng->png_filter(lp_mat_frame(s), filters, raw.data())) return false;
    // png_deflate_claim: a window no larger than the data needs
    int window_bits = 15;
    if (raw_len <= 16384) {
        unsigned half = 1u << (window_bits - 1);
        while (raw_len + 262 <= half) { half >>= 1; --window_bits; }
    }
    if (window_bits < 9) window_bits = 9; // zlib's own lower bound for deflate
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level < 0 ? 1 : level, Z_DEFLATED, window_bits, 8, strategy) != Z_OK) return false;
    std::vector<uint8_t> comp(deflateBound(&z, (uLong)raw_len) + 64);
    z.next_in = raw.data();
    z.avail_in = (uInt)raw_len;
    z.next_out = comp.data();
    z.avail_out = (uInt)comp.size();
    const int zr = deflate(&z, Z_FINISH);
    const size_t clen = comp.size() - z.avail_out;
    deflateEnd(&z);
    if (zr != Z_STREAM_END) return false;
    std::vector<uint8_t> out;
    out.reserve(clen + 128);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    out.insert(out.end(), sig, sig + 8);
    const uint32_t W = (uint32_t)s->cols, H = (uint32_t)s->rows;
    const uint8_t ihdr[13] = {(uint8_t)(W >> 24), (uint8_t)(W >> 16), (uint8_t)(W >> 8), (uint8_t)W, (uint8_t)(H >> 24), (uint8_t)(H >> 16), (uint8_t)(H >> 8), (uint8_t)H,
                              8, (uint8_t)(cn == 1 ? 0 : cn == 3 ? 2 : 6), 0, 0, 0};
    png_put_chunk(out, "IHDR", ihdr, 13);
    for (size_t at = 0; at < clen; at += 8192) png_put_chunk(out, "IDAT", comp.data() + at, std::min<size_t>(8192, clen - at)); // libpng's zbuffer size
    png_put_chunk(out, "IEND", nullptr, 0);
    LpMat* d = e->dst;
    const size_t cap = (size_t)(d->datalimit - d->datastart), len = out.size();
    if (len <= cap && d->datastart) {
        memcpy(d->datastart, out.data(), len);
        d->data = d->datastart;
    } else { // cv::imencode into a too-small Mat reallocates (opencv.go:890-895 -> ErrBufTooSmall)
        d->own = out;
        d->data = d->datastart = d->own.data();
        d->datalimit = d->data + len;
    }
    d->rows = (int)len; d->cols = 1; d->type = CV_8U; d->step = 1;
    d->dev_valid = false;
    return true;
}

extern "C" {

int opencv_type_depth(int type) { return cv_depth_bytes(type) * 8; }          // opencv.cpp:83-86
int opencv_type_channels(int type) { return cv_channels(type); }              // opencv.cpp:88-91
int opencv_type_convert_depth(int t, int depth) { return (depth & 7) + ((cv_channels(t) - 1) << 3); } // opencv.cpp:93-96

opencv_mat opencv_mat_create(int width, int height, int type) // opencv.cpp:22-25
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto m = new LpMat();
    m->rows = height; m->cols = width; m->type = type;
    m->step = (size_t)width * cv_elem_size(type);
    m->own.assign(m->step * (size_t)height, 0);
    m->data = m->datastart = m->own.data();
    m->datalimit = m->data + m->own.size();
    return m;
}
LP_ABI_CATCH("opencv_mat_create", return nullptr)

opencv_mat opencv_mat_create_from_data(int width, int height, int type, void* data, size_t data_len) // opencv.cpp:27-36
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    size_t total = (size_t)width * height * cv_elem_size(type);
    if (total > data_len) return NULL;
    auto m = new LpMat();
    m->rows = height; m->cols = width; m->type = type;
    m->step = (size_t)width * cv_elem_size(type);
    m->data = m->datastart = (uint8_t*)data;
    m->datalimit = (uint8_t*)data + data_len;
    return m;
}
LP_ABI_CATCH("opencv_mat_create_from_data", return nullptr)

opencv_mat opencv_mat_create_empty_from_data(int length, void* data) // opencv.cpp:38-49
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = new LpMat();
    m->rows = 0; m->cols = 1; m->type = CV_8U; m->step = 1;
    m->data = m->datastart = (uint8_t*)data;
    m->datalimit = m->data + length;
    return m;
}
LP_ABI_CATCH("opencv_mat_create_empty_from_data", return nullptr)

bool opencv_mat_set_row_stride(opencv_mat mat, size_t stride) // opencv.cpp:51-75
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (m->step == stride) return true;
    size_t ws = (size_t)m->cols * cv_elem_size(m->type);
    if (stride < ws) return false;
    if (m->step != ws) return false;
    if (m->datastart + stride * (size_t)m->rows > m->datalimit) return false;
    if (!lp_mat_host_current(m)) return false;
    m->step = stride;
    m->dev_valid = false;
    return true;
}
LP_ABI_CATCH("opencv_mat_set_row_stride", return false)

void opencv_mat_release(opencv_mat mat) { delete static_cast<LpMat*>(mat); } // opencv.cpp:77-81

int opencv_mat_get_width(const opencv_mat mat) { return static_cast<const LpMat*>(mat)->cols; }   // opencv.cpp:223-227
int opencv_mat_get_height(const opencv_mat mat) { return static_cast<const LpMat*>(mat)->rows; }  // opencv.cpp:229-233
void* opencv_mat_get_data(const opencv_mat mat) // opencv.cpp:235-239
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(const_cast<void*>((const void*)mat));
    lp_mat_host_current(m); // a caller that asks for the pointer is about to look at the pixels
    return m->data;
}
LP_ABI_CATCH("opencv_mat_get_data", return nullptr)

opencv_mat opencv_mat_crop(const opencv_mat src, int x, int y, int width, int height) // opencv.cpp:210-215
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto s = static_cast<const LpMat*>(src);
    if (x < 0 || y < 0 || width < 0 || height < 0 || (int64_t)x + width > s->cols || (int64_t)y + height > s->rows) {
        // cv::Mat(Rect) asserts here and the reference would abort; refuse instead
        fprintf(stderr, "lilliput_hip: opencv_mat_crop rectangle outside the matrix\n");
        return NULL;
    }
    if (s->lazy && s->lazy->has_resize && !lp_mat_materialize(const_cast<LpMat*>(s))) return NULL; // a view of a resized frame: computed first
    auto m = new LpMat();
    m->rows = height; m->cols = width; m->type = s->type; m->step = s->step;
    m->data = s->data + (size_t)y * s->step + (size_t)x * cv_elem_size(s->type);
    m->datastart = s->datastart;
    m->datalimit = s->datalimit;
    if (s->lazy) { // a view of pixels that do not exist yet: the chain with the rectangle appended (rectangles compose)
        m->lazy = std::make_shared<LpLazy>(*s->lazy);
        m->lazy->cx = (s->lazy->has_crop ? s->lazy->cx : 0) + x;
        m->lazy->cy = (s->lazy->has_crop ? s->lazy->cy : 0) + y;
        m->lazy->cw = width; m->lazy->ch = height;
        m->lazy->has_crop = true;
        return m;
    }
    if (s->dev && s->dev_valid) { // a view of the parent's device mirror, like cv::Mat(Rect) is a view of its data
        m->dev = s->dev;
        m->dev_off = s->dev_off + (size_t)y * s->dev_step + (size_t)x * cv_elem_size(s->type);
        m->dev_step = s->dev_step;
        m->dev_valid = true;
        m->dev_shared = true;
        m->host_stale = s->host_stale;
    }
    return m;
}
LP_ABI_CATCH("opencv_mat_crop", return nullptr)

void opencv_mat_resize(const opencv_mat src, opencv_mat dst, int width, int height, int interpolation) // opencv.cpp:196-208
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto s = static_cast<LpMat*>(const_cast<void*>((const void*)src));
    auto d = static_cast<LpMat*>(dst);
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !s || !d || width <= 0 || height <= 0 || s->rows <= 0 || s->cols <= 0) { fprintf(stderr, "lilliput_hip: opencv_mat_resize failed (no device / empty matrix)\n"); return; }
    if (interpolation != CV_INTER_AREA) { fprintf(stderr, "lilliput_hip: opencv_mat_resize supports CV_INTER_AREA only\n"); return; }
    if (cv_depth_bytes(s->type) != 1) { fprintf(stderr, "lilliput_hip: opencv_mat_resize supports 8-bit matrices only\n"); return; }
    if (s->lazy && !s->lazy->has_resize && s != d && defer_on()) { // the source is a recorded chain: so is the result
        lp_mat_reshape(d, height, width, s->type);
        d->lazy = std::make_shared<LpLazy>(*s->lazy);
        d->lazy->has_resize = true;
        d->lazy->rw = width; d->lazy->rh = height;
        d->dev.reset(); d->dev_valid = false; d->dev_shared = false; d->host_stale = false;
        return;
    }
    if (!lp_mat_to_device(s, eng)) return;
    lp_mat_reshape(d, height, width, s->type);
    d->lazy.reset();
    if (!dev_resize(s, d, width, height, eng)) return;
    lp_mat_to_host(d, eng);
}
LP_ABI_CATCH("opencv_mat_resize", return)

void opencv_mat_orientation_transform(CVImageOrientation orientation, opencv_mat mat) // opencv.cpp:217-221
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    int o = (int)orientation;
    if (!m || o <= 1 || o > 8 || m->rows <= 0 || m->cols <= 0) return; // cv::ExifTransform: TL and unknown values are no-ops
    const bool swap = o >= 5;
    const int nr = swap ? m->cols : m->rows, nc = swap ? m->rows : m->cols;
    // the pixels stay in the caller's buffer (tightly packed), unlike cv::transpose which reallocates
    const size_t need = (size_t)nr * nc * cv_elem_size(m->type);
    if (m->lazy && m->lazy->orientation == 1 && !m->lazy->has_crop && !m->lazy->has_resize && defer_on()) { // recorded, not executed
        if ((size_t)(m->datalimit - m->data) < need) { fprintf(stderr, "lilliput_hip: orientation transform: buffer too small\n"); return; }
        m->lazy->orientation = o;
        m->rows = nr; m->cols = nc; m->step = (size_t)nc * cv_elem_size(m->type);
        return;
    }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(m, eng)) { fprintf(stderr, "lilliput_hip: orientation transform failed (no device)\n"); return; }
    if ((size_t)(m->datalimit - m->data) < need) { fprintf(stderr, "lilliput_hip: orientation transform: buffer too small\n"); return; }
    if (!dev_orient(m, o, eng)) return;
    m->step = (size_t)nc * cv_elem_size(m->type);
    lp_mat_to_host(m, eng);
}
LP_ABI_CATCH("opencv_mat_orientation_transform", return)

// The host side of the pixel hand-over (lilliput_hip_pixels_header): rows from a frame decoded elsewhere become the Mat's content;
// they reach the device with the next opencv_* call, like a WebP frame decoded by libwebp.
extern "C" int lilliput_hip_mat_set_pixels(opencv_mat mat, const void* pixels, size_t stride)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!m || !pixels || !m->data || m->rows <= 0 || m->cols <= 0) return -1;
    const size_t rowb = (size_t)m->cols * cv_elem_size(m->type);
    if (stride < rowb || (size_t)(m->datalimit - m->data) < m->step * (size_t)(m->rows - 1) + rowb) return -1;
    for (int y = 0; y < m->rows; y++) memcpy(m->data + (size_t)y * m->step, (const uint8_t*)pixels + (size_t)y * stride, rowb);
    m->lazy.reset();
    m->dev_valid = false;
    m->host_stale = false;
    return 0;
}
LP_ABI_CATCH("lilliput_hip_mat_set_pixels", return -1)

void opencv_mat_reset(opencv_mat mat) // opencv.cpp:471-477
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!m) return;
    const size_t rowb = (size_t)m->cols * cv_elem_size(m->type);
    for (int y = 0; y < m->rows; y++) memset(m->data + (size_t)y * m->step, 0, rowb);
    m->lazy.reset();
    m->dev_valid = false;
    m->host_stale = false;
}
LP_ABI_CATCH("opencv_mat_reset", return)

void opencv_mat_set_color(opencv_mat mat, int red, int green, int blue, int alpha) // opencv.cpp:488-496
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!m) return;
    const int cn = cv_channels(m->type);
    const uint8_t v[4] = {(uint8_t)blue, (uint8_t)green, (uint8_t)red, (uint8_t)(alpha >= 0 ? alpha : 0)};
    for (int y = 0; y < m->rows; y++)
        for (int x = 0; x < m->cols; x++)
            for (int c = 0; c < cn && c < 4; c++) m->data[(size_t)y * m->step + (size_t)x * cn + c] = v[c];
    m->lazy.reset();
    m->dev_valid = false;
    m->host_stale = false;
}
LP_ABI_CATCH("opencv_mat_set_color", return)

static int composite_common(LpMat* s, LpMat* d, int xOffset, int yOffset, int width, int height, int kind)
{
    if (!d || (kind != 2 && (!s || s->rows <= 0 || s->cols <= 0)) || d->rows <= 0 || d->cols <= 0) return OPENCV_ERROR_NULL_MATRIX;
    if (xOffset < 0 || yOffset < 0 || xOffset + width > d->cols || yOffset + height > d->rows) return OPENCV_ERROR_OUT_OF_BOUNDS;
    if (width <= 0 || height <= 0) return OPENCV_ERROR_INVALID_DIMENSIONS;
    const int dcn = cv_channels(d->type);
    if (kind == 2) { if (dcn != 3 && dcn != 4) return OPENCV_ERROR_INVALID_CHANNEL_COUNT; }
    else {
        const int scn = cv_channels(s->type);
        if (kind == 0 && ((scn != 1 && scn != 3 && scn != 4) || (dcn != 3 && dcn != 4))) return OPENCV_ERROR_INVALID_CHANNEL_COUNT;
        if (kind == 1 && scn != dcn && !((scn == 3 && dcn == 4) || (scn == 4 && dcn == 3) || (scn == 1 && dcn == 3) || (scn == 1 && dcn == 4)))
            return OPENCV_ERROR_INVALID_CHANNEL_COUNT;
        if (s->cols != width || s->rows != height) {
            // the reference would cv::resize(INTER_LINEAR) here; ops.go:552-573 never asks for it
            return kind == 0 ? OPENCV_ERROR_ALPHA_BLENDING_FAILED : OPENCV_ERROR_COPY_FAILED;
        }
    }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) return OPENCV_ERROR_UNKNOWN;
    if (!lp_mat_to_device(d, eng)) return OPENCV_ERROR_UNKNOWN;
    if (kind != 2 && !lp_mat_to_device(s, eng)) return OPENCV_ERROR_UNKNOWN;
    LpCompositeOp op;
    memset(&op, 0, sizeof(op));
    op.dst = lp_mat_frame(d);
    if (kind != 2) op.src = lp_mat_frame(s);
    op.kind = (uint32_t)kind;
    op.x0 = (uint32_t)xOffset; op.y0 = (uint32_t)yOffset; op.w = (uint32_t)width; op.h = (uint32_t)height;
    if (eng->composite(op)) return kind == 0 ? OPENCV_ERROR_ALPHA_BLENDING_FAILED : OPENCV_ERROR_UNKNOWN;
    if (lazy_host() || d->host_stale) { d->host_stale = true; return OPENCV_SUCCESS; }
    // write the ROI rows back into the caller's buffer
    const size_t es = cv_elem_size(d->type);
    if (hipMemcpy2DAsync(d->data + (size_t)yOffset * d->step + (size_t)xOffset * es, d->step,
                         (uint8_t*)d->dev->p + d->dev_off + (size_t)yOffset * d->dev_step + (size_t)xOffset * es, d->dev_step, (size_t)width * es,
                         (size_t)height, hipMemcpyDeviceToHost, eng->stream()) != hipSuccess)
        return OPENCV_ERROR_UNKNOWN;
    if (eng->sync()) return OPENCV_ERROR_UNKNOWN;
    return OPENCV_SUCCESS;
}

int opencv_mat_clear_to_transparent(opencv_mat mat, int xOffset, int yOffset, int width, int height) // opencv.cpp:508-543
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto m = static_cast<LpMat*>(mat);
    if (!m) return OPENCV_ERROR_NULL_MATRIX;
    if (xOffset < 0 || yOffset < 0 || xOffset + width > m->cols || yOffset + height > m->rows) return OPENCV_ERROR_OUT_OF_BOUNDS;
    if (width <= 0 || height <= 0) return OPENCV_ERROR_INVALID_DIMENSIONS;
    return composite_common(nullptr, m, xOffset, yOffset, width, height, 2);
}
LP_ABI_CATCH("opencv_mat_clear_to_transparent", return OPENCV_ERROR_UNKNOWN)

int opencv_copy_to_region_with_alpha(opencv_mat src, opencv_mat dst, int xOffset, int yOffset, int width, int height) // opencv.cpp:556-667
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    return composite_common(static_cast<LpMat*>(src), static_cast<LpMat*>(dst), xOffset, yOffset, width, height, 0);
}
LP_ABI_CATCH("opencv_copy_to_region_with_alpha", return OPENCV_ERROR_UNKNOWN)

int opencv_copy_to_region(opencv_mat src, opencv_mat dst, int xOffset, int yOffset, int width, int height) // opencv.cpp:680-752
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    return composite_common(static_cast<LpMat*>(src), static_cast<LpMat*>(dst), xOffset, yOffset, width, height, 1);
}
LP_ABI_CATCH("opencv_copy_to_region", return OPENCV_ERROR_UNKNOWN)

// ---- decoder (opencv.cpp:99-171)
opencv_decoder opencv_decoder_create(const opencv_mat buf)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto m = static_cast<const LpMat*>(buf);
    if (!m || !m->data) return NULL;
    const size_t len = (size_t)m->cols * (size_t)m->rows * cv_elem_size(m->type);
    // cv::findDecoder signature check: the JPEG and PNG signatures are served by this build
    static const uint8_t png_sig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
    const bool jpeg = len >= 3 && m->data[0] == 0xFF && m->data[1] == 0xD8 && m->data[2] == 0xFF;
    const bool png = len >= 8 && memcmp(m->data, png_sig, 8) == 0;
    const bool bmp = len >= 2 && m->data[0] == 'B' && m->data[1] == 'M'; // cv::BmpDecoder's signature: the two letters, nothing more
    const bool pxm = lp_pxm_signature(m->data, len); // cv::PxMDecoder's: 'P', '1'..'6', a white-space character
    if (!jpeg && !png && !bmp && !pxm) return NULL;
    auto d = new LpDecoder();
    d->data = m->data;
    d->len = len;
    d->is_png = png;
    d->is_bmp = bmp;
    d->is_pxm = pxm;
    return d;
}
LP_ABI_CATCH("opencv_decoder_create", return nullptr)

const char* opencv_decoder_get_description(const opencv_decoder d)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!d) return nullptr;
    auto p = static_cast<const LpDecoder*>(d);
    return p->is_png ? "PNG" : p->is_bmp ? "BMP" : p->is_pxm ? "PXM" : "JPEG";
}
LP_ABI_CATCH("opencv_decoder_get_description", return nullptr)
// LILLIPUT_HIP_DEFER_KEEP_SERVED=1: a chain that has been encoded once keeps a copy of its source at Close as well, so that a caller
// who closes the decoder and THEN encodes the same framebuffer a second time (or reads it) still gets pixels, as with the reference's
// framebuffer of real pixels. Off by default: ops.go never does that, and the copy is 4 MB of host traffic per request for nobody.
static bool keep_served_sources()
{
    static const bool on = [] { const char* e = getenv("LILLIPUT_HIP_DEFER_KEEP_SERVED"); return e && atoi(e) != 0; }();
    return on;
}
// Deferred chains that still read `d`'s bytes when the decoder lets go of them (release, set_source): see opencv_decoder_release
static void lp_decoder_detach_chains(LpDecoder* d)
{
    for (auto& w : d->lazies)
        if (auto src = w.lock()) {
            if (src->p != d->data || !src->keep.empty()) continue;
            if (src->served && !keep_served_sources()) src->p = nullptr;
            else {
                src->keep.assign(d->data, d->data + d->len);
                src->p = src->keep.data();
                g_defer_stats[3]++;
            }
        }
    d->lazies.clear();
}
// opencv.hpp:68 declares this entry and the reference never defines or calls it; it is here so that the header's every prototype
// links. Meaning taken from cv::ImageDecoder::setSource(const Mat&): the decoder chosen at create() now reads `buf` -- no signature
// check (OpenCV has none either: a buffer of another format fails at read_header), header state forgotten.
bool opencv_decoder_set_source(opencv_decoder dd, const opencv_mat buf)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<LpDecoder*>(dd);
    auto m = static_cast<const LpMat*>(buf);
    if (!d || !m || !m->data) return false;
    lp_decoder_detach_chains(d);
    d->data = m->data;
    d->len = (size_t)m->cols * (size_t)m->rows * cv_elem_size(m->type);
    d->parsed = false;
    d->parse_rc = 0;
    d->hdr = LpJpegHeader();
    d->png = LpPngInfo();
    d->bmp = LpBmpInfo();
    d->pxm = LpPxmInfo();
    d->png_channels = 0;
    return true;
}
LP_ABI_CATCH("opencv_decoder_set_source", return false)

void opencv_decoder_release(opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<LpDecoder*>(dd);
    if (!d) return;
    // Deferred chains still read this decoder's bytes, and after Close the caller may free or reuse its buffer (opencv.go:663-667).
    // A chain that has NOT produced anything yet gets its own copy of the bytes. A chain that has been served -- ops.go's Transform:
    // the framebuffers of the ImageOps still carry the record when the caller closes the decoder, and nothing will ever look at them
    // again before the next DecodeTo replaces it -- just loses its source: copying 4 MB per request for nobody would cost more host
    // time than the whole transform. Should somebody ask such a Mat for pixels after all, lp_mat_materialize fails loudly.
    lp_decoder_detach_chains(d);
    delete d;
}
LP_ABI_CATCH("opencv_decoder_release", return)

bool opencv_decoder_read_header(opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto d = static_cast<LpDecoder*>(dd);
    if (!d) return false;
    if (d->parsed) return d->parse_rc == LP_PARSE_OK;
    if (d->is_pxm) { // cv::PxMDecoder::readHeader
        d->parsed = true;
        d->parse_rc = lp_pxm_read_info(d->data, d->len, d->pxm) ? LP_PARSE_OK : LP_PARSE_NOT_JPEG;
        return d->parse_rc == LP_PARSE_OK;
    }
    if (d->is_bmp) { // cv::BmpDecoder::readHeader
        d->parsed = true;
        d->parse_rc = lp_bmp_read_info(d->data, d->len, d->bmp) ? LP_PARSE_OK : LP_PARSE_NOT_JPEG;
        return d->parse_rc == LP_PARSE_OK;
    }
    if (d->is_png) { // cv::PngDecoder::readHeader: png_read_info, then the Mat type from colour type / tRNS / bit depth
        d->parsed = true;
        d->parse_rc = lp_png_read_info(d->data, d->len, d->png) ? LP_PARSE_OK : LP_PARSE_NOT_JPEG;
        const int ct = d->png.color_type;
        d->png_channels = (ct == 2 || ct == 3) ? (d->png.num_trans > 0 ? 4 : 3) : (ct == 4 || ct == 6) ? 4 : 1;
        return d->parse_rc == LP_PARSE_OK;
    }
    d->parse_rc = lp_jpeg_parse(d->data, d->len, &d->hdr);
    d->parsed = true;
    if (d->parse_rc == LP_PARSE_UNSUPPORTED) {
        lp_set_error("JPEG feature outside the device path (arithmetic coding / lossless / 12-bit / sampling factors above 2)");
        fprintf(stderr, "lilliput_hip: %s\n", g_last_error.c_str());
    }
    return d->parse_rc == LP_PARSE_OK;
}
LP_ABI_CATCH("opencv_decoder_read_header", return false)

int opencv_decoder_get_width(const opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<const LpDecoder*>(dd);
    return d->is_png ? (int)d->png.width : d->is_bmp ? d->bmp.width : d->is_pxm ? d->pxm.width : (int)d->hdr.j.width;
}
LP_ABI_CATCH("opencv_decoder_get_width", return 0)
int opencv_decoder_get_height(const opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<const LpDecoder*>(dd);
    return d->is_png ? (int)d->png.height : d->is_bmp ? d->bmp.height : d->is_pxm ? d->pxm.height : (int)d->hdr.j.height;
}
LP_ABI_CATCH("opencv_decoder_get_height", return 0)
int opencv_decoder_get_pixel_type(const opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<const LpDecoder*>(dd);
    if (d->is_png) return (d->png.depth == 16 ? 2 /* CV_16U */ : 0) + ((d->png_channels - 1) << 3); // the Go side demotes 16-bit types (opencv.go:255-257)
    if (d->is_bmp) return (d->bmp.channels - 1) << 3; // CV_8UC1 / C3 / C4
    if (d->is_pxm) return ((d->pxm.channels - 1) << 3) | (d->pxm.maxval > 255 ? 2 : 0); // CV_8UC1 / CV_8UC3, CV_16UC1 / CV_16UC3 for samples above 255 (the Go layer demotes the depth)
    return d->hdr.j.ncomp == 1 ? CV_8UC1 : CV_8UC3;
}
LP_ABI_CATCH("opencv_decoder_get_pixel_type", return 0)
int opencv_decoder_get_orientation(const opencv_decoder dd)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<const LpDecoder*>(dd);
    return d->is_png || d->is_bmp || d->is_pxm ? 1 : (int)d->hdr.j.orientation; // a PNG's eXIf chunk is only looked at while the pixels are read, after lilliput has asked
}
LP_ABI_CATCH("opencv_decoder_get_orientation", return 0)

// cv::PngDecoder::readData into an 8-bit Mat of the announced channel count: chunk walk + inflate on the
// host (serial, like libpng + zlib-ng in the reference), filter reversal and pixel expansion on the device.
static bool png_read_data(LpDecoder* d, LpMat* m)
{
    const LpPngInfo& pi = d->png;
    if (m->rows != (int)pi.height || m->cols != (int)pi.width || cv_channels(m->type) != d->png_channels || cv_depth_bytes(m->type) != 1) return false;
    LpBytes filtered;
    if (!lp_png_read_idat(d->data, d->len, pi, filtered)) { lp_set_error("PNG image data is damaged"); return false; }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !mat_new_dev(m)) return false;
    LpPngOp op;
    memset(&op, 0, sizeof(op));
    op.dst = lp_mat_frame(m);
    op.depth = (uint32_t)pi.depth;
    op.color_type = (uint32_t)pi.color_type;
    const int bits = pi.depth * lp_png_channels_in_file(pi.color_type);
    op.bpp = (uint32_t)(bits >= 8 ? bits / 8 : 1);
    op.has_key = pi.color_type == 2 && pi.num_trans > 0;
    for (int c = 0; c < 3; c++) op.key[c] = pi.trans_key[c];
    uint64_t off = 0;
    for (int p = 0; p < (pi.interlace ? 7 : 1); p++) {
        LpPngPass& ps = op.pass[op.npass++];
        lp_png_pass_geometry(pi, p, &ps.pw, &ps.ph, &ps.x0, &ps.y0, &ps.dx, &ps.dy);
        ps.row_bytes = (uint32_t)(((size_t)ps.pw * bits + 7) / 8);
        ps.off = off;
        if (ps.pw && ps.ph) off += (uint64_t)ps.ph * (ps.row_bytes + 1);
        else ps.pw = ps.ph = 0;
    }
    uint8_t pal[1024];
    memset(pal, 0, sizeof(pal)); // indices past the palette read as black, like libpng's zero-filled 256-entry table
    for (int i = 0; i < 256; i++) pal[4 * i + 3] = 255;
    for (int i = 0; i < pi.num_palette; i++) { pal[4 * i] = pi.palette[i][2]; pal[4 * i + 1] = pi.palette[i][1]; pal[4 * i + 2] = pi.palette[i][0]; }
    if (pi.color_type == 3) for (int i = 0; i < pi.num_trans; i++) pal[4 * i + 3] = pi.trans_alpha[i];
    if (eng->png_decode(op, filtered.data(), filtered.size(), pal)) { lp_set_error(eng->last_error()); return false; }
    m->dev_valid = true;
    return lp_mat_to_host(m, eng);
}

bool opencv_decoder_read_data(opencv_decoder dd, opencv_mat dst)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    auto d = static_cast<LpDecoder*>(dd);
    auto m = static_cast<LpMat*>(dst);
    if (!d || !m) return false;
    if (!d->parsed && !opencv_decoder_read_header(dd)) return false;
    if (d->parse_rc != LP_PARSE_OK) return false;
    m->lazy.reset(); // whatever the Mat was, it is this frame now
    if (d->is_png) return png_read_data(d, m);
    if (d->is_pxm) { // cv::PxMDecoder::readData into the 8-bit Mat the Go layer hands over (lp_pxm.h)
        const LpPxmInfo& pi = d->pxm;
        if (m->rows != pi.height || m->cols != pi.width || cv_channels(m->type) != pi.channels || cv_depth_bytes(m->type) != 1 || !m->data) return false;
        if (m->step < (size_t)pi.width * pi.channels) return false;
        const bool ok = lp_pxm_read_data(d->data, d->len, pi, m->data, m->step);
        m->dev_valid = false;
        m->host_stale = false;
        if (!ok) lp_set_error("PBM / PGM / PPM image data is damaged");
        return ok;
    }
    if (d->is_bmp) { // cv::BmpDecoder::readData: rows unpacked on the host, straight into the Mat (lp_bmp.h)
        const LpBmpInfo& bi = d->bmp;
        if (m->rows != bi.height || m->cols != bi.width || cv_channels(m->type) != bi.channels || cv_depth_bytes(m->type) != 1 || !m->data) return false;
        if (m->step < (size_t)bi.width * bi.channels) return false;
        const bool ok = lp_bmp_read_data(d->data, d->len, bi, m->data, m->step);
        m->dev_valid = false; // the host copy is the frame now; it reaches the device with the next opencv_* call
        m->host_stale = false;
        if (!ok) lp_set_error("BMP image data is damaged");
        return ok;
    }
    const LpJpeg& j = d->hdr.j;
    const int cn = j.ncomp == 1 ? 1 : 3;
    if (m->rows != (int)j.height || m->cols != (int)j.width || cv_channels(m->type) != cn || cv_depth_bytes(m->type) != 1) return false;
    m->lazy.reset();
    if (!d->hdr.scan_path && defer_on() && lilliput_hip_device_count() > 0) { // a closed baseline stream: cannot fail to decode, so nothing is lost by not doing it yet
        auto z = std::make_shared<LpLazy>();
        z->src = std::make_shared<LpLazySrc>();
        z->src->p = d->data; z->src->len = d->len;
        z->src->enter();
        z->hdr_w = (int)j.width; z->hdr_h = (int)j.height; z->hdr_orientation = (int)j.orientation;
        d->lazies.erase(std::remove_if(d->lazies.begin(), d->lazies.end(), [](const std::weak_ptr<LpLazySrc>& w) { return w.expired(); }), d->lazies.end());
        d->lazies.push_back(z->src);
        m->lazy = z;
        m->dev.reset(); m->dev_valid = false; m->dev_shared = false; m->host_stale = false;
        g_defer_stats[0]++;
        return true;
    }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng) return false;
    if (!mat_new_dev(m)) return false;
    LpFrame f = lp_mat_frame(m);
    LpJpegSrc src{d->data, d->len};
    int st = 0;
    int rc = eng->decode_jpegs(&src, 1, &d->hdr, &f, &st);
    if (rc || st) { lp_set_error(eng->last_error()); return false; }
    m->dev_valid = true;
    return lp_mat_to_host(m, eng);
}
LP_ABI_CATCH("opencv_decoder_read_data", return false)

// Test access (no device work): would the PNG's image data be accepted? Returns the number of inflated bytes, or -1.
extern "C" long lilliput_hip_png_inflate_check(const void* data, size_t len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    LpPngInfo pi;
    LpBytes filtered;
    if (!lp_png_read_info((const uint8_t*)data, len, pi) || !lp_png_read_idat((const uint8_t*)data, len, pi, filtered)) return -1;
    return (long)filtered.size();
}
LP_ABI_CATCH("lilliput_hip_png_inflate_check", return -1)

// Test access: the filtered rows the image data inflates to (what the un-filter kernel is given). Returns their size, -1 when the file
// is rejected, -2 when `cap` is too small.
extern "C" long lilliput_hip_png_inflate_bytes(const void* data, size_t len, uint8_t* out, size_t cap)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    LpPngInfo pi;
    LpBytes filtered;
    if (!lp_png_read_info((const uint8_t*)data, len, pi) || !lp_png_read_idat((const uint8_t*)data, len, pi, filtered)) return -1;
    if (filtered.size() > cap) return -2;
    memcpy(out, filtered.data(), filtered.size());
    return (long)filtered.size();
}
LP_ABI_CATCH("lilliput_hip_png_inflate_bytes", return -1)
// Test access: which inflater lp_png_read_idat tries first: 1 the library's own (default), 0 zlib only. Returns the previous setting.
extern "C" int lilliput_hip_png_set_inflater(int own) { return lp_png_set_inflater(own); }
// Test access: lp_inflate_exact on a caller's buffer (copied behind the padding the bit reader wants)
extern "C" int lilliput_hip_inflate_exact(const void* in, size_t in_len, uint8_t* out, size_t out_len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    std::vector<uint8_t> z(in_len + LP_INFLATE_PAD, 0);
    memcpy(z.data(), in, in_len);
    std::vector<uint8_t> o(out_len + 1, 0xa5); // one guard byte: the decoder must not write past out_len
    const int r = lp_inflate_exact(z.data(), in_len, o.data(), out_len);
    if (o[out_len] != 0xa5) return -1;
    if (r == 1 && out_len) memcpy(out, o.data(), out_len);
    return r;
}
LP_ABI_CATCH("lilliput_hip_inflate_exact", return 0)

// Test access: the checksum routines of lp_inflate.cpp (which = 0 Adler-32, 1 CRC-32), same conventions as zlib's
extern "C" uint32_t lilliput_hip_checksum(int which, uint32_t seed, const void* p, size_t n)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    return which ? lp_crc32(seed, (const uint8_t*)p, n) : lp_adler32(seed, (const uint8_t*)p, n);
}
LP_ABI_CATCH("lilliput_hip_checksum", return 0)

// Test access (no device work): cv::PxMDecoder's answer for a file -- 0 decoded (w, h, the decoder's type, 8-bit pixels of its channels), 1 header
// refused (or not its signature), 2 data refused (what was written before stays), -1 cap
extern "C" int lilliput_hip_pxm_decode(const void* data, size_t len, int* w, int* h, int* type, uint8_t* out, size_t cap)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    LpPxmInfo pi;
    if (!lp_pxm_read_info((const uint8_t*)data, len, pi)) return 1;
    *w = pi.width; *h = pi.height; *type = ((pi.channels - 1) << 3) | (pi.maxval > 255 ? 2 : 0);
    if ((size_t)pi.width * pi.height * pi.channels > cap) return -1;
    return lp_pxm_read_data((const uint8_t*)data, len, pi, out, (size_t)pi.width * pi.channels) ? 0 : 2;
}
LP_ABI_CATCH("lilliput_hip_pxm_decode", return -1)

// Test access (no device work): cv::BmpDecoder's answer for a file -- 0 decoded (w, h, channels, pixels), 1 header refused, 2 data refused, -1 cap
extern "C" int lilliput_hip_bmp_decode(const void* data, size_t len, int* w, int* h, int* channels, uint8_t* out, size_t cap)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    LpBmpInfo bi;
    if (!lp_bmp_read_info((const uint8_t*)data, len, bi)) return 1;
    *w = bi.width; *h = bi.height; *channels = bi.channels;
    if ((size_t)bi.width * bi.height * bi.channels > cap) return -1;
    return lp_bmp_read_data((const uint8_t*)data, len, bi, out, (size_t)bi.width * bi.channels) ? 0 : 2;
}
LP_ABI_CATCH("lilliput_hip_bmp_decode", return -1)

// ---- encoder (opencv.cpp:173-194)
opencv_encoder opencv_encoder_create(const char* ext, opencv_mat dst)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!ext || !dst) return NULL;
    std::string e(ext);
    for (auto& c : e) c = (char)tolower(c);
    const bool png = e == ".png";
    if (!png && e != ".jpeg" && e != ".jpg" && e != ".jpe") return NULL;
    auto enc = new LpEncoder();
    enc->dst = static_cast<LpMat*>(dst);
    enc->png = png;
    return enc;
}
LP_ABI_CATCH("opencv_encoder_create", return nullptr)

void opencv_encoder_release(opencv_encoder e) { delete static_cast<LpEncoder*>(e); }

bool opencv_encoder_write(opencv_encoder ee, const opencv_mat src, const int* opt, size_t opt_len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    lp_abi_test_fault();
    auto e = static_cast<LpEncoder*>(ee);
    auto s = static_cast<LpMat*>(const_cast<void*>((const void*)src));
    if (!e || !s || s->rows <= 0 || s->cols <= 0) return false;
    if (e->png) return lp_png_encoder_write(e, s, opt, opt_len);
    int quality = 95; // cv::JpegEncoder default
    bool progressive = false;
    for (size_t i = 0; i + 1 < opt_len; i += 2) {
        if (opt[i] == CV_IMWRITE_JPEG_QUALITY) quality = opt[i + 1] < 0 ? 0 : opt[i + 1] > 100 ? 100 : opt[i + 1];
        else if (opt[i] == CV_IMWRITE_JPEG_PROGRESSIVE) progressive = opt[i + 1] != 0; // cv::JpegEncoder: jpeg_simple_progression
    }
    LpMat* d = e->dst;
    const size_t cap = (size_t)(d->datalimit - d->datastart);
    // A recorded chain that arrives while few others are in flight and the dispatchers are idle (a handful of goroutines, or a quiet moment) runs on
    // the caller's own thread as a resident batch of one (lp_lone_batch_transform) -- no hand-over to a dispatcher thread and back, no stager /
    // compute thread pair (profiles/r06_one_image.md). LILLIPUT_HIP_DEFER_INLINE=0: always through the dispatchers.
    // "few" = at most LILLIPUT_HIP_DEFER_INLINE_MAX (default 8) requests of deferred Part A in flight, from the read_data that recorded a chain to the
    // first time it is served; "idle" = nothing queued for or inside a dispatcher, so the first request that goes there pulls the ones behind it
    // along: 8 callers all on their own threads 4.1 k images/s, all through the dispatchers 2.5 k; 64 callers 8 / 10 k -- and a mix of the two routes
    // is slower than either (profiles/r06_part_a.md sections 4-6).
    const bool lone = s->lazy && defer_inline_on() && lp_part_a_in_flight() <= lp_lone_inline_max() && lp_coalesce_busy() == 0;
    struct ServedScope { LpLazySrc* p; ~ServedScope() { if (p) p->leave(); } } served_scope{s->lazy ? s->lazy->src.get() : nullptr};
    std::shared_ptr<LpLazySrc> keep_src = s->lazy ? s->lazy->src : nullptr; // (the scope's pointer stays valid)
    if (s->lazy && !lone && quality > 0 && d->datastart && cap) { // a recorded chain: decode -> orientation -> crop -> resize -> encode as ONE item of the batched path
        lilliput_batch_options bo;
        if (lazy_plan_options(*s->lazy, quality, progressive, &bo)) {
            size_t n = 0;
            const std::shared_ptr<LpLazySrc> src = s->lazy->src;
            // a chain whose decoder was closed after it had been served once has lost its bytes (opencv_decoder_release): no batched
            // route for it -- the eager route below reports the broken contract (lp_mat_materialize) instead of reading a null source
            const int st = src->p ? lp_coalesce_transform_status(lp_current_device(), src->p, src->len, d->datastart, cap, bo, &n) : LILLIPUT_ERR_INVALID_IMAGE;
            if (st == LILLIPUT_OK && n > 0 && n <= cap) {
                src->served = true;
                d->data = d->datastart;
                d->rows = (int)n; d->cols = 1; d->type = CV_8U; d->step = 1;
                d->dev_valid = false;
                g_defer_stats[1]++;
                return true;
            }
            // anything else (a result larger than the caller's buffer: the pointer must change; a dispatcher that is shutting down;
            // a device error): the eager route below reproduces the direct behaviour
        }
    }
    if (s->lazy && lone && lp_lone_batch_enabled() && quality > 0 && d->datastart && cap && s->lazy->src->p) { // ... on this thread, as a resident batch of one
        lilliput_batch_options bo;
        if (lazy_plan_options(*s->lazy, quality, progressive, &bo)) {
            const std::shared_ptr<LpLazySrc> src = s->lazy->src;
            size_t n = 0;
            if (lp_lone_batch_transform(lp_current_device(), src->p, src->len, d->datastart, cap, bo, &n) == LILLIPUT_OK) {
                src->served = true;
                d->data = d->datastart;
                d->rows = (int)n; d->cols = 1; d->type = CV_8U; d->step = 1;
                d->dev_valid = false;
                g_defer_stats[1]++;
                return true;
            }
            // anything else: the eager route below, whose behaviour is the direct one's (errors, a result beyond the caller's buffer)
        }
    }
    LpEngineLease lease;
    LpEngine* eng = lease.get();
    if (!eng || !lp_mat_to_device(s, eng)) return false;
    LpEncodeReq rq;
    rq.src = lp_mat_frame(s);
    rq.quality = quality;
    rq.out_cap = (size_t)s->rows * s->cols * 4 + 4096; // device-side bound; the caller's capacity is checked below
    int st = 0;
    uint32_t len = 0;
    std::vector<uint8_t> prog;
    if (progressive) {
        if (eng->encode_jpeg_progressive(rq, prog) || prog.empty()) return false;
        len = (uint32_t)prog.size();
    } else if (eng->encode_jpegs(&rq, 1, &st, &len) || st || !len)
        return false;
    if (len <= cap && d->datastart) {
        if (progressive) memcpy(d->datastart, prog.data(), len);
        else if (eng->encoded_copy(0, d->datastart, cap)) return false;
        d->data = d->datastart;
    } else {
        // cv::imencode into a too-small Mat reallocates: the data pointer changes and Go reports ErrBufTooSmall (opencv.go:890-895)
        if (progressive) d->own = prog;
        else {
            d->own.assign(len, 0);
            if (eng->encoded_copy(0, d->own.data(), len)) return false;
        }
        d->data = d->datastart = d->own.data();
        d->datalimit = d->data + len;
    }
    d->rows = (int)len; d->cols = 1; d->type = CV_8U; d->step = 1;
    d->dev_valid = false;
    return true;
}
LP_ABI_CATCH("opencv_encoder_write", return false)

} // extern "C"

