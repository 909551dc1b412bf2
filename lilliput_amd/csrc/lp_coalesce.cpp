// lp_coalesce.cpp -- see lp_coalesce.h.
#include "lp_coalesce.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "lp_abi_guard.h"

namespace {

std::atomic<int> g_in_flight{0};
std::atomic<int> g_busy{0}; // requests queued for, or being served by, the dispatchers (lp_coalesce_busy)
thread_local int t_suppress = 0;

int env_int(const char* name, int dflt, int lo, int hi)
{
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v < lo ? lo : v > hi ? hi : v;
}
int threshold() { static const int v = env_int("LILLIPUT_HIP_COALESCE", 3, 0, 1 << 20); return v; }
int n_workers() { static const int v = env_int("LILLIPUT_HIP_COALESCE_WORKERS", 4, 1, 16); return v; }
int idle_ms() { static const int v = env_int("LILLIPUT_HIP_COALESCE_IDLE_MS", 1000, 1, 1 << 30); return v; }
size_t max_take() { static const int v = env_int("LILLIPUT_HIP_COALESCE_MAX", 16, 1, 1024); return (size_t)v; }
// Dispatchers beyond the first n_workers() that join in only while at least extra_at() requests are WAITING (about 128 in flight with the defaults):
// four dispatchers serve 12 ... 64 callers best, six to eight serve 256 best -- 9.9-10.1 k against 11.8-12.5 k images/s (profiles/r06_part_a.md section 7)
int n_extra() { static const int v = env_int("LILLIPUT_HIP_COALESCE_EXTRA", 4, 0, 16); return v; }
size_t extra_at() { static const int v = env_int("LILLIPUT_HIP_COALESCE_EXTRA_AT", 64, 1, 1 << 20); return (size_t)v; }

struct Req {
    const void* src; size_t len; void* dst; size_t cap;
    lilliput_batch_options opt;
    int status = -1;
    size_t out_len = 0;
    bool done = false;
    std::mutex mu;
    std::condition_variable cv;
};

bool same_options(const lilliput_batch_options& a, const lilliput_batch_options& b)
{
    return a.width == b.width && a.height == b.height && a.resize_method == b.resize_method && a.normalize_orientation == b.normalize_orientation &&
           a.jpeg_quality == b.jpeg_quality && a.jpeg_progressive == b.jpeg_progressive;
}

// The dispatchers of one device.
struct Dispatch {
    int device;
    std::mutex mu;
    std::condition_variable cv;
    std::condition_variable cv_extra;   // the extra dispatchers wait here (see n_extra)
    std::deque<Req*> q;
    std::vector<std::thread> th;
    bool stop = false;

    explicit Dispatch(int dev) : device(dev) {}

    void body(bool extra = false)
    {
        std::condition_variable& cv = extra ? cv_extra : this->cv;
        LpCoalesceSuppress inner; // whatever this batch calls back into the one-image path stays on this thread
        lilliput_hip_batch batch = nullptr;
        std::vector<Req*> take;
        std::vector<lilliput_batch_item> items;
        for (;;) {
            take.clear();
            {
                std::unique_lock<std::mutex> lk(mu);
                // an idle dispatcher gives its batch -- engines, streams, the arenas of its largest chunk -- back after a while: what the
                // process holds follows the calls in flight, as with the engine pool of the direct route
                while (!stop && (extra ? q.size() < extra_at() : q.empty())) {
                    if (!batch) { cv.wait(lk); continue; }
                    if (cv.wait_for(lk, std::chrono::milliseconds(idle_ms())) == std::cv_status::timeout && (extra ? q.size() < extra_at() : q.empty()) && !stop) {
                        lk.unlock();
                        lilliput_hip_batch_destroy(batch);
                        batch = nullptr;
                        lk.lock();
                    }
                }
                if (stop) break;
                // the oldest request and every waiting one with the same options, in arrival order
                // how many: at most LILLIPUT_HIP_COALESCE_MAX (16 since round 6: 64 callers are served faster by four dispatchers with 16 requests
                // each than by two with 32 -- 8.7-9.7 k against 5.8-7.8 k images/s, p99 10-14 ms against 20-70; a limit that grows with the
                // queue was measured and bought nothing at 256 callers: profiles/r06_part_a.md)
                const size_t limit = max_take();
                const lilliput_batch_options key = q.front()->opt;
                for (auto it = q.begin(); it != q.end() && take.size() < limit;) {
                    if (same_options((*it)->opt, key)) { take.push_back(*it); it = q.erase(it); }
                    else ++it;
                }
            }
            if (!batch) batch = lilliput_hip_batch_create(device);
            items.assign(take.size(), lilliput_batch_item());
            for (size_t i = 0; i < take.size(); i++) {
                memset(&items[i], 0, sizeof(items[i]));
                items[i].src = take[i]->src; items[i].src_len = take[i]->len;
                items[i].dst = take[i]->dst; items[i].dst_cap = take[i]->cap;
                items[i].status = LILLIPUT_ERR_DEVICE;
            }
            if (batch) (void)lilliput_hip_batch_transform(batch, items.data(), items.size(), &take[0]->opt);
            for (size_t i = 0; i < take.size(); i++) {
                Req* r = take[i];
                std::lock_guard<std::mutex> lk(r->mu);
                r->status = batch ? items[i].status : LILLIPUT_ERR_DEVICE;
                r->out_len = items[i].dst_len;
                r->done = true;
                r->cv.notify_one(); // under the lock: the requester destroys *r as soon as it has seen `done`, which it cannot before the lock is released
            }
        }
        if (batch) lilliput_hip_batch_destroy(batch);
    }

    void start()
    {
        for (int i = 0; i < n_workers(); i++) th.emplace_back([this] { body(); });
        for (int i = 0; i < n_extra(); i++) th.emplace_back([this] { body(true); });
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(mu); if (stop) return; stop = true; }
        cv.notify_all();
        cv_extra.notify_all();
        for (auto& t : th) t.join();
        th.clear();
        // requests that were still queued: back to their callers, who take the direct route
        for (Req* r : q) { std::lock_guard<std::mutex> lk(r->mu); r->status = LILLIPUT_ERR_DEVICE; r->done = true; r->cv.notify_one(); }
        q.clear();
    }
};

// Pinned staging slots for the requests' sources. A request's bytes usually sit in pageable memory (a Go []byte). Left there, the
// dispatcher's ingest thread copies all of a dispatch's sources into its pinned slot one after the other before the first transfer can
// start -- 16 x 4.2 MB, 3 - 5 ms on the critical path of every dispatch. Instead every CALLER copies its own source into a pinned slot
// before it queues the request (64 callers copy in parallel, off the dispatchers' path) and the batch reads it in place like any
// pinned source (lp_hostmem.h). Slots are power-of-two sized arenas from lilliput_hip_host_alloc, kept on free lists per size class
// and device, at most LILLIPUT_HIP_COALESCE_PINNED_MB in all; when the bound is reached, or for sources above 64 MiB, the request goes
// in as it is. OFF by default (0): on the measurement boxes the container is granted 16 CPUs, and 64 / 256 callers copying 4.2 MB each
// with plain memcpy cost more CPU time than the dispatchers' streaming-store stagers do for the same bytes -- 5.1 k / 6.1 k images/s
// with the slots against 5.4 k / 9.3 k without (profiles/r04_a_service.md). For hosts with CPUs to spare.
struct StagePool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_slots; // (device, class bytes) -> idle slots
    size_t total = 0;
};
StagePool& stage_pool()
{
    static StagePool* p = new StagePool(); // never destroyed (slots are given back to the runtime at exit by the registry's hook)
    return *p;
}
size_t stage_cap_bytes() { static const size_t v = (size_t)env_int("LILLIPUT_HIP_COALESCE_PINNED_MB", 0, 0, 1 << 20) << 20; return v; }
size_t stage_class(size_t n) { size_t c = (size_t)256 << 10; while (c < n) c <<= 1; return c; }
void* stage_acquire(int device, size_t len, size_t* cls)
{
    if (len > ((size_t)64 << 20) || !stage_cap_bytes()) return nullptr;
    const size_t c = stage_class(len + 64);
    *cls = c;
    StagePool& P = stage_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto& fl = P.free_slots[std::make_pair(device, c)];
        if (!fl.empty()) { void* p = fl.back(); fl.pop_back(); return p; }
        if (P.total + c > stage_cap_bytes()) return nullptr;
        P.total += c;
    }
    void* p = lilliput_hip_host_alloc(c, device);
    if (!p) { std::lock_guard<std::mutex> lk(P.mu); P.total -= c; }
    return p;
}
void stage_release(int device, size_t cls, void* p)
{
    StagePool& P = stage_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    P.free_slots[std::make_pair(device, cls)].push_back(p);
}
void stage_pool_free_all()
{
    StagePool& P = stage_pool();
    std::map<std::pair<int, size_t>, std::vector<void*>> all;
    { std::lock_guard<std::mutex> lk(P.mu); all.swap(P.free_slots); P.total = 0; }
    for (auto& kv : all) for (void* p : kv.second) lilliput_hip_host_free(p);
}

struct Registry {
    std::mutex mu;
    std::map<int, std::unique_ptr<Dispatch>> by_device;
    bool closed = false;
};
Registry& registry()
{
    static Registry* r = new Registry(); // never destroyed: used from an atexit handler
    return *r;
}

Dispatch* dispatch_for(int device)
{
    Registry& R = registry();
    std::lock_guard<std::mutex> lk(R.mu);
    if (R.closed) return nullptr;
    auto it = R.by_device.find(device);
    if (it != R.by_device.end()) return it->second.get();
    // The dispatchers hold engines (streams, arenas) that must go before the HIP runtime does: an atexit handler registered after the
    // runtime's own (the runtime is up: the caller has parsed a header through the one-image ABI... make sure) runs before it.
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return nullptr; }
    static bool hooked = false;
    if (!hooked) {
        hooked = true;
        atexit([] {
            Registry& R2 = registry();
            std::map<int, std::unique_ptr<Dispatch>> all;
            { std::lock_guard<std::mutex> lk2(R2.mu); R2.closed = true; all.swap(R2.by_device); }
            // the dispatchers are stopped and then LEAKED, like the registry itself: a thread that is still inside a Transform at process
            // exit may hold a Dispatch* from dispatch_for() and is about to lock its mutex or queue a request into it
            for (auto& kv : all) { kv.second->shutdown(); (void)kv.second.release(); }
            stage_pool_free_all();
        });
    }
    auto d = std::unique_ptr<Dispatch>(new Dispatch(device));
    d->start();
    Dispatch* p = d.get();
    R.by_device[device] = std::move(d);
    return p;
}

} // namespace

LpTransformInFlight::LpTransformInFlight() { now = g_in_flight.fetch_add(1, std::memory_order_relaxed) + 1; }
LpTransformInFlight::~LpTransformInFlight() { g_in_flight.fetch_sub(1, std::memory_order_relaxed); }

LpCoalesceSuppress::LpCoalesceSuppress() { prev = t_suppress; t_suppress = 1; }
LpCoalesceSuppress::~LpCoalesceSuppress() { t_suppress = prev; }

int lp_coalesce_busy() { return g_busy.load(std::memory_order_relaxed); }
bool lp_coalesce_suppressed() { return t_suppress != 0; }

bool lp_coalesce_wanted(int in_flight)
{
    const int t = threshold();
    return t > 0 && !t_suppress && in_flight >= t;
}

int lp_coalesce_transform_status(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& opt, size_t* out_len)
{
    *out_len = 0;
    if (!src || !len) return LILLIPUT_ERR_INVALID_IMAGE; // nothing to stage (a deferred chain whose source was dropped never gets here)
    Dispatch* D = dispatch_for(device);
    if (!D) return LILLIPUT_ERR_DEVICE;
    Req r;
    r.src = src; r.len = len; r.dst = dst; r.cap = cap; r.opt = opt;
    struct Busy { Busy() { g_busy.fetch_add(1, std::memory_order_relaxed); } ~Busy() { g_busy.fetch_sub(1, std::memory_order_relaxed); } } busy;
    // the caller's own copy into pinned memory (see StagePool); a source that is pinned already travels as it is
    size_t cls = 0;
    void* slot = lilliput_hip_host_is_pinned(src, len) ? nullptr : stage_acquire(device, len, &cls);
    if (slot) { memcpy(slot, src, len); r.src = slot; }
    bool queued = false, many = false;
    {
        std::lock_guard<std::mutex> lk(D->mu);
        if (!D->stop) { D->q.push_back(&r); queued = true; many = D->q.size() >= extra_at(); }
    }
    if (queued) {
        D->cv.notify_one();
        if (many) D->cv_extra.notify_one();
        std::unique_lock<std::mutex> lk(r.mu);
        r.cv.wait(lk, [&] { return r.done; });
    }
    if (slot) stage_release(device, cls, slot);
    if (!queued) return LILLIPUT_ERR_DEVICE;
    if (r.status == LILLIPUT_OK) *out_len = r.out_len;
    return r.status;
}

bool lp_coalesce_transform(int device, const void* src, size_t len, void* dst, size_t cap, const lilliput_batch_options& opt, size_t* out_len)
{
    return lp_coalesce_transform_status(device, src, len, dst, cap, opt, out_len) == LILLIPUT_OK;
}

// Part B: one image through the shared dispatchers, whatever the number of calls in flight -- the entry point a Go build's Transform
// calls for a static source with JPEG output. The answer is the batched path's for that item.
extern "C" int lilliput_hip_transform_one(int device, const void* src, size_t src_len, const lilliput_batch_options* opt, void* dst, size_t dst_cap, size_t* dst_len)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    size_t n = 0;
    if (dst_len) *dst_len = 0;
    if (!src || !src_len || !opt || !dst || !dst_cap) return LILLIPUT_ERR_INVALID_IMAGE;
    if (t_suppress) return LILLIPUT_ERR_DEVICE; // called from inside a batch: it would wait for itself
    const int rc = lp_coalesce_transform_status(device, src, src_len, dst, dst_cap, *opt, &n);
    if (dst_len) *dst_len = n;
    return rc;
}
LP_ABI_CATCH("lilliput_hip_transform_one", return LILLIPUT_ERR_DEVICE)
