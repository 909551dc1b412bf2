// lp_hostmem.cpp -- see lp_hostmem.h.
#include "lp_hostmem.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>

#include "../../include/lilliput_hip.h"
#include "lp_guard.h"
#include "lp_abi_guard.h"

namespace {

enum Kind { kArena, kExplicit, kTemp };
struct Entry { uintptr_t end; Kind kind; uint32_t refs; ptrdiff_t dev_delta; };
ptrdiff_t device_delta(void* host) // device alias of a pinned / registered host address
{
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) { (void)hipGetLastError(); return 0; }
    return (ptrdiff_t)((uintptr_t)d - (uintptr_t)host);
}

struct Table {
    std::mutex mu;
    std::map<uintptr_t, Entry> by_start;    // disjoint ranges
    // the entry that covers [a, b), or end()
    std::map<uintptr_t, Entry>::iterator covering(uintptr_t a, uintptr_t b)
    {
        auto it = by_start.upper_bound(a);
        if (it == by_start.begin()) return by_start.end();
        --it;
        return (it->first <= a && b <= it->second.end) ? it : by_start.end();
    }
    bool overlaps(uintptr_t a, uintptr_t b)
    {
        auto it = by_start.lower_bound(a);
        if (it != by_start.end() && it->first < b) return true;
        if (it == by_start.begin()) return false;
        --it;
        return it->second.end > a;
    }
};
Table& table()
{
    static Table* t = new Table(); // never destroyed: batches may be torn down from atexit handlers
    return *t;
}

uintptr_t page_size()
{
    static const uintptr_t p = (uintptr_t)sysconf(_SC_PAGESIZE) > 0 ? (uintptr_t)sysconf(_SC_PAGESIZE) : 4096u;
    return p;
}

size_t register_min()
{
    static const size_t v = getenv("LILLIPUT_HIP_REGISTER_MIN") ? (size_t)strtoull(getenv("LILLIPUT_HIP_REGISTER_MIN"), nullptr, 10) : (size_t)64 << 10;
    return v;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

static int g_ingest_mode = -1; // -1: read LILLIPUT_HIP_INGEST on first use
static LpIngestMode parse_ingest(const char* e)
{
    if (!e || !*e || !strcmp(e, "auto") || !strcmp(e, "pinned")) return LP_INGEST_PINNED_ONLY;
    if (!strcmp(e, "register")) return LP_INGEST_REGISTER;
    if (!strcmp(e, "staged")) return LP_INGEST_STAGED;
    fprintf(stderr, "lilliput_hip: LILLIPUT_HIP_INGEST=%s is not one of auto | pinned | register | staged; using auto\n", e);
    return LP_INGEST_PINNED_ONLY;
}
LpIngestMode lp_ingest_mode()
{
    int m = __atomic_load_n(&g_ingest_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        m = (int)parse_ingest(getenv("LILLIPUT_HIP_INGEST"));
        __atomic_store_n(&g_ingest_mode, m, __ATOMIC_RELAXED);
    }
    return (LpIngestMode)m;
}
extern "C" int lilliput_hip_set_ingest_mode(const char* mode)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    const int prev = (int)lp_ingest_mode();
    __atomic_store_n(&g_ingest_mode, (int)parse_ingest(mode), __ATOMIC_RELAXED);
    return prev;
}
LP_ABI_CATCH("lilliput_hip_set_ingest_mode", return 0)

bool lp_host_is_pinned(const void* p, size_t n, ptrdiff_t* dev_delta, uintptr_t* base)
{
    if (!p || !n) return false;
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.covering((uintptr_t)p, (uintptr_t)p + n);
    if (it == t.by_start.end() || it->second.kind == kTemp) return false; // a temporary registration belongs to whoever holds it
    if (dev_delta) *dev_delta = it->second.dev_delta;
    if (base) *base = it->first;
    return true;
}

bool LpPinScope::add(const void* p, size_t n, ptrdiff_t* dev_delta, uintptr_t* base)
{
    if (!p || !n) return false;
    const uintptr_t lo = (uintptr_t)p, hi = lo + n, ps = page_size();
    const uintptr_t a = lo & ~(ps - 1), b = (hi + ps - 1) & ~(ps - 1);
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.covering(lo, hi);
    if (it != t.by_start.end()) {
        if (it->second.kind == kTemp) { it->second.refs++; held_.push_back(it->first); }
        if (dev_delta) *dev_delta = it->second.dev_delta;
        if (base) *base = it->first;
        return true;
    }
    if (n < register_min() || t.overlaps(a, b)) return false; // shares pages with a live registration: the staged route
    const double t0 = now_ms();
    if (hipHostRegister((void*)a, b - a, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError(); // e.g. a read-only mapping: not an error of the call, the item is staged instead
        return false;
    }
    reg_ms_ += now_ms() - t0;
    reg_bytes_ += b - a;
    const ptrdiff_t dd = device_delta((void*)a);
    t.by_start[a] = Entry{b, kTemp, 1, dd};
    held_.push_back(a);
    if (dev_delta) *dev_delta = dd;
    if (base) *base = a;
    return true;
}

void LpPinScope::release()
{
    reg_bytes_ = 0;
    reg_ms_ = 0;
    if (held_.empty()) return;
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    for (uintptr_t a : held_) {
        auto it = t.by_start.find(a);
        if (it == t.by_start.end() || it->second.kind != kTemp) continue;
        if (--it->second.refs == 0) {
            (void)hipHostUnregister((void*)a);
            t.by_start.erase(it);
        }
    }
    held_.clear();
}

// ---- NUMA
static bool numa_enabled()
{
    static const bool on = !(getenv("LILLIPUT_HIP_NUMA") && atoi(getenv("LILLIPUT_HIP_NUMA")) == 0);
    return on;
}

int lp_device_numa_node(int device)
{
    if (!numa_enabled()) return -1;
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    int node = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, device) == hipSuccess) {
        for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
        if (FILE* f = fopen(path.c_str(), "r")) {
            if (fscanf(f, "%d", &node) != 1) node = -1;
            fclose(f);
        }
    } else
        (void)hipGetLastError();
    cache[device] = node;
    return node;
}

int lp_bind_thread_near(int device)
{
    const int node = lp_device_numa_node(device);
    if (node < 0) return -1;
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return -1;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return -1;
    int n = 0;
    for (char* s = list; *s && *s != '\n';) { // "0-63,128-191"
        char* e = nullptr;
        const long a = strtol(s, &e, 10);
        if (e == s) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); n++; }
        if (*e != ',') break;
        s = e + 1;
    }
    if (n < 2) return -1; // a cpuset that leaves (nearly) nothing of the node: stay where the caller put us
    return sched_setaffinity(0, sizeof(want), &want) == 0 ? node : -1;
}

// ---- Part B: pinned arenas and long-lived registrations
extern "C" void* lilliput_hip_host_alloc(size_t bytes, int device)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!bytes) return nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    void* p = nullptr;
    // ROCr places the pages on the NUMA node closest to the current device; mapped + portable: every device of the node may copy from it
    const int e = lp_pinned_malloc(&p, bytes, true, "host_alloc");
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e || !p) return nullptr;
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    const ptrdiff_t dd = device_delta(p);
    t.by_start[(uintptr_t)p] = Entry{(uintptr_t)p + bytes, kArena, 1, dd};
    return p;
}
LP_ABI_CATCH("lilliput_hip_host_alloc", return nullptr)

extern "C" void lilliput_hip_host_free(void* p)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!p) return;
    {
        Table& t = table();
        std::lock_guard<std::mutex> lk(t.mu);
        auto it = t.by_start.find((uintptr_t)p);
        if (it == t.by_start.end() || it->second.kind != kArena) return; // not ours
        t.by_start.erase(it);
    }
    lp_pinned_free(p);
}
LP_ABI_CATCH("lilliput_hip_host_free", return)

extern "C" int lilliput_hip_host_register(void* p, size_t bytes)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!p || !bytes) return LILLIPUT_ERR_INVALID_IMAGE;
    const uintptr_t ps = page_size(), a = (uintptr_t)p & ~(ps - 1), b = ((uintptr_t)p + bytes + ps - 1) & ~(ps - 1);
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.covering((uintptr_t)p, (uintptr_t)p + bytes);
    if (it != t.by_start.end() && it->second.kind != kTemp) return LILLIPUT_OK; // already pinned for good
    if (t.overlaps(a, b)) return LILLIPUT_ERR_DEVICE;
    if (hipHostRegister((void*)a, b - a, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return LILLIPUT_ERR_DEVICE; }
    t.by_start[a] = Entry{b, kExplicit, 1, device_delta((void*)a)};
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_hip_host_register", return LILLIPUT_ERR_DEVICE)

extern "C" int lilliput_hip_host_unregister(void* p)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    if (!p) return LILLIPUT_ERR_INVALID_IMAGE;
    const uintptr_t ps = page_size(), a = (uintptr_t)p & ~(ps - 1);
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.by_start.find(a);
    if (it == t.by_start.end() || it->second.kind != kExplicit) return LILLIPUT_ERR_INVALID_IMAGE;
    (void)hipHostUnregister((void*)a);
    t.by_start.erase(it);
    return LILLIPUT_OK;
}
LP_ABI_CATCH("lilliput_hip_host_unregister", return LILLIPUT_ERR_DEVICE)

extern "C" int lilliput_hip_host_is_pinned(const void* p, size_t bytes) { return lp_host_is_pinned(p, bytes) ? 1 : 0; }

// true when the container grants fewer CPUs than it shows (a cgroup CPU quota below the affinity mask: the gpurun boxes show 256 hardware
// threads and grant 16): spinning waits then burn granted CPU time that the host codecs need (profiles/r04_j_firehose_host.md)
bool lp_cpu_quota_limited()
{
    static const bool v = [] {
        double shown = 1, quota = -1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) shown = (double)CPU_COUNT(&set);
        else shown = (double)sysconf(_SC_NPROCESSORS_ONLN);
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            double per = 0;
            if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            double q = -1, per = 0;
            if (fscanf(g, "%lf", &q) != 1) q = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) != 1) per = 0; fclose(h); }
            if (q > 0 && per > 0) quota = q / per;
        }
        return quota > 0 && quota < shown * 0.75;
    }();
    return v;
}

unsigned lp_usable_cpus_per_device()
{
    static const unsigned v = [] {
        double cpus = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = (double)CPU_COUNT(&set);
        else cpus = (double)sysconf(_SC_NPROCESSORS_ONLN);
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64] = {0};
            double per = 0;
            if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) cpus = std::min(cpus, atof(q) / per);
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
            double q = -1, per = 0;
            if (fscanf(g, "%lf", &q) != 1) q = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) != 1) per = 0; fclose(h); }
            if (q > 0 && per > 0) cpus = std::min(cpus, q / per);
        }
        int share = 0;
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) share = atoi(e);
        if (share <= 0) { if (hipGetDeviceCount(&share) != hipSuccess) { (void)hipGetLastError(); share = 1; } }
        cpus /= (double)std::max(1, share);
        return (unsigned)std::max(1.0, cpus + 0.5);
    }();
    return v;
}
