// lp_guard.cpp -- see lp_guard.h.
#include "lp_guard.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/lilliput_hip.h"
#include "lp_abi_guard.h"

namespace {

const uint8_t kCanary = 0xC5;

size_t guard_align() // 0 = off
{
    static const size_t a = [] {
        const char* e = getenv("LILLIPUT_HIP_GUARD");
        if (!e || !*e) return (size_t)0;
        long v = atol(e);
        if (v <= 0) return (size_t)0;
        if (v == 1) v = 64;
        size_t p = 16; // dwordx4 accesses of the kernels need 16; hipMalloc itself promises 256
        while (p < (size_t)v && p < 4096) p <<= 1;
        return p;
    }();
    return a;
}
bool guard_log()
{
    static const bool on = getenv("LILLIPUT_HIP_GUARD_LOG") && atoi(getenv("LILLIPUT_HIP_GUARD_LOG")) != 0;
    return on;
}
size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DevRec { void* base; size_t reserved, mapped, body, bytes; hipMemGenericAllocationHandle_t handle; const char* tag; int device; };
struct PinRec { void* base; size_t total, inner, body, bytes; const char* tag; };
struct State {
    std::mutex mu;
    std::map<uintptr_t, DevRec> dev;
    std::map<uintptr_t, PinRec> pin;
    size_t n_alloc = 0, n_violations = 0, live_bytes = 0, peak_bytes = 0;
};
State& state()
{
    static State* s = new State(); // never destroyed: buffers are freed from atexit handlers too
    return *s;
}

int guarded_dev_malloc(void** out, size_t bytes, const char* tag)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return 1; }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) { (void)hipGetLastError(); return 1; }
    const size_t A = guard_align();
    const size_t body = up(bytes ? bytes : 1, A);
    const size_t mapped = up(body + 256, gran);            // at least 256 canary bytes in front of the buffer
    const size_t reserved = mapped + 2 * gran;
    void* base = nullptr;
    if (hipMemAddressReserve(&base, reserved, gran, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return 1; }
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemAddressFree(base, reserved); return 1; }
    uint8_t* lo = static_cast<uint8_t*>(base) + gran;
    if (hipMemMap(lo, mapped, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); (void)hipMemAddressFree(base, reserved); return 1; }
    // every device of the process may touch the buffer (the node API copies between engines of different devices through host memory
    // only, but hipMalloc'ed memory is peer-visible once peers are enabled; keep to the owner here)
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(lo, mapped, &acc, 1) != hipSuccess) {
        (void)hipGetLastError(); (void)hipMemUnmap(lo, mapped); (void)hipMemRelease(h); (void)hipMemAddressFree(base, reserved);
        return 1;
    }
    uint8_t* p = lo + (mapped - body);
    {   // the canary band: written by a synchronous copy and read back at once, so that a band that is wrong at the free is known to have been
        // right here (the first run of this mode filled it with hipMemset and found whole first pages of fresh mappings "overwritten")
        std::vector<uint8_t> band(mapped - body, kCanary), back(mapped - body, 0);
        const hipError_t e1 = hipMemcpy(lo, band.data(), band.size(), hipMemcpyHostToDevice);
        const hipError_t e2 = hipMemcpy(back.data(), lo, back.size(), hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || e2 != hipSuccess || memcmp(band.data(), back.data(), band.size()) != 0) {
            (void)hipGetLastError();
            fprintf(stderr, "[lilliput_hip guard] canary band of %s (%zu B) did not take: write %d, read %d\n", tag ? tag : "?", bytes, (int)e1, (int)e2);
        }
    }
    State& s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    s.dev[(uintptr_t)p] = DevRec{base, reserved, mapped, body, bytes, h, tag, device};
    s.n_alloc++;
    s.live_bytes += mapped;
    if (s.live_bytes > s.peak_bytes) s.peak_bytes = s.live_bytes;
    if (guard_log())
        fprintf(stderr, "[lilliput_hip guard] dev  #%zu %-14s %12zu B  [%p, %p)  unmapped from %p\n", s.n_alloc, tag ? tag : "?", bytes, (void*)p, (void*)(p + body), (void*)(lo + mapped));
    *out = p;
    return 0;
}

bool guarded_dev_free(void* p)
{
    DevRec r;
    {
        State& s = state();
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.dev.find((uintptr_t)p);
        if (it == s.dev.end()) return false;
        r = it->second;
        s.dev.erase(it);
        s.live_bytes -= r.mapped;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != r.device) (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();
    uint8_t* lo = static_cast<uint8_t*>(r.base) + (r.reserved - r.mapped) / 2;
    const size_t gap = r.mapped - r.body;
    std::vector<uint8_t> h(gap);
    if (hipMemcpy(h.data(), lo, gap, hipMemcpyDeviceToHost) == hipSuccess) {
        size_t bad = 0, first = 0, last = 0;
        for (size_t i = 0; i < gap; i++)
            if (h[i] != kCanary) { if (!bad) first = i; bad++; last = i; }
        if (bad) {
            char hex[3 * 16 + 1] = {0};
            for (size_t i = 0; i < 16 && first + i < gap; i++) snprintf(hex + 3 * i, 4, "%02x ", h[first + i]);
            fprintf(stderr, "[lilliput_hip guard] CANARY of %s (%zu B at %p) overwritten: %zu bytes in [-%zu, -%zu) before the buffer (band %zu B, mapping starts at -%zu); first bytes: %s\n",
                    r.tag ? r.tag : "?", r.bytes, p, bad, gap - first, gap - last - 1, gap, gap, hex);
            State& s = state();
            std::lock_guard<std::mutex> lk(s.mu);
            s.n_violations++;
        }
    } else
        (void)hipGetLastError();
    (void)hipMemUnmap(lo, r.mapped);
    (void)hipMemRelease(r.handle);
    // The address range is NOT given back: a later reservation must never land on an address some kernel may still hold (a stale pointer
    // then faults instead of reading a stranger's buffer), and no mapping is ever made twice at one address. 47 bits of address space
    // outlast any run of this mode. LILLIPUT_HIP_GUARD_REUSE_VA=1 frees the range as the first version did.
    static const bool reuse_va = getenv("LILLIPUT_HIP_GUARD_REUSE_VA") && atoi(getenv("LILLIPUT_HIP_GUARD_REUSE_VA")) != 0;
    if (reuse_va) (void)hipMemAddressFree(r.base, r.reserved);
    if (prev >= 0 && prev != r.device) (void)hipSetDevice(prev);
    return true;
}

int guarded_pinned_malloc(void** out, size_t bytes, bool portable, const char* tag)
{
    const size_t page = (size_t)sysconf(_SC_PAGESIZE) > 0 ? (size_t)sysconf(_SC_PAGESIZE) : 4096;
    const size_t A = guard_align();
    const size_t body = up(bytes ? bytes : 1, A);
    const size_t inner = up(body + 256, page);
    const size_t total = inner + 2 * page;
    void* base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return 1;
    uint8_t* lo = static_cast<uint8_t*>(base) + page;
    (void)mprotect(base, page, PROT_NONE);
    (void)mprotect(lo + inner, page, PROT_NONE);
    memset(lo, kCanary, inner - body);
    if (hipHostRegister(lo, inner, hipHostRegisterMapped | (portable ? hipHostRegisterPortable : 0u)) != hipSuccess) {
        (void)hipGetLastError();
        munmap(base, total);
        return 1;
    }
    uint8_t* p = lo + (inner - body);
    State& s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    s.pin[(uintptr_t)p] = PinRec{base, total, inner, body, bytes, tag};
    s.n_alloc++;
    if (guard_log())
        fprintf(stderr, "[lilliput_hip guard] host #%zu %-14s %12zu B  [%p, %p)  no access from %p\n", s.n_alloc, tag ? tag : "?", bytes, (void*)p, (void*)(p + body), (void*)(lo + inner));
    *out = p;
    return 0;
}

bool guarded_pinned_free(void* p)
{
    PinRec r;
    {
        State& s = state();
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.pin.find((uintptr_t)p);
        if (it == s.pin.end()) return false;
        r = it->second;
        s.pin.erase(it);
    }
    uint8_t* lo = static_cast<uint8_t*>(r.base) + (r.total - r.inner) / 2;
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister(lo);
    size_t bad = 0;
    for (size_t i = 0; i < r.inner - r.body; i++) bad += lo[i] != kCanary;
    if (bad) {
        fprintf(stderr, "[lilliput_hip guard] CANARY of pinned %s (%zu B at %p) overwritten: %zu bytes before the buffer\n", r.tag ? r.tag : "?", r.bytes, p, bad);
        State& s = state();
        std::lock_guard<std::mutex> lk(s.mu);
        s.n_violations++;
    }
    munmap(r.base, r.total);
    return true;
}

} // namespace

bool lp_guard_on() { return guard_align() != 0; }

int lp_dev_malloc(void** p, size_t bytes, const char* tag)
{
    if (guard_align()) return guarded_dev_malloc(p, bytes, tag);
    if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

void lp_dev_free(void* p)
{
    if (!p) return;
    if (guard_align() && guarded_dev_free(p)) return;
    (void)hipFree(p);
}

int lp_pinned_malloc(void** p, size_t bytes, bool portable, const char* tag)
{
    if (guard_align()) return guarded_pinned_malloc(p, bytes, portable, tag);
    if (hipHostMalloc(p, bytes, hipHostMallocMapped | (portable ? hipHostMallocPortable : 0u)) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

void lp_pinned_free(void* p)
{
    if (!p) return;
    if (guard_align() && guarded_pinned_free(p)) return;
    (void)hipHostFree(p);
}

// out[0] = guard alignment (0: the mode is off), out[1] = allocations so far, out[2] = canary violations seen at frees,
// out[3] = peak of mapped device bytes
extern "C" void lilliput_hip_guard_stats(size_t out[4])
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    State& s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    out[0] = guard_align(); out[1] = s.n_alloc; out[2] = s.n_violations; out[3] = s.peak_bytes;
}
LP_ABI_CATCH("lilliput_hip_guard_stats", return)
