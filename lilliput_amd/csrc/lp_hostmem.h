// lp_hostmem.h -- where the caller's encoded bytes are, as far as the DMA engines are concerned.
//
// The reference decodes from the caller's []byte in place (/root/reference/opencv.cpp:99-171: opencv_decoder_create wraps the Go
// slice, nothing is copied). The batched path keeps that: a source that already sits in pinned memory -- because the service read it
// into a lilliput_hip_host_alloc arena, or registered its receive buffers once with lilliput_hip_host_register -- is copied to the
// device straight from where it is. Any other source goes through the engine's pinned slot (one host memcpy) -- or, with
// LILLIPUT_HIP_INGEST=register, is pinned for the duration of the call (hipHostRegister on its page range, every distinct range once
// however many items share it) and copied from there too, except for sources that cannot be pinned (tiny ones, ranges that share a
// page with a live registration, a refused hipHostRegister). Per-call registration is opt-in because the driver makes it slower than
// the copy it saves (measured: profiles/r03_a_ingest.md).
//
// One process-wide table of page ranges answers "is [p, p + n) pinned?" and counts the users of a temporary registration, so the same
// buffer appearing twice in a batch -- or in two batches running at once -- is registered exactly once (ROCclr aborts the process on
// a second hipHostRegister of the same pages: "Memobj map does not have ptr").
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

enum LpIngestMode {
    LP_INGEST_REGISTER = 0,   // known-pinned sources direct, large pageable ones registered for the call, the rest staged. OPT-IN: on this driver a
                              // first-time hipHostRegister of 4 MB costs 0.3 - 2 ms (14 GB/s at best, serialised in the runtime) -- several times
                              // slower than copying the bytes through a pinned slot (profiles/r03_a_ingest.md)
    LP_INGEST_STAGED = 1,     // everything through the pinned slot (the round-2 pipeline; A/B measurements)
    LP_INGEST_PINNED_ONLY = 2 // the default ("auto"): known-pinned sources direct, everything else staged (no per-call registration)
};
LpIngestMode lp_ingest_mode();      // LILLIPUT_HIP_INGEST = auto (= pinned) | register | staged

// True when every byte of [p, p + n) lies in memory this process has pinned through this library (arena or explicit registration).
// *dev_delta (optional): device address of a byte = its host address + *dev_delta.
// *base (optional): start of the pinned range that holds the bytes (two sources with the same base may be fetched with one copy).
bool lp_host_is_pinned(const void* p, size_t n, ptrdiff_t* dev_delta = nullptr, uintptr_t* base = nullptr);

// Temporary registrations of one chunk of a batch. add() tries to make [p, p + n) DMA-able and says whether it is; release() drops
// what add() took (the last user of a range unregisters it). Not thread-safe by itself -- one scope per upload slot -- but any number
// of scopes may be active at once.
class LpPinScope {
public:
    ~LpPinScope() { release(); }
    bool add(const void* p, size_t n, ptrdiff_t* dev_delta = nullptr, uintptr_t* base = nullptr);
    void release();
    size_t registered_bytes() const { return reg_bytes_; }  // bytes this scope registered itself (not those it found pinned)
    double register_ms() const { return reg_ms_; }
private:
    std::vector<uintptr_t> held_;   // start addresses of the table entries this scope holds a reference on
    size_t reg_bytes_ = 0;
    double reg_ms_ = 0;
};

// NUMA placement of the ingest threads: the CPUs of the node the device hangs off (empty when unknown or switched off with
// LILLIPUT_HIP_NUMA=0). lp_bind_thread_near(device) pins the calling thread to them; returns the node or -1.
int lp_device_numa_node(int device);
int lp_bind_thread_near(int device);

// Host CPUs this process can really use at once: the hardware threads it may run on (affinity mask), cut down to the container's cgroup
// CPU quota when there is one (cpu.max; the gpurun boxes show 256 threads and grant 16 -- scripts/host_scale.cpp), and shared between the
// ranks of a node (LOCAL_WORLD_SIZE under torchrun, else the visible devices). Worker pools of host codecs are sized from it.
unsigned lp_usable_cpus_per_device();
bool lp_cpu_quota_limited(); // the container grants clearly fewer CPUs than it shows (cgroup quota below the affinity mask)
