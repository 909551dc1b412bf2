// lp_engine.cpp -- see lp_engine.h.
#include "lp_engine.h"
#include "lp_hostmem.h"

#include <emmintrin.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>

#include "lp_huff_core.h"
#include "lp_launch.h"
#include "lp_jpeg_progenc.h"
#include "lp_prog_host.h"
#include "lp_abi_guard.h"

void lp_encode_upload_tables(const uint16_t code[4][256], const uint8_t len[4][256]);

// The last error text is per THREAD: the ingest stager (upload_layout / upload_commit) and the compute thread (run_decode ...) of a
// batch part work on the same engine at the same time, and each reports its own failure.
std::string& LpEngine::err_ref()
{
    static thread_local std::string e;
    return e;
}
#define err_ (LpEngine::err_ref())

// The HIP runtime multiplexes its streams over GPU_MAX_HW_QUEUES hardware queues, four by default. A batch runs four engines, each
// with a compute stream, next to the copy stream(s) of the ingest pipeline; with four queues a compute stream ends up behind a copy
// stream's barrier packets and one engine of four decodes at a third of the others' rate (measured: 8.9 k -> 10.6 k images/s end to
// end with eight queues). The variable is read when the runtime initialises, i.e. at the first HIP call of the process; a value the
// user has set is left alone, and LILLIPUT_HIP_KEEP_RUNTIME_ENV=1 keeps the library from touching the environment at all (the effect
// is process-wide: documented in include/lilliput_hip.h).
namespace {
struct LpRuntimeEnv {
    LpRuntimeEnv() { if (!(getenv("LILLIPUT_HIP_KEEP_RUNTIME_ENV") && atoi(getenv("LILLIPUT_HIP_KEEP_RUNTIME_ENV")))) setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} lp_runtime_env;
}

// ------------------------------------------------------------------------------------------------
// Growing an arena replaces its block. The old block is NOT freed on the spot: hipFree / hipHostFree synchronise the whole device, and
// stages of a chunk are enqueued back to back without host synchronisation, so a kernel still in flight may hold the old address -- with
// four engines and the batch's host workers sharing the GPU a growth in the middle of a call stalled everything for up to a second
// (one chunk of a mixed-size batch: 940 ms). Retired blocks wait in a list that is emptied at the quiet points: the end of a batch /
// node call and engine destruction (lp_retired_collect). Growth is geometric (x 1.5), so the waste is bounded.
namespace {
struct Retired {
    std::mutex mu;
    std::vector<std::pair<int, void*>> dev, host; // (owning device, block): the list is process-wide, engines of several devices feed it
};
int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
Retired& retired()
{
    static Retired* r = new Retired(); // never destroyed: engines may be torn down from atexit handlers
    return *r;
}
// guard mode (lp_guard.h): exactly what was asked for, so that the byte after it is the unmapped page
size_t grown(size_t cap, size_t bytes) { return lp_guard_on() ? bytes : std::max(bytes + bytes / 8 + 256, cap + cap / 2); }
}
void lp_retired_collect()
{
    std::vector<std::pair<int, void*>> d, h;
    {
        Retired& r = retired();
        std::lock_guard<std::mutex> lk(r.mu);
        d.swap(r.dev);
        h.swap(r.host);
    }
    if (d.empty() && h.empty()) return;
    // every device that owns a block is synchronised before its blocks go: whatever was enqueued against them has run (the list is
    // process-wide; round 3 synchronised only the caller's current device and relied on hipFree doing the rest)
    const int prev = current_device();
    std::vector<int> devs;
    for (auto& e : d) if (std::find(devs.begin(), devs.end(), e.first) == devs.end()) devs.push_back(e.first);
    for (auto& e : h) if (std::find(devs.begin(), devs.end(), e.first) == devs.end()) devs.push_back(e.first);
    for (int dev : devs) {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); continue; }
        (void)hipDeviceSynchronize();
        for (auto& e : d) if (e.first == dev) lp_dev_free(e.second);
        for (auto& e : h) if (e.first == dev) lp_pinned_free(e.second);
    }
    (void)hipSetDevice(prev);
}

// ------------------------------------------------------------------------------------------------
// Stage profile (lilliput_hip_stage_profile, include/lilliput_hip.h): device time and algorithmic bytes of the launches behind the
// one-image entry points, by kernel name -- what bench.py's roofline of the PNG / WebP / animated workloads is computed from. Off by
// default; when on, every probed launch is bracketed by two events and waited for (measurement legs only, never the timed region).
namespace {
struct StageRow { double ms = 0; double bytes = 0; uint64_t calls = 0; };
struct StageTable {
    std::mutex mu;
    std::map<std::string, StageRow> rows;
};
StageTable& stage_table()
{
    static StageTable* t = new StageTable();
    return *t;
}
std::atomic<int> g_stage_profile{0};
struct LpStageProbe {
    hipStream_t st;
    const char* name;
    double bytes;
    hipEvent_t a = nullptr, b = nullptr;
    LpStageProbe(hipStream_t stream, const char* n, double by) : st(stream), name(n), bytes(by)
    {
        if (!g_stage_profile.load(std::memory_order_relaxed)) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); a = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    ~LpStageProbe()
    {
        if (!a) return;
        float ms = 0;
        if (b && hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) {
            StageTable& T = stage_table();
            std::lock_guard<std::mutex> lk(T.mu);
            StageRow& r = T.rows[name];
            r.ms += ms; r.bytes += bytes; r.calls++;
        } else
            (void)hipGetLastError();
        (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};
}
extern "C" int lilliput_hip_stage_profile(int on)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    const int prev = g_stage_profile.exchange(on ? 1 : 0);
    if (on && !prev) {
        StageTable& T = stage_table();
        std::lock_guard<std::mutex> lk(T.mu);
        T.rows.clear();
    }
    return prev;
}
LP_ABI_CATCH("lilliput_hip_stage_profile", return 0)
extern "C" size_t lilliput_hip_stage_profile_read(char* out, size_t cap)
try { // LP_ABI_GUARD: nothing unwinds through the C ABI (lp_abi_guard.h)
    std::string s;
    {
        StageTable& T = stage_table();
        std::lock_guard<std::mutex> lk(T.mu);
        char line[256];
        for (const auto& kv : T.rows) {
            snprintf(line, sizeof(line), "%s\t%llu\t%.6f\t%.0f\n", kv.first.c_str(), (unsigned long long)kv.second.calls, kv.second.ms, kv.second.bytes);
            s += line;
        }
    }
    if (out && cap) {
        const size_t n = std::min(cap - 1, s.size());
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return s.size();
}
LP_ABI_CATCH("lilliput_hip_stage_profile_read", return 0)

LpDevBuf::~LpDevBuf() { if (p) lp_dev_free(p); }
bool LpDevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return true;
    const size_t want = grown(cap, bytes);
    void* np = nullptr;
    if (lp_dev_malloc(&np, want, tag)) { // out of memory: give the retired blocks back first, then try once more
        lp_retired_collect();
        if (lp_dev_malloc(&np, want, tag)) return false;
    }
    if (p) {
        Retired& r = retired();
        std::lock_guard<std::mutex> lk(r.mu);
        r.dev.emplace_back(current_device(), p);
    }
    p = np;
    cap = want;
    return true;
}
LpPinned::~LpPinned() { if (p) lp_pinned_free(p); }
bool LpPinned::ensure(size_t bytes)
{
    if (bytes <= cap && p) return true;
    const size_t want = grown(cap, bytes);
    void* np = nullptr;
    if (lp_pinned_malloc(&np, want, false, tag)) return false;
    if (p) { // kernels read and write these blocks through their device alias
        Retired& r = retired();
        std::lock_guard<std::mutex> lk(r.mu);
        r.host.emplace_back(current_device(), p);
    }
    p = np;
    if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) dev = p; // unified addressing: the same pointer
    cap = want;
    return true;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

LpEngine::LpEngine(int device) : device_(device)
{
    // names for the guard mode's allocation log (lp_guard.h)
#define LP_TAG(b) b.tag = #b
    LP_TAG(d_imgs_); LP_TAG(d_states_); LP_TAG(d_clean_); LP_TAG(d_rst_); LP_TAG(d_chunk_); LP_TAG(d_ckpt_); LP_TAG(d_exit_); LP_TAG(d_spec_exit_); LP_TAG(d_entry_);
    LP_TAG(d_tot_); LP_TAG(d_spec_tot_); LP_TAG(d_prefix_); LP_TAG(d_changed_); LP_TAG(d_coef_); LP_TAG(d_wide_); LP_TAG(d_wide_id_); LP_TAG(d_dc_); LP_TAG(d_dcpart_);
    LP_TAG(d_planes_); LP_TAG(d_frames_desc_); LP_TAG(h_small_); LP_TAG(h_out_); LP_TAG(h_dstate_); LP_TAG(h_desc_); LP_TAG(h_xfer_in_); LP_TAG(h_xfer_out_); LP_TAG(d_pscans_); LP_TAG(d_pstreams_);
    LP_TAG(d_pstates_); LP_TAG(d_pcoef_); LP_TAG(d_pdeps_); LP_TAG(d_pprog_); LP_TAG(heap_); LP_TAG(d_ops_); LP_TAG(d_taps_); LP_TAG(d_ranges_); LP_TAG(d_fops_); LP_TAG(d_aops_); LP_TAG(d_ataps_);
    LP_TAG(d_aranges_); LP_TAG(d_tone_); LP_TAG(d_jobs_); LP_TAG(d_estates_); LP_TAG(d_ecoef_); LP_TAG(d_blkbits_); LP_TAG(d_bits_); LP_TAG(d_hdrs_); LP_TAG(d_out_);
    LP_TAG(d_packed_); LP_TAG(d_pkoff_);
#undef LP_TAG
    for (auto& u : up_) { u.d_raw.tag = "up.d_raw"; u.d_huffs.tag = "up.d_huffs"; u.d_phuffs.tag = "up.d_phuffs"; u.stage.tag = "up.stage"; u.pcoef.tag = "up.pcoef"; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { err_ = "no HIP device visible"; return; }
    if (device_ < 0 || device_ >= n) { err_ = "HIP device index out of range"; return; }
    if (!check(hipSetDevice(device_), "hipSetDevice")) return;
    {   // LILLIPUT_HIP_BLOCKING_SYNC=1: host threads that wait for the device sleep instead of spinning (hipDeviceScheduleBlockingSync). A spin
        // costs nothing on a host with idle cores and a whole CPU per waiting caller inside a CPU quota (profiles/r04_a_service.md)
        // Default: on inside a CPU quota that is clearly smaller than the CPUs the container shows (lp_cpu_quota_limited), off otherwise.
        static const bool blocking = getenv("LILLIPUT_HIP_BLOCKING_SYNC") ? atoi(getenv("LILLIPUT_HIP_BLOCKING_SYNC")) != 0 : lp_cpu_quota_limited();
        static std::atomic<uint64_t> done_mask{0};
        if (blocking && device_ < 64 && !(done_mask.fetch_or(1ull << device_) & (1ull << device_)))
            if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    }
    // lp_engine_stream_priority_hint (lp_engine.h): the engines of the batches that serve lone callers alternate between the default and the
    // high stream priority -- the runtime keeps a pool of hardware queues per priority, so eight such engines do not pile up on the default four
    // (profiles/r06_part_a.md section 6)
    const int prio_hint = lp_engine_stream_priority_hint(-1);
    int least = 0, greatest = 0;
    if (prio_hint > 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
        if (!check(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest), "hipStreamCreate")) return;
    } else if (!check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate")) return;
    if (!check(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking), "hipStreamCreate")) return;
    for (auto& e : ev_)
        if (!check(hipEventCreate(&e), "hipEventCreate")) return;
    ok_ = true;
}

int lp_engine_stream_priority_hint(int set)
{
    thread_local int hint = 0;
    const int prev = hint;
    if (set >= 0) hint = set;
    return prev;
}

LpEngine::~LpEngine()
{
    lp_retired_collect();
    if (stream_) { (void)hipStreamSynchronize(stream_); }
    if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); }
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto& u : up_) {
        if (u.ready) (void)hipEventDestroy(u.ready);
        for (auto& e : u.ready_x) if (e) (void)hipEventDestroy(e);
    }
    if (stream_) (void)hipStreamDestroy(stream_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
}

void LpEngine::mark(int i)
{
    if (!timing_) return;
    static const bool drain = getenv("LILLIPUT_HIP_TIMING_SYNC") != nullptr && atoi(getenv("LILLIPUT_HIP_TIMING_SYNC")) != 0;
    if (drain) (void)hipStreamSynchronize(stream_);
    (void)hipEventRecord(ev_[i], stream_);
}

bool LpEngine::check(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    fprintf(stderr, "lilliput_hip: %s\n", err_.c_str());
    return false;
}

// Small descriptor upload on the compute stream without the copy engine (see lp_launch_copy_small). The bytes are staged in a
// pinned ring; a wrap waits for the stream, so nothing in flight is overwritten. dst must have room for bytes rounded up to 16.
bool LpEngine::h2d_small(void* dst, const void* src, size_t bytes)
{
    if (!bytes) return true;
    const size_t kRing = 8u << 20, need = align_up(bytes, 256);
    if (need > kRing / 2) // large lists take the ordinary path
        return check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_), "H2D descriptors");
    if (!h_desc_.p && !h_desc_.ensure(kRing)) { err_ = "pinned allocation failed"; return false; } // mapped: the kernel reads it through its device alias
    if (desc_used_ + need > kRing) {
        if (!check(hipStreamSynchronize(stream_), "descriptor ring sync")) return false;
        desc_used_ = 0;
    }
    uint8_t* stage = h_desc_.as<uint8_t>() + desc_used_;
    memcpy(stage, src, bytes);
    desc_used_ += need;
    lp_launch_copy_small(stream_, dst, static_cast<uint8_t*>(h_desc_.dev) + (stage - h_desc_.as<uint8_t>()), bytes);
    return true;
}

bool LpEngine::small_flush(SmallBatch& sb)
{
    if (sb.n) lp_launch_small_ops(stream_, sb.ops, sb.n);
    sb.n = 0;
    return true;
}
bool LpEngine::small_copy(SmallBatch& sb, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return true;
    const size_t kRing = 8u << 20, need = align_up(bytes, 256);
    if (need > kRing / 2) return check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_), "H2D descriptors"); // in stream order before the flush: fine
    if (!h_desc_.p && !h_desc_.ensure(kRing)) { err_ = "pinned allocation failed"; return false; }
    if (desc_used_ + need > kRing) { // the ring wraps: what is pending still reads its old places
        small_flush(sb);
        if (!check(hipStreamSynchronize(stream_), "descriptor ring sync")) return false;
        desc_used_ = 0;
    }
    uint8_t* stage = h_desc_.as<uint8_t>() + desc_used_;
    memcpy(stage, src, bytes);
    desc_used_ += need;
    if (sb.n == LP_SMALL_SEGS) small_flush(sb);
    sb.ops.s[sb.n++] = LpSmallSeg{dst, static_cast<uint8_t*>(h_desc_.dev) + (stage - h_desc_.as<uint8_t>()), (uint32_t)((bytes + 15) / 16), 0u};
    return true;
}
bool LpEngine::small_zero(SmallBatch& sb, void* dst, size_t bytes)
{
    if (!bytes) return true;
    if (bytes > (64u << 20) || ((uintptr_t)dst & 15u)) return check(hipMemsetAsync(dst, 0, bytes, stream_), "memset");
    if (sb.n == LP_SMALL_SEGS) small_flush(sb);
    sb.ops.s[sb.n++] = LpSmallSeg{dst, nullptr, (uint32_t)(bytes / 16), (uint32_t)(bytes & 15u)};
    return true;
}
bool LpEngine::small_d2h(SmallBatch& sb, const LpPinned& pin, void* host, const void* dev, size_t bytes)
{
    if (!bytes) return true;
    if (sb.n == LP_SMALL_SEGS) small_flush(sb);
    sb.ops.s[sb.n++] = LpSmallSeg{static_cast<uint8_t*>(pin.dev) + (static_cast<uint8_t*>(host) - pin.as<uint8_t>()), dev, (uint32_t)((bytes + 15) / 16), 0u};
    return true;
}

bool LpEngine::h2d_any(void* dst, const void* src, size_t bytes, bool dst_has_slack)
{
    if (!bytes) return true;
    // dst_has_slack: the destination is a descriptor arena allocated with room behind the payload (ensure(... + 64)): the 16-byte
    // groups of k_copy_small may spill up to 15 surplus bytes there, so an odd size need not take the synchronising route below
    if (dst_has_slack && bytes <= (2u << 20) && ((uintptr_t)dst & 15u) == 0) return h2d_small(dst, src, bytes);
    // k_copy_small moves 16-byte groups: it needs a 16-byte aligned destination and writes the size rounded UP to 16 (the surplus bytes
    // are whatever the pinned ring held). Fine for the descriptor arenas it was written for; a destination that is a view with an odd
    // offset, or one sized to the byte, takes the copy engine instead.
    if (bytes <= (2u << 20) && ((uintptr_t)dst & 15u) == 0 && (bytes & 15u) == 0) return h2d_small(dst, src, bytes);
    // the pinned buffer is reused by the next large transfer of this engine: wait for whatever still reads it
    if (!check(hipStreamSynchronize(stream_), "transfer buffer sync")) return false;
    if (!h_xfer_in_.ensure(bytes + 64)) { err_ = "pinned allocation failed"; return false; }
    memcpy(h_xfer_in_.p, src, bytes);
    return check(hipMemcpyAsync(dst, h_xfer_in_.p, bytes, hipMemcpyHostToDevice, stream_), "H2D (pinned transfer buffer)");
}

const uint8_t* LpEngine::d2h_begin(const void* dev, size_t bytes, size_t slot_off)
{
    if (!h_xfer_out_.ensure(slot_off + bytes + 64)) { err_ = "pinned allocation failed"; return nullptr; }
    uint8_t* h = h_xfer_out_.as<uint8_t>() + slot_off;
    if (!check(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, stream_), "D2H (pinned transfer buffer)")) return nullptr;
    return h;
}

// Small device -> pinned host download by a kernel (the counterpart of h2d_small); `host` lies inside the pinned buffer `pin`.
void LpEngine::d2h_small(const LpPinned& pin, void* host, const void* dev, size_t bytes)
{
    lp_launch_copy_small(stream_, static_cast<uint8_t*>(pin.dev) + (static_cast<uint8_t*>(host) - pin.as<uint8_t>()), dev, bytes);
}

size_t LpEngine::device_bytes() const
{
    size_t n = 0;
    for (const LpUpload& u : up_) n += u.d_raw.cap + u.d_huffs.cap + u.d_phuffs.cap;
    for (const LpDevBuf* b : {&d_imgs_, &d_states_, &d_clean_, &d_rst_, &d_chunk_, &d_ckpt_, &d_exit_, &d_spec_exit_, &d_entry_, &d_tot_, &d_spec_tot_, &d_prefix_, &d_changed_,
                              &d_coef_, &d_wide_, &d_wide_id_, &d_dc_, &d_dcpart_, &d_planes_, &d_frames_desc_, &d_pscans_, &d_pstreams_, &d_pstates_, &d_pcoef_, &d_pdeps_, &d_pprog_, &heap_,
                              &d_ops_, &d_taps_, &d_ranges_, &d_fops_, &d_aops_, &d_ataps_, &d_aranges_, &d_tone_, &d_jobs_, &d_estates_, &d_ecoef_, &d_blkbits_, &d_bits_, &d_hdrs_, &d_out_, &d_packed_, &d_pkoff_})
        n += b->cap;
    return n;
}

int LpEngine::sync() { return check(hipStreamSynchronize(stream_), "hipStreamSynchronize") ? LP_OK : LP_ERR_DEVICE; }

bool LpEngine::heap_reserve(size_t bytes)
{
    if (bytes <= heap_.cap) return true;
    (void)hipStreamSynchronize(stream_);
    heap_used_ = 0;
    return heap_.ensure(bytes);
}

uint8_t* LpEngine::heap_alloc(size_t bytes)
{
    size_t off = align_up(heap_used_, 256);
    if (off + bytes > heap_.cap) return nullptr;
    heap_used_ = off + bytes;
    return heap_.as<uint8_t>() + off;
}

// ------------------------------------------------------------------------------------------------
// decode
static uint32_t pick_S(size_t max_ecs_bytes)
{
    size_t bits = max_ecs_bytes * 8;
    if (bits >= (1u << 22)) return 16384;
    if (bits >= (1u << 19)) return 4096;
    if (bits >= (1u << 16)) return 1024;
    return 256;
}

size_t LpEngine::resident_round(size_t max_raw_len)
{
    if (!ok_ || hipSetDevice(device_) != hipSuccess) return 112;
    static thread_local uint32_t slots = 0;
    if (!slots) slots = lp_huff_write_slots();
    const uint32_t S = S_cfg_ ? S_cfg_ : pick_S(max_raw_len);
    const uint64_t nsub = ((uint64_t)max_raw_len * 8 + S - 1) / S + 1;
    const uint64_t wgs = (nsub + 255) / 256;
    return (size_t)std::max<uint64_t>(16, std::min<uint64_t>(512, slots / std::max<uint64_t>(1, wgs)));
}

// The unstuff kernels read whole 4 KiB chunks (16 bytes per lane, unconditionally): the last chunk of the last segment reaches up to 4095
// bytes past the data, so the arena carries that much padding. (With only the growth slack of LpDevBuf a small upload -- three small
// files -- read past its allocation: a GPU memory fault whenever the block happened to end a mapped region, one run in four of the
// two-slot node test.)
static const size_t kRawPad = 4096 + 64;

// Descriptors and arena layout of one set of sources (slot `slot`): every entropy-coded segment becomes a piece at a 16-byte
// aligned arena offset followed by 32 zero bytes. whole = size the slot's pinned buffer for the whole set (staged uploads).
static int layout_set(LpUpload& u, const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs)
{
    u.src.assign((size_t)n, LpJpeg());
    u.huffs.clear();
    u.prog.assign((size_t)n, std::vector<LpUpload::ProgScanUp>());
    u.phuffs.clear();
    // Where each scan-path image's entropy decode runs (lp_prog_host.h): 0 host threads, 1 device (a wave per progressive scan, lanes for
    // sequential ones), 2 device lanes throughout (the generic kernel: the tested reference of the wave decoder).
    u.prog_mode = u.force_host_scans ? 0 : lp_prog_entropy_mode();
    u.prog_dev.assign((size_t)n, 0);
    u.any_prog_dev = false;
    {
        size_t cand = 0;
        for (int i = 0; i < n; i++) {
            if (!hdrs[i].scan_path || hdrs[i].arith) continue; // a QM-coded scan only has a host decoder (lp_arith_host.h)
            if (hdrs[i].decode_fails) continue;                 // a multi-scan file without its EOI: the host route reports what the reference does
            if (hdrs[i].ref_smooths) continue;                  // libjpeg smooths this file: the filter runs on the host, behind the host threads' scans (lp_prog_smooth)
            bool seq = false;
            for (const LpProgScanHost& sh : hdrs[i].scans) seq = seq || sh.s.sequential;
            if (u.prog_mode < 0 && seq) continue;               // auto: the wave decoder's files only
            u.prog_dev[(size_t)i] = 1;
            cand++;
        }
        // auto: a scan is a serial chain and one wave walks it ~3x slower than a host core, so the device only wins with enough
        // chains side by side (measured break-even: profiles/r06_progressive.md); below it the host threads keep the set
        if (u.prog_mode < 0 && std::max<size_t>(cand, u.prog_in_call) < lp_prog_device_min_images()) cand = 0;
        if (u.prog_mode == 0 || !cand) u.prog_dev.assign((size_t)n, 0);
        u.any_prog_dev = u.prog_mode != 0 && cand != 0;
        if (u.prog_mode < 0) u.prog_mode = 1;
    }
    u.pcoef_off.assign((size_t)n, 0);
    u.perr.assign((size_t)n, 0);
    u.pcoef_total = 0;
    u.host_tasks.clear();
    u.pieces.clear();
    std::vector<uint32_t> lev;
    std::map<uint64_t, std::vector<uint32_t>> phuff_by_hash;
    size_t raw_bytes = 0;
    for (int i = 0; i < n; i++) {
        LpJpeg j = hdrs[i].j;
        // Huffman set dedupe (most batches share one set)
        uint32_t hi = 0;
        for (; hi < u.huffs.size(); hi++)
            if (memcmp(&u.huffs[hi], &hdrs[i].huff, sizeof(LpHuffSet)) == 0) break;
        if (hi == u.huffs.size()) u.huffs.push_back(hdrs[i].huff);
        j.huff_idx = hi;
        j.raw_off = raw_bytes;
        j.raw_skip = 0;
        j.scan_path = hdrs[i].scan_path ? 1 : 0;
        if (hdrs[i].scan_path && !u.prog_dev[(size_t)i]) { // hybrid mode: host threads decode the scans into a pinned coefficient buffer
            j.raw_len = 0;
            u.pcoef_off[(size_t)i] = u.pcoef_total;
            for (int c = 0; c < j.ncomp; c++) u.pcoef_total += (size_t)j.bw[c] * j.bh[c] * 64;
            const size_t coef_elems = u.pcoef_total - u.pcoef_off[(size_t)i];
            lp_prog_levels(hdrs[i].scans, lev);
            if (hdrs[i].decode_fails) u.perr[(size_t)i] |= 8u; // a multi-scan file without its EOI: the reference's read_data fails (lp_jpeg_parse.h)
            for (size_t q = 0; q < hdrs[i].scans.size(); q++)
                u.host_tasks.push_back(LpProgHostTask{srcs[i].data, srcs[i].len, &hdrs[i].scans[q], nullptr, lev[q], &u.perr[(size_t)i], !hdrs[i].one_pass, coef_elems});
        } else if (hdrs[i].scan_path) { // every scan is a stream of its own
            j.raw_len = 0;
            lp_prog_levels(hdrs[i].scans, lev);
            for (const LpProgScanHost& sh : hdrs[i].scans) {
                LpUpload::ProgScanUp up;
                up.s = sh.s;
                up.level = lev[u.prog[(size_t)i].size()];
                // (eight bytes a step: byte by byte this hash was 4.7 ms of a 64-file chunk's layout -- 640 tables of 7 KB)
                uint64_t hash = 1469598103934665603ull;
                static_assert(sizeof(LpProgHuff) % 8 == 0, "hashed in 64-bit words");
                for (size_t q = 0; q < sizeof(LpProgHuff); q += 8) {
                    uint64_t w;
                    memcpy(&w, reinterpret_cast<const uint8_t*>(&sh.tables) + q, 8);
                    hash = (hash ^ w) * 1099511628211ull;
                    hash ^= hash >> 29;
                }
                uint32_t found = 0xffffffffu;
                for (uint32_t cand : phuff_by_hash[hash])
                    if (memcmp(&u.phuffs[cand], &sh.tables, sizeof(LpProgHuff)) == 0) { found = cand; break; }
                if (found == 0xffffffffu) {
                    found = (uint32_t)u.phuffs.size();
                    u.phuffs.push_back(sh.tables);
                    phuff_by_hash[hash].push_back(found);
                }
                up.s.huff = found;
                up.raw_off = raw_bytes;
                up.raw_len = (uint32_t)sh.ecs_len;
                u.pieces.push_back(LpUpload::Piece{raw_bytes, srcs[i].data + sh.ecs_off, sh.ecs_len, srcs[i].data, srcs[i].len, raw_bytes, 0, false, i, (int)u.prog[(size_t)i].size(), 0});
                raw_bytes = align_up(raw_bytes + up.raw_len + 32, 16);
                u.prog[(size_t)i].push_back(up);
            }
        } else {
            j.raw_len = (uint32_t)hdrs[i].ecs_len;
            u.pieces.push_back(LpUpload::Piece{raw_bytes, srcs[i].data + hdrs[i].ecs_off, hdrs[i].ecs_len, srcs[i].data, srcs[i].len, raw_bytes, 0, false, i, -1, 0});
            raw_bytes = align_up(raw_bytes + j.raw_len + 32, 16);
        }
        j.nchunks = (j.raw_len + 4095) / 4096;
        u.src[(size_t)i] = j;
    }
    u.raw_bytes = raw_bytes;
    u.stage_bytes = raw_bytes;
    u.direct_bytes = 0;
    u.copied_bytes = 0;
    for (const LpUpload::Piece& pc : u.pieces) u.copied_bytes += pc.len;
    return LP_OK;
}

// scan-path images of the set in hybrid mode: the scans' entropy decode on host threads, into the set's pinned coefficient buffer
static bool host_scan_decode(LpUpload& u, int n, const LpJpegHeader* hdrs)
{
    if (u.host_tasks.empty()) return true;
    if (!u.pcoef.ensure(u.pcoef_total * 2 + 64)) return false;
    memset(u.pcoef.p, 0, u.pcoef_total * 2);
    size_t t = 0;
    for (int i = 0; i < n; i++)
        if (hdrs[i].scan_path && !u.prog_dev[(size_t)i]) // (the images layout_set made tasks for: a set may hold device-decoded ones too)
            for (size_t q = 0; q < hdrs[i].scans.size(); q++) u.host_tasks[t++].coef = u.pcoef.as<int16_t>() + u.pcoef_off[(size_t)i];
    lp_prog_host_run(u.host_tasks, 0);
    for (int i = 0; i < n; i++)
        if (hdrs[i].scan_path && hdrs[i].ref_smooths && !u.prog_dev[(size_t)i]) lp_prog_smooth(hdrs[i], u.pcoef.as<int16_t>() + u.pcoef_off[(size_t)i]);
    return true;
}

int LpEngine::upload_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    LpUpload& u = up_[0];
    u_ = &u;
    u.staged_whole = false;
    layout_set(u, srcs, n, hdrs);
    if (!host_scan_decode(u, n, hdrs)) { err_ = "pinned allocation failed"; return LP_ERR_DEVICE; }
    const size_t raw_bytes = u.raw_bytes;
    const size_t kStage = 256u << 20; // pinned staging window
    if (!u.d_huffs.ensure(sizeof(LpHuffSet) * std::max<size_t>(1, u.huffs.size())) || !u.d_raw.ensure(raw_bytes + kRawPad) ||
        !u.d_phuffs.ensure(sizeof(LpProgHuff) * std::max<size_t>(1, u.phuffs.size())) ||
        !u.stage.ensure(std::min(raw_bytes, kStage) + (16u << 20) + 64)) {
        err_ = "device allocation failed";
        return LP_ERR_DEVICE;
    }
    uint8_t* stage = u.stage.as<uint8_t>();
    size_t win_begin = 0; // arena offset of the first byte staged in the current window
    auto flush = [&](size_t win_end) -> bool {
        if (win_end == win_begin) return true;
        if (!check(hipMemcpyAsync(u.d_raw.as<uint8_t>() + win_begin, stage, win_end - win_begin, hipMemcpyHostToDevice, stream_), "H2D raw")) return false;
        if (!check(hipStreamSynchronize(stream_), "upload sync")) return false;
        win_begin = win_end;
        return true;
    };
    for (const LpUpload::Piece& pc : u.pieces) {
        const size_t end = pc.arena_off + pc.len + 32;
        if (end - win_begin > u.stage.cap - 64) {
            if (!flush(pc.arena_off)) return LP_ERR_DEVICE;
            if (end - win_begin > u.stage.cap - 64 && !u.stage.ensure(end - win_begin + 64)) return LP_ERR_DEVICE;
            stage = u.stage.as<uint8_t>();
        }
        memcpy(stage + (pc.arena_off - win_begin), pc.src, pc.len);
        memset(stage + (pc.arena_off - win_begin) + pc.len, 0, 32);
    }
    if (!flush(raw_bytes)) return LP_ERR_DEVICE;
    if (!u.huffs.empty() &&
        !check(hipMemcpyAsync(u.d_huffs.p, u.huffs.data(), sizeof(LpHuffSet) * u.huffs.size(), hipMemcpyHostToDevice, stream_), "H2D huffs"))
        return LP_ERR_DEVICE;
    if (!u.phuffs.empty() &&
        !check(hipMemcpyAsync(u.d_phuffs.p, u.phuffs.data(), sizeof(LpProgHuff) * u.phuffs.size(), hipMemcpyHostToDevice, stream_), "H2D scan tables"))
        return LP_ERR_DEVICE;
    // the staging buffer is reused by the next upload: wait for the copies
    if (!check(hipStreamSynchronize(stream_), "upload sync")) return LP_ERR_DEVICE;
    return LP_OK;
}

int LpEngine::upload_layout(int slot, const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (slot < 0 || slot >= LP_UPLOAD_SLOTS) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    LpUpload& u = up_[slot];
    u.pins.release(); // the set this slot held before has been decoded: its copies are long done
    u.staged_whole = true;
    layout_set(u, srcs, n, hdrs);
    // How every segment travels (lp_hostmem.h): straight from the caller's pages when they are pinned -- already, or for the duration
    // of this set -- else through the slot's pinned buffer. The item's whole buffer is registered, so the scans of a multi-scan file
    // share one registration.
    const LpIngestMode mode = lp_ingest_mode();
    size_t so = 0, direct = 0, copied = 0;
    for (LpUpload::Piece& pc : u.pieces) {
        pc.direct = false;
        pc.pin_base = 0;
        if (mode != LP_INGEST_STAGED && pc.len) {
            ptrdiff_t delta = 0;
            if (lp_host_is_pinned(pc.src, pc.len, &delta, &pc.pin_base)) pc.direct = true;
            else if (mode == LP_INGEST_REGISTER && u.pins.add(pc.item, pc.item_len, &delta, &pc.pin_base)) pc.direct = true;
            pc.dev_delta = delta;
        }
    }
    // Second layout pass. Pinned sources that lie next to each other in HOST memory (files received back to back into one arena) keep
    // their spacing in the device arena, so that one copy-engine transfer fetches the whole run -- 32 transfers of ~4 MB per set cost
    // 7 % of the link rate against one of 135 MB (profiles/r03_a_ingest.md). A segment that follows another this way starts wherever
    // the host spacing puts it; the unstuff kernels take the misalignment as LpJpeg::raw_skip.
    static const bool by_kernel = getenv("LILLIPUT_HIP_DIRECT_COPY") && !strcmp(getenv("LILLIPUT_HIP_DIRECT_COPY"), "kernel");
    static const size_t max_gap = getenv("LILLIPUT_HIP_MERGE_GAP") ? (size_t)strtoull(getenv("LILLIPUT_HIP_MERGE_GAP"), nullptr, 10) : (size_t)256 << 10;
    size_t off = 0;
    for (size_t q = 0; q < u.pieces.size(); q++) {
        LpUpload::Piece& pc = u.pieces[q];
        const LpUpload::Piece* pv = q ? &u.pieces[q - 1] : nullptr;
        // (the scans of a progressive file lie a table segment apart: one transfer per file -- or per run of files -- instead of ten; a
        // 256-image set of 1024 x 1024 progressive files was 2 560 transfers, 51 ms of the stager's time)
        const bool follows = !by_kernel && pv && pc.direct && pv->direct && pc.pin_base && pc.pin_base == pv->pin_base &&
                             pc.src >= pv->src + pv->len && (size_t)(pc.src - (pv->src + pv->len)) <= max_gap;
        pc.arena_off = follows ? pv->arena_off + (size_t)(pc.src - pv->src) : align_up(off, 16);
        off = pc.arena_off + pc.len + 32;
        if (pc.scan < 0) {
            u.src[(size_t)pc.img].raw_off = pc.arena_off & ~(size_t)15;
            u.src[(size_t)pc.img].raw_skip = (uint32_t)(pc.arena_off & 15);
            u.src[(size_t)pc.img].nchunks = (u.src[(size_t)pc.img].raw_skip + u.src[(size_t)pc.img].raw_len + 4095) / 4096;
        } else
            u.prog[(size_t)pc.img][(size_t)pc.scan].raw_off = pc.arena_off;
        if (pc.direct) { direct += pc.len; continue; }
        pc.stage_off = so;
        so = align_up(so + pc.len + 32, 16);
        copied += pc.len;
    }
    u.raw_bytes = align_up(off, 16);
    u.stage_bytes = so;
    u.direct_bytes = direct;
    u.copied_bytes = copied;
    if (!u.ready && !check(hipEventCreateWithFlags(&u.ready, hipEventDisableTiming), "hipEventCreate")) return LP_ERR_DEVICE;
    if (!u.d_huffs.ensure(sizeof(LpHuffSet) * std::max<size_t>(1, u.huffs.size())) || !u.d_raw.ensure(u.raw_bytes + kRawPad) ||
        !u.d_phuffs.ensure(sizeof(LpProgHuff) * std::max<size_t>(1, u.phuffs.size())) ||
        !u.stage.ensure(align_up(u.stage_bytes + 64, 256) + sizeof(LpHuffSet) * u.huffs.size() + sizeof(LpProgHuff) * u.phuffs.size() + 64 +
                        (sizeof(LpGatherPiece) + 4) * (u.pieces.size() + 2))) {
        err_ = "device allocation failed";
        return LP_ERR_DEVICE;
    }
    if (!host_scan_decode(u, n, hdrs)) { err_ = "pinned allocation failed"; return LP_ERR_DEVICE; }
    return LP_OK;
}

void LpEngine::upload_release_pins()
{
    for (auto& u : up_) u.pins.release();
}

// Staging copy into the pinned slot with streaming stores: the destination is written once and next read by the DMA engine, so
// allocating its lines in the cache (a read-for-ownership of every line, which glibc's memcpy does for pieces of this size) is a
// third of the DRAM traffic of the copy for nothing -- per GPU the stager moves ~50 GB/s, and eight ranks share two sockets.
static void stream_copy(uint8_t* dst /* 16-byte aligned */, const uint8_t* src, size_t n)
{
    static const bool plain = getenv("LILLIPUT_HIP_STAGE_PLAIN_MEMCPY") != nullptr;
    if (plain || n < 4096 || ((uintptr_t)dst & 15)) { memcpy(dst, src, n); return; }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a); _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c); _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}

void LpEngine::upload_copy(int slot, size_t p0, size_t p1)
{
    LpUpload& u = up_[slot];
    uint8_t* stage = u.stage.as<uint8_t>();
    for (size_t q = p0; q < p1 && q < u.pieces.size(); q++) {
        const LpUpload::Piece& pc = u.pieces[q];
        if (pc.direct) continue;
        stream_copy(stage + pc.stage_off, pc.src, pc.len);
        memset(stage + pc.stage_off + pc.len, 0, 32);
    }
}

int LpEngine::upload_commit(int slot, hipStream_t on, const hipStream_t* extra, int n_extra)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    LpUpload& u = up_[slot];
    hipStream_t copy_stream_ = on ? on : this->copy_stream_;
    // the tables travel from the tail of the pinned buffer too: a copy from pageable memory would hold this thread until the copy
    // engine has worked through everything queued before it
    uint8_t* tab = u.stage.as<uint8_t>() + align_up(u.stage_bytes + 64, 256);
    const size_t hb = sizeof(LpHuffSet) * u.huffs.size(), pb = sizeof(LpProgHuff) * u.phuffs.size();
    if (hb) memcpy(tab, u.huffs.data(), hb);
    if (pb) memcpy(tab + hb, u.phuffs.data(), pb);
    // One copy per pinned source segment; the staged ones leave the slot's buffer in runs (a run = neighbours in the arena that are
    // neighbours in the buffer: a set without a pinned source is still ONE copy). The unstuff kernels mask by the segment's length,
    // so what follows a directly copied segment in the arena (the previous set's bytes) is never looked at.
    uint8_t* d_raw = u.d_raw.as<uint8_t>();
    const uint8_t* stage = u.stage.as<uint8_t>();
    n_extra = std::max(0, std::min(3, extra ? n_extra : 0));
    hipStream_t queues[4] = {copy_stream_, nullptr, nullptr, nullptr};
    for (int k = 0; k < n_extra; k++) queues[k + 1] = extra[k];
    bool used[4] = {true, false, false, false};
    size_t rr = 0;
    // How the segments that are read in place travel: LILLIPUT_HIP_DIRECT_COPY = sdma (default: copy-engine transfers, one per run of
    // neighbours) | kernel (k_gather_raw, one launch per set: fast alone, but its workgroups take CU time from the decode kernels --
    // 10.7 k images/s with 32 workgroups, 6.9 k with 256, against 10.8 k for per-segment transfers; profiles/r03_a_ingest.md)
    static const bool by_kernel = getenv("LILLIPUT_HIP_DIRECT_COPY") && !strcmp(getenv("LILLIPUT_HIP_DIRECT_COPY"), "kernel");
    static const uint32_t gather_wgs = getenv("LILLIPUT_HIP_GATHER_WGS") ? (uint32_t)std::max(1, atoi(getenv("LILLIPUT_HIP_GATHER_WGS"))) : 32u;
    if (by_kernel && u.direct_bytes) {
        // descriptors behind the tables in the slot's pinned buffer (mapped: the kernel reads them through its device alias)
        size_t nd = 0;
        for (const LpUpload::Piece& pc : u.pieces) nd += pc.direct && pc.len ? 1 : 0;
        uint8_t* base = tab + align_up(hb + pb, 16);
        LpGatherPiece* gp = reinterpret_cast<LpGatherPiece*>(base);
        uint32_t* tf = reinterpret_cast<uint32_t*>(base + sizeof(LpGatherPiece) * nd);
        uint32_t tiles = 0, q = 0;
        for (const LpUpload::Piece& pc : u.pieces) {
            if (!pc.direct || !pc.len) continue;
            gp[q] = LpGatherPiece{pc.src + pc.dev_delta, (uint64_t)pc.arena_off, (uint32_t)pc.len, 0u};
            tf[q++] = tiles;
            tiles += (uint32_t)((pc.len + 4095) / 4096);
        }
        tf[q] = tiles;
        const uint8_t* dev_base = static_cast<const uint8_t*>(u.stage.dev) + (base - u.stage.as<uint8_t>());
        lp_launch_gather_raw(copy_stream_, reinterpret_cast<const LpGatherPiece*>(dev_base), reinterpret_cast<const uint32_t*>(dev_base + sizeof(LpGatherPiece) * nd), (uint32_t)nd,
                             d_raw, std::min<uint32_t>(gather_wgs, std::max<uint32_t>(1u, tiles)));
    }
    for (size_t q = 0; q < u.pieces.size();) {
        const LpUpload::Piece& pc = u.pieces[q];
        if (pc.direct) {
            size_t r = q + 1;
            const uint8_t* end = pc.src + pc.len;
            while (!by_kernel && r < u.pieces.size() && u.pieces[r].direct && u.pieces[r].pin_base == pc.pin_base && u.pieces[r].src >= end &&
                   u.pieces[r].arena_off - pc.arena_off == (size_t)(u.pieces[r].src - pc.src)) { // a run of host neighbours: one transfer
                end = u.pieces[r].src + u.pieces[r].len;
                r++;
            }
            if (!by_kernel && end > pc.src) {
                const size_t w = rr++ % (size_t)(n_extra + 1);
                used[w] = true;
                if (!check(hipMemcpyAsync(d_raw + pc.arena_off, pc.src, (size_t)(end - pc.src), hipMemcpyHostToDevice, queues[w]), "H2D raw (caller's pages)")) return LP_ERR_DEVICE;
            }
            q = by_kernel ? q + 1 : r;
            continue;
        }
        size_t r = q + 1, end = pc.stage_off + pc.len + 32;
        while (r < u.pieces.size() && !u.pieces[r].direct && u.pieces[r].arena_off - pc.arena_off == u.pieces[r].stage_off - pc.stage_off) {
            end = u.pieces[r].stage_off + u.pieces[r].len + 32;
            r++;
        }
        if (!check(hipMemcpyAsync(d_raw + pc.arena_off, stage + pc.stage_off, end - pc.stage_off, hipMemcpyHostToDevice, copy_stream_), "H2D raw")) return LP_ERR_DEVICE;
        q = r;
    }
    if (hb && !check(hipMemcpyAsync(u.d_huffs.p, tab, hb, hipMemcpyHostToDevice, copy_stream_), "H2D huffs")) return LP_ERR_DEVICE;
    if (pb && !check(hipMemcpyAsync(u.d_phuffs.p, tab + hb, pb, hipMemcpyHostToDevice, copy_stream_), "H2D scan tables")) return LP_ERR_DEVICE;
    if (!check(hipEventRecord(u.ready, copy_stream_), "hipEventRecord")) return LP_ERR_DEVICE;
    u.ready_n = 0;
    for (int k = 1; k <= n_extra; k++) {
        if (!used[k]) continue;
        hipEvent_t& e = u.ready_x[u.ready_n];
        if (!e && !check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return LP_ERR_DEVICE;
        if (!check(hipEventRecord(e, queues[k]), "hipEventRecord")) return LP_ERR_DEVICE;
        u.ready_n++;
    }
    return LP_OK;
}

// Lay out the working arenas for images [first, first+n) of the uploaded set and run every decode stage.
int LpEngine::run_decode(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame, bool defer)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    if (n <= 0 || (size_t)(first + n) > u_->src.size()) return LP_ERR_INVALID_IMAGE;
    if (u_->staged_whole) { // the set's H2D copies (copy queues)
        if (!check(hipStreamWaitEvent(stream_, u_->ready, 0), "hipStreamWaitEvent")) return LP_ERR_DEVICE;
        for (int k = 0; k < u_->ready_n; k++)
            if (!check(hipStreamWaitEvent(stream_, u_->ready_x[k], 0), "hipStreamWaitEvent")) return LP_ERR_DEVICE;
    }
    h_imgs_.assign(u_->src.begin() + first, u_->src.begin() + first + n);
    size_t max_ecs = 0;
    for (auto& j : h_imgs_) max_ecs = std::max<size_t>(max_ecs, j.raw_len);
    S_ = S_cfg_ ? S_cfg_ : pick_S(max_ecs);
    vr_ = LP_VERIFY_ROUNDS;
    if (!S_cfg_) {
        // Small launches (the one-image ABI) are bound by the length of a lane's serial walk, not by throughput: cut the
        // subsequences shorter until the launch has enough lanes to occupy a good part of the device.
        // 128 k lanes = two waves per SIMD: the 32-image chunks of the ingest pipeline decode with 8 192-bit subsequences (measured
        // 11.05 k against 10.8 k images/s end to end), the 112-image chunks of a resident batch keep 16 384
        static const uint32_t want_lanes = getenv("LILLIPUT_HIP_LAT_LANES") ? (uint32_t)atoi(getenv("LILLIPUT_HIP_LAT_LANES")) : 131072u;
        // (floor: 512 / 1 024 / 2 048 / 4 096 / 8 192 bits -> 2.28 / 1.95 / 1.52 / 1.79 / 2.33 ms for one 4096 x 4096 Transform through the
        // one-image ABI, 1.45 / 1.50 / 1.23 / 1.62 / 1.63 ms for a 1300 x 1942 one: below 2 048 the verify rounds outgrow what the
        // shorter walks save)
        static const uint32_t min_S = getenv("LILLIPUT_HIP_MIN_S") ? (uint32_t)atoi(getenv("LILLIPUT_HIP_MIN_S")) : 2048u;
        uint64_t bits = 0;
        for (auto& j : h_imgs_) bits += (uint64_t)j.raw_len * 8;
        // A deferred decode (the chunks of the ingest pipeline) has a fixed number of verify rounds queued behind it and pays for a
        // subsequence that does not settle within them with a second pass: below 4 096 bits the self-synchronisation distance of
        // 4096 x 4096 sources (p99 ~800 symbols) spans several subsequences. Measured, 4 / 8 images per call: 5.3 / 6.4 ms with the
        // floor at 1 024, 2.5 / 4.3 ms at 4 096 (profiles/r03_c_final.md).
        // ... for the large sources that was measured on. A SMALL deferred launch (one 512 x 512 file through the batched entry point: 228
        // lanes of 4 096 bits = ONE workgroup, k_huff_spec 122 us + k_huff_verify 170 + k_huff_write 337 of a 0.9 ms call, profiles/
        // r06_one_image.md) is bound by the length of the lanes' walks like the one-image ABI is: below LILLIPUT_HIP_DEFER_SMALL_KBIT
        // (default 40 960 = 5 MB of entropy-coded data in the launch: one 4096 x 4096 q90 file) the floor is the one-image ABI's.
        static const uint64_t defer_small_bits = (uint64_t)(getenv("LILLIPUT_HIP_DEFER_SMALL_KBIT") ? atoi(getenv("LILLIPUT_HIP_DEFER_SMALL_KBIT")) : 40960) << 10;
        // ... and a small launch may go below the one-image floor when MORE verify rounds are queued behind it (vr_: an idle round costs
        // ~4 us, a host-driven second pass ~0.8 ms): LILLIPUT_HIP_SMALL_S / LILLIPUT_HIP_SMALL_ROUNDS, measured in profiles/r06_one_image.md
        static const uint32_t small_S = getenv("LILLIPUT_HIP_SMALL_S") ? (uint32_t)atoi(getenv("LILLIPUT_HIP_SMALL_S")) : 2048u;
        static const uint32_t small_rounds = getenv("LILLIPUT_HIP_SMALL_ROUNDS") ? (uint32_t)std::min(LP_VERIFY_MAX, std::max(1, atoi(getenv("LILLIPUT_HIP_SMALL_ROUNDS")))) : 6u;
        // (six since late round 6: with four, 40 % of 4096 x 4096 q90 sources -- 16 k lanes of 2 048 bits -- still moved an exit state in the fourth round and
        // paid the host-driven pass and the second decode: a lone call 2.7 ms instead of 1.1; two idle rounds cost 9 us. profiles/r06_one_image.md)
        const bool small_launch = bits <= defer_small_bits;
        static const uint32_t big_rounds = getenv("LILLIPUT_HIP_VERIFY_ROUNDS") ? (uint32_t)std::min(LP_VERIFY_MAX, std::max(1, atoi(getenv("LILLIPUT_HIP_VERIFY_ROUNDS")))) : (uint32_t)LP_VERIFY_ROUNDS;
        vr_ = small_launch ? small_rounds : big_rounds;
        const uint32_t floor_S = small_launch ? std::min(min_S, small_S) : defer ? std::max(min_S, 4096u) : min_S;
        // Small files (round 5): pick_S sizes by the largest file alone and hands 1 024 or 256 bits to sources of a few KB -- whose
        // entropy streams then need a verify round per subsequence the self-synchronisation distance spans (22 rounds for 128 x 128
        // sources, against the four that are queued). The floor holds for them too: a chunk of many small files gets its lanes from
        // the number of files, not from cutting each of them finer.
        S_ = std::max(S_, floor_S);
        while (S_ > floor_S && bits / S_ < want_lanes) S_ >>= 1;
    }
    if (S_ % 32 || S_ < 64 || S_ > 32768) { err_ = "bad subsequence size"; return LP_ERR_DEVICE; }
    sched_ = lp_make_sched(S_, C_cfg_ ? C_cfg_ : 256); // checkpoint schedule of the speculative pass (see LpCkSched)
    K_ = sched_.K;
    size_t clean_words = 0, coef_elems = 0, plane_bytes = LP_AREA_SLACK, pcoef_elems = 0; // the planes start behind a slack the area kernels' unclamped window loads may touch (lp_area_core.h)
    uint32_t max_pchunks = 0;
    h_pstreams_.clear();
    std::vector<std::pair<uint32_t, LpProgScan>> leveled; // (dependency level, scan)
    tot_sub_ = tot_chunks_ = tot_rst_ = 0;
    max_chunks_ = max_sub_ = max_bw_ = max_rows_ = max_w_ = max_h_ = max_mcus_ = 0;
    bool any_frame = false, any_generic = false, any_420 = false, any_baseline = false;
    for (size_t i = 0; i < h_imgs_.size(); i++) {
        LpJpeg& j = h_imgs_[i];
        any_baseline = any_baseline || !j.scan_path;
        j.chunk_off = tot_chunks_;
        tot_chunks_ += j.nchunks;
        j.clean_off = clean_words;                                  // multiple of 4 words: the bit reader loads 16 bytes at a time
        j.clean_cap_words = (j.raw_len / 4 + 64 + 3) / 4 * 4;       // + slack for the zeroed tail and reads past the end
        clean_words += j.clean_cap_words;
        j.sub_off = tot_sub_;
        // Per-image subsequence size. (Rounding it so that the count fills whole 256-lane workgroups was measured and
        // rejected: workgroups that all start and drain together are 3-5 % slower than the ragged tail they replace.)
        j.sub_bits = S_;
        j.sub_cap = (uint32_t)(((uint64_t)j.raw_len * 8 + S_ - 1) / S_) + 1;
        tot_sub_ += j.sub_cap;
        j.rst_off = tot_rst_;
        j.rst_cap = j.dri ? (j.mcus_x * j.mcus_y + j.dri - 1) / j.dri + 2 : 2;
        tot_rst_ += j.rst_cap;
        if (j.scan_path) {
            // Scans that touch the same coefficients of the same component must run in file order (a refinement needs what came
            // before it); all others are independent (lp_prog_levels). Host mode: ups is empty, the coefficients are ready.
            const std::vector<LpUpload::ProgScanUp>& ups = u_->prog[(size_t)first + i];
            j.coef_off = pcoef_elems;
            for (int c = 0; c < j.ncomp; c++) pcoef_elems += (size_t)j.bw[c] * j.bh[c] * 64;
            for (size_t a = 0; a < ups.size(); a++) {
                const LpProgScan& sa = ups[a].s;
                LpJpeg ps;
                memset(&ps, 0, sizeof(ps));
                ps.raw_off = ups[a].raw_off & ~(uint64_t)15; // (a scan fetched in one transfer with its neighbours starts wherever the file puts it)
                ps.raw_skip = (uint32_t)(ups[a].raw_off & 15);
                ps.raw_len = ups[a].raw_len;
                ps.nchunks = (ps.raw_skip + ps.raw_len + 4095) / 4096;
                ps.chunk_off = tot_chunks_;
                tot_chunks_ += ps.nchunks;
                ps.clean_off = clean_words;
                ps.clean_cap_words = (ps.raw_len / 4 + 64 + 3) / 4 * 4;
                clean_words += ps.clean_cap_words;
                ps.sub_bits = S_;
                ps.sub_cap = 1;
                ps.rst_off = tot_rst_;
                ps.rst_cap = sa.dri ? (sa.mcux * sa.mcuy + sa.dri - 1) / sa.dri + 2 : 2;
                tot_rst_ += ps.rst_cap;
                // the unstuff kernels check a stream's restart markers (one per interval, RST0..7 in turn) when it looks like an image of its
                // own: anything else is the host route's (jdmarker.c jpeg_resync_to_restart, lp_jbits.h)
                ps.dri = sa.dri; ps.mcus_x = sa.mcux; ps.mcus_y = sa.mcuy; ps.total_blocks = 1;
                max_pchunks = std::max(max_pchunks, ps.nchunks);
                LpProgScan sc = sa;
                sc.img = (uint32_t)i;
                sc.stream = (uint32_t)h_pstreams_.size();
                sc.coef_off = j.coef_off;
                h_pstreams_.push_back(ps);
                leveled.push_back(std::make_pair(ups[a].level, sc));
            }
        } else {
            j.coef_off = coef_elems; // a multiple of 8 blocks: a group of eight DC values is one aligned 16-byte store (DevSink)
            coef_elems += (((size_t)j.total_blocks + 7) & ~(size_t)7) * 64;
        }
        uint32_t rows = 0;
        for (int c = 0; c < j.ncomp; c++) {
            j.plane_off[c] = plane_bytes;
            plane_bytes = align_up(plane_bytes + (size_t)j.bw[c] * 8 * j.bh[c] * 8, 16);
            max_bw_ = std::max(max_bw_, j.bw[c]);
            rows += j.bh[c];
        }
        max_rows_ = std::max(max_rows_, rows);
        max_chunks_ = std::max(max_chunks_, j.nchunks);
        max_sub_ = std::max(max_sub_, j.sub_cap);
        max_mcus_ = std::max(max_mcus_, j.mcus_x * j.mcus_y);
        if (!want_frame || want_frame[i]) {
            any_frame = true;
            const bool f420 = j.ncomp == 3 && !j.generic_sampling && j.colorspace == 2 && j.hs[0] == 2 && j.vs[0] == 2 && j.width > 4;
            any_420 = any_420 || f420;
            any_generic = any_generic || !f420;
            max_w_ = std::max(max_w_, j.width);
            max_h_ = std::max(max_h_, j.height);
        }
    }
    std::stable_sort(leveled.begin(), leveled.end(), [](const std::pair<uint32_t, LpProgScan>& x, const std::pair<uint32_t, LpProgScan>& y) { return x.first < y.first; });
    h_pscans_.clear();
    h_plevel_first_.clear();
    for (size_t q = 0; q < leveled.size(); q++) {
        while (h_plevel_first_.size() <= leveled[q].first) h_plevel_first_.push_back((uint32_t)q);
        h_pscans_.push_back(leveled[q].second);
    }
    h_plevel_first_.push_back((uint32_t)leveled.size());
    // Pipelined launch (lp_kernels_prog.hip): every level of the range in one grid, each scan staying one MCU row behind the scans whose
    // coefficients it refines -- an image then costs its longest scan instead of the sum of its levels' longest. Needs the wave decoder for
    // every scan (no sequential ones) and at most LP_PROG_MAX_DEPS direct dependencies per scan (libjpeg's scripts have one or two).
    h_pdeps_.clear();
    {
        static const bool pipeline_on = !(getenv("LILLIPUT_HIP_PROG_PIPELINE") && atoi(getenv("LILLIPUT_HIP_PROG_PIPELINE")) == 0);
        bool can = pipeline_on && u_->prog_mode == 1 && h_plevel_first_.size() > 2 && !leveled.empty();
        for (const auto& lv : leveled) can = can && !lv.second.sequential;
        if (can) {
            auto conflicts = [](const LpProgScan& a, const LpProgScan& b, int* va, int* vb) {
                if (!(a.Ss <= b.Se && b.Ss <= a.Se)) return false;
                for (uint32_t x = 0; x < a.ns; x++)
                    for (uint32_t y = 0; y < b.ns; y++)
                        if (a.comp[x] == b.comp[y]) { *va = a.vs[x]; *vb = b.vs[y]; return true; }
                return false;
            };
            h_pdeps_.assign(leveled.size(), LpProgDep());
            std::vector<std::vector<uint32_t>> of_img(h_imgs_.size());
            for (uint32_t q = 0; q < leveled.size() && can; q++) {
                const LpProgScan& a = leveled[q].second;
                std::vector<uint32_t> dep; // earlier scans of the image this one conflicts with (all of a lower level)
                int va, vb;
                for (uint32_t b : of_img[a.img])
                    if (leveled[b].first < leveled[q].first && conflicts(a, leveled[b].second, &va, &vb)) dep.push_back(b);
                std::vector<uint32_t> direct; // minus those another dependency already stays behind
                for (uint32_t c : dep) {
                    bool implied = false;
                    for (uint32_t b : dep)
                        implied = implied || (b != c && leveled[c].first < leveled[b].first && conflicts(leveled[b].second, leveled[c].second, &va, &vb));
                    if (!implied) direct.push_back(c);
                }
                if (direct.size() > LP_PROG_MAX_DEPS) { can = false; break; }
                LpProgDep& d = h_pdeps_[q];
                memset(&d, 0, sizeof(d));
                d.ndep = (uint32_t)direct.size();
                for (size_t k = 0; k < direct.size(); k++) {
                    conflicts(a, leveled[direct[k]].second, &va, &vb);
                    d.scan[k] = direct[k];
                    d.vs_self[k] = (uint8_t)va;
                    d.vs_dep[k] = (uint8_t)vb;
                }
                of_img[a.img].push_back(q);
            }
        }
        if (!can) h_pdeps_.clear();
        else {
            // Launch order = ticket order: a wave takes the next scan in line and a chip holds ~1 000 of these waves at full speed (one per
            // SIMD: k_prog_wave's LDS footprint keeps it at that -- chains that share a SIMD share its one scalar issue slot per four
            // cycles), so the scans on an image's CRITICAL PATH go first: by the length of the longest dependency chain that starts at the
            // scan (entropy-coded bytes as the measure), longest first. A scan's chain is longer than that of any scan that waits for it, so
            // producers still come before their consumers.
            const size_t ns = leveled.size();
            std::vector<uint64_t> cp(ns, 0), below(ns, 0);
            for (size_t q = ns; q-- > 0;) {
                cp[q] = (uint64_t)h_pstreams_[leveled[q].second.stream].raw_len + 1u + below[q];
                for (uint32_t k = 0; k < h_pdeps_[q].ndep; k++) below[h_pdeps_[q].scan[k]] = std::max(below[h_pdeps_[q].scan[k]], cp[q]);
            }
            std::vector<uint32_t> order(ns), pos(ns);
            for (size_t q = 0; q < ns; q++) order[q] = (uint32_t)q;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cp[x] > cp[y]; }); // (ties keep level order)
            for (size_t q = 0; q < ns; q++) pos[order[q]] = (uint32_t)q;
            std::vector<LpProgDep> nd(ns);
            for (size_t q = 0; q < ns; q++) {
                h_pscans_[q] = leveled[order[q]].second;
                nd[q] = h_pdeps_[order[q]];
                for (uint32_t k = 0; k < nd[q].ndep; k++) nd[q].scan[k] = pos[nd[q].scan[k]];
            }
            h_pdeps_.swap(nd);
        }
    }
    const size_t nstreams = h_pstreams_.size();
    if (pcoef_elems && !d_pcoef_.ensure(pcoef_elems * 2 + 64)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    if (nstreams && !(d_pstreams_.ensure(sizeof(LpJpeg) * nstreams) && d_pstates_.ensure(sizeof(LpJpegState) * nstreams) &&
                      d_pscans_.ensure(sizeof(LpProgScan) * nstreams) && d_pcoef_.ensure(pcoef_elems * 2 + 64) &&
                      d_pdeps_.ensure(sizeof(LpProgDep) * nstreams + 64) && d_pprog_.ensure(4 * (nstreams + 1) + 64))) {
        err_ = "device allocation failed";
        return LP_ERR_DEVICE;
    }
    bool a = d_imgs_.ensure(sizeof(LpJpeg) * (size_t)n) && d_states_.ensure(sizeof(LpJpegState) * (size_t)n) && d_clean_.ensure(clean_words * 4 + 4096) &&
             d_rst_.ensure((size_t)tot_rst_ * 4 + 64) && d_chunk_.ensure((size_t)tot_chunks_ * 8 + 64) &&
             d_ckpt_.ensure((size_t)tot_sub_ * K_ * sizeof(LpCkptPk) + 64) && d_exit_.ensure((size_t)tot_sub_ * sizeof(LpSubState) + 64) &&
             d_spec_exit_.ensure((size_t)tot_sub_ * sizeof(LpSubState) + 64) && d_entry_.ensure((size_t)tot_sub_ * sizeof(LpSubState) + 64) &&
             d_tot_.ensure((size_t)tot_sub_ * sizeof(LpSubSum) + 64) && d_spec_tot_.ensure((size_t)tot_sub_ * sizeof(LpSubSum) + 64) &&
             d_prefix_.ensure((size_t)tot_sub_ * sizeof(LpSubSum) + 64) && d_changed_.ensure(64) && h_dstate_.ensure(64 + sizeof(LpJpegState) * (size_t)n + 64) && d_coef_.ensure(coef_elems + 64) && d_wide_.ensure(coef_elems * 2 + 64) && d_wide_id_.ensure(coef_elems / 16 + 64) && d_dc_.ensure(coef_elems / 32 + 64) && d_dcpart_.ensure((size_t)n * lp_dc_scan_max_ranges() * 16 + 64) &&
             d_planes_.ensure(plane_bytes + LP_AREA_SLACK) && d_frames_desc_.ensure(sizeof(LpFrame) * (size_t)n) &&
             h_small_.ensure(std::max<size_t>(4096, sizeof(LpJpegState) * (size_t)n));
    if (!a) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    const LpJpeg* di = d_imgs_.as<LpJpeg>();
    LpJpegState* ds = d_states_.as<LpJpegState>();
    {   // image descriptors up, states and the verify rounds' counters cleared: one launch
        SmallBatch sb;
        if (!small_copy(sb, d_imgs_.p, h_imgs_.data(), sizeof(LpJpeg) * (size_t)n) || !small_zero(sb, ds, sizeof(LpJpegState) * (size_t)n) ||
            !small_zero(sb, d_changed_.p, 4 * (vr_ + 1)) || !small_flush(sb))
            return LP_ERR_DEVICE;
    }
    // LILLIPUT_HIP_DEBUG_SYNC=1: synchronise after every stage and name it on stderr (a faulting kernel aborts the process at
    // the next synchronisation, so the last line printed is the stage before the culprit)
    static const bool dbg = getenv("LILLIPUT_HIP_DEBUG_SYNC") != nullptr;
    auto stage = [&](const char* name) {
        if (!dbg) return;
        fprintf(stderr, "[lilliput_hip] stage %s ...\n", name);
        (void)hipStreamSynchronize(stream_);
        fprintf(stderr, "[lilliput_hip] stage %s done (%s)\n", name, hipGetErrorString(hipGetLastError()));
    };
    mark(0);
    lp_launch_unstuff(stream_, di, (uint32_t)n, max_chunks_, u_->d_raw.as<uint8_t>(), d_chunk_.as<uint2>(), ds, d_clean_.as<uint32_t>(),
                      d_rst_.as<uint32_t>());
    stage("unstuff");
    mark(1);
    LpHuffArgs ha;
    ha.imgs = di; ha.states = ds; ha.huffs = u_->d_huffs.as<LpHuffSet>();
    ha.nimg = (uint32_t)n; ha.max_sub = max_sub_; ha.tot_sub = tot_sub_;
    ha.clean = d_clean_.as<uint32_t>(); ha.rst = d_rst_.as<uint32_t>();
    ha.ckpts = d_ckpt_.as<LpCkptPk>();
    ha.spec_exit = d_spec_exit_.as<LpSubState>(); ha.spec_total = d_spec_tot_.as<LpSubSum>();
    ha.cur_exit = d_exit_.as<LpSubState>(); ha.cur_total = d_tot_.as<LpSubSum>();
    ha.entry_used = d_entry_.as<LpSubState>(); ha.prefix = d_prefix_.as<LpSubSum>();
    ha.changed = d_changed_.as<uint32_t>(); ha.coef8 = d_coef_.as<int8_t>(); ha.wide = d_wide_.as<int16_t>(); ha.wide_id = d_wide_id_.as<uint32_t>(); ha.dc16 = d_dc_.as<int16_t>();
    ha.sched = sched_;
    lp_launch_huff_spec(stream_, ha);
    stage("huff_spec");
    mark(8);
    // Verify rounds back to back, no host round trip in between: round r counts the exit states it moved into changed[r] and a
    // round that follows an idle one returns at once. The last counter is looked at when the decode is collected (finish_decode);
    // streams that need more than LP_VERIFY_ROUNDS rounds (tiny subsequences, hostile data) continue there under host control.
    // (the counters were cleared with the descriptor upload above)
    for (uint32_t r = 0; r < vr_; r++) lp_launch_huff_verify(stream_, ha, r);
    mark(9);
    stage("huff_verify");
    lp_launch_sub_scan(stream_, ha);
    stage("sub_scan");
    mark(10);
    lp_launch_huff_write(stream_, ha);
    stage("huff_write");
    lp_launch_dc_scan(stream_, di, (uint32_t)n, max_mcus_, d_dc_.as<int16_t>(), d_dcpart_.p);
    stage("dc_scan");
    if (nstreams && !check(hipMemsetAsync(d_pcoef_.p, 0, pcoef_elems * 2, stream_), "memset coefficients")) return LP_ERR_DEVICE;
    if (pcoef_elems) { // hybrid mode: the coefficients were decoded at upload time
        for (int i = 0; i < n; i++) {
            const LpJpeg& j = h_imgs_[(size_t)i];
            if (!j.scan_path || u_->prog_dev[(size_t)first + i]) continue;
            size_t ne = 0;
            for (int c = 0; c < j.ncomp; c++) ne += (size_t)j.bw[c] * j.bh[c] * 64;
            if (!check(hipMemcpyAsync(d_pcoef_.as<int16_t>() + j.coef_off, u_->pcoef.as<int16_t>() + u_->pcoef_off[(size_t)first + i], ne * 2, hipMemcpyHostToDevice, stream_), "H2D coefficients"))
                return LP_ERR_DEVICE;
        }
    }
    if (nstreams) { // progressive images: unstuff every scan, then the scans level by level into the zeroed int16 arena
        if (!check(hipMemcpyAsync(d_pstreams_.p, h_pstreams_.data(), sizeof(LpJpeg) * nstreams, hipMemcpyHostToDevice, stream_), "H2D scan streams") ||
            !check(hipMemcpyAsync(d_pscans_.p, h_pscans_.data(), sizeof(LpProgScan) * nstreams, hipMemcpyHostToDevice, stream_), "H2D scans") ||
            !check(hipMemsetAsync(d_pstates_.p, 0, sizeof(LpJpegState) * nstreams, stream_), "memset scan states"))
            return LP_ERR_DEVICE;
        lp_launch_unstuff(stream_, d_pstreams_.as<LpJpeg>(), (uint32_t)nstreams, max_pchunks, u_->d_raw.as<uint8_t>(), d_chunk_.as<uint2>(),
                          d_pstates_.as<LpJpegState>(), d_clean_.as<uint32_t>(), d_rst_.as<uint32_t>());
        stage("prog_unstuff");
        if (!h_pdeps_.empty()) { // every level in one grid
            if (!check(hipMemcpyAsync(d_pdeps_.p, h_pdeps_.data(), sizeof(LpProgDep) * nstreams, hipMemcpyHostToDevice, stream_), "H2D scan dependencies") ||
                !check(hipMemsetAsync(d_pprog_.p, 0, 4 * (nstreams + 1), stream_), "memset scan progress"))
                return LP_ERR_DEVICE;
            lp_launch_prog_wave(stream_, d_pscans_.as<LpProgScan>(), 0, (uint32_t)nstreams, d_pstreams_.as<LpJpeg>(), d_pstates_.as<LpJpegState>(), u_->d_phuffs.as<LpProgHuff>(),
                                d_clean_.as<uint32_t>(), d_rst_.as<uint32_t>(), d_pcoef_.as<int16_t>(), d_pdeps_.as<LpProgDep>(), d_pprog_.as<uint32_t>());
        } else
        for (size_t l = 0; l + 1 < h_plevel_first_.size(); l++) {
            const uint32_t f = h_plevel_first_[l], cnt = h_plevel_first_[l + 1] - f;
            bool any_seq = false, any_prog = false;
            for (uint32_t q = f; q < f + cnt; q++) (h_pscans_[q].sequential ? any_seq : any_prog) = true;
            const bool waves = u_->prog_mode == 1;
            // one WAVE per progressive scan (lp_kernels_prog.hip); sequential scans, and everything in lanes mode, one LANE per scan:
            // few lanes: one per wave (a lane alone on its SIMD runs fastest); many: pack them so that the grid stays a few waves per SIMD
            if (waves && any_prog)
                lp_launch_prog_wave(stream_, d_pscans_.as<LpProgScan>(), f, cnt, d_pstreams_.as<LpJpeg>(), d_pstates_.as<LpJpegState>(), u_->d_phuffs.as<LpProgHuff>(),
                                    d_clean_.as<uint32_t>(), d_rst_.as<uint32_t>(), d_pcoef_.as<int16_t>(), nullptr, nullptr);
            if (!waves || any_seq) {
                const uint32_t lpw = std::min<uint32_t>(64u, std::max<uint32_t>(1u, (cnt + 4095u) / 4096u));
                lp_launch_prog_scans(stream_, d_pscans_.as<LpProgScan>(), f, cnt, lpw, waves, d_pstreams_.as<LpJpeg>(), d_pstates_.as<LpJpegState>(),
                                     u_->d_phuffs.as<LpProgHuff>(), d_clean_.as<uint32_t>(), d_rst_.as<uint32_t>(), d_pcoef_.as<int16_t>());
            }
        }
        stage("prog_scans");
    }
    mark(2);
    lp_launch_idct(stream_, di, ds, (uint32_t)n, max_bw_, max_rows_, d_coef_.as<int8_t>(), d_wide_.as<int16_t>(), d_wide_id_.as<uint32_t>(), d_dc_.as<int16_t>(), d_planes_.as<uint8_t>(),
                   (any_baseline ? 1u : 0u) | (pcoef_elems ? 2u : 0u), d_pcoef_.as<int16_t>());
    stage("idct");
    mark(3);
    // frames
    for (int i = 0; i < n; i++) {
        const LpJpeg& j = h_imgs_[(size_t)i];
        LpFrame& f = frames[i];
        f.w = j.width; f.h = j.height; f.cn = j.ncomp == 1 ? 1 : 3; f.stride = f.w * f.cn;
        if (want_frame && !want_frame[i]) { f.off = 0; continue; }
        if (!f.off) {
            uint8_t* p = heap_alloc((size_t)f.stride * f.h);
            if (!p) { err_ = "frame heap exhausted"; return LP_ERR_DEVICE; }
            f.off = (uint64_t)(uintptr_t)p;
        }
    }
    if (any_frame) {
        if (!h2d_small(d_frames_desc_.p, frames, sizeof(LpFrame) * (size_t)n)) return LP_ERR_DEVICE;
        lp_launch_ycc_to_frame(stream_, di, (uint32_t)n, max_w_, max_h_, any_generic, any_420, d_planes_.as<uint8_t>(), d_frames_desc_.as<LpFrame>(), nullptr);
    }
    mark(4);
    pend_ = Pending{true, first, n, nstreams, pcoef_elems, any_baseline, any_frame, any_generic, any_420, frames, ha, vr_};
    {
        SmallBatch sb;
        small_d2h(sb, h_dstate_, h_dstate_.as<uint8_t>() + 64, ds, sizeof(LpJpegState) * (size_t)n);
        small_d2h(sb, h_dstate_, h_dstate_.p, d_changed_.p, 4 * (vr_ + 1));
        small_flush(sb);
    }
    return defer ? LP_OK : finish_decode(status);
}

// Second half of a decode: wait for the stream, finish the verification under host control if the enqueued rounds did not settle
// it (then everything behind the verify stage is enqueued again), and turn the per-image device states into statuses.
// Returns LP_RETRY when a deferred decode had to redo its tail: what the caller enqueued behind it read unfinished planes.
// launches whose queued verify rounds did not settle the exit states (host-driven rounds + the stages behind them once more + LP_RETRY): process-wide
static std::atomic<uint64_t> g_decode_redone{0};
extern "C" uint64_t lilliput_hip_decode_redone_count() { return g_decode_redone.load(std::memory_order_relaxed); }

int LpEngine::finish_decode(int* status)
{
    if (!pend_.active) return LP_ERR_DEVICE;
    pend_.active = false;
    const int n = pend_.n, first = pend_.first;
    const size_t nstreams = pend_.nstreams;
    if (!check(hipStreamSynchronize(stream_), "decode sync")) return LP_ERR_DEVICE;
    if (!check(hipGetLastError(), "decode kernels")) return LP_ERR_DEVICE;
    const uint32_t* h_changed = h_dstate_.as<uint32_t>();
    uint32_t rounds = 1;
    const uint32_t vr = pend_.vr; // verify rounds that were queued behind the speculative pass
    for (uint32_t r = 0; r + 1 < vr; r++) rounds += h_changed[r] ? 1u : 0u;
    bool redone = false;
    if (h_changed[vr - 1] != 0) {
        const LpHuffArgs& ha = pend_.ha;
        for (;;) { // round index vr: its gate reads the previous counter, which is non-zero here
            if (!check(hipMemsetAsync(d_changed_.as<uint32_t>() + vr, 0, 4, stream_), "memset changed")) return LP_ERR_DEVICE;
            lp_launch_huff_verify(stream_, ha, vr);
            if (!check(hipMemcpyAsync(h_small_.p, d_changed_.as<uint32_t>() + vr, 4, hipMemcpyDeviceToHost, stream_), "D2H changed")) return LP_ERR_DEVICE;
            if (!check(hipStreamSynchronize(stream_), "verify sync")) return LP_ERR_DEVICE;
            rounds++;
            if (*h_small_.as<uint32_t>() == 0) break;
            // every round makes at least one more subsequence final, so max_sub_ rounds always suffice; real streams need 1-3
            if (rounds > max_sub_ + 1 || rounds >= 100000) { err_ = "entropy decode did not converge"; return LP_ERR_DECODE_FAILED; }
        }
        // the stages behind the verification ran on unsettled exit states: once more
        lp_launch_reset_tail_state(stream_, d_states_.as<LpJpegState>(), (uint32_t)n);
        lp_launch_sub_scan(stream_, ha);
        lp_launch_huff_write(stream_, ha);
        lp_launch_dc_scan(stream_, d_imgs_.as<LpJpeg>(), (uint32_t)n, max_mcus_, d_dc_.as<int16_t>(), d_dcpart_.p);
        lp_launch_idct(stream_, d_imgs_.as<LpJpeg>(), d_states_.as<LpJpegState>(), (uint32_t)n, max_bw_, max_rows_, d_coef_.as<int8_t>(), d_wide_.as<int16_t>(),
                       d_wide_id_.as<uint32_t>(), d_dc_.as<int16_t>(), d_planes_.as<uint8_t>(), (pend_.any_baseline ? 1u : 0u) | (pend_.pcoef_elems ? 2u : 0u), d_pcoef_.as<int16_t>());
        if (pend_.any_frame)
            lp_launch_ycc_to_frame(stream_, d_imgs_.as<LpJpeg>(), (uint32_t)n, max_w_, max_h_, pend_.any_generic, pend_.any_420, d_planes_.as<uint8_t>(), d_frames_desc_.as<LpFrame>(), nullptr);
        d2h_small(h_dstate_, h_dstate_.as<uint8_t>() + 64, d_states_.p, sizeof(LpJpegState) * (size_t)n);
        if (!check(hipStreamSynchronize(stream_), "decode sync")) return LP_ERR_DEVICE;
        if (!check(hipGetLastError(), "decode kernels")) return LP_ERR_DEVICE;
        redone = true;
    }
    tm_.verify_rounds = rounds;
    h_states_.resize((size_t)n);
    memcpy(h_states_.data(), h_dstate_.as<uint8_t>() + 64, sizeof(LpJpegState) * (size_t)n);
    if (nstreams) { // a scan that failed to unstuff fails its image
        h_pstates_.resize(nstreams);
        if (!check(hipMemcpy(h_pstates_.data(), d_pstates_.p, sizeof(LpJpegState) * nstreams, hipMemcpyDeviceToHost), "D2H scan states")) return LP_ERR_DEVICE;
        // a scan the device gave up on (a marker inside it, restart markers out of turn, a code that matches nothing, data that runs out ...):
        // the image is the host route's, which does with such a stream what libjpeg does (LpEngine::scan_gave_up)
        for (const LpProgScan& sc : h_pscans_)
            if (h_pstates_[sc.stream].error) h_states_[sc.img].error |= 64u;
        uint64_t dev = 0, gave_up = 0;
        for (int i = 0; i < n; i++)
            if (h_imgs_[(size_t)i].scan_path && u_->prog_dev[(size_t)first + i]) { dev++; gave_up += (h_states_[(size_t)i].error & 64u) ? 1u : 0u; }
        lp_prog_count(dev, gave_up, h_pscans_.size());
    }
    for (int i = 0; i < n; i++)
        if (h_imgs_[(size_t)i].scan_path && !u_->prog_dev[(size_t)first + i]) h_states_[(size_t)i].error |= u_->perr[(size_t)first + i];
    int rc = LP_OK;
    for (int i = 0; i < n; i++) {
        status[i] = h_states_[(size_t)i].error ? LP_ERR_DECODE_FAILED : LP_OK;
        if (status[i]) rc = status[i];
    }
    if (timing_) {
        (void)hipEventElapsedTime(&tm_.unstuff_ms, ev_[0], ev_[1]);
        (void)hipEventElapsedTime(&tm_.huff_ms, ev_[1], ev_[2]);
        (void)hipEventElapsedTime(&tm_.idct_ms, ev_[2], ev_[3]);
        (void)hipEventElapsedTime(&tm_.color_ms, ev_[3], ev_[4]);
        (void)hipEventElapsedTime(&tm_.huff_spec_ms, ev_[1], ev_[8]);
        (void)hipEventElapsedTime(&tm_.huff_verify_ms, ev_[8], ev_[9]);
        (void)hipEventElapsedTime(&tm_.huff_scan_ms, ev_[9], ev_[10]);
        (void)hipEventElapsedTime(&tm_.huff_write_ms, ev_[10], ev_[2]);
    }
    if (redone) g_decode_redone.fetch_add(1, std::memory_order_relaxed);
    return redone ? LP_RETRY : rc;
}

int LpEngine::decode_uploaded(int first, int n, LpFrame* frames, int* status, const uint8_t* want_frame)
{
    const int rc = run_decode(first, n, frames, status, want_frame, false);
    return rc == LP_RETRY ? LP_OK : rc; // nothing was enqueued behind the decode: the redone tail is simply the result
}

int LpEngine::decode_begin(int first, int n, LpFrame* frames, const uint8_t* want_frame)
{
    return run_decode(first, n, frames, nullptr, want_frame, true);
}

int LpEngine::copy_coefs(int i, int comp, int16_t* dst, size_t cap_elems)
{
    // test access: the arena holds the blocks in decode order; return component `comp` as [by][bx][64]
    const LpJpeg& j = h_imgs_[(size_t)i];
    const size_t ne = (size_t)j.bw[comp] * j.bh[comp] * 64;
    if (ne > cap_elems) return LP_ERR_BUF_TOO_SMALL;
    if (j.scan_path) { // already [by][bx], every block in zigzag order
        size_t base = 0;
        for (int c = 0; c < comp; c++) base += (size_t)j.bw[c] * j.bh[c] * 64;
        std::vector<int16_t> t(ne);
        if (!check(hipMemcpyAsync(t.data(), d_pcoef_.as<int16_t>() + j.coef_off + base, ne * 2, hipMemcpyDeviceToHost, stream_), "D2H coefs")) return LP_ERR_DEVICE;
        const int rc = sync();
        if (rc) return rc;
        static const uint8_t zz[80] = LP_ZIGZAG_INIT;
        for (size_t q = 0; q < ne; q++) dst[(q & ~(size_t)63) | zz[q & 63]] = t[q];
        return LP_OK;
    }
    const size_t nb = j.total_blocks;
    std::vector<int8_t> c8(nb * 64);
    std::vector<uint32_t> wid(nb);
    std::vector<int16_t> dcs(nb);
    if (!check(hipMemcpyAsync(dcs.data(), d_dc_.as<int16_t>() + j.coef_off / 64, nb * 2, hipMemcpyDeviceToHost, stream_), "D2H dc")) return LP_ERR_DEVICE;
    if (!check(hipMemcpyAsync(c8.data(), d_coef_.as<int8_t>() + j.coef_off, c8.size(), hipMemcpyDeviceToHost, stream_), "D2H coefs")) return LP_ERR_DEVICE;
    if (!check(hipMemcpyAsync(wid.data(), d_wide_id_.as<uint32_t>() + j.coef_off / 64, nb * 4, hipMemcpyDeviceToHost, stream_), "D2H wide ids")) return LP_ERR_DEVICE;
    int rc = sync();
    if (rc) return rc;
    const uint32_t n_wide = std::min<uint32_t>(h_states_[(size_t)i].n_wide, (uint32_t)nb); // never more slots than blocks
    std::vector<int16_t> wide((size_t)n_wide * 64);
    if (n_wide) {
        if (!check(hipMemcpyAsync(wide.data(), d_wide_.as<int16_t>() + j.coef_off, wide.size() * 2, hipMemcpyDeviceToHost, stream_), "D2H wide")) return LP_ERR_DEVICE;
        if ((rc = sync())) return rc;
    }
    std::vector<int16_t> all(nb * 64);
    for (size_t q = 0; q < nb; q++)
        for (int e = 0; e < 64; e++) {
            const int8_t v = c8[q * 64 + e];
            const int nat = ((e & 7) << 3) | (e >> 3); // blocks are stored transposed
            all[q * 64 + nat] = v == -128 && wid[q] < n_wide ? wide[(size_t)wid[q] * 64 + e] : v; // -128 = escape to the wide slot
        }
    for (size_t q = 0; q < nb; q++) all[q * 64] = dcs[q];
    const uint32_t hs = j.hs[comp], vs = j.vs[comp];
    for (uint32_t by = 0; by < j.bh[comp]; by++)
        for (uint32_t bx = 0; bx < j.bw[comp]; bx++) {
            const size_t blk = ((size_t)(by / vs) * j.mcus_x + bx / hs) * j.bpm + j.blk_first[comp] + (by % vs) * hs + (bx % hs);
            memcpy(dst + ((size_t)by * j.bw[comp] + bx) * 64, all.data() + blk * 64, 128);
        }
    return LP_OK;
}

int LpEngine::copy_plane(int i, int comp, uint8_t* dst, size_t cap)
{
    const LpJpeg& j = h_imgs_[(size_t)i];
    size_t nb = (size_t)j.bw[comp] * 8 * j.bh[comp] * 8;
    if (nb > cap) return LP_ERR_BUF_TOO_SMALL;
    if (!check(hipMemcpyAsync(dst, d_planes_.as<uint8_t>() + j.plane_off[comp], nb, hipMemcpyDeviceToHost, stream_), "D2H plane")) return LP_ERR_DEVICE;
    return sync();
}

int LpEngine::decode_jpegs(const LpJpegSrc* srcs, int n, const LpJpegHeader* hdrs, LpFrame* frames, int* status)
{
    int rc = upload_jpegs(srcs, n, hdrs);
    if (rc) return rc;
    {
        double db = 0;
        for (int i = 0; i < n; i++) db += (double)srcs[i].len + 4.5 * hdrs[i].j.width * hdrs[i].j.height * (hdrs[i].j.ncomp == 1 ? 1.0 / 1.5 : 1.0);
        LpStageProbe probe_(stream_, "JPEG decode (k_unstuff .. k_idct, k_ycc_to_frame)", db);
        rc = run_decode(0, n, frames, status, nullptr, false);
    }
    if (rc == LP_RETRY) rc = LP_OK;
    if (rc == LP_ERR_DEVICE) return rc; // anything else is the status of an image that failed
    // A baseline stream that came up short of blocks (truncated upload, damaged data) is not an error to libjpeg: it warns, feeds zero
    // bits to the MCU at hand and leaves the following MCUs untouched (grey). The serial scan decoder implements exactly that rule
    // (lp_prog_core.h, pinned against libjpeg on damaged files), so such an image is decoded once more through it.
    for (int i = 0; i < n; i++) {
        if (status[i] != LP_ERR_DECODE_FAILED || hdrs[i].scan_path) continue;
        LpJpegHeader again;
        if (lp_jpeg_parse_opts(srcs[i].data, srcs[i].len, &again, true) != LP_PARSE_OK || !again.scan_path) continue;
        int st = 0;
        if (upload_jpegs(&srcs[i], 1, &again)) continue;
        const int r2 = run_decode(0, 1, &frames[i], &st, nullptr, false);
        if (r2 != LP_ERR_DEVICE) status[i] = st;
    }
    // a progressive image the device's scan decoders gave up on: once more with its scans on host threads
    std::vector<int> gave_up;
    for (int i = 0; i < n; i++)
        if (status[i] == LP_ERR_DECODE_FAILED && hdrs[i].scan_path && scan_gave_up(i)) gave_up.push_back(i);
    for (int i : gave_up) {
        int st = 0;
        for (int s = 0; s < LP_UPLOAD_SLOTS; s++) up_[s].force_host_scans = true;
        const int ru = upload_jpegs(&srcs[i], 1, &hdrs[i]);
        const int r2 = ru ? LP_ERR_DEVICE : run_decode(0, 1, &frames[i], &st, nullptr, false);
        for (int s = 0; s < LP_UPLOAD_SLOTS; s++) up_[s].force_host_scans = false;
        if (r2 != LP_ERR_DEVICE) status[i] = st;
    }
    rc = LP_OK;
    for (int i = 0; i < n; i++)
        if (status[i]) rc = status[i];
    return rc;
}

// ------------------------------------------------------------------------------------------------
// orientation / resize / composite
int LpEngine::orient(const LpOrientOp* ops, int n)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    if (!d_ops_.ensure(std::max(sizeof(LpOrientOp), sizeof(LpResizeOp)) * (size_t)n + 64)) return LP_ERR_DEVICE;
    uint32_t mw = 0, mh = 0;
    double orient_bytes = 0;
    for (int i = 0; i < n; i++) { mw = std::max(mw, std::max(ops[i].src.w, ops[i].src.h)); mh = mw; orient_bytes += 2.0 * ops[i].src.w * ops[i].src.h * ops[i].src.cn; }
    if (!h2d_any(d_ops_.p, ops, sizeof(LpOrientOp) * (size_t)n, true)) return LP_ERR_DEVICE;
    // the op array is pageable host memory: make sure the copy has been consumed before the caller frees it
    { LpStageProbe probe_(stream_, "k_orient", (double)(orient_bytes)); lp_launch_orient(stream_, d_ops_.as<LpOrientOp>(), (uint32_t)n, mw, mh, nullptr, nullptr); }
    if (!check(hipStreamSynchronize(stream_), "orient sync")) return LP_ERR_DEVICE;
    return check(hipGetLastError(), "orient kernel") ? LP_OK : LP_ERR_DEVICE;
}

int lp_resize_mode(int sw, int sh, int dw, int dh, int* iscale_x, int* iscale_y)
{
    // cv::resize (modules/imgproc/src/resize.cpp): dsize == ssize -> copyTo; INTER_AREA with both scales >= 1 ->
    // integer scales: resizeAreaFast_, else resizeArea_; otherwise bilinear with area-style coefficients.
    *iscale_x = *iscale_y = 1;
    if (sw == dw && sh == dh) return 0;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (scale_x >= 1 && scale_y >= 1) {
        int ix = (int)lrint(scale_x), iy = (int)lrint(scale_y);
        *iscale_x = ix; *iscale_y = iy;
        bool fast = fabs(scale_x - ix) < DBL_EPSILON && fabs(scale_y - iy) < DBL_EPSILON;
        return fast ? 1 : 2;
    }
    return 3;
}

int lp_area_tab(int ssize, int dsize, std::vector<LpTap>& taps, std::vector<uint32_t>& ranges)
{
    // resize.cpp computeResizeAreaTab; ranges[dx]..ranges[dx+1] index the taps of destination dx
    double scale = 1. / ((double)dsize / ssize);
    const uint32_t base = (uint32_t)taps.size();
    for (int dx = 0; dx < dsize; dx++) {
        ranges.push_back((uint32_t)taps.size() - base);
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) taps.push_back(LpTap{(uint32_t)(sx1 - 1), (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; sx++) taps.push_back(LpTap{(uint32_t)sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) taps.push_back(LpTap{(uint32_t)sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
    ranges.push_back((uint32_t)taps.size() - base);
    return (int)(taps.size() - base);
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }

int LpEngine::resize(const LpResizeReq* reqs, int n, LpFrame* dsts, int* status)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    std::vector<LpResizeOp> ops((size_t)n);
    std::vector<LpTap> taps;
    std::vector<uint32_t> ranges;
    std::map<std::pair<int, int>, std::pair<uint32_t, uint32_t>> cache; // (ssize,dsize) -> (tap_off, range_off)
    uint32_t modes = 0, mdw = 0, mdh = 0, area3_mask = 0;
    for (int i = 0; i < n; i++) {
        const LpResizeReq& r = reqs[i];
        LpResizeOp& op = ops[(size_t)i];
        memset(&op, 0, sizeof(op));
        status[i] = LP_OK;
        if (r.crop_w == 0 || r.crop_h == 0 || r.crop_x + r.crop_w > r.src.w || r.crop_y + r.crop_h > r.src.h || r.dst_w == 0 || r.dst_h == 0) {
            status[i] = LP_ERR_INVALID_IMAGE;
            op.mode = 99;
            continue;
        }
        op.src = r.src;
        op.src.off += (uint64_t)r.crop_y * r.src.stride + (uint64_t)r.crop_x * r.src.cn;
        op.src.w = r.crop_w; op.src.h = r.crop_h;
        op.dst = dsts[i];
        op.dst.w = r.dst_w; op.dst.h = r.dst_h; op.dst.cn = r.src.cn;
        if (!op.dst.stride) op.dst.stride = r.dst_w * r.src.cn;
        dsts[i] = op.dst;
        int ix, iy;
        op.mode = (uint32_t)lp_resize_mode((int)r.crop_w, (int)r.crop_h, (int)r.dst_w, (int)r.dst_h, &ix, &iy);
        op.iscale_x = (uint32_t)ix; op.iscale_y = (uint32_t)iy;
        op.inv_area = 1.f / (float)(ix * iy);
        if (op.mode == 2) {
            auto kx = std::make_pair((int)r.crop_w, (int)r.dst_w), ky = std::make_pair((int)r.crop_h, (int)r.dst_h);
            for (int ax = 0; ax < 2; ax++) {
                auto key = ax ? ky : kx;
                auto it = cache.find(key);
                if (it == cache.end()) {
                    uint32_t to = (uint32_t)taps.size(), ro = (uint32_t)ranges.size();
                    lp_area_tab(key.first, key.second, taps, ranges);
                    it = cache.emplace(key, std::make_pair(to, ro)).first;
                }
                if (ax) { op.ytab_off = it->second.first; op.yrange_off = it->second.second; }
                else { op.xtab_off = it->second.first; op.xrange_off = it->second.second; }
            }
            if (r.src.cn == 3) { // k_resize_area3<MAXT>: every column's taps must be consecutive source columns, at most MAXT of them
                uint32_t mx = 0;
                bool contiguous = true;
                for (uint32_t dx = 0; dx < r.dst_w && contiguous; dx++) {
                    const uint32_t t0 = ranges[op.xrange_off + dx], t1 = ranges[op.xrange_off + dx + 1];
                    mx = std::max(mx, t1 - t0);
                    for (uint32_t k = t0 + 1; k < t1; k++) contiguous = contiguous && taps[op.xtab_off + k].si == taps[op.xtab_off + k - 1].si + 1;
                    contiguous = contiguous && t1 > t0;
                }
                op.fast = !contiguous ? 0u : mx <= 6 ? 6u : mx <= 10 ? 10u : mx <= 18 ? 18u : mx <= 34 ? 34u : mx <= 66 ? 66u : 0u;
                if (op.fast) area3_mask |= op.fast == 6 ? 1u : op.fast == 10 ? 2u : op.fast == 18 ? 4u : op.fast == 34 ? 8u : 16u;
            }
        } else if (op.mode == 3) {
            // resizeGeneric_ tables for INTER_AREA with an up-scaling axis (area_mode coefficients, 11-bit fixed point)
            double inv_x = (double)r.dst_w / r.crop_w, inv_y = (double)r.dst_h / r.crop_h, sc_x = 1. / inv_x, sc_y = 1. / inv_y;
            op.xrange_off = (uint32_t)ranges.size();
            uint32_t xmax = r.dst_w;
            for (uint32_t dx = 0; dx < r.dst_w; dx++) {
                int sx = (int)floor(dx * sc_x);
                float fx = (float)((dx + 1) - (sx + 1) * inv_x);
                fx = fx <= 0 ? 0.f : fx - floorf(fx);
                if (sx + 1 >= (int)r.crop_w) {
                    xmax = std::min(xmax, dx);
                    if (sx >= (int)r.crop_w - 1) { fx = 0; sx = (int)r.crop_w - 1; }
                }
                ranges.push_back((uint32_t)sx);
                ranges.push_back((uint32_t)(int32_t)(short)cv_round_f((1.f - fx) * 2048));
                ranges.push_back((uint32_t)(int32_t)(short)cv_round_f(fx * 2048));
            }
            op.xmax = xmax;
            op.yrange_off = (uint32_t)ranges.size();
            for (uint32_t dy = 0; dy < r.dst_h; dy++) {
                int sy = (int)floor(dy * sc_y);
                float fy = (float)((dy + 1) - (sy + 1) * inv_y);
                fy = fy <= 0 ? 0.f : fy - floorf(fy);
                ranges.push_back((uint32_t)sy);
                ranges.push_back((uint32_t)(int32_t)(short)cv_round_f((1.f - fy) * 2048));
                ranges.push_back((uint32_t)(int32_t)(short)cv_round_f(fy * 2048));
            }
        }
        if (!(op.mode == 2 && op.fast)) modes |= 1u << op.mode;
        mdw = std::max(mdw, r.dst_w);
        mdh = std::max(mdh, r.dst_h);
    }
    if (!d_ops_.ensure(sizeof(LpResizeOp) * (size_t)n + 64) || !d_taps_.ensure(sizeof(LpTap) * taps.size() + 64) || !d_ranges_.ensure(4 * ranges.size() + 64))
        return LP_ERR_DEVICE;
    if (!h2d_any(d_ops_.p, ops.data(), sizeof(LpResizeOp) * (size_t)n, true)) return LP_ERR_DEVICE;
    if (!taps.empty() && !h2d_any(d_taps_.p, taps.data(), sizeof(LpTap) * taps.size(), true)) return LP_ERR_DEVICE;
    if (!ranges.empty() && !h2d_any(d_ranges_.p, ranges.data(), 4 * ranges.size(), true)) return LP_ERR_DEVICE;
    mark(5);
    {
        double rb = 0;
        for (int i = 0; i < n; i++) rb += (double)reqs[i].crop_w * reqs[i].crop_h * reqs[i].src.cn + (double)reqs[i].dst_w * reqs[i].dst_h * reqs[i].src.cn;
        LpStageProbe probe_(stream_, "k_resize_* (crop + INTER_AREA from a frame)", rb);
        lp_launch_resize(stream_, d_ops_.as<LpResizeOp>(), (uint32_t)n, modes & 15u, area3_mask, mdw, mdh, d_taps_.as<LpTap>(), d_ranges_.as<uint32_t>(), nullptr, nullptr);
    }
    mark(6);
    if (!check(hipStreamSynchronize(stream_), "resize sync")) return LP_ERR_DEVICE;
    if (!check(hipGetLastError(), "resize kernels")) return LP_ERR_DEVICE;
    if (timing_) (void)hipEventElapsedTime(&tm_.resize_ms, ev_[5], ev_[6]);
    return LP_OK;
}

// Which thread-per-box kernel takes a fused op, if any (LpFusedOp::fast and the launch-mask bit of lp_launch_resample_fused).
bool lp_fused_op_is_fast(const LpFusedOp& op, const LpJpeg& j, uint32_t* fast_out, uint32_t* mask_bit)
{
    // the 4:2:0 thread-per-pixel kernel needs aligned, even boxes (see k_resample_420)
    const bool swapped = op.dxy != 0;
    const uint32_t U = swapped ? op.dst.h : op.dst.w;
    const int32_t stepx = swapped ? op.dyx : op.dxx;  // source-x step between neighbouring boxes: +-rw
    const bool rw_ok = op.rw == 8 || op.rw == 16 || op.rw == 32;
    const bool steps = stepx == (int32_t)op.rw || stepx == -(int32_t)op.rw;
    const bool aligned = rw_ok && (op.x0 % (int32_t)op.rw) == 0 && steps && op.dst.cn == 3;
    const bool ycc = j.ncomp == 3 && j.colorspace == 2;
    const uint32_t rwbit = op.rw == 8 ? 1u : op.rw == 16 ? 2u : 4u;
    // 2- and 4-pixel boxes of a 4:2:0 source (a 512 x 512 or 1024 x 1024 file and a 256 x 256 thumbnail): k_resample_420_small walks tiles of
    // eight luma columns = four or two boxes, so the boxes must tile that grid exactly
    const uint32_t nb = op.rw == 2 ? 4u : op.rw == 4 ? 2u : 0u;
    // (mirrored orientations: op.x0 is the box of destination 0, the one at the HIGHEST x; the tile's origin is NB - 1 boxes below it)
    const bool small_ok = nb && steps && op.dst.cn == 3 && (U % nb) == 0 && ((stepx > 0 ? op.x0 : op.x0 + (int32_t)op.rw) % 8) == 0;
    uint32_t fast = 0, bit = 0;
    if (j.ncomp == 1 && op.dst.cn == 1) {                                                      // k_resample_gray
        fast = 0x1000u;
        bit = 0x4000u;
    } else if (ycc && small_ok && j.hs[0] == 2 && j.vs[0] == 2 && !(op.rh & 1) && !(op.y0 & 1)) {   // k_resample_420_small<8 / rw>
        fast = op.rw / 2;
        bit = op.rw == 2 ? 0x1000u : 0x2000u;
    } else if (ycc && aligned && j.hs[0] == 2 && j.vs[0] == 2 && !(op.rh & 1) && !(op.y0 & 1)) { // k_resample_420<rw / 2>
        fast = op.rw / 2;
        bit = rwbit;
    } else if (ycc && aligned && j.vs[0] == 1 && (j.hs[0] == 1 || j.hs[0] == 2)) {         // k_resample_hv1<rw, hs>: 4:4:4 / 4:2:2
        fast = 0x100u * (1u + j.hs[0]) + op.rw;
        bit = rwbit << (j.hs[0] == 1 ? 4 : 8);
    }
    if (fast_out) *fast_out = fast;
    if (mask_bit) *mask_bit = bit;
    return fast != 0;
}

int LpEngine::fused_resample(const LpFusedOp* ops_in, int n)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (n <= 0) return LP_OK;
    if (!d_fops_.ensure(sizeof(LpFusedOp) * (size_t)n)) return LP_ERR_DEVICE;
    std::vector<LpFusedOp>& ops = h_fops_; // stays alive: the copy below is not waited for
    ops.assign(ops_in, ops_in + n);
    uint32_t max_px = 0, fast_mask = 0, fast_grid = 0;
    bool general = false;
    for (auto& op : ops) {
        const LpJpeg& j = h_imgs_[op.img];
        const bool swapped = op.dxy != 0;
        const uint32_t U = swapped ? op.dst.h : op.dst.w, V = swapped ? op.dst.w : op.dst.h;
        uint32_t bit = 0;
        if (lp_fused_op_is_fast(op, j, &op.fast, &bit)) {
            fast_mask |= bit;
            fast_grid = std::max(fast_grid, V * ((U + 255u) / 256u));
        } else {
            op.fast = 0;
            general = true;
            max_px = std::max(max_px, op.dst.w * op.dst.h);
        }
    }
    if (!h2d_small(d_fops_.p, ops.data(), sizeof(LpFusedOp) * (size_t)n)) return LP_ERR_DEVICE;
    mark(11);
    lp_launch_resample_fused(stream_, d_imgs_.as<LpJpeg>(), d_fops_.as<LpFusedOp>(), (uint32_t)n, max_px, general, fast_mask, fast_grid, d_planes_.as<uint8_t>());
    mark(7);
    // not waited for: the encode (or whatever reads the thumbnails next) is enqueued behind it; resample_ms() reads the events
    if (!check(hipGetLastError(), "fused resample kernel")) return LP_ERR_DEVICE;
    tm_.resize_ms = 0;
    fused_timed_ = timing_;
    return LP_OK;
}

static uint32_t area_bucket_of(const std::vector<LpTap>& taps, const std::vector<uint32_t>& ranges, uint32_t tab_off, uint32_t range_off, uint32_t dsize)
{
    uint32_t mx = 0;
    for (uint32_t d = 0; d < dsize; d++) {
        const uint32_t t0 = ranges[range_off + d], t1 = ranges[range_off + d + 1];
        if (t1 <= t0) return 0;
        mx = std::max(mx, t1 - t0);
        for (uint32_t k = t0 + 1; k < t1; k++)
            if (taps[tab_off + k].si != taps[tab_off + k - 1].si + 1) return 0;
    }
    return mx <= 6 ? 6u : mx <= 10 ? 10u : mx <= 18 ? 18u : mx <= 34 ? 34u : mx <= 66 ? 66u : 0u;
}

int lp_area_sampling(const LpJpeg& j)
{
    if (j.ncomp != 3 || j.generic_sampling || j.colorspace != 2) return -1;
    if (j.hs[0] == 1 && j.vs[0] == 1) return 0;
    if (j.hs[0] == 2 && j.width > 4) return j.vs[0] == 2 ? 2 : j.vs[0] == 1 ? 1 : -1; // jdsample.c: no fancy upsampling up to 4 pixels of width
    return -1;
}

uint32_t lp_area420_bucket(int ssize, int dsize, bool transposed)
{
    if (ssize <= 0 || dsize <= 0 || dsize >= ssize) return 0;
    std::vector<LpTap> taps;
    std::vector<uint32_t> ranges;
    lp_area_tab(ssize, dsize, taps, ranges);
    const uint32_t b = area_bucket_of(taps, ranges, 0, 0, (uint32_t)dsize);
    return transposed && b > 34 ? 0 : b;
}

uint32_t lp_area420_bucket_int(int scale, bool transposed)
{
    if (scale < 2 || scale > 66) return 0;
    const uint32_t b = scale <= 6 ? 6u : scale <= 10 ? 10u : scale <= 18 ? 18u : scale <= 34 ? 34u : 66u;
    return transposed && b > 34 ? 0 : b;
}

// resizeAreaFast_'s boxes as a tap table: destination dx takes the `scale` source columns from dx * scale, weight 1 each
static void area_tab_int(int dsize, int scale, std::vector<LpTap>& taps, std::vector<uint32_t>& ranges)
{
    const uint32_t base = (uint32_t)taps.size();
    for (int dx = 0; dx < dsize; dx++) {
        ranges.push_back((uint32_t)taps.size() - base);
        for (int k = 0; k < scale; k++) taps.push_back(LpTap{(uint32_t)(dx * scale + k), 1.f});
    }
    ranges.push_back((uint32_t)taps.size() - base);
}

int LpEngine::area_resample(const LpAreaReq* reqs, int n, bool after_fused)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (n <= 0) return LP_OK;
    std::vector<LpArea420Op>& ops = h_aops_; // these three stay alive: the copies below are not waited for
    std::vector<LpTap>& taps = h_ataps_;
    std::vector<uint32_t>& ranges = h_aranges_;
    ops.assign((size_t)n, LpArea420Op{});
    taps.clear();
    ranges.clear();
    std::map<std::pair<int, int>, std::pair<uint32_t, uint32_t>> cache; // (ssize, dsize) -> (tap_off, range_off)
    uint32_t mask[3] = {0, 0, 0}, mdw = 0, mdh = 0;
    for (int i = 0; i < n; i++) {
        const LpAreaReq& r = reqs[i];
        const LpJpeg& j = h_imgs_[r.img];
        LpArea420Op& op = ops[(size_t)i];
        const int ss = lp_area_sampling(j);
        if (ss < 0 || (r.xstep != 1 && r.xstep != -1) ||
            (r.ystep != 1 && r.ystep != -1) || !r.dst.off || !r.dst.w || !r.dst.h) {
            err_ = "area_resample: not a YCbCr 4:2:0 / 4:2:2 / 4:4:4 image";
            return LP_ERR_DEVICE;
        }
        op.img = r.img; op.x0 = r.x0; op.y0 = r.y0; op.xstep = r.xstep; op.ystep = r.ystep;
        op.transposed = r.transposed ? 1u : 0u;
        op.dst = r.dst; op.dst.cn = 3;
        if (!op.dst.stride) op.dst.stride = r.dst.w * 3;
        // integer scales: unit taps (keyed by a negative source size: -scale) and the finish of resizeAreaFast_
        const bool integer = r.int_x && r.int_y;
        op.post = integer ? 1.f / (float)(r.int_x * r.int_y) : 1.f;
        op.half_up = integer && r.int_x == 2 && r.int_y == 2 ? 1u : 0u;
        const auto kx = std::make_pair(integer ? -(int)r.int_x : (int)r.crop_w, (int)r.dst.w), ky = std::make_pair(integer ? -(int)r.int_y : (int)r.crop_h, (int)r.dst.h);
        for (int ax = 0; ax < 2; ax++) {
            const auto key = ax ? ky : kx;
            auto it = cache.find(key);
            if (it == cache.end()) {
                const uint32_t to = (uint32_t)taps.size(), ro = (uint32_t)ranges.size();
                if (integer) area_tab_int(key.second, -key.first, taps, ranges);
                else lp_area_tab(key.first, key.second, taps, ranges);
                it = cache.emplace(key, std::make_pair(to, ro)).first;
            }
            if (ax) { op.ytab_off = it->second.first; op.yrange_off = it->second.second; }
            else { op.xtab_off = it->second.first; op.xrange_off = it->second.second; }
        }
        // the axis that runs along source x sets the instantiation
        op.maxt = op.transposed ? area_bucket_of(taps, ranges, op.ytab_off, op.yrange_off, r.dst.h) : area_bucket_of(taps, ranges, op.xtab_off, op.xrange_off, r.dst.w);
        if (!op.maxt || (op.transposed && op.maxt > 34)) { err_ = "area_resample: no kernel for this many taps"; return LP_ERR_DEVICE; }
        const uint32_t bit = op.maxt == 6 ? 0u : op.maxt == 10 ? 1u : op.maxt == 18 ? 2u : op.maxt == 34 ? 3u : 4u;
        mask[ss] |= op.transposed ? 1u << (10u + bit + (r.xstep < 0 ? 4u : 0u)) : 1u << (bit + (r.xstep < 0 ? 5u : 0u));
        mdw = std::max(mdw, r.dst.w);
        mdh = std::max(mdh, r.dst.h);
    }
    if (!d_aops_.ensure(sizeof(LpArea420Op) * (size_t)n) || !d_ataps_.ensure(sizeof(LpTap) * taps.size() + 64) || !d_aranges_.ensure(4 * ranges.size() + 64))
        return LP_ERR_DEVICE;
    {
        SmallBatch sb;
        if (!small_copy(sb, d_aops_.p, ops.data(), sizeof(LpArea420Op) * (size_t)n) || !small_copy(sb, d_ataps_.p, taps.data(), sizeof(LpTap) * taps.size()) ||
            !small_copy(sb, d_aranges_.p, ranges.data(), 4 * ranges.size()) || !small_flush(sb))
            return LP_ERR_DEVICE;
    }
    if (!(after_fused && fused_timed_)) mark(11); // behind a fused_resample: its start event stands
    lp_launch_area_420(stream_, d_imgs_.as<LpJpeg>(), d_aops_.as<LpArea420Op>(), (uint32_t)n, mask, mdw, mdh, d_ataps_.as<LpTap>(), d_aranges_.as<uint32_t>(),
                       d_planes_.as<uint8_t>());
    mark(7);
    if (!check(hipGetLastError(), "area resample kernel")) return LP_ERR_DEVICE;
    tm_.resize_ms = 0;
    fused_timed_ = timing_;
    return LP_OK;
}

// Test access: lp_area420_pixel / lp_area420t_pixel on the host (what k_area_420 / k_area_420t run per thread), so that the order of
// operations can be compared with the oracle without a GPU. Not a product path: nothing in the library calls it.
template <int MAXT, int SS, bool FLIP>
static void area420_host_run(const LpAreaPlanes& P, const LpArea420Op& op, const std::vector<LpTap>& taps, const std::vector<uint32_t>& ranges, uint8_t* out)
{
    for (uint32_t dy = 0; dy < op.dst.h; dy++)
        for (uint32_t dx = 0; dx < op.dst.w; dx++) {
            const uint32_t x0 = ranges[op.xrange_off + dx], x1 = ranges[op.xrange_off + dx + 1];
            const uint32_t y0 = ranges[op.yrange_off + dy], y1 = ranges[op.yrange_off + dy + 1];
            uint8_t* o = out + ((size_t)dy * op.dst.w + dx) * 3;
            float wt[MAXT];
            if (op.transposed) {
                const LpTap* yt = taps.data() + op.ytab_off + y0;
                for (int k = 0; k < MAXT; k++) wt[k] = (uint32_t)k < y1 - y0 ? yt[k].alpha : 0.f;
                const int32_t si0 = (int32_t)yt[0].si;
                const int32_t xa = FLIP ? op.x0 - si0 - (MAXT - 1) : op.x0 + si0;
                lp_area420t_pixel<MAXT, SS, FLIP>(P, xa, wt, taps.data() + op.xtab_off, x0, x1, op.y0, op.ystep, o);
            } else {
                const LpTap* xt = taps.data() + op.xtab_off + x0;
                for (int k = 0; k < MAXT; k++) wt[k] = (uint32_t)k < x1 - x0 ? xt[k].alpha : 0.f;
                const int32_t si0 = (int32_t)xt[0].si;
                const int32_t xa = FLIP ? op.x0 - si0 - (MAXT - 1) : op.x0 + si0;
                lp_area420_pixel<MAXT, SS, FLIP>(P, xa, wt, taps.data() + op.ytab_off, y0, y1, op.y0, op.ystep, o);
            }
        }
}

// crop_* in oriented coordinates, w / h the decoded (un-oriented) size
static void area420_place(int orientation, int w, int h, int crop_x, int crop_y, int32_t* x0, int32_t* xstep, int32_t* y0, int32_t* ystep, uint32_t* transposed)
{
    // cv::ExifTransform inverse: which source column / row an oriented index names
    if (orientation <= 4) {
        const bool fx = orientation == 2 || orientation == 3, fy = orientation == 3 || orientation == 4;
        *transposed = 0;
        *x0 = fx ? w - 1 - crop_x : crop_x; *xstep = fx ? -1 : 1;
        *y0 = fy ? h - 1 - crop_y : crop_y; *ystep = fy ? -1 : 1;
    } else {
        // oriented row oy is source column oy (5, 6) or w - 1 - oy (7, 8); oriented column ox is source row ox (5, 8) or h - 1 - ox (6, 7)
        const bool fc = orientation == 7 || orientation == 8, fr = orientation == 6 || orientation == 7;
        *transposed = 1;
        *x0 = fc ? w - 1 - crop_y : crop_y; *xstep = fc ? -1 : 1;
        *y0 = fr ? h - 1 - crop_x : crop_x; *ystep = fr ? -1 : 1;
    }
}

void lp_area420_place(int orientation, int w, int h, int crop_x, int crop_y, LpAreaReq* rq)
{
    area420_place(orientation, w, h, crop_x, crop_y, &rq->x0, &rq->xstep, &rq->y0, &rq->ystep, &rq->transposed);
}

extern "C" int lilliput_hip_area420_host(const uint8_t* py, const uint8_t* pb, const uint8_t* pr, uint32_t stride_y, uint32_t stride_c, int w, int h, int sampling,
                                         int orientation, int crop_x, int crop_y, int crop_w, int crop_h, int dst_w, int dst_h, uint8_t* out)
{
    if (orientation < 1 || orientation > 8 || sampling < 0 || sampling > 2 || (sampling && w <= 4) || (stride_y & 3) || (stride_c & 3)) return 1;
    int ix, iy;
    if (lp_resize_mode(crop_w, crop_h, dst_w, dst_h, &ix, &iy) != 2) return 1;
    std::vector<LpTap> taps;
    std::vector<uint32_t> ranges;
    LpArea420Op op{};
    op.xtab_off = 0; op.xrange_off = 0;
    lp_area_tab(crop_w, dst_w, taps, ranges);
    op.ytab_off = (uint32_t)taps.size(); op.yrange_off = (uint32_t)ranges.size();
    lp_area_tab(crop_h, dst_h, taps, ranges);
    area420_place(orientation, w, h, crop_x, crop_y, &op.x0, &op.xstep, &op.y0, &op.ystep, &op.transposed);
    op.maxt = op.transposed ? area_bucket_of(taps, ranges, op.ytab_off, op.yrange_off, (uint32_t)dst_h) : area_bucket_of(taps, ranges, 0, 0, (uint32_t)dst_w);
    if (!op.maxt || (op.transposed && op.maxt > 34)) return 1;
    op.dst.w = (uint32_t)dst_w; op.dst.h = (uint32_t)dst_h;
    const bool flip = op.xstep < 0;
    LpAreaPlanes P{py, pb, pr, stride_y, stride_c, sampling ? (w + 1) >> 1 : w, sampling == 2 ? (h + 1) >> 1 : h};
#define LP_AREA_HOST2(T, S) if (flip) area420_host_run<T, S, true>(P, op, taps, ranges, out); else area420_host_run<T, S, false>(P, op, taps, ranges, out)
#define LP_AREA_HOST(T) case T: if (sampling == 2) { LP_AREA_HOST2(T, 2); } else if (sampling == 1) { LP_AREA_HOST2(T, 1); } else { LP_AREA_HOST2(T, 0); } break
    switch (op.maxt) { LP_AREA_HOST(6); LP_AREA_HOST(10); LP_AREA_HOST(18); LP_AREA_HOST(34); LP_AREA_HOST(66); default: return 1; }
#undef LP_AREA_HOST
#undef LP_AREA_HOST2
    return 0;
}

float LpEngine::fused_resample_ms()
{
    float ms = 0;
    if (fused_timed_ && hipEventSynchronize(ev_[7]) == hipSuccess) (void)hipEventElapsedTime(&ms, ev_[11], ev_[7]);
    fused_timed_ = false;
    return ms;
}

int LpEngine::composite(const LpCompositeOp& op)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    { LpStageProbe probe_(stream_, "k_composite", (double)((size_t)op.w * op.h * (op.src.cn + 2 * op.dst.cn))); lp_launch_composite(stream_, op, nullptr, nullptr); }
    if (!check(hipStreamSynchronize(stream_), "composite sync")) return LP_ERR_DEVICE;
    return check(hipGetLastError(), "composite kernel") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::gather_samples(const LpFrame& f, const uint32_t* idx, uint32_t w, uint32_t h, uint8_t* out)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t ib = ((size_t)(w + h) * 4 + 255) & ~(size_t)255, ob = (size_t)w * h * f.cn;
    if (!d_ops_.ensure(ib + ob + 64)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    uint8_t* base = d_ops_.as<uint8_t>();
    if (!h2d_any(base, idx, (size_t)(w + h) * 4)) return LP_ERR_DEVICE;
    lp_launch_gather_samples(stream_, f, reinterpret_cast<const uint32_t*>(base), w, h, base + ib);
    const uint8_t* got = d2h_begin(base + ib, ob);
    if (!got) return LP_ERR_DEVICE;
    if (!check(hipStreamSynchronize(stream_), "gather sync")) return LP_ERR_DEVICE;
    memcpy(out, got, ob);
    return check(hipGetLastError(), "gather kernel") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::webp_yuv420(const LpFrame& f, const LpWebpYuvTab& tab, uint8_t* y, uint8_t* u, uint8_t* v, bool* translucent)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t yb = (size_t)f.w * f.h, cb = (size_t)((f.w + 1) / 2) * ((f.h + 1) / 2);
    const size_t tab_b = (sizeof(LpWebpYuvTab) + 255) & ~(size_t)255, y_b = (yb + 255) & ~(size_t)255, c_b = (cb + 255) & ~(size_t)255;
    if (!d_ops_.ensure(tab_b + 256 + y_b + 2 * c_b + 64)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    uint8_t* base = d_ops_.as<uint8_t>();
    uint32_t* flag = h_small_.ensure(4096) ? h_small_.as<uint32_t>() : nullptr;
    if (!flag) return LP_ERR_DEVICE;
    if (!h2d_any(base, &tab, sizeof(tab))) return LP_ERR_DEVICE;
    if (!check(hipMemsetAsync(base + tab_b, 0, 256, stream_), "memset webp flag")) return LP_ERR_DEVICE;
    uint8_t *dy = base + tab_b + 256, *du = dy + y_b, *dv = du + c_b;
    { LpStageProbe probe_(stream_, "k_webp_yuv420", (double)((size_t)f.w * f.h * f.cn + yb + 2 * cb)); lp_launch_webp_yuv420(stream_, f, reinterpret_cast<const LpWebpYuvTab*>(base), dy, du, dv, reinterpret_cast<uint32_t*>(base + tab_b)); }
    // the three planes lie behind each other in the arena (256-byte aligned): one transfer into the pinned buffer, three memcpys out of it
    const uint8_t* got = d2h_begin(dy, y_b + 2 * c_b);
    if (!got) return LP_ERR_DEVICE;
    if (!check(hipMemcpyAsync(flag, base + tab_b, 4, hipMemcpyDeviceToHost, stream_), "D2H webp flag")) return LP_ERR_DEVICE;
    if (!check(hipStreamSynchronize(stream_), "webp yuv sync")) return LP_ERR_DEVICE;
    memcpy(y, got, yb);
    memcpy(u, got + y_b, cb);
    memcpy(v, got + y_b + c_b, cb);
    *translucent = *flag != 0;
    return check(hipGetLastError(), "webp yuv kernel") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::png_decode(LpPngOp op, const uint8_t* filtered, size_t n, const uint8_t* palette_bgra)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t pal_off = (n + LP_PNG_MARGIN + 255) & ~(size_t)255;
    const size_t sync_b = lp_png_sync_bytes(op);
    if (!d_planes_.ensure(LP_PNG_MARGIN + pal_off + 1024 + 256 + sync_b)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    uint8_t* base = d_planes_.as<uint8_t>() + LP_PNG_MARGIN; // the stream, with LP_PNG_MARGIN readable bytes before it and (up to the palette) behind it
    if (n && !h2d_any(base, filtered, n)) return LP_ERR_DEVICE;
    if (!h2d_any(base + pal_off, palette_bgra, 1024)) return LP_ERR_DEVICE;
    if (!check(hipMemsetAsync(base + pal_off + 1024, 0, 256 + sync_b, stream_), "memset png flag")) return LP_ERR_DEVICE; // the error flag, the un-filter kernel's tickets and its mailboxes (no stale tag may match)
    op.data_off = (uint64_t)(uintptr_t)base;
    op.palette_off = (uint64_t)(uintptr_t)(base + pal_off);
    op.error_off = (uint64_t)(uintptr_t)(base + pal_off + 1024);
    op.sync_off = (uint64_t)(uintptr_t)(base + pal_off + 1024 + 256);
    { LpStageProbe probe_(stream_, "k_png_unfilter + k_png_convert", (double)(3 * n + (size_t)op.dst.w * op.dst.h * op.dst.cn)); lp_launch_png(stream_, op); }
    uint32_t* flag = h_small_.ensure(4096) ? h_small_.as<uint32_t>() : nullptr;
    if (!flag) return LP_ERR_DEVICE;
    if (!check(hipMemcpyAsync(flag, base + pal_off + 1024, 4, hipMemcpyDeviceToHost, stream_), "D2H png flag")) return LP_ERR_DEVICE;
    if (!check(hipStreamSynchronize(stream_), "png sync")) return LP_ERR_DEVICE;
    if (!check(hipGetLastError(), "png kernels")) return LP_ERR_DEVICE;
    if (*flag & 2u) { err_ = "PNG: the un-filter kernel gave up waiting for a band (device trouble)"; return LP_ERR_DEVICE; }
    if (*flag) { err_ = "PNG: bad adaptive filter value"; return LP_ERR_DECODE_FAILED; }
    return LP_OK;
}

int LpEngine::png_filter(const LpFrame& src, uint32_t filters, uint8_t* out)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t n = (size_t)src.h * ((size_t)src.w * src.cn + 1);
    if (!d_packed_.ensure(n + 64)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    LpPngEncOp op;
    memset(&op, 0, sizeof(op));
    op.src = src;
    op.out_off = (uint64_t)(uintptr_t)d_packed_.p;
    op.filters = filters;
    { LpStageProbe probe_(stream_, "k_png_filter", (double)((size_t)src.w * src.h * src.cn + n)); lp_launch_png_filter(stream_, op); }
    const uint8_t* got = d2h_begin(d_packed_.p, n);
    if (!got) return LP_ERR_DEVICE;
    if (!check(hipStreamSynchronize(stream_), "png filter sync")) return LP_ERR_DEVICE;
    memcpy(out, got, n);
    return check(hipGetLastError(), "png filter kernel") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::tonemap(const LpFrame& f, int transfer, int primaries, const uint16_t* d_src16, int depth)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    if (f.cn != 3 && f.cn != 4) { err_ = "tone map: 3 or 4 channels"; return LP_ERR_INVALID_IMAGE; }
    const size_t npix = (size_t)f.w * f.h;
    if (!npix) return LP_OK;
    const size_t stats_bytes = (size_t)LP_TONE_MAX_WG * LP_TONE_STATS * sizeof(double);
    if (!d_packed_.ensure(npix * 3 * sizeof(float) + stats_bytes)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    LpToneOp op;
    memset(&op, 0, sizeof(op));
    op.f = f;
    op.stats = (uint64_t)(uintptr_t)d_packed_.p;
    op.img = op.stats + stats_bytes;
    op.transfer = transfer; op.primaries = primaries;
    op.src16 = (uint64_t)(uintptr_t)d_src16; op.depth = (uint32_t)depth;
    std::vector<double> part((size_t)LP_TONE_MAX_WG * LP_TONE_STATS);
    double mn = 0, mx = 0, sum[5];
    // one pass + the fold of its per-workgroup partials
    auto run = [&](int pass) -> bool {
        uint32_t nwg = 0;
        lp_launch_tonemap(stream_, op, pass, &nwg);
        if (pass == 3) return check(hipStreamSynchronize(stream_), "tone map sync") && check(hipGetLastError(), "tone map kernels");
        if (!check(hipMemcpyAsync(part.data(), d_packed_.p, (size_t)nwg * LP_TONE_STATS * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H tone stats")) return false;
        if (!check(hipStreamSynchronize(stream_), "tone map sync") || !check(hipGetLastError(), "tone map kernels")) return false;
        mn = part[0]; mx = part[1];
        for (int k = 0; k < 5; k++) sum[k] = 0;
        for (uint32_t w = 0; w < nwg; w++) {
            const double* q = &part[(size_t)w * LP_TONE_STATS];
            mn = std::min(mn, q[0]); mx = std::max(mx, q[1]);
            for (int k = 0; k < 5; k++) sum[k] += q[2 + k];
        }
        return true;
    };
    // cv::Tonemap's linear map: (v - min) / (max - min) as convertTo(alpha, beta); identity when the image is flat
    auto normalise = [&]() {
        if (mx - mn > DBL_EPSILON) { op.a = (float)(1.0 / (mx - mn)); op.b = (float)(-mn / (mx - mn)); }
        else { op.a = 1.0f; op.b = 0.0f; }
    };
    if (!run(0)) return LP_ERR_DEVICE;
    normalise();
    if (!run(1)) return LP_ERR_DEVICE;
    {   // cv::TonemapReinhard::process: key of the log-luminance histogram, global adaptation levels
        const float color_adapt = 0.3f;
        const double n = (double)npix;
        const float log_mean = (float)(sum[0] / n);
        const double key = (float)((mx - log_mean) / (mx - mn));
        op.map_key = 0.3f + 0.7f * powf((float)key, 1.4f);
        op.intensity = expf(-0.6f);
        const float gray_mean = (float)(sum[1] / n);
        for (int c = 0; c < 3; c++) op.glob[c] = color_adapt * (float)(sum[2 + c] / n) + (1.0f - color_adapt) * gray_mean;
    }
    if (!run(2)) return LP_ERR_DEVICE;
    normalise();
    return run(3) ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::tonemap_host8(uint8_t* pixels, int w, int h, int cn, int transfer, int primaries)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t bytes = (size_t)w * h * cn;
    if (!d_tone_.ensure(bytes)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    if (!check(hipMemcpyAsync(d_tone_.p, pixels, bytes, hipMemcpyHostToDevice, stream_), "H2D tone map")) return LP_ERR_DEVICE;
    LpFrame f;
    f.off = (uint64_t)(uintptr_t)d_tone_.p; f.w = (uint32_t)w; f.h = (uint32_t)h; f.stride = (uint32_t)(w * cn); f.cn = (uint32_t)cn;
    if (int rc = tonemap(f, transfer, primaries)) return rc;
    if (!check(hipMemcpyAsync(pixels, d_tone_.p, bytes, hipMemcpyDeviceToHost, stream_), "D2H tone map")) return LP_ERR_DEVICE;
    return check(hipStreamSynchronize(stream_), "tone map sync") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::tonemap_host(const uint16_t* src, uint8_t* dst, int w, int h, int depth, int transfer, int primaries)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t n = (size_t)w * h * 3, dst_off = (n * 2 + 255) & ~(size_t)255;
    if (!d_tone_.ensure(dst_off + n)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    if (!check(hipMemcpyAsync(d_tone_.p, src, n * 2, hipMemcpyHostToDevice, stream_), "H2D tone map")) return LP_ERR_DEVICE;
    LpFrame f;
    f.off = (uint64_t)(uintptr_t)d_tone_.p + dst_off; f.w = (uint32_t)w; f.h = (uint32_t)h; f.stride = (uint32_t)(w * 3); f.cn = 3;
    if (int rc = tonemap(f, transfer, primaries, d_tone_.as<uint16_t>(), depth)) return rc;
    if (!check(hipMemcpyAsync(dst, d_tone_.as<uint8_t>() + dst_off, n, hipMemcpyDeviceToHost, stream_), "D2H tone map")) return LP_ERR_DEVICE;
    return check(hipStreamSynchronize(stream_), "tone map sync") ? LP_OK : LP_ERR_DEVICE;
}

int LpEngine::gif_frame(LpGifFrameOp op, const uint8_t* indices, size_t n_indices, const uint8_t* palette_bgra)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    const size_t pal_off = (n_indices + 255) & ~(size_t)255;
    if (!d_ops_.ensure(pal_off + 1024)) { err_ = "device allocation failed"; return LP_ERR_DEVICE; }
    uint8_t* base = d_ops_.as<uint8_t>();
    if (n_indices && !h2d_any(base, indices, n_indices)) return LP_ERR_DEVICE;
    if (!h2d_any(base + pal_off, palette_bgra, 1024)) return LP_ERR_DEVICE;
    op.index_off = (uint64_t)(uintptr_t)base;
    op.palette_off = (uint64_t)(uintptr_t)(base + pal_off);
    { LpStageProbe probe_(stream_, "k_gif_frame", (double)(n_indices + 2.0 * op.canvas.w * op.canvas.h * 4)); lp_launch_gif_frame(stream_, op); }
    if (!check(hipStreamSynchronize(stream_), "gif frame sync")) return LP_ERR_DEVICE; // the host buffers are the caller's
    return check(hipGetLastError(), "gif frame kernel") ? LP_OK : LP_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------------
// encode
static const uint8_t kStdLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                      69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                                      81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                        99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                        99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t kZig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static void quant_table(int quality, const uint8_t* std_tbl, uint16_t* q)
{
    // jcparam.c jpeg_set_quality(q, TRUE): jpeg_quality_scaling + jpeg_add_quant_table(force_baseline)
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        long t = ((long)std_tbl[i] * scale + 50L) / 100L;
        if (t <= 0) t = 1;
        if (t > 255) t = 255;
        q[i] = (uint16_t)t;
    }
}

size_t lp_build_jpeg_header(int W, int H, int ncomp, int quality, uint8_t* o, uint16_t qt[2][64])
{
    quant_table(quality, kStdLumaQ, qt[0]);
    quant_table(quality, kStdChromaQ, qt[1]);
    size_t n = 0;
    auto b = [&](int v) { o[n++] = (uint8_t)v; };
    auto w = [&](int v) { o[n++] = (uint8_t)(v >> 8); o[n++] = (uint8_t)v; };
    b(0xFF); b(0xD8);
    b(0xFF); b(0xE0); w(16); b('J'); b('F'); b('I'); b('F'); b(0); b(1); b(1); b(0); w(1); w(1); b(0); b(0);
    for (int t = 0; t < (ncomp == 1 ? 1 : 2); t++) {
        b(0xFF); b(0xDB); w(67); b(t);
        for (int z = 0; z < 64; z++) b(qt[t][kZig[z]]);
    }
    b(0xFF); b(0xC0); w(8 + 3 * ncomp); b(8); w(H); w(W); b(ncomp);
    if (ncomp == 1) { b(1); b(0x11); b(0); }
    else { b(1); b(0x22); b(0); b(2); b(0x11); b(1); b(3); b(0x11); b(1); }
    auto dht = [&](int id, const uint8_t* bits, const uint8_t* vals) {
        int tot = 0;
        for (int l = 1; l <= 16; l++) tot += bits[l];
        b(0xFF); b(0xC4); w(19 + tot); b(id);
        for (int l = 1; l <= 16; l++) b(bits[l]);
        for (int i = 0; i < tot; i++) b(vals[i]);
    };
    dht(0x00, lp_std_huff_bits[0], lp_std_huff_dc_vals);
    dht(0x10, lp_std_huff_bits[1], lp_std_huff_ac_luma);
    if (ncomp == 3) { dht(0x01, lp_std_huff_bits[2], lp_std_huff_dc_vals); dht(0x11, lp_std_huff_bits[3], lp_std_huff_ac_chroma); }
    b(0xFF); b(0xDA); w(6 + 2 * ncomp); b(ncomp); b(1); b(0x00);
    if (ncomp == 3) { b(2); b(0x11); b(3); b(0x11); }
    b(0); b(63); b(0);
    return n;
}

void lp_encode_init_tables()
{
    uint16_t code[4][256];
    uint8_t len[4][256];
    memset(code, 0, sizeof(code));
    memset(len, 0, sizeof(len));
    const uint8_t* vals[4] = {lp_std_huff_dc_vals, lp_std_huff_ac_luma, lp_std_huff_dc_vals, lp_std_huff_ac_chroma};
    for (int t = 0; t < 4; t++) {
        int c = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            for (int i = 0; i < lp_std_huff_bits[t][l]; i++, k++, c++) { code[t][vals[t][k]] = (uint16_t)c; len[t][vals[t][k]] = (uint8_t)l; }
            c <<= 1;
        }
    }
    lp_encode_upload_tables(code, len);
}

int LpEngine::encode_jpegs(const LpEncodeReq* reqs, int n, int* status, uint32_t* out_len)
{
    if (!ok_) return LP_ERR_DEVICE;
    if (!check(hipSetDevice(device_), "hipSetDevice")) return LP_ERR_DEVICE;
    if (!enc_tables_ready_) {
        static std::mutex mu; // engines on several host threads share the device-side table symbol
        std::lock_guard<std::mutex> lk(mu);
        lp_encode_init_tables();
        enc_tables_ready_ = true;
    }
    h_jobs_.assign((size_t)n, LpEncJob());
    h_big_.clear();
    std::vector<uint8_t> hdrs;
    size_t coef_elems = 0, bits_words = 0, out_bytes = 0;
    uint32_t tot_blocks = 0, max_blocks = 0;
    for (int i = 0; i < n; i++) {
        LpEncJob& j = h_jobs_[(size_t)i];
        memset(&j, 0, sizeof(j));
        const LpEncodeReq& r = reqs[i];
        status[i] = LP_OK;
        out_len[i] = 0;
        if (r.src.w == 0 || r.src.h == 0 || r.src.w > 65535 || r.src.h > 65535 || (r.src.cn != 1 && r.src.cn != 3 && r.src.cn != 4)) {
            status[i] = LP_ERR_INVALID_IMAGE;
            continue; // total_blocks == 0: every kernel skips the job
        }
        j.src = r.src;
        j.ncomp = r.src.cn == 1 ? 1 : 3;
        const uint32_t mcu = j.ncomp == 1 ? 8 : 16;
        j.mcus_x = (r.src.w + mcu - 1) / mcu;
        j.mcus_y = (r.src.h + mcu - 1) / mcu;
        j.bpm = j.ncomp == 1 ? 1 : 6;
        j.total_blocks = j.mcus_x * j.mcus_y * j.bpm;
        j.wib = (r.src.w + 7) / 8;
        j.hib = (r.src.h + 7) / 8;
        j.blk_off = tot_blocks;
        tot_blocks += j.total_blocks;
        max_blocks = std::max(max_blocks, j.total_blocks);
        j.coef_off = coef_elems;
        coef_elems += (size_t)j.total_blocks * 64;
        uint8_t hb[1024];
        j.hdr_off = (uint32_t)hdrs.size();
        j.hdr_len = (uint32_t)lp_build_jpeg_header((int)r.src.w, (int)r.src.h, (int)j.ncomp, r.quality, hb, j.qt);
        hdrs.insert(hdrs.end(), hb, hb + j.hdr_len);
        size_t cap = std::min<size_t>(r.out_cap, (size_t)j.total_blocks * 448 + 4096);
        j.out_cap = (uint32_t)std::min<size_t>(cap, 0xfffffff0u);
        j.out_off = out_bytes;
        out_bytes = align_up(out_bytes + j.out_cap, 16);
        j.bits_off = bits_words;
        j.bits_cap_words = (uint32_t)std::min<size_t>((size_t)j.total_blocks * 54 + 16, (size_t)j.out_cap / 4 + 16);
        bits_words += j.bits_cap_words;
    }
    h_estates_.assign((size_t)n, LpEncState());
    if (!d_jobs_.ensure(sizeof(LpEncJob) * (size_t)n) || !d_estates_.ensure(sizeof(LpEncState) * (size_t)n) || !d_ecoef_.ensure(coef_elems * 2 + 64) ||
        !d_blkbits_.ensure((size_t)tot_blocks * 4 + 64) || !d_bits_.ensure(bits_words * 4 + 64) || !d_hdrs_.ensure(hdrs.size() + 64) ||
        !d_out_.ensure(out_bytes + 64) || !h_small_.ensure(std::max<size_t>(4096, sizeof(LpEncState) * (size_t)n + 64)))
        return LP_ERR_DEVICE;
    // where every stream goes in the pinned output buffer (a slot bounds the usual size; a stream that outgrows it is fetched from the
    // output arena by encoded_fetch_all): host values only, so the offsets travel with the other descriptors
    std::vector<uint32_t>& pk = h_pk_;
    pk.assign((size_t)n + 1, 0);
    size_t pk_total = 0;
    h_out_off_.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        const LpEncJob& j = h_jobs_[(size_t)i];
        const size_t slot = align_up(std::min<size_t>(j.out_cap, std::max<size_t>(16384, (size_t)j.src.w * j.src.h * j.ncomp / 2 + 4096)), 16) + 16;
        h_out_off_[(size_t)i] = pk_total;
        pk[(size_t)i] = (uint32_t)pk_total;
        pk_total += j.total_blocks ? slot : 0;
    }
    pk[(size_t)n] = (uint32_t)pk_total;
    if (!enc_fdct_only_ && (pk_total > 0xffffff00ull || !h_out_.ensure(pk_total + 64) || !d_pkoff_.ensure(((size_t)n + 1) * 4 + 64))) return LP_ERR_DEVICE;
    {   // jobs, headers and pack offsets up, bit buffers and states cleared: one launch
        SmallBatch sb;
        if (!small_copy(sb, d_jobs_.p, h_jobs_.data(), sizeof(LpEncJob) * (size_t)n) || (!hdrs.empty() && !small_copy(sb, d_hdrs_.p, hdrs.data(), hdrs.size())) ||
            (!enc_fdct_only_ && !small_copy(sb, d_pkoff_.p, pk.data(), ((size_t)n + 1) * 4)) || !small_zero(sb, d_bits_.p, bits_words * 4 + 64) ||
            !small_zero(sb, d_estates_.p, sizeof(LpEncState) * (size_t)n) || !small_flush(sb))
            return LP_ERR_DEVICE;
    }
    if (enc_fdct_only_) { // progressive output: the entropy coding happens on the host
        lp_launch_enc_fdct(stream_, d_jobs_.as<LpEncJob>(), (uint32_t)n, max_blocks, d_ecoef_.as<int16_t>());
        if (!check(hipStreamSynchronize(stream_), "fdct sync")) return LP_ERR_DEVICE;
        return check(hipGetLastError(), "fdct kernel") ? LP_OK : LP_ERR_DEVICE;
    }
    mark(5);
    {
        double eb = 0;
        for (int i = 0; i < n; i++) eb += (double)h_jobs_[(size_t)i].src.w * h_jobs_[(size_t)i].src.h * h_jobs_[(size_t)i].src.cn * 1.1;
        LpStageProbe probe_(stream_, "k_enc_* (JPEG encode)", eb);
        lp_launch_encode(stream_, d_jobs_.as<LpEncJob>(), d_estates_.as<LpEncState>(), (uint32_t)n, max_blocks, nullptr, d_ecoef_.as<int16_t>(),
                         d_blkbits_.as<uint32_t>(), d_bits_.as<uint32_t>(), d_hdrs_.as<uint8_t>(), d_out_.as<uint8_t>());
    }
    mark(6);
    {   // results: every stream into its slot of the pinned output buffer, and the states -- one wait for both
        lp_launch_enc_pack(stream_, d_jobs_.as<LpEncJob>(), d_estates_.as<LpEncState>(), (uint32_t)n, d_pkoff_.as<uint32_t>(), d_out_.as<uint8_t>(),
                           static_cast<uint8_t*>(h_out_.dev));
    }
    d2h_small(h_small_, h_small_.p, d_estates_.p, sizeof(LpEncState) * (size_t)n);
    if (!check(hipStreamSynchronize(stream_), "encode sync")) return LP_ERR_DEVICE;
    if (!check(hipGetLastError(), "encode kernels")) return LP_ERR_DEVICE;
    if (timing_) (void)hipEventElapsedTime(&tm_.encode_ms, ev_[5], ev_[6]);
    memcpy(h_estates_.data(), h_small_.p, sizeof(LpEncState) * (size_t)n);
    int rc = LP_OK;
    for (int i = 0; i < n; i++) {
        if (status[i]) { rc = status[i]; continue; }
        const LpEncState& st = h_estates_[(size_t)i];
        if (st.error || st.out_len == 0) { status[i] = LP_ERR_BUF_TOO_SMALL; rc = status[i]; continue; }
        out_len[i] = st.out_len;
    }
    return rc;
}

int LpEngine::encode_jpeg_progressive(const LpEncodeReq& req, std::vector<uint8_t>& out)
{
    int st = 0;
    uint32_t len = 0;
    enc_fdct_only_ = true;
    const int rc = encode_jpegs(&req, 1, &st, &len);
    enc_fdct_only_ = false;
    if (rc) return rc;
    if (st) return st;
    const LpEncJob& j = h_jobs_[0];
    std::vector<int16_t> coef((size_t)j.total_blocks * 64);
    if (!check(hipMemcpyAsync(coef.data(), d_ecoef_.as<int16_t>() + j.coef_off, coef.size() * 2, hipMemcpyDeviceToHost, stream_), "D2H coefficients")) return LP_ERR_DEVICE;
    const int src = sync();
    if (src) return src;
    if (!lp_jpeg_encode_progressive((int)j.src.w, (int)j.src.h, (int)j.ncomp, req.quality, coef.data(), out)) { err_ = "coefficient out of range"; return LP_ERR_INVALID_IMAGE; }
    return LP_OK;
}

int LpEngine::encode_jpegs_progressive(const LpEncodeReq* reqs, int n, int* status, std::vector<std::vector<uint8_t>>& outs)
{
    outs.assign((size_t)n, std::vector<uint8_t>());
    if (n <= 0) return LP_OK;
    std::vector<uint32_t> len((size_t)n, 0);
    enc_fdct_only_ = true;
    const int rc = encode_jpegs(reqs, n, status, len.data());
    enc_fdct_only_ = false;
    if (rc == LP_ERR_DEVICE) return rc;
    size_t total = 0;
    for (int i = 0; i < n; i++) total = std::max(total, (size_t)h_jobs_[(size_t)i].coef_off + (size_t)h_jobs_[(size_t)i].total_blocks * 64);
    std::vector<int16_t> coef(total);
    if (total && !check(hipMemcpyAsync(coef.data(), d_ecoef_.p, total * 2, hipMemcpyDeviceToHost, stream_), "D2H coefficients")) return LP_ERR_DEVICE;
    const int src = sync();
    if (src) return src;
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            if (status[i]) continue;
            const LpEncJob& j = h_jobs_[(size_t)i];
            if (!lp_jpeg_encode_progressive((int)j.src.w, (int)j.src.h, (int)j.ncomp, reqs[i].quality, coef.data() + j.coef_off, outs[(size_t)i])) status[i] = LP_ERR_INVALID_IMAGE;
        }
    };
    const int nt = std::min(n, 8);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    int out_rc = LP_OK;
    for (int i = 0; i < n; i++) if (status[i]) out_rc = status[i];
    return out_rc;
}

const uint8_t* LpEngine::encoded_device_ptr(int i) const { return d_out_.as<uint8_t>() + h_jobs_[(size_t)i].out_off; }

int LpEngine::encoded_copy(int i, uint8_t* dst, size_t cap)
{
    const uint32_t len = h_estates_[(size_t)i].out_len;
    if (len == 0 || len > cap) return LP_ERR_BUF_TOO_SMALL;
    if (!check(hipMemcpyAsync(dst, encoded_device_ptr(i), len, hipMemcpyDeviceToHost, stream_), "D2H jpeg")) return LP_ERR_DEVICE;
    return sync();
}

int LpEngine::encoded_fetch_all()
{
    // encode_jpegs has already placed every stream that fits its slot in the pinned buffer; the rest come from the output arena
    const size_t n = h_jobs_.size();
    for (size_t i = 0; i < n; i++) {
        const uint32_t len = h_estates_[i].out_len;
        if (!len || len <= h_pk_[i + 1] - h_pk_[i]) continue;
        h_big_.emplace_back(i, std::vector<uint8_t>(len));
        if (!check(hipMemcpyAsync(h_big_.back().second.data(), encoded_device_ptr((int)i), len, hipMemcpyDeviceToHost, stream_), "D2H jpeg")) return LP_ERR_DEVICE;
    }
    return h_big_.empty() ? LP_OK : sync();
}

const uint8_t* LpEngine::encoded_host(int i) const
{
    for (const auto& b : h_big_)
        if (b.first == (size_t)i) return b.second.data();
    return h_out_.as<uint8_t>() + h_out_off_[(size_t)i];
}
