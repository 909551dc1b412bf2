// ingest_micro.hip -- how should pinned source bytes travel to the device? (round 3, after `bench.py --ingest pinned` came out SLOWER
// than the staged pipeline: 10.7 k vs 11.6 k images/s with 32 copies of ~4 MB per chunk instead of one copy of 135 MB)
//   A. one 128 MiB hipMemcpyAsync from pinned memory
//   B. 32 x 4 MiB copies on one stream, sources 64-byte aligned / sources at odd offsets (an entropy-coded segment starts wherever the
//      file's headers end)
//   C. the same 32 copies spread over 2 / 4 streams
//   D. a gather KERNEL that reads the mapped pinned pieces itself (16-byte loads over the link, re-aligned on the way) and writes the
//      arena: one launch per chunk, G workgroups
//   E. hipHostRegister / hipHostUnregister of a 4 MiB malloc'ed buffer, first and repeated
//   F. host staging copies (pageable -> pinned, streaming stores) with T persistent threads: does the rate collapse with many threads?
// Build: hipcc -O3 --offload-arch=gfx950 -o ingest_micro ingest_micro.hip -lpthread
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Piece { const uint8_t* src; uint64_t dst_off; uint32_t len; uint32_t pad; };

// Every workgroup walks the pieces in a grid-stride over 4 KiB tiles; a lane moves 16 bytes per tile step. The source is read with
// aligned 16-byte loads (two per lane when it is misaligned, combined with a funnel shift), the destination is 16-byte aligned.
__global__ void __launch_bounds__(256) k_gather(const Piece* __restrict__ pcs, int npieces, uint8_t* __restrict__ arena, const uint32_t* __restrict__ tile_first)
{
    // tile_first[p] = index of piece p's first tile; tile_first[npieces] = total
    const uint32_t total = tile_first[npieces];
    for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
        int lo = 0, hi = npieces; // piece of tile t (binary search, wave-uniform)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_first[mid] <= t) lo = mid; else hi = mid; }
        const Piece pc = pcs[lo];
        const uint32_t off = (t - tile_first[lo]) * 4096u + threadIdx.x * 16u;
        if (off >= pc.len) continue;
        const uintptr_t s = (uintptr_t)pc.src + off;
        const uint32_t mis = (uint32_t)(s & 15u);
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* a = (const u32x4*)(s - mis);
        const u32x4 q0 = __builtin_nontemporal_load(a);
        uint4 v0 = make_uint4(q0.x, q0.y, q0.z, q0.w);
        uint4 out = v0;
        if (mis) {
            const u32x4 q1 = __builtin_nontemporal_load(a + 1);
            const uint4 v1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
            const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const uint32_t wd = mis >> 2, sh = (mis & 3u) * 8u;
            uint32_t r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = w[wd + i], y = w[wd + i + 1];
                r[i] = sh ? (x >> sh) | (y << (32u - sh)) : x;
            }
            out = make_uint4(r[0], r[1], r[2], r[3]);
        }
        *(uint4*)(arena + pc.dst_off + off) = out;
    }
}

static void stream_copy(uint8_t* dst, const uint8_t* src, size_t n)
{
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

int main()
{
    const size_t PIECE = 4200000, NP = 32, SLOT = 4608u << 10; // pieces of the size of a 4096 x 4096 q90 entropy-coded segment
    uint8_t *h, *d;
    CK(hipHostMalloc((void**)&h, NP * SLOT + 4096, hipHostMallocMapped | hipHostMallocPortable));
    CK(hipMalloc((void**)&d, NP * SLOT + (1 << 20)));
    for (size_t i = 0; i < NP * SLOT; i += 4096) h[i] = (uint8_t)i;
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, double bytes, auto&& fn) {
        fn(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            const double t0 = now_ms();
            fn();
            CK(hipDeviceSynchronize());
            best = std::min(best, now_ms() - t0);
        }
        printf("  %-86s %7.2f ms  %6.1f GB/s\n", what, best, bytes / best / 1e6);
        fflush(stdout);
    };
    printf("== H2D of one 32-image chunk (%zu pieces of %zu bytes) from pinned memory\n", NP, PIECE);
    timed("A. one copy of the whole chunk", (double)NP * PIECE, [&] { CK(hipMemcpyAsync(d, h, NP * PIECE, hipMemcpyHostToDevice, st[0])); });
    timed("B. 32 copies on one stream, sources 4 MiB apart (aligned)", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT, h + p * SLOT, PIECE, hipMemcpyHostToDevice, st[0])); });
    timed("B'. 32 copies on one stream, sources at +623 bytes (unaligned), destinations aligned", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT, h + p * SLOT + 623, PIECE - 623, hipMemcpyHostToDevice, st[0])); });
    timed("B''. the same, destinations misaligned the same way (+623)", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT + 623, h + p * SLOT + 623, PIECE - 623, hipMemcpyHostToDevice, st[0])); });
    timed("C2. 32 unaligned copies round-robin over 2 streams", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT, h + p * SLOT + 623, PIECE - 623, hipMemcpyHostToDevice, st[p & 1])); });
    timed("C4. 32 unaligned copies round-robin over 4 streams", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT, h + p * SLOT + 623, PIECE - 623, hipMemcpyHostToDevice, st[p & 3])); });
    timed("C4a. 32 ALIGNED copies round-robin over 4 streams", (double)NP * PIECE, [&] { for (size_t p = 0; p < NP; p++) CK(hipMemcpyAsync(d + p * SLOT, h + p * SLOT, PIECE, hipMemcpyHostToDevice, st[p & 3])); });
    {
        Piece* pcs; uint32_t* tf; uint8_t* hd = nullptr;
        CK(hipHostMalloc((void**)&pcs, sizeof(Piece) * NP, hipHostMallocMapped));
        CK(hipHostMalloc((void**)&tf, 4 * (NP + 1), hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void**)&hd, h, 0));
        for (int mis : {0, 623}) {
            uint32_t t = 0;
            for (size_t p = 0; p < NP; p++) { pcs[p] = Piece{hd + p * SLOT + mis, (uint64_t)(p * SLOT), (uint32_t)(PIECE - mis), 0}; tf[p] = t; t += (uint32_t)((PIECE - mis + 4095) / 4096); }
            tf[NP] = t;
            for (int G : {32, 64, 128, 256, 512, 1024}) {
                char what[128];
                snprintf(what, sizeof(what), "D. gather kernel, %4d workgroups of 256, sources %s", G, mis ? "at +623 bytes" : "aligned");
                timed(what, (double)NP * PIECE, [&] { hipLaunchKernelGGL(k_gather, dim3(G), dim3(256), 0, st[0], pcs, (int)NP, d, tf); });
            }
        }
        // correctness of the re-aligning gather
        std::vector<uint8_t> back(PIECE);
        CK(hipMemcpy(back.data(), d + 3 * SLOT, PIECE - 623, hipMemcpyDeviceToHost));
        printf("  gather check: %s\n", memcmp(back.data(), h + 3 * SLOT + 623, PIECE - 623) == 0 ? "bytes identical" : "MISMATCH");
    }
    printf("== hipHostRegister / hipHostUnregister of a 4 MiB malloc'ed (touched) buffer\n");
    {
        const size_t n = 4u << 20;
        uint8_t* p = (uint8_t*)aligned_alloc(4096, n);
        memset(p, 1, n);
        for (int rep = 0; rep < 4; rep++) {
            const double t0 = now_ms();
            const hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
            const double t1 = now_ms();
            const hipError_t u = e == hipSuccess ? hipHostUnregister(p) : e;
            const double t2 = now_ms();
            printf("  #%d register %.3f ms (%s), unregister %.3f ms (%s)\n", rep, t1 - t0, hipGetErrorString(e), t2 - t1, hipGetErrorString(u));
        }
        // 8 threads registering their own buffers at the same time
        std::vector<std::thread> th;
        std::atomic<int> go{0};
        double per[8] = {0};
        for (int k = 0; k < 8; k++)
            th.emplace_back([&, k] {
                uint8_t* q = (uint8_t*)aligned_alloc(4096, n);
                memset(q, 2, n);
                while (!go.load()) std::this_thread::yield();
                const double t0 = now_ms();
                for (int rep = 0; rep < 8; rep++) { if (hipHostRegister(q, n, hipHostRegisterDefault) == hipSuccess) (void)hipHostUnregister(q); }
                per[k] = (now_ms() - t0) / 8;
                free(q);
            });
        go = 1;
        for (auto& t : th) t.join();
        printf("  8 threads at once: %.3f .. %.3f ms per register + unregister pair\n", *std::min_element(per, per + 8), *std::max_element(per, per + 8));
        free(p);
    }
    printf("== host staging copies into pinned slots (persistent threads, 4 MiB pieces, streaming stores vs memcpy)\n");
    for (int nt : {4, 12, 24, 48, 96, 192}) {
        for (int plain = 0; plain < 2; plain++) {
            std::atomic<int> go{0}, ready{0};
            std::atomic<bool> stop{false};
            std::atomic<size_t> bytes{0};
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++)
                th.emplace_back([&, plain] {
                    const size_t n = 4u << 20, nsrc = 8;
                    std::vector<uint8_t*> src(nsrc);
                    for (auto& s : src) { s = (uint8_t*)aligned_alloc(4096, n); memset(s, 3, n); }
                    uint8_t* dst = nullptr;
                    if (hipHostMalloc((void**)&dst, 2 * n, hipHostMallocMapped) != hipSuccess) dst = nullptr;
                    ready++;
                    while (!go.load()) std::this_thread::yield();
                    size_t done = 0;
                    for (size_t it = 0; dst && !stop.load(); it++) {
                        if (plain) memcpy(dst + (it & 1) * n, src[it % nsrc], n); else stream_copy(dst + (it & 1) * n, src[it % nsrc], n);
                        done += n;
                    }
                    bytes += done;
                    if (dst) (void)hipHostFree(dst);
                    for (auto& s : src) free(s);
                });
            while (ready.load() < nt) std::this_thread::yield();
            const double t0 = now_ms();
            go = 1;
            std::this_thread::sleep_for(std::chrono::milliseconds(1500));
            stop = true;
            for (auto& t : th) t.join();
            printf("  %3d threads, %s: %.1f GB/s\n", nt, plain ? "memcpy          " : "streaming stores", bytes.load() / (now_ms() - t0) / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
