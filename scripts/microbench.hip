// microbench.hip -- small hardware facts the design leans on, measured on the GPU box (results quoted in DESIGN.md / profiles/):
//   1. VALU issue rate of a wave64 instruction on gfx950 (2 vs 4 cycles per instruction per SIMD)
//   2. H2D bandwidth from pinned memory (one and two streams), from pageable memory, and D2H
//   3. host memcpy bandwidth pageable -> pinned with T threads (the staging step of the ingest pipeline)
//   4. staging threads and H2D running together (what the pipelined batch does)
//   5. hipHostRegister cost of a 4 MB buffer
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench microbench.hip -lpthread
#include <hip/hip_runtime.h>
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

// ---- 1. VALU issue: ITER x 16 independent integer adds per lane, no memory traffic; waves_per_simd co-resident waves.
template <int DEP>
__global__ void __launch_bounds__(1024) k_valu(uint32_t* out, int iters, uint64_t* cycles)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = blockIdx.x | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static void valu_bench()
{
    uint32_t* d_out; uint64_t* d_cyc;
    CK(hipMalloc(&d_out, 1024 * 1024 * 4)); CK(hipMalloc(&d_cyc, 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000; // x 32 v_add per iteration
    printf("== VALU issue (v_add_u32, 8 independent chains, %d instructions per wave)\n", iters * 32);
    for (int threads : {64, 256, 512, 1024}) {
        for (int blocks : {1, 256}) {
            k_valu<0><<<blocks, threads>>>(d_out, 100, d_cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            k_valu<0><<<blocks, threads>>>(d_out, iters, d_cyc);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            const double ninst = (double)iters * 32;
            const int waves_per_simd = (threads / 64 + 3) / 4;
            printf("  %4d threads/WG x %3d WGs (%d wave(s) per SIMD): %.3f ms, s_memtime %llu ticks -> %.2f ticks per wave-instruction, "
                   "%.2f per instruction per SIMD; wall-clock %.2f ns per wave-instruction\n",
                   threads, blocks, waves_per_simd, ms, (unsigned long long)cyc, cyc / ninst, cyc / ninst / waves_per_simd, ms * 1e6 / ninst);
        }
    }
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("  device clock attribute: %d kHz (s_memtime ticks at a constant 100 MHz on gfx9; use wall-clock x clock for cycles)\n", clk);
    CK(hipFree(d_out)); CK(hipFree(d_cyc));
}

// ---- 2-4. copies
static void copy_bench()
{
    const size_t SZ = 512ull << 20;
    uint8_t *pin0, *pin1, *dev0, *dev1;
    double t = now_ms();
    CK(hipHostMalloc((void**)&pin0, SZ, hipHostMallocDefault));
    double t_pin = now_ms() - t;
    CK(hipHostMalloc((void**)&pin1, SZ, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev0, SZ)); CK(hipMalloc((void**)&dev1, SZ));
    uint8_t* page = (uint8_t*)malloc(SZ * 2);
    for (size_t i = 0; i < SZ * 2; i += 4096) page[i] = (uint8_t)i;
    memset(pin0, 1, SZ); memset(pin1, 2, SZ);
    printf("== copies (512 MiB buffers; hipHostMalloc of 512 MiB took %.1f ms)\n", t_pin);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto gbs = [&](double bytes, double ms) { return bytes / ms / 1e6; };
    for (size_t sz : {4ull << 20, 64ull << 20, 512ull << 20}) {
        CK(hipMemcpyAsync(dev0, pin0, sz, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        const int reps = sz < (64ull << 20) ? 50 : 5;
        t = now_ms();
        for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(dev0, pin0, sz, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        double ms = (now_ms() - t) / reps;
        printf("  H2D pinned, 1 stream, %4zu MiB: %.2f ms  %.1f GB/s\n", sz >> 20, ms, gbs(sz, ms));
    }
    {
        t = now_ms();
        for (int r = 0; r < 5; r++) { CK(hipMemcpyAsync(dev0, pin0, SZ, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(dev1, pin1, SZ, hipMemcpyHostToDevice, s1)); }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        double ms = (now_ms() - t) / 5;
        printf("  H2D pinned, 2 streams x 512 MiB concurrently: %.2f ms  %.1f GB/s aggregate\n", ms, gbs(2.0 * SZ, ms));
        t = now_ms();
        for (int r = 0; r < 5; r++) { CK(hipMemcpyAsync(dev0, pin0, SZ, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pin1, dev1, SZ, hipMemcpyDeviceToHost, s1)); }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        ms = (now_ms() - t) / 5;
        printf("  H2D + D2H concurrently (512 MiB each): %.2f ms  %.1f GB/s each way\n", ms, gbs(SZ, ms));
        t = now_ms();
        for (int r = 0; r < 3; r++) CK(hipMemcpy(dev0, page, SZ, hipMemcpyHostToDevice));
        ms = (now_ms() - t) / 3;
        printf("  H2D pageable (hipMemcpy, runtime staging), 512 MiB: %.2f ms  %.1f GB/s\n", ms, gbs(SZ, ms));
    }
    // host memcpy pageable -> pinned in 4 MiB pieces (one "image" per piece), T threads
    const size_t PIECE = 4ull << 20, NP = SZ / PIECE;
    for (int T : {1, 2, 4, 8, 16, 32, 64}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            std::atomic<size_t> next{0};
            t = now_ms();
            std::vector<std::thread> th;
            for (int k = 0; k < T; k++)
                th.emplace_back([&] { for (;;) { size_t i = next.fetch_add(1); if (i >= NP) break; memcpy(pin0 + i * PIECE, page + ((i * 2 + rep) % (2 * NP)) * PIECE, PIECE); } });
            for (auto& x : th) x.join();
            best = std::min(best, now_ms() - t);
        }
        printf("  host memcpy pageable -> pinned, %2d threads, 512 MiB in 4 MiB pieces: %.2f ms  %.1f GB/s\n", T, best, gbs(SZ, best));
    }
    // pipeline: T threads stage 64 MiB slices into alternating pinned buffers while the previous slice is in flight to the device
    for (int T : {8, 16, 32}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        const size_t SLICE = 64ull << 20, NS = 32; // 2 GiB in all
        hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        t = now_ms();
        for (size_t s = 0; s < NS; s++) {
            uint8_t* pin = (s & 1) ? pin1 : pin0;
            if (s >= 2) CK(hipEventSynchronize(ev[s & 1]));
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            const size_t np = SLICE / PIECE;
            for (int k = 0; k < T; k++)
                th.emplace_back([&] { for (;;) { size_t i = next.fetch_add(1); if (i >= np) break; memcpy(pin + i * PIECE, page + ((s * np + i) % (2 * NP)) * PIECE, PIECE); } });
            for (auto& x : th) x.join();
            CK(hipMemcpyAsync(dev0 + (s & 7) * SLICE, pin, SLICE, hipMemcpyHostToDevice, s0));
            CK(hipEventRecord(ev[s & 1], s0));
        }
        CK(hipStreamSynchronize(s0));
        double ms = now_ms() - t;
        printf("  staged pipeline (%2d threads stage 64 MiB slices, double-buffered H2D): 2 GiB in %.1f ms  %.1f GB/s end to end\n", T, ms, gbs((double)SLICE * NS, ms));
    }
    // hipHostRegister of a caller buffer
    {
        uint8_t* buf = (uint8_t*)aligned_alloc(4096, 4 << 20);
        memset(buf, 3, 4 << 20);
        t = now_ms();
        for (int r = 0; r < 20; r++) { CK(hipHostRegister(buf, 4 << 20, hipHostRegisterDefault)); CK(hipHostUnregister(buf)); }
        printf("  hipHostRegister + Unregister of 4 MiB: %.3f ms per pair\n", (now_ms() - t) / 20);
        free(buf);
    }
    printf("  host threads: %u\n", std::thread::hardware_concurrency());
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, clock %d kHz, mem clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate);
    if (argc < 2 || strcmp(argv[1], "copy")) valu_bench();
    if (argc < 2 || strcmp(argv[1], "valu")) copy_bench();
    return 0;
}
