// ingest_scale.cpp -- rehearsal of the HOST side of the ingest pipeline for N = 1, 2, 4, 8 engine sets (= GPUs of a node), without
// any GPU work: how many GB/s of entropy-coded bytes can the host make DMA-ready, staged (header walk + streaming memcpy into pinned
// slots, the round-2 pipeline) against zero-copy (header walk + hipHostRegister / hipHostUnregister of the caller's pages)?
// One MI355X needs ~49 GB/s of it for the 4096 x 4096 workload (link-bound), eight need ~390 GB/s.
//
//   hipcc -O2 -std=c++17 scripts/ingest_scale.cpp -Ililliput_amd/csrc -Llilliput_amd -llilliput_hip -lpthread -o scripts/ingest_scale
//   LD_LIBRARY_PATH=lilliput_amd scripts/ingest_scale <dir with synth_*.jpg> [seconds per point]
//
// Every set runs what a device's four engines run in lp_batch.cpp: four stager threads (each with two helpers in staged mode), chunks
// of 32 images. Set k is bound to NUMA node k mod nodes (a node's GPUs hang off both sockets); its sources are first-touched there.
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <emmintrin.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "lp_jpeg_parse.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<std::vector<int>> numa_cpus()
{
    std::vector<std::vector<int>> nodes;
    for (int n = 0; n < 64; n++) {
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", n);
        FILE* f = fopen(path, "r");
        if (!f) break;
        char list[4096] = {0};
        if (!fgets(list, sizeof(list), f)) { fclose(f); break; }
        fclose(f);
        std::vector<int> cpus;
        for (char* s = list; *s && *s != '\n';) {
            char* e = nullptr;
            long a = strtol(s, &e, 10), b = a;
            if (e == s) break;
            if (*e == '-') b = strtol(e + 1, &e, 10);
            for (long c = a; c <= b; c++) cpus.push_back((int)c);
            if (*e != ',') break;
            s = e + 1;
        }
        if (!cpus.empty()) nodes.push_back(cpus);
    }
    return nodes;
}

static void bind_to(const std::vector<int>& cpus)
{
    if (cpus.empty()) return;
    cpu_set_t s;
    CPU_ZERO(&s);
    for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &s);
    (void)sched_setaffinity(0, sizeof(s), &s);
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

struct Src { uint8_t* p; size_t len; };

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <dir with *.jpg> [seconds per point]\n", argv[0]); return 2; }
    const double secs = argc > 2 ? atof(argv[2]) : 2.0;
    std::vector<std::string> files;
    if (DIR* d = opendir(argv[1])) {
        while (dirent* e = readdir(d)) { std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".jpg") files.push_back(std::string(argv[1]) + "/" + n); }
        closedir(d);
    }
    std::sort(files.begin(), files.end());
    if (files.empty()) { fprintf(stderr, "no .jpg files in %s\n", argv[1]); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "no HIP device (pinned memory needs the runtime)\n"); return 2; }
    const auto nodes = numa_cpus();
    printf("# host: %u hardware threads, %zu NUMA nodes; %zu distinct sources; %.1f s per point\n", std::thread::hardware_concurrency(), nodes.size(), files.size(), secs);
    const int kStagers = 4, kChunk = 32;
    printf("| mode | engine sets (GPUs) | host threads | GB/s made DMA-ready | per set | needed at the 1-GPU link rate (49 GB/s each) |\n|---|---:|---:|---:|---:|---:|\n");
    // mode 1 (register / unregister the caller's pages per chunk) is kept for reference only: re-registering the SAME buffers in a loop
    // hits the runtime's cache of pinned ranges and reports a rate no first-time registration reaches (profiles/r03_a_ingest.md section 3);
    // run it with a third argument
    for (int mode = 0; mode < (argc > 3 ? 2 : 1); mode++) {
        for (int N : {1, 2, 4, 8}) {
            const int team = mode == 0 ? 3 : 1;
            std::atomic<size_t> bytes{0};
            std::atomic<bool> go{false}, stop{false};
            std::atomic<int> ready{0};
            std::vector<std::thread> th;
            for (int set = 0; set < N; set++)
                for (int st = 0; st < kStagers; st++)
                    th.emplace_back([&, set, st] {
                        const std::vector<int> cpus = nodes.empty() ? std::vector<int>() : nodes[(size_t)set % nodes.size()];
                        bind_to(cpus);
                        // this stager's sources: its own copies (distinct pages per thread, first-touched on the set's node) of a slice of the file list
                        std::vector<Src> mine;
                        const size_t per = std::max<size_t>(8, std::min<size_t>(kChunk, files.size() / (size_t)(N * kStagers)));
                        for (size_t k = 0; k < per; k++) {
                            const std::string& fn = files[((size_t)(set * kStagers + st) * per + k) % files.size()];
                            FILE* f = fopen(fn.c_str(), "rb");
                            if (!f) continue;
                            fseek(f, 0, SEEK_END);
                            const size_t len = (size_t)ftell(f);
                            fseek(f, 0, SEEK_SET);
                            uint8_t* p = (uint8_t*)aligned_alloc(4096, (len + 4095) & ~(size_t)4095);
                            if (fread(p, 1, len, f) != len) { fclose(f); free(p); continue; }
                            fclose(f);
                            mine.push_back(Src{p, len});
                        }
                        size_t chunk_bytes = 0;
                        for (const Src& s : mine) chunk_bytes += s.len + 64;
                        uint8_t* slot[2] = {nullptr, nullptr};
                        if (mode == 0)
                            for (auto& s : slot)
                                if (hipHostMalloc((void**)&s, chunk_bytes + 4096, hipHostMallocMapped) != hipSuccess) s = nullptr;
                        ready++;
                        while (!go.load()) std::this_thread::yield();
                        size_t done = 0;
                        for (size_t it = 0; !stop.load(); it++) {
                            std::vector<LpJpegHeader> hdrs(mine.size());
                            for (size_t k = 0; k < mine.size(); k++) (void)lp_jpeg_parse(mine[k].p, mine[k].len, &hdrs[k]);   // the header walk of the chunk
                            if (mode == 0) {
                                uint8_t* dst = slot[it & 1];
                                if (!dst) break;
                                std::vector<size_t> off(mine.size() + 1, 0);
                                for (size_t k = 0; k < mine.size(); k++) off[k + 1] = (off[k] + hdrs[k].ecs_len + 32 + 15) & ~(size_t)15;
                                auto part = [&](size_t t) {
                                    for (size_t k = mine.size() * t / (size_t)team; k < mine.size() * (t + 1) / (size_t)team; k++)
                                        stream_copy(dst + off[k], mine[k].p + hdrs[k].ecs_off, hdrs[k].ecs_len);
                                };
                                std::vector<std::thread> helpers;
                                for (int t = 1; t < team; t++) helpers.emplace_back([&, t] { bind_to(cpus); part((size_t)t); });
                                part(0);
                                for (auto& h : helpers) h.join();
                            } else {
                                for (const Src& s : mine) if (hipHostRegister(s.p, (s.len + 4095) & ~(size_t)4095, hipHostRegisterDefault) != hipSuccess) (void)hipGetLastError();
                                for (const Src& s : mine) (void)hipHostUnregister(s.p);
                            }
                            for (size_t k = 0; k < mine.size(); k++) done += hdrs[k].ecs_len;
                        }
                        bytes += done;
                        for (auto& s : slot) if (s) (void)hipHostFree(s);
                        for (const Src& s : mine) free(s.p);
                    });
            while (ready.load() < N * kStagers) std::this_thread::yield();
            const double t0 = now_s();
            go = true;
            std::this_thread::sleep_for(std::chrono::duration<double>(secs));
            stop = true;
            for (auto& t : th) t.join();
            const double dt = now_s() - t0, gbs = bytes.load() / dt / 1e9;
            printf("| %s | %d | %d | %.1f | %.1f | %d |\n", mode == 0 ? "staged (memcpy into pinned slots)" : "zero-copy (register / unregister the caller's pages)", N, N * kStagers * team, gbs,
                   gbs / N, 49 * N);
            fflush(stdout);
        }
    }
    return 0;
}
