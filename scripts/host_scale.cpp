// host_scale.cpp -- why does host staging fall from 299 GB/s at 12 threads to 41 GB/s at 96 (profiles/r03_a_ingest.md section 4)?
// Throughput that FALLS as threads are added is contention, not bandwidth. This probe separates the suspects:
//   destination memory : plain malloc (first touch by the copying thread) | hipHostMalloc (what the engines' upload slots are) |
//                        hipHostMalloc + NUMA-user flag | anonymous mmap + hipHostRegister | the same on 2 MiB huge pages (THP)
//   thread placement   : unbound | one thread per physical core, spread over the sockets, each next to its memory
//   the box itself     : cgroup CPU quota and throttle counters, automatic NUMA balancing (hint faults on pinned pages), THP settings,
//                        where the pages of a pinned buffer really are (numa_maps), /proc/vmstat deltas per point
// T threads each copy a 4 MiB source into their own 4 MiB destination in a loop (streaming stores, the stager's stream_copy) for a
// fixed time; the table is GB/s of bytes copied for the whole box, with the slowest and fastest thread.
//   hipcc -O2 -std=c++17 scripts/host_scale.cpp -lpthread -o scripts/host_scale ; scripts/host_scale [seconds per point]
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::string slurp(const char* path)
{
    std::string s;
    if (FILE* f = fopen(path, "r")) { char b[4096]; size_t n; while ((n = fread(b, 1, sizeof(b), f)) > 0) s.append(b, n); fclose(f); }
    while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
    return s;
}
static std::map<std::string, long long> vmstat()
{
    std::map<std::string, long long> m;
    if (FILE* f = fopen("/proc/vmstat", "r")) { char k[128]; long long v; while (fscanf(f, "%127s %lld", k, &v) == 2) m[k] = v; fclose(f); }
    return m;
}
static void stream_copy(uint8_t* dst, const uint8_t* src, size_t n)
{
    for (size_t i = 0; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a); _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c); _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    _mm_sfence();
}
// physical cores as (cpu id) lists per core, in an order that alternates sockets
static std::vector<int> core_order()
{
    std::map<std::pair<int, int>, int> first; // (package, core) -> lowest cpu
    const int n = (int)sysconf(_SC_NPROCESSORS_CONF);
    for (int c = 0; c < n; c++) {
        char p[160];
        snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", c);
        const std::string pk = slurp(p);
        snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/topology/core_id", c);
        const std::string co = slurp(p);
        if (pk.empty() || co.empty()) continue;
        const auto key = std::make_pair(atoi(pk.c_str()), atoi(co.c_str()));
        if (!first.count(key)) first[key] = c;
    }
    std::map<int, std::vector<int>> by_pkg;
    for (auto& kv : first) by_pkg[kv.first.first].push_back(kv.second);
    std::vector<int> order;
    for (size_t i = 0;; i++) {
        bool any = false;
        for (auto& kv : by_pkg) if (i < kv.second.size()) { order.push_back(kv.second[i]); any = true; }
        if (!any) break;
    }
    return order;
}

enum Dst { PLAIN, HIP_MALLOC, HIP_NUMA_USER, MMAP_REGISTER, THP_REGISTER, HIP_NONCOHERENT };
static const char* dst_name(Dst d)
{
    switch (d) {
    case PLAIN: return "malloc (not pinned)";
    case HIP_MALLOC: return "hipHostMalloc(Mapped)";
    case HIP_NUMA_USER: return "hipHostMalloc(Mapped|NumaUser)";
    case MMAP_REGISTER: return "mmap + hipHostRegister";
    case THP_REGISTER: return "mmap 2 MiB THP + hipHostRegister";
    default: return "hipHostMalloc(Mapped|NonCoherent)";
    }
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    int ndev = 0;
    (void)hipGetDeviceCount(&ndev);
    printf("## the box\n");
    printf("cpus online %ld, cgroup cpu.max = [%s], cpuset = [%s]\n", sysconf(_SC_NPROCESSORS_ONLN), slurp("/sys/fs/cgroup/cpu.max").c_str(), slurp("/sys/fs/cgroup/cpuset.cpus.effective").c_str());
    printf("kernel.numa_balancing = [%s], THP enabled = [%s], THP defrag = [%s]\n", slurp("/proc/sys/kernel/numa_balancing").c_str(), slurp("/sys/kernel/mm/transparent_hugepage/enabled").c_str(),
           slurp("/sys/kernel/mm/transparent_hugepage/defrag").c_str());
    for (int n = 0; n < 8; n++) {
        char p[128];
        snprintf(p, sizeof(p), "/sys/devices/system/node/node%d/cpulist", n);
        const std::string l = slurp(p);
        if (l.empty()) break;
        snprintf(p, sizeof(p), "/sys/devices/system/node/node%d/meminfo", n);
        const std::string mi = slurp(p);
        const size_t at = mi.find("MemTotal:");
        printf("NUMA node %d: cpus %s; %s\n", n, l.c_str(), at == std::string::npos ? "" : mi.substr(at, mi.find('\n', at) - at).c_str());
    }
    printf("cgroup cpu.stat before: %s\n\n", slurp("/sys/fs/cgroup/cpu.stat").c_str());
    const std::vector<int> cores = core_order();
    const size_t PIECE = 4u << 20;
    printf("## T threads, each copying a 4 MiB source into its own 4 MiB destination with streaming stores, %.1f s per point\n", secs);
    printf("| destination | placement | threads | GB/s (box) | slowest / fastest thread GB/s | numa_hint_faults | pgfault | throttled periods |\n|---|---|---:|---:|---|---:|---:|---:|\n");
    for (Dst kind : {PLAIN, HIP_MALLOC, HIP_NUMA_USER, HIP_NONCOHERENT, MMAP_REGISTER, THP_REGISTER}) {
        if (kind != PLAIN && ndev <= 0) continue;
        for (int bound = 0; bound < 2; bound++) {
            for (int T : {4, 12, 24, 48, 96, 192}) {
                if (bound && T > (int)cores.size()) continue;
                if (kind != PLAIN && kind != HIP_MALLOC && (T == 4 || T == 24 || T == 48)) continue; // the variants: the ends of the curve
                std::vector<double> rate((size_t)T, 0.0);
                std::atomic<int> ready{0};
                std::atomic<bool> go{false}, stop{false};
                std::vector<std::thread> th;
                std::vector<std::string> where((size_t)T);
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        if (bound) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(cores[(size_t)t % cores.size()], &s); (void)sched_setaffinity(0, sizeof(s), &s); }
                        uint8_t* src = (uint8_t*)aligned_alloc(4096, PIECE);
                        memset(src, t + 1, PIECE);
                        uint8_t* dst = nullptr;
                        void* map = nullptr;
                        size_t map_len = 0;
                        if (kind == PLAIN) dst = (uint8_t*)aligned_alloc(4096, PIECE);
                        else if (kind == HIP_MALLOC) { if (hipHostMalloc((void**)&dst, PIECE, hipHostMallocMapped) != hipSuccess) dst = nullptr; }
                        else if (kind == HIP_NUMA_USER) { if (hipHostMalloc((void**)&dst, PIECE, hipHostMallocMapped | hipHostMallocNumaUser) != hipSuccess) dst = nullptr; }
                        else if (kind == HIP_NONCOHERENT) { if (hipHostMalloc((void**)&dst, PIECE, hipHostMallocMapped | hipHostMallocNonCoherent) != hipSuccess) dst = nullptr; }
                        else {
                            map_len = kind == THP_REGISTER ? 2 * PIECE : PIECE;
                            map = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                            if (map != MAP_FAILED) {
                                dst = (uint8_t*)map;
                                if (kind == THP_REGISTER) { dst = (uint8_t*)(((uintptr_t)map + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1)); (void)madvise(dst, PIECE, MADV_HUGEPAGE); }
                                memset(dst, 0, PIECE);
                                if (hipHostRegister(dst, PIECE, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); dst = nullptr; }
                            }
                        }
                        if (dst) memset(dst, 0, PIECE);
                        ready++;
                        while (!go.load()) std::this_thread::yield();
                        size_t n = 0;
                        const double t0 = now_s();
                        while (dst && !stop.load()) { stream_copy(dst, src, PIECE); n++; }
                        rate[(size_t)t] = dst ? n * (double)PIECE / (now_s() - t0) / 1e9 : 0.0;
                        if (kind == PLAIN) free(dst);
                        else if (kind == HIP_MALLOC || kind == HIP_NUMA_USER || kind == HIP_NONCOHERENT) { if (dst) (void)hipHostFree(dst); }
                        else { if (dst) (void)hipHostUnregister(dst); if (map && map != MAP_FAILED) munmap(map, map_len); }
                        free(src);
                    });
                while (ready.load() < T) std::this_thread::yield();
                const auto v0 = vmstat();
                const std::string c0 = slurp("/sys/fs/cgroup/cpu.stat");
                go = true;
                std::this_thread::sleep_for(std::chrono::duration<double>(secs));
                stop = true;
                for (auto& t : th) t.join();
                const auto v1 = vmstat();
                const std::string c1 = slurp("/sys/fs/cgroup/cpu.stat");
                auto field = [](const std::string& s, const char* k) { const size_t at = s.find(k); return at == std::string::npos ? 0ll : atoll(s.c_str() + at + strlen(k)); };
                double sum = 0, lo = 1e30, hi = 0;
                for (double r : rate) { sum += r; lo = std::min(lo, r); hi = std::max(hi, r); }
                auto d = [&](const char* k) { return (v1.count(k) ? v1.at(k) : 0) - (v0.count(k) ? v0.at(k) : 0); };
                printf("| %s | %s | %d | %.1f | %.2f / %.2f | %lld | %lld | %lld |\n", dst_name(kind), bound ? "1 / physical core, sockets alternating" : "unbound", T, sum, lo, hi,
                       d("numa_hint_faults"), d("pgfault"), field(c1, "nr_throttled ") - field(c0, "nr_throttled "));
                fflush(stdout);
            }
        }
    }
    // where does a pinned buffer live?
    if (ndev > 0) {
        uint8_t* p = nullptr;
        if (hipHostMalloc((void**)&p, 64u << 20, hipHostMallocMapped) == hipSuccess) {
            memset(p, 1, 64u << 20);
            char key[32];
            snprintf(key, sizeof(key), "%012lx", (unsigned long)(uintptr_t)p);
            printf("\n## numa_maps of a 64 MiB hipHostMalloc buffer at %p (touched by the main thread)\n", (void*)p);
            if (FILE* f = fopen("/proc/self/numa_maps", "r")) {
                char line[1024];
                while (fgets(line, sizeof(line), f)) if (!strncmp(line, key, 12)) printf("%s", line);
                fclose(f);
            }
            (void)hipHostFree(p);
        }
    }
    printf("\ncgroup cpu.stat after: %s\n", slurp("/sys/fs/cgroup/cpu.stat").c_str());
    return 0;
}
