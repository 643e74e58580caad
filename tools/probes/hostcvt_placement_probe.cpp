// What bounds the host converter pool's rate (fp64 rows -> fp32 / bf16 operand images into page-locked staging) on the GPU box?
// The boundary call converts config 5's 537 MB of K and V in ~4.0 ms (~134 GB/s of fp64 source) with 16, 32 or 64 pool threads
// alike; this probe runs the library's own row converter (sdpa_host_cvt_rows, C ABI) from PERSISTENT C++ threads (a spin barrier
// starts a round: no thread start-up in the figures) under several placements:
//   unpinned | pinned one thread per physical core on the source's NUMA node | spread over both nodes | SMT siblings
// and several thread counts, for a source array first-touched by the main thread (what a numpy caller hands over) and for a
// source first-touched IN PARALLEL by the threads that will read it (pages on both nodes).
//   hostcvt_placement_probe <libsdpa_hip.so> [rows cols]
// No GPU work except hipHostMalloc for the destination (falls back to malloc when there is no device).
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef int (*cvt_fn)(const double *, void *, long, int, int, int, double, int);
typedef void *(*alloc_fn)(size_t);

static std::vector<int> parse_cpulist(const char *path) {
    std::vector<int> out;
    FILE *f = fopen(path, "r");
    if (!f) return out;
    char buf[4096];
    if (fgets(buf, sizeof buf, f)) {
        char *p = buf;
        while (*p && *p != '\n') {
            int a = (int)strtol(p, &p, 10), b = a;
            if (*p == '-') b = (int)strtol(p + 1, &p, 10);
            for (int i = a; i <= b; ++i) out.push_back(i);
            if (*p == ',') ++p;
        }
    }
    fclose(f);
    return out;
}
static int node_of_page(void *p) {
    int status = -1;
    void *pages[1] = {(void *)((uintptr_t)p & ~(uintptr_t)4095)};
    long rc = syscall(SYS_move_pages, 0, 1UL, pages, nullptr, &status, 0);
    return rc == 0 ? status : -100;
}

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s <libsdpa_hip.so> [rows cols]\n", argv[0]); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    cvt_fn cvt = (cvt_fn)dlsym(h, "sdpa_host_cvt_rows");
    alloc_fn halloc = (alloc_fn)dlsym(h, "sdpa_host_alloc");
    if (!cvt) { printf("no sdpa_host_cvt_rows\n"); return 2; }
    const long rows = argc > 3 ? atol(argv[2]) : 131072;      // K and V of config 5 together: 131072 x 512 fp64 = 537 MB
    const int cols = argc > 3 ? atoi(argv[3]) : 512;
    const size_t n = (size_t)rows * cols;
    // topology
    std::vector<std::vector<int>> node_cpus;
    for (int nd = 0; nd < 8; ++nd) {
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
        std::vector<int> c = parse_cpulist(path);
        if (c.empty()) break;
        node_cpus.push_back(c);
    }
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    sched_getaffinity(0, sizeof allowed, &allowed);
    printf("hardware_concurrency %u, allowed cpus %d, numa nodes %zu", std::thread::hardware_concurrency(), CPU_COUNT(&allowed), node_cpus.size());
    for (size_t nd = 0; nd < node_cpus.size(); ++nd) printf(", node%zu: %zu cpus (%d..%d)", nd, node_cpus[nd].size(), node_cpus[nd].front(), node_cpus[nd].back());
    {
        FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
        char buf[128] = "?";
        if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
        buf[strcspn(buf, "\n")] = 0;
        printf(", cgroup cpu.max: %s\n", buf);
    }
    // which cpu is the SMT sibling of which: thread_siblings_list of cpu c
    auto siblings = [&](int c) {
        char path[160];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        return parse_cpulist(path);
    };
    // physical cores of a node: cpus that are the first of their sibling list
    auto phys = [&](int nd) {
        std::vector<int> out;
        for (int c : node_cpus[nd]) {
            std::vector<int> s = siblings(c);
            if ((s.empty() || s.front() == c) && CPU_ISSET(c, &allowed)) out.push_back(c);
        }
        return out;
    };
    double *src = (double *)aligned_alloc(4096, n * sizeof(double));
    void *dst = halloc ? halloc(n * 4) : nullptr;
    const bool pinned = dst != nullptr;
    if (!dst) dst = aligned_alloc(4096, n * 4);
    memset(dst, 0, n * 4);
    // one configuration: T threads (created, pinned, then started together by a spin barrier -- no start-up in the figure), `reps`
    // rounds, the best one; thread i runs on cpus[i % size] (empty = wherever the scheduler puts it)
    auto run = [&](int T, const std::vector<int> &cpus, int kind, bool first_touch, long item_rows, int reps) {
        std::atomic<int> go{0}, done{0};
        std::atomic<long> next{0};
        std::atomic<bool> quit{false};
        auto body = [&](int id) {
            if (!cpus.empty()) {
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET(cpus[id % cpus.size()], &set);
                pthread_setaffinity_np(pthread_self(), sizeof set, &set);
            }
            int seen = 0;
            for (;;) {
                while (go.load(std::memory_order_acquire) == seen) __builtin_ia32_pause();
                seen = go.load(std::memory_order_acquire);
                if (quit) return;
                for (;;) {
                    const long r0 = next.fetch_add(item_rows);
                    if (r0 >= rows) break;
                    const long nr = std::min(item_rows, rows - r0);
                    if (first_touch) {
                        for (size_t i = (size_t)r0 * cols; i < (size_t)(r0 + nr) * cols; ++i) src[i] = (double)(i % 977) * 1e-3 - 0.4;
                    } else if (kind == 0) {
                        cvt(src + (size_t)r0 * cols, (float *)dst + (size_t)r0 * cols, nr, cols, cols, 0, 1.0, 2);
                    } else {
                        cvt(src + (size_t)r0 * cols, (unsigned short *)dst + (size_t)r0 * cols, nr, cols, cols, 1, 1.0, 2);
                    }
                }
                done.fetch_add(1, std::memory_order_release);
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i) th.emplace_back(body, i);
        usleep(20000);
        double best = 1e9;
        for (int rep = 0; rep < reps; ++rep) {
            next.store(0); done.store(0);
            const double t0 = now();
            go.fetch_add(1, std::memory_order_release);
            while (done.load(std::memory_order_acquire) < T) __builtin_ia32_pause();
            best = std::min(best, now() - t0);
        }
        quit = true;
        go.fetch_add(1, std::memory_order_release);
        for (auto &t : th) t.join();
        return best;
    };
    for (int touch = 0; touch < 2; ++touch) {
        // (re)place the source's pages: by the main thread, or by 64 threads spread over both nodes
        if (touch == 0) {
            for (size_t i = 0; i < n; ++i) src[i] = (double)(i % 977) * 1e-3 - 0.4;
        } else {
            free(src);
            src = (double *)aligned_alloc(4096, n * sizeof(double));
            std::vector<int> spread;
            for (size_t nd = 0; nd < node_cpus.size(); ++nd) { std::vector<int> p = phys((int)nd); spread.insert(spread.end(), p.begin(), p.end()); }
            run((int)spread.size(), spread, 0, true, 16, 1);
        }
        const int src_node = node_of_page(src), src_node_mid = node_of_page(src + n / 2), src_node_end = node_of_page(src + n - 512);
        printf("== source %s: pages on node %d / %d / %d (first, middle, last); destination %s on node %d\n",
               touch ? "first-touched by 64+ threads over both nodes" : "first-touched by the main thread", src_node, src_node_mid, src_node_end,
               pinned ? "page-locked (sdpa_host_alloc)" : "malloc", node_of_page(dst));
        struct Placement { std::string name; std::vector<int> cpus; };
        std::vector<Placement> pls;
        pls.push_back({"unpinned", {}});
        const int sn = src_node >= 0 && src_node < (int)node_cpus.size() ? src_node : 0;
        pls.push_back({"one per physical core, source's node", phys(sn)});
        if (node_cpus.size() > 1) {
            pls.push_back({"one per physical core, OTHER node", phys(1 - sn)});
            std::vector<int> both, a = phys(0), b = phys(1);
            for (size_t i = 0; i < std::max(a.size(), b.size()); ++i) { if (i < a.size()) both.push_back(a[i]); if (i < b.size()) both.push_back(b[i]); }
            pls.push_back({"alternating over both nodes' physical cores", both});
        }
        for (const Placement &pl : pls) {
            if (pl.name != "unpinned" && pl.cpus.empty()) continue;
            for (int kind = 0; kind < 2; ++kind) {
                for (int T : {8, 16, 32, 64, 128}) {
                    if (!pl.cpus.empty() && T > (int)pl.cpus.size()) continue;
                    const double best = run(T, pl.cpus, kind, false, 16, 4);
                    printf("%-46s %-4s %3d threads: %.3f ms = %.1f GB/s of fp64 source\n", pl.name.c_str(), kind ? "bf16" : "f32", T, best, n * 8.0 / best / 1e6);
                }
            }
        }
        // item size at 32 threads, unpinned, bf16
        for (long item : {4L, 16L, 64L, 256L}) {
            const double best = run(32, {}, 1, false, item, 4);
            printf("unpinned bf16 32 threads, items of %ld rows (%ld KiB of source): %.3f ms = %.1f GB/s\n", item, item * cols * 8 / 1024, best, n * 8.0 / best / 1e6);
        }
    }
    return 0;
}
